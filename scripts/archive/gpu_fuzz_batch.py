"""Randomised batches against single calls, bit for bit: 2 .. 32 frames of different cloud sizes (down to a few points, some frames losing every
point to the prune), chains of 4 .. 512 nodes, with / without the LLE term (the library's own H), priors, visibility weighting, fixed iteration
counts and the stopping rule, both precisions.  usage: python scripts/gpu_fuzz_batch.py [n_batches] [first_seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from trackdlo_amd import synth, binding as B
P = synth.LAUNCH_PARAMS
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
s0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
bad = 0; frames = 0
ctxs = {}
for seed in range(s0, s0 + n):
    rng = np.random.default_rng(91000 + seed)
    M = int(rng.choice([rng.integers(4, 65), rng.integers(65, 200), rng.integers(200, 513)], p=[0.7, 0.2, 0.1]))
    F = int(rng.integers(2, 33 if M <= 64 else 9)); prec = int(rng.integers(0, 2))
    lle = bool(rng.integers(0, 2)); tol = float(rng.choice([0.0, 2e-4])); iters = int(rng.integers(1, 9)) if tol == 0 else int(rng.choice([6, 30, 50]))
    vis_on = bool(rng.integers(0, 2)) and M >= 12 and not lle
    use_pri = bool(rng.integers(0, 2))
    key = (M > 64, M > 200)
    cap_n = 1 << 13
    ctx = ctxs.get(key)
    if ctx is None:
        ctx = ctxs[key] = B.Context(device=0, max_frames=32 if M <= 64 else 8, max_points=cap_n, max_nodes=64 if M <= 64 else (200 if M <= 200 else 512))
    pr = B.make_params(P["beta_pre_proc"] if lle else P["beta"], P["lambda_pre_proc"] if lle else P["lambda_"], P["lle_weight"], P["mu"], iters, tol, lle,
                       float(rng.choice([1.0, 3.0])) if use_pri else 0.0, P["k_vis"] if vis_on else 0.0, P["visibility_threshold"], prec)
    Ys, s2s = [], []
    vext = None; pri = None
    for f in range(F):
        N = int(rng.choice([rng.integers(1, 200), rng.integers(200, cap_n)]))
        X, Y0, v = synth.scene(N, M, config=1200 + seed, frame=f, occlude=(0.4, 0.6) if vis_on else None, noise=float(rng.choice([0.001, 0.003])))
        if rng.random() < 0.05: X = X + np.array([0.0, 0.0, 3.0])                 # this frame loses every point to the prune
        if len(X) == 0: X = Y0[:1].copy()
        if vis_on: vext = synth.extend_visible(v, M, synth.geodesic_coord(Y0))
        ctx.set_cloud(f, X)
        Ys.append(Y0); s2s.append(float(rng.choice([0.0, 1e-4, 2e-5])))
    if use_pri:
        idx = np.sort(rng.choice(M, size=max(1, M // 4), replace=False))
        pri = np.concatenate([idx[:, None].astype(float), Ys[0][idx] + rng.normal(0, 0.003, size=(len(idx), 3))], axis=1)
    single = [ctx.cpd_lle_resident(f, Ys[f], s2s[f], pr, priors=pri, visible_nodes=vext, check=False) for f in range(F)]
    try:
        out = ctx.cpd_lle_batch(Ys, s2s, pr, priors=pri, visible_nodes=vext)
        brc = 0
    except B.TdloError as e:
        out = None; brc = e.code if hasattr(e, "code") else -99
    worst_single = min(g["rc"] for g in single)
    frames += F
    if out is None:
        if worst_single == 0:
            bad += 1; print(f"BATCH FAILED ({brc}) WHERE EVERY SINGLE CALL SUCCEEDED: seed {seed} M {M} F {F} prec {prec} lle {lle} tol {tol} iters {iters}", flush=True)
        continue
    for f in range(F):
        st = out["stats"][f]
        ok = st["status"] == single[f]["status"] if "status" in single[f] else True
        if single[f]["rc"] == 0:
            ok = ok and np.array_equal(out["Y"][f], single[f]["Y"]) and out["sigma2"][f] == single[f]["sigma2"] and st["iters"] == single[f]["iters"] and st["n_kept"] == single[f]["n_kept"]
        if not ok:
            bad += 1
            print(f"MISMATCH seed {seed} frame {f}/{F} M {M} prec {prec} lle {lle} vis {vis_on} pri {use_pri} tol {tol} iters {iters}: single rc {single[f]['rc']} iters {single[f]['iters']} "
                  f"batch status {st['status']} iters {st['iters']}  |dY| {np.abs(out['Y'][f] - single[f]['Y']).max():.2e}", flush=True)
            break
print(f"{n} batches from seed {s0} ({frames} frames): {bad} not bit-identical to the single calls")
