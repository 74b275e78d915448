"""Randomised N-split registrations (one-shot exchange, R contexts on this GPU, one thread each) against the unsplit call: 1 .. 8 shards of uneven
sizes (some empty after the prune), chains of 4 .. 300 nodes, with / without the LLE term, visibility weighting, priors, the stopping rule, both
precisions.  Every rank must end with the same bits; fp64 mode within 1e-11 m of the unsplit registration (the shards' sums are added in another
order), fp32 mode within its stated tolerance; same iteration count and kept points.
usage: GPU_MAX_HW_QUEUES=16 python scripts/gpu_fuzz_split.py [n_cases] [first_seed]"""
import os, sys, queue, threading
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")      # ranks stacked on one GPU must not share a hardware queue (they wait for each other's flags)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from trackdlo_amd import synth, binding as B
P = synth.LAUNCH_PARAMS
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
s0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
bad = 0; compared = 0; both_failed = 0
for seed in range(s0, s0 + n):
    rng = np.random.default_rng(93000 + seed)
    R = int(rng.integers(1, 9)); M = int(rng.choice([rng.integers(4, 65), rng.integers(65, 301)], p=[0.75, 0.25])); prec = int(rng.integers(0, 2))
    lle = bool(rng.integers(0, 2)); tol = float(rng.choice([0.0, 2e-4])); iters = int(rng.integers(1, 9)) if tol == 0 else int(rng.choice([6, 30]))
    vis_on = bool(rng.integers(0, 2)) and M >= 12 and not lle
    use_pri = bool(rng.integers(0, 2))
    N = int(rng.integers(R * 8, 30000))
    X, Y0, v = synth.scene(N, M, config=1400 + seed, occlude=(0.4, 0.6) if vis_on else None, noise=float(rng.choice([0.001, 0.003])), outliers=int(rng.integers(0, 10)))
    vext = synth.extend_visible(v, M, synth.geodesic_coord(Y0)) if vis_on else None
    nX = len(X)
    cuts = np.sort(rng.integers(0, nX + 1, size=R - 1)) if R > 1 else np.array([], dtype=int)
    bounds = np.concatenate([[0], cuts, [nX]]).astype(int)
    if rng.random() < 0.2 and R > 1:                                   # one shard loses every point to the prune
        r = int(rng.integers(0, R)); X = X.copy(); X[bounds[r]:bounds[r + 1]] += np.array([0.0, 0.0, 4.0])
    pri = None
    if use_pri:
        idx = np.sort(rng.choice(M, size=max(1, M // 4), replace=False))
        pri = np.concatenate([idx[:, None].astype(float), Y0[idx] + rng.normal(0, 0.003, size=(len(idx), 3))], axis=1)
    pr = B.make_params(P["beta_pre_proc"] if lle else P["beta"], P["lambda_pre_proc"] if lle else P["lambda_"], P["lle_weight"], P["mu"], iters, tol, lle,
                       float(rng.choice([1.0, 3.0])) if use_pri else 0.0, P["k_vis"] if vis_on else 0.0, P["visibility_threshold"], prec)
    s2 = float(rng.choice([0.0, 1e-4, 2e-5]))
    c1 = B.Context(device=0, max_points=max(1024, nX), max_nodes=max(64, M))
    a = c1.cpd_lle(X, Y0, s2, pr, priors=pri, visible_nodes=vext, check=False)
    c1.close()
    ctxs = [B.Context(device=0, max_points=max(1024, int((bounds[1:] - bounds[:-1]).max()) + 1), max_nodes=max(64, M)) for _ in range(R)]
    inboxes = [c.xch_create(R, max(64, M)) for c in ctxs]
    out = queue.Queue()
    def work(r):
        try:
            ctxs[r].xch_bind(r, inboxes)
            shard = X[bounds[r]:bounds[r + 1]]
            if len(shard) == 0: shard = np.array([[0.0, 0.0, 9.0]])    # (a shard needs a cloud; this point is pruned)
            ctxs[r].set_cloud(0, shard)
            out.put((r, ctxs[r].split_run(Y0, s2, pr, priors=pri, visible_nodes=vext, check=False)))
        except Exception as e:
            out.put((r, dict(rc=-99, err=repr(e))))
    th = [threading.Thread(target=work, args=(r,)) for r in range(R)]
    [t.start() for t in th]; [t.join(timeout=120) for t in th]
    if any(t.is_alive() for t in th):
        print("HANG seed", seed, "R", R, "M", M, flush=True); os._exit(3)
    o = dict(out.get() for _ in range(R))
    for c in ctxs: c.close()
    tag = f"seed {seed} R {R} M {M} N {nX} prec {prec} lle {lle} vis {vis_on} pri {use_pri} tol {tol} iters {iters}"
    rcs = [o[r]["rc"] for r in range(R)]
    if a["rc"] != 0 or any(rcs):
        if not (a["rc"] != 0 and all(rc != 0 for rc in rcs)):
            bad += 1; print("ERROR MISMATCH", tag, "plain rc", a["rc"], "shards", rcs, [o[r].get("err") for r in range(R) if o[r]["rc"] == -99], flush=True)
        else: both_failed += 1
        continue
    compared += 1
    ty, ts = ((1e-5, 1e-3), (1e-11, 1e-9))[prec]
    kept = sum(o[r]["n_kept"] for r in range(R))
    dy = max(float(np.abs(o[r]["Y"] - a["Y"]).max()) for r in range(R)); ds = max(abs(o[r]["sigma2"] - a["sigma2"]) / a["sigma2"] for r in range(R))
    same = all(np.array_equal(o[r]["Y"], o[0]["Y"]) and o[r]["sigma2"] == o[0]["sigma2"] for r in range(R))
    its = all(o[r]["iters"] == a["iters"] and o[r]["converged"] == a["converged"] for r in range(R))
    if not (same and its and kept == a["n_kept"] and dy <= ty and ds <= ts):
        bad += 1
        print("MISMATCH", tag, f"same bits on all ranks {same} iterations {[o[r]['iters'] for r in range(R)]} vs {a['iters']} kept {kept} vs {a['n_kept']} dY {dy:.2e} dsigma2 {ds:.2e}", flush=True)
print(f"{n} split registrations from seed {s0}: {compared} compared, {both_failed} refused by both forms, {bad} outside")
