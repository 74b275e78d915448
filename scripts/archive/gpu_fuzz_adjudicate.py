"""For the draws of scripts/gpu_fuzz_chain.py outside the fp64 gate: whose error is it?  The oracle's faithful QR solve, its diagnostic
quadruple-precision solve of the same double-precision system, the dense GPU eliminations and the chain smoother, all on the same draw."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from trackdlo_amd import binding as B, synth
from oracle import ref_cpu
for seed in [int(a) for a in sys.argv[1:]]:
    rng = np.random.default_rng(77000 + seed)
    M = int(rng.choice([rng.integers(4, 65), rng.integers(65, 200), rng.integers(200, 513)], p=[0.7, 0.2, 0.1]))
    N = int(rng.integers(200, 9000)); iters = int(rng.integers(1, 12))
    vis = bool(rng.integers(0, 2)) and M >= 12
    use_pri = bool(rng.integers(0, 2))
    X, Y0, v = synth.scene(N, M, config=500 + seed, frame=seed, noise=float(rng.choice([0.0005, 0.002, 0.004])), occlude=(0.35, 0.55) if vis else None,
                           outliers=int(rng.integers(0, 20)), shift=(0.0, float(rng.uniform(0, 0.008)), float(rng.uniform(-0.003, 0.003))))
    vext = synth.extend_visible(v, M, synth.geodesic_coord(Y0)) if vis else None
    kw = dict(beta=float(rng.choice([0.1, 0.35, 0.6, 3.0])), lambda_=float(rng.choice([1.0, 500.0, 50000.0])), lle_weight=10.0,
              mu=float(rng.choice([0.05, 0.1, 0.3])), max_iter=iters, tol=float(rng.choice([0.0, 2e-4])), include_lle=False, alpha=0.0,
              k_vis=50.0 if vis else 0.0, visibility_threshold=0.008)
    pri = None
    if use_pri:
        idx = np.sort(rng.choice(M, size=max(1, M // 4), replace=False))
        pri = np.concatenate([idx[:, None].astype(float), Y0[idx] + rng.normal(0, 0.003, size=(len(idx), 3))], axis=1)
        kw["alpha"] = float(rng.choice([1.0, 3.0]))
    s2 = float(rng.choice([0.0, 1e-4, 2e-5]))
    o = ref_cpu.cpd_lle(X, Y0, s2, priors=pri, visible_nodes=vext, **kw)
    with ref_cpu.extended_solver():
        ox = ref_cpu.cpd_lle(X, Y0, s2, priors=pri, visible_nodes=vext, **kw)
    pr = B.make_params(kw["beta"], kw["lambda_"], kw["lle_weight"], kw["mu"], kw["max_iter"], kw["tol"], False, kw["alpha"], kw["k_vis"], kw["visibility_threshold"], 1)
    res = {}
    for dense in (False, True):
        prev = B.mstep_dense(dense)
        ctx = B.Context(device=0, max_points=1 << 14, max_nodes=512)
        res[dense] = ctx.cpd_lle(X, Y0, s2, pr, priors=pri, visible_nodes=vext, check=False)
        ctx.close(); B.mstep_dense(prev)
    d = lambda a, b: float(np.abs(a["Y"] - b["Y"]).max())
    print(f"seed {seed} M={M} beta={kw['beta']} lambda={kw['lambda_']}: |QR - quad| {d(o, ox):.1e}  |chain - QR| {d(res[False], o):.1e}  |chain - quad| {d(res[False], ox):.1e}  "
          f"|dense GPU - QR| {d(res[True], o):.1e}  |dense GPU - quad| {d(res[True], ox):.1e}  |chain - dense GPU| {d(res[False], res[True]):.1e}")
