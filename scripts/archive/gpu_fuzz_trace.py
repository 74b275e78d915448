"""sigma2 per iteration, oracle against product, for one draw of scripts/gpu_fuzz_chain.py (FUZZ_N etc. as there).
usage: FUZZ_N=1,150 python scripts/gpu_fuzz_trace.py seed"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from trackdlo_amd import binding as B, synth
from oracle import ref_cpu
NRANGE = [int(v) for v in os.environ["FUZZ_N"].split(",")] if os.environ.get("FUZZ_N") else None
seed = int(sys.argv[1])
rng = np.random.default_rng(77000 + seed)
M = int(rng.choice([rng.integers(4, 65), rng.integers(65, 200), rng.integers(200, 513)], p=[0.7, 0.2, 0.1]))
N = int(rng.integers(200, 9000)); iters = int(rng.integers(1, 12))
if NRANGE: N = int(rng.integers(NRANGE[0], NRANGE[1] + 1))
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
print("M", M, "N", len(X), "iters", iters, "sigma2 in", s2, kw)
ctx = B.Context(device=0, max_points=1 << 14, max_nodes=512)
for dense in (False, True):
    B.mstep_dense(dense)
    for it in range(1, iters + 1):
        k = dict(kw, max_iter=it, tol=0.0)
        o = ref_cpu.cpd_lle(X, Y0, s2, priors=pri, visible_nodes=vext, **k)
        g = ctx.cpd_lle(X, Y0, s2, B.make_params(k["beta"], k["lambda_"], k["lle_weight"], k["mu"], it, 0.0, False, k["alpha"], k["k_vis"], k["visibility_threshold"], 1),
                        priors=pri, visible_nodes=vext, check=False)
        print(f"{'dense' if dense else 'chain'} it {it}: oracle sigma2 {o['sigma2']:.6e} kept {o['n_kept']}  product rc {g['rc']} sigma2 {g['sigma2']:.6e} kept {g['n_kept']}  |dY| {np.abs(g['Y'] - o['Y']).max():.2e}", flush=True)
B.mstep_dense(False)
