"""Randomised sweep of reg (utils.cpp:21-82, plain GMM-EM: the node initialisation of the ROS node) against the oracle: cloud sizes 5 .. 60 000,
2 .. 400 centroids, mu, iteration counts, clouds off the origin.  usage: python scripts/gpu_fuzz_reg.py [n_cases] [first_seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from trackdlo_amd import binding as B, synth
from oracle import ref_cpu
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
s0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
ctx = B.Context(device=0, max_points=1 << 16, max_nodes=512)
bad = 0; worst = (0.0, None); degen = 0
for seed in range(s0, s0 + n):
    rng = np.random.default_rng(66000 + seed)
    M = int(rng.choice([rng.integers(2, 40), rng.integers(40, 400)], p=[0.8, 0.2]))
    N = int(rng.choice([rng.integers(5, 300), rng.integers(300, 60000)]))
    mu = float(rng.choice([0.0, 0.05, 0.3])); iters = int(rng.integers(0, 40))
    X, _, _ = synth.scene(N, max(4, min(M, 60)), config=1700 + seed, frame=seed, noise=float(rng.choice([0.0005, 0.003])))
    X = X - np.array([0.0, 0.0, float(rng.choice([0.0, 0.6]))])
    try:
        Yo, so = ref_cpu.reg(X, M, mu=mu, max_iter=iters)
        ok_o = bool(np.all(np.isfinite(Yo)) and np.isfinite(so) and so > 1e-14)
    except Exception:
        Yo = None; ok_o = False
    try:
        Yg, sg = ctx.reg(X, M, mu=mu, max_iter=iters)
        ok_g = True
    except B.TdloError as e:
        ok_g = False; err = str(e)
    if not ok_o:
        degen += 1                      # the reference divides by a collapsed sigma2 / produces NaN: nothing to compare; the product must only come back
        continue
    if not ok_g:
        bad += 1; print(f"PRODUCT FAILED seed {seed} M {M} N {N} mu {mu} iters {iters}: {err}", flush=True); continue
    dy = float(np.abs(Yg - Yo).max()); ds = abs(sg - so) / so
    if dy > worst[0]: worst = (dy, (seed, M, N, mu, iters))
    if dy > 1e-9 or ds > 1e-9:
        bad += 1; print(f"MISMATCH seed {seed} M {M} N {N} mu {mu} iters {iters}: dY {dy:.2e} dsigma2 {ds:.2e} sigma2 {so:.3e}", flush=True)
print(f"{n} reg cases from seed {s0}: {bad} outside (1e-9 m, 1e-9), {degen} where the oracle's own result is not finite or collapsed; worst |dY| {worst[0]:.2e} at {worst[1]}")
