"""Randomised sweep of the visibility pre-pass (trackdlo_node.cpp:257-277, :345-360) against the oracle: chain lengths 4 .. 512, clouds of 1 .. 60 000
points, random occlusions, thresholds and gap-fill distances; node distances to 1e-12 m, the visible and extended index sets exactly.
usage: python scripts/gpu_fuzz_prepass.py [n_cases] [first_seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from trackdlo_amd import binding as B, synth
from oracle import ref_cpu
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
s0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
ctx = B.Context(device=0, max_points=1 << 16, max_nodes=512)
bad = 0; near = 0
for seed in range(s0, s0 + n):
    rng = np.random.default_rng(68000 + seed)
    M = int(rng.choice([rng.integers(4, 65), rng.integers(65, 513)], p=[0.7, 0.3]))
    N = int(rng.choice([rng.integers(1, 200), rng.integers(200, 60000)]))
    occl = None
    k = int(rng.integers(0, 4))
    if k == 1: occl = (0.0, float(rng.uniform(0.1, 0.5)))
    elif k == 2: occl = (float(rng.uniform(0.5, 0.9)), 1.0)
    elif k == 3:
        a = float(rng.uniform(0.1, 0.7)); occl = (a, a + float(rng.uniform(0.02, 0.3)))
    X, Y0, _ = synth.scene(N, M, config=1900 + seed, frame=seed, occlude=occl, noise=float(rng.choice([0.0005, 0.002, 0.006])), outliers=int(rng.integers(0, 10)),
                           shift=(0.0, float(rng.uniform(0, 0.01)), 0.0))
    if len(X) == 0: continue
    coord = synth.geodesic_coord(Y0)
    thr = float(rng.choice([0.002, 0.008, 0.02])); dvis = float(rng.choice([0.0, 0.03, 0.06, 0.5]))
    ctx.set_cloud(0, X)
    d, vis, ext = ctx.visibility_prepass(0, Y0, thr, dvis, coord)
    do, viso, exto = ref_cpu.visibility_prepass(X, Y0, thr, dvis, coord)
    ok = np.abs(d - do).max() <= 1e-12 and np.array_equal(vis, viso) and np.array_equal(ext, exto)
    if not ok:
        # a node whose distance sits within rounding of the threshold may legitimately fall on either side: say so instead of counting it
        edge = np.abs(do - thr).min() <= 1e-12
        if edge: near += 1
        else:
            bad += 1; print(f"MISMATCH seed {seed} M {M} N {len(X)} thr {thr} dvis {dvis}: max|dd| {np.abs(d - do).max():.2e} visible {len(vis)}/{len(viso)} extended {len(ext)}/{len(exto)}", flush=True)
print(f"{n} pre-pass cases from seed {s0}: {bad} differ, {near} with a node distance within 1e-12 m of the threshold")
