"""For the draws of scripts/gpu_fuzz_band.py outside the fp64 gate: whose error is it?  The oracle's faithful QR solve, its diagnostic
quadruple-precision solve of the same double-precision system, the banded L D L^T and the dense pivoted GPU eliminations, all on the same draw.
usage: python scripts/gpu_fuzz_band_adjudicate.py seed [seed ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from trackdlo_amd import binding as B
from oracle import ref_cpu
from gpu_fuzz_band import draw, params
for seed in [int(a) for a in sys.argv[1:]]:
    X, Y0, H, kw, pri, s2 = draw(seed)
    o = ref_cpu.cpd_lle(X, Y0, s2, priors=pri, H=H, **kw)
    with ref_cpu.extended_solver():
        ox = ref_cpu.cpd_lle(X, Y0, s2, priors=pri, H=H, **kw)
    res = {}
    for dense in (False, True):
        prev = B.mstep_lle_dense(dense)
        ctx = B.Context(device=0, max_points=1 << 14, max_nodes=512)
        res[dense] = ctx.cpd_lle(X, Y0, s2, params(kw), priors=pri, H=H, check=False)
        ctx.close(); B.mstep_lle_dense(prev)
    d = lambda a, b: float(np.abs(a["Y"] - b["Y"]).max())
    print(f"seed {seed} M={len(Y0)} H max {np.abs(H).max():.1e} iters {o['iters']}/{ox['iters']}/{res[False]['iters']}/{res[True]['iters']}: |QR - quad| {d(o, ox):.1e}  |band - QR| {d(res[False], o):.1e}  "
          f"|band - quad| {d(res[False], ox):.1e}  |dense GPU - QR| {d(res[True], o):.1e}  |dense GPU - quad| {d(res[True], ox):.1e}  |band - dense GPU| {d(res[False], res[True]):.1e}", flush=True)
