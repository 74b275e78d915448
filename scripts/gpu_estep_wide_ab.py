"""fp64 E-step of long chains: batches with a wide node window lane = node (tdlo_estep_wide.h, the default) against thread = point throughout (TDLO_ESTEP_WIDE=0).
Results after 1 .. 50 iterations side by side (the two forms differ in the order of additions: 1e-16 relative), whole-call time at C5, and a smaller case against
the CPU oracle.   usage: python scripts/gpu_estep_wide_ab.py [N] [M] [rounds]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from trackdlo_amd import binding as B, synth
P = synth.LAUNCH_PARAMS
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
M = int(sys.argv[2]) if len(sys.argv) > 2 else 300
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
X, Y0, _ = synth.scene(N, M, config=5)


def ctx_with(mode):
    os.environ["TDLO_ESTEP_WIDE"] = str(mode)
    c = B.Context(max_points=N, max_nodes=M, timing=False)
    os.environ.pop("TDLO_ESTEP_WIDE")
    c.set_sort_reuse(False)
    c.set_cloud(0, X)
    return c


MODES = [int(a) for a in os.environ.get("MODES", "0,1").split(",")]
cs = {m: ctx_with(m) for m in MODES}
for iters in (1, 2, 3, 5, 8, 50):
    pr = B.make_params(P['beta'], P['lambda_'], P['lle_weight'], P['mu'], iters, 0.0, False, precision=B.PREC_F64)
    o = {m: cs[m].cpd_lle_resident(0, Y0, 0.0, pr) for m in MODES}; o = {0: o[MODES[0]], 1: o[MODES[-1]]}
    dY = float(np.abs(np.asarray(o[0]["Y"]) - np.asarray(o[1]["Y"])).max())
    print(f"iters {iters:2d}: status {o[0]['status']} {o[1]['status']}  sigma2 {o[0]['sigma2']:.12e} {o[1]['sigma2']:.12e}  rel {abs(o[0]['sigma2'] - o[1]['sigma2']) / o[0]['sigma2']:.1e}  max|dY| {dY:.2e}", flush=True)
pr = B.make_params(P['beta'], P['lambda_'], P['lle_weight'], P['mu'], 50, 0.0, False, precision=B.PREC_F64)
for r in range(rounds):
    for m in MODES:
        c = cs[m]
        for _ in range(3): c.cpd_lle_resident(0, Y0, 0.0, pr)
        t0 = time.perf_counter()
        K = 20
        for _ in range(K): c.cpd_lle_resident(0, Y0, 0.0, pr)
        dt = (time.perf_counter() - t0) / K
        print(f"round {r} wide={m}: {dt * 1e3:.4f} ms per call  {50 / dt:9.0f} EM it/s", flush=True)
for c in cs.values(): c.close()
