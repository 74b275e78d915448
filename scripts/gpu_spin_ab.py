"""The spin-ahead loop of one frame (TDLO_SPIN_AHEAD=1: E-steps on a second stream, kernels parked on device words) against the ordinary loop at C2:
same bits (the kernels compute the same things in the same order), whole-call rate, alternating rounds on one box.
usage: python scripts/gpu_spin_ab.py [rounds]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from trackdlo_amd import binding as B, synth
P = synth.LAUNCH_PARAMS
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
N, M = 50000, 50
pr = B.make_params(P['beta'], P['lambda_'], P['lle_weight'], P['mu'], 50, 0.0, False)
pairs = [synth.scene(N, M, config=2, frame=f)[:2] for f in range(2)]


def ctx_with(mode):
    os.environ["TDLO_SPIN_AHEAD"] = str(mode)
    c = B.Context(max_frames=2, max_points=N, max_nodes=M, timing=False)
    c.set_sort_reuse(False)
    for k, (X, _) in enumerate(pairs):
        c.set_cloud(k, X)
    return c


ref = None
for mode in (0, 1):
    c = ctx_with(mode)
    out = [c.cpd_lle_resident(k, pairs[k][1], 0.0, pr) for k in (0, 1, 0)]
    print(f"spin={mode}: spin calls {int(c.lib.tdlo_debug_route_count(c.h, 11))}  iters {[o['iters'] for o in out]} status {[o['status'] for o in out]}  sigma2 {out[0]['sigma2']:.6e}", flush=True)
    if ref is None: ref = out
    else:
        for a, b in zip(ref, out):
            print("   same bits as the ordinary loop:", bool(np.array_equal(a["Y"], b["Y"]) and a["sigma2"] == b["sigma2"]), " max |dY|", float(np.abs(a["Y"] - b["Y"]).max()))
    c.close()
for r in range(rounds):
    for mode in (0, 1):
        c = ctx_with(mode)
        for i in range(30): c.cpd_lle_resident(i & 1, pairs[i & 1][1], 0.0, pr)
        c.synchronize(); t0 = time.perf_counter()
        n = 600
        for i in range(n): c.cpd_lle_resident(i & 1, pairs[i & 1][1], 0.0, pr)
        c.synchronize(); dt = time.perf_counter() - t0
        print(f"round {r} spin={mode}: {n * 50 / dt:9.0f} EM it/s   {dt / n * 1e3:.4f} ms per call  ({dt / n / 50 * 1e6:.2f} us per iteration incl. prune)", flush=True)
        c.close()
