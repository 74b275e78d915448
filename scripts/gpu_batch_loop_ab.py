"""A batch's loop as ONE launch (k_batch_loop, TDLO_BATCH_PERSIST=1; an experiment, OFF by default) against the launch-per-step loop on stream groups (=0): same bits, whole-call rate,
alternating rounds on one box.   usage: python scripts/gpu_batch_loop_ab.py [F] [N] [rounds] [iters]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from trackdlo_amd import binding as B, synth
P = synth.LAUNCH_PARAMS
F = int(sys.argv[1]) if len(sys.argv) > 1 else 32
N = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 50
M = 50
pr = B.make_params(P['beta'], P['lambda_'], P['lle_weight'], P['mu'], iters, 0.0, False)
scenes = [synth.scene(N, M, config=2, frame=f)[:2] for f in range(F)]
Ys = [y for _, y in scenes]


def ctx_with(mode):
    os.environ["TDLO_BATCH_PERSIST"] = str(mode)
    c = B.Context(max_frames=F, max_points=N, max_nodes=M, timing=False)
    c.set_sort_reuse(False)
    for f, (X, _) in enumerate(scenes):
        c.set_cloud(f, X)
    return c


ref = None
for mode in (0, 1):
    c = ctx_with(mode)
    outs = [c.cpd_lle_batch(Ys, [0.0] * F, pr) for _ in range(2)]
    calls = int(c.lib.tdlo_debug_route_count(c.h, 12)); fb = int(c.lib.tdlo_debug_route_count(c.h, 13))
    o = outs[-1]
    print(f"persist={mode}: loop-kernel calls {calls} fallbacks {fb}  iters {sorted(set(s['iters'] for s in o['stats']))} status {sorted(set(s['status'] for s in o['stats']))} sigma2[0] {o['sigma2'][0]:.6e}", flush=True)
    assert np.array_equal(np.asarray(outs[0]["Y"]), np.asarray(outs[1]["Y"]))
    if ref is None: ref = o
    else: print("   same bits as the launch-per-step loop:", bool(np.array_equal(np.asarray(ref["Y"]), np.asarray(o["Y"])) and np.array_equal(ref["sigma2"], o["sigma2"])),
                " max |dY|", float(np.abs(np.asarray(ref["Y"]) - np.asarray(o["Y"])).max()), flush=True)
    c.close()
for r in range(rounds):
    for mode in (0, 1):
        c = ctx_with(mode)
        for i in range(5): c.cpd_lle_batch(Ys, [0.0] * F, pr)
        c.synchronize(); t0 = time.perf_counter()
        n = 40
        for i in range(n): c.cpd_lle_batch(Ys, [0.0] * F, pr)
        c.synchronize(); dt = time.perf_counter() - t0
        print(f"round {r} persist={mode}: {n * F * iters / dt:10.0f} EM it/s   {dt / n * 1e3:.4f} ms per call", flush=True)
        c.close()
