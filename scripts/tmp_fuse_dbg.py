import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from trackdlo_amd import binding as B, synth
P = synth.LAUNCH_PARAMS
N, M = 5000, 45
X, Y0, v = synth.scene(N, M, config=72)
for nth in (1, 2, 3):
    os.environ["TDLO_FUSE_FORCE_TIMEOUT"] = str(nth)
    c = B.Context(device=0, timing=False, max_points=N, max_nodes=64)
    os.environ.pop("TDLO_FUSE_FORCE_TIMEOUT")
    c.set_sort_reuse(False); c.set_cloud(0, X)
    p = B.make_params(P["beta"], P["lambda_"], P["lle_weight"], P["mu"], 5, 0.0, False)
    for k in range(4):
        t0 = time.perf_counter(); g = c.cpd_lle_resident(0, Y0, 0.0, p); dt = time.perf_counter() - t0
        print(nth, k, round(dt, 4), c.route_counts(), g["iters"], g["rc"], flush=True)
    c.close()
