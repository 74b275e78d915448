import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from trackdlo_amd import binding as B, synth
P = synth.LAUNCH_PARAMS
M = 30
ctx = B.Context(device=0, timing=False)
depth, mask, cam, Y0 = synth.depth_scene(M, config=9, frame=3)
a = (cam["fx"], cam["fy"], cam["cx"], cam["cy"])
coord = synth.geodesic_coord(Y0)
X, n, _ = ctx.depth_to_cloud(0, depth, mask, *a, 0.008)
vis = np.arange(M, dtype=np.int32)
def mk():
    t = B.trackdlo(M, P["visibility_threshold"], P["beta"], P["lambda_"], P["alpha"], P["k_vis"], P["mu"], 50, P["tol"], P["beta_pre_proc"], P["lambda_pre_proc"], P["lle_weight"], ctx=ctx)
    t.initialize_nodes(Y0); t.initialize_geodesic_coord(coord); return t
def rate(fn, n=300):
    for _ in range(20): fn()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    return (time.perf_counter() - t0) * 1e3 / n
t1 = mk(); r1 = rate(lambda: t1.tracking_step(X, vis, vis)); s1 = [s["iters"] for s in t1.last_stats]
ctx.depth_to_cloud(0, depth, mask, *a, 0.008, fetch=False)
t2 = mk(); r2 = rate(lambda: t2.tracking_step(None, vis, vis)); s2 = [s["iters"] for s in t2.last_stats]
print(f"N={n} M={M}: tracking_step(X from host) {r1:.4f} ms iters {s1}; tracking_step(None: resident cloud) {r2:.4f} ms iters {s2}; routes {ctx.route_counts()}")
os.environ["TDLO_TRACK_PROFILE"] = "1"
