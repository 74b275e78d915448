"""The parts of a device-born frame (trackdlo_node.cpp:195-369), ms per call each: depth -> cloud (pinned images), visibility pre-pass, tracking_step on the resident cloud."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from trackdlo_amd import binding as B, synth
P = synth.LAUNCH_PARAMS
M = 30
ctx = B.Context(device=0, timing=False)
for shape in ((480, 640), (720, 1280)):
    depth, mask, cam, Y0 = synth.depth_scene(M, config=9, frame=3, rows=shape[0], cols=shape[1])
    a = (cam["fx"], cam["fy"], cam["cx"], cam["cy"])
    coord = synth.geodesic_coord(Y0)
    trk = B.trackdlo(M, P["visibility_threshold"], P["beta"], P["lambda_"], P["alpha"], P["k_vis"], P["mu"], 50, P["tol"], P["beta_pre_proc"], P["lambda_pre_proc"], P["lle_weight"], ctx=ctx)
    trk.initialize_nodes(Y0); trk.initialize_geodesic_coord(coord)
    d, m = ctx.image_buffers(*shape); d[:] = depth; m[:] = mask
    def rate(fn, n=300):
        for _ in range(20): fn()
        t0 = time.perf_counter()
        for _ in range(n): fn()
        return (time.perf_counter() - t0) * 1e3 / n
    ctx.depth_to_cloud(0, d, m, *a, 0.008, fetch=False)
    Yc = trk.get_tracking_result()
    _, vis, vext = ctx.visibility_prepass(0, Yc, P["visibility_threshold"], 0.06, coord)
    t_cloud = rate(lambda: ctx.depth_to_cloud(0, d, m, *a, 0.008, fetch=False))
    t_vis = rate(lambda: ctx.visibility_prepass(0, Yc, P["visibility_threshold"], 0.06, coord))
    t_trk = rate(lambda: trk.tracking_step(None, vis, vext))
    print(f"{shape[1]}x{shape[0]}: depth_to_cloud {t_cloud:.4f}  visibility_prepass {t_vis:.4f}  tracking_step (resident cloud, {len(vis)}/{M} visible) {t_trk:.4f} ms   routes {ctx.route_counts()}", flush=True)
ctx.close()
