import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from trackdlo_amd import binding as B, synth
ctx = B.Context(max_points=1 << 16)
d, m, cam, _ = synth.depth_scene(30, config=9)
a = (cam['fx'], cam['fy'], cam['cx'], cam['cy'])
for _ in range(3): ctx.depth_to_cloud(0, d, m, *a, 0.008, fetch=False)
t = time.perf_counter()
for _ in range(50): ctx.depth_to_cloud(0, d, m, *a, 0.008, fetch=False)
print(f"depth_to_cloud 640x480 ({int(np.count_nonzero(m))} masked px): {(time.perf_counter()-t)/50*1e3:.3f} ms per call")
# whole frame born on the device: depth image -> cloud -> visibility pre-pass -> tracking_step on the resident cloud
P = synth.LAUNCH_PARAMS
M = 30
_, _, _, Y0 = synth.depth_scene(M, config=9)
coord = synth.geodesic_coord(Y0)
trk = B.trackdlo(M, P['visibility_threshold'], P['beta'], P['lambda_'], P['alpha'], P['k_vis'], P['mu'], 50, P['tol'], P['beta_pre_proc'], P['lambda_pre_proc'], P['lle_weight'], ctx=ctx)
trk.initialize_nodes(Y0); trk.initialize_geodesic_coord(coord)
def frame():
    ctx.depth_to_cloud(0, d, m, *a, 0.008, fetch=False)
    _, vis, vext = ctx.visibility_prepass(0, trk.get_tracking_result(), 0.02, 0.06, coord)
    trk.tracking_step(None, vis, vext)
for _ in range(5): frame()
t = time.perf_counter()
for _ in range(50): frame()
print(f"device-born frame (depth -> cloud -> visibility -> tracking_step), M={M}: {(time.perf_counter()-t)/50*1e3:.3f} ms per frame")
