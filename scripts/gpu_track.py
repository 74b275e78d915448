import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from trackdlo_amd import binding as B, synth
P = synth.LAUNCH_PARAMS
ctx = B.Context(max_points=1 << 16, timing=os.environ.get('TIMING', '1') != '0')      # TIMING=0: without the stream markers behind loop_ms / total_ms (the C API's default)
N, M = 5000, 45
X, Y0, _ = synth.scene(N, M, config=2)
coord = synth.geodesic_coord(Y0)
trk = B.trackdlo(M, P['visibility_threshold'], P['beta'], P['lambda_'], P['alpha'], P['k_vis'], P['mu'], 50, P['tol'], P['beta_pre_proc'], P['lambda_pre_proc'], P['lle_weight'], ctx=ctx)
trk.initialize_nodes(Y0); trk.initialize_geodesic_coord(coord)
vis = np.arange(M)
for _ in range(5): trk.tracking_step(X, vis, vis)
t = time.perf_counter()
for _ in range(200): trk.tracking_step(X, vis, vis)
dt = (time.perf_counter() - t) / 200
print(f"tracking_step N={N} M={M}: {dt*1e3:.3f} ms/frame; stats pre {trk.last_stats[0]['iters']} it loop {trk.last_stats[0]['loop_ms']:.3f} total {trk.last_stats[0]['total_ms']:.3f} host {trk.last_stats[0]['host_ms']:.3f} | main {trk.last_stats[1]['iters']} it loop {trk.last_stats[1]['loop_ms']:.3f} total {trk.last_stats[1]['total_ms']:.3f} host {trk.last_stats[1]['host_ms']:.3f}")
