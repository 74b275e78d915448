"""Production-like timing: tol = 2e-4 (early exit), tracking_step with two registrations, frames/s."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from trackdlo_amd import binding as B, synth
P = synth.LAUNCH_PARAMS
ctx = B.Context(max_frames=32, max_points=1 << 16)
for N, M in ((50000, 50), (5000, 45)):
    X, Y0, _ = synth.scene(N, M, config=2)
    pr = B.make_params(P['beta'], P['lambda_'], P['lle_weight'], P['mu'], 50, P['tol'], False)
    ctx.set_cloud(0, X)
    for _ in range(3): g = ctx.cpd_lle_resident(0, Y0, 0.0, pr)
    t = time.perf_counter()
    for _ in range(20): g = ctx.cpd_lle_resident(0, Y0, 0.0, pr)
    dt = (time.perf_counter() - t) / 20
    print(f"cpd_lle tol=2e-4 N={N} M={M}: iters={g['iters']} conv={g['converged']} host_ms/call={dt*1e3:.3f} loop_ms={g['loop_ms']:.3f} total_ms={g['total_ms']:.3f}")
    coord = synth.geodesic_coord(Y0)
    trk = B.trackdlo(M, P['visibility_threshold'], P['beta'], P['lambda_'], P['alpha'], P['k_vis'], P['mu'], 50, P['tol'], P['beta_pre_proc'], P['lambda_pre_proc'], P['lle_weight'], ctx=ctx)
    trk.initialize_nodes(Y0); trk.initialize_geodesic_coord(coord)
    vis = np.arange(M)
    for _ in range(3): trk.tracking_step(X, vis, vis)
    t = time.perf_counter()
    for _ in range(20): trk.tracking_step(X, vis, vis)
    dt = (time.perf_counter() - t) / 20
    print(f"tracking_step N={N} M={M}: {dt*1e3:.3f} ms/frame ({1/dt:.1f} frames/s) iters pre/main = {trk.last_stats[0]['iters']}/{trk.last_stats[1]['iters']}")
# batch throughput: F frames of C2 concurrently, tol = 0
N, M = 50000, 50
pr = B.make_params(P['beta'], P['lambda_'], P['lle_weight'], P['mu'], 50, 0.0, False)
for F in (1, 2, 4, 8, 16, 32):
    Ys = []
    for f in range(F):
        X, Y0, _ = synth.scene(N, M, config=2, frame=f); ctx.set_cloud(f, X); Ys.append(Y0)
    ctx.cpd_lle_batch(Ys, [0.0] * F, pr)
    t = time.perf_counter()
    for _ in range(5): out = ctx.cpd_lle_batch(Ys, [0.0] * F, pr)
    dt = (time.perf_counter() - t) / 5
    print(f"batch F={F}: {dt*1e3:.3f} ms per batch call, {F*50/dt:.0f} EM it/s, loop_ms={out['stats'][0]['loop_ms']:.3f}")
