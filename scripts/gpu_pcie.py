"""PCIe-inclusive rate at C2: tdlo_cpd_lle with the cloud handed over as a host buffer on every call."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trackdlo_amd import binding as B, synth
P = synth.LAUNCH_PARAMS
ctx = B.Context(max_points=1 << 16, timing=False)      # wall-clock per call: without the optional stream markers
X, Y0, _ = synth.scene(50000, 50, config=2)
pr = B.make_params(P['beta'], P['lambda_'], P['lle_weight'], P['mu'], 50, 0.0, False)
for _ in range(5): ctx.cpd_lle(X, Y0, 0.0, pr)
t = time.perf_counter()
for _ in range(200): g = ctx.cpd_lle(X, Y0, 0.0, pr)
dt = (time.perf_counter() - t) / 200
print(f"cpd_lle with host cloud (1.2 MB H2D per call): {dt*1e3:.3f} ms per call = {50/dt:.0f} EM it/s")
ctx.set_cloud(0, X)
t = time.perf_counter()
for _ in range(200): g = ctx.cpd_lle_resident(0, Y0, 0.0, pr)
dt2 = (time.perf_counter() - t) / 200
print(f"resident cloud: {dt2*1e3:.3f} ms per call = {50/dt2:.0f} EM it/s")
import numpy as np
d, m, cam, _ = synth.depth_scene(30, config=9)
for _ in range(3): ctx.depth_to_cloud(0, d, m, cam['fx'], cam['fy'], cam['cx'], cam['cy'], 0.008, fetch=False)
t = time.perf_counter()
for _ in range(20): ctx.depth_to_cloud(0, d, m, cam['fx'], cam['fy'], cam['cx'], cam['cy'], 0.008, fetch=False)
print(f"depth_to_cloud 640x480 ({int(np.count_nonzero(m))} masked px): {(time.perf_counter()-t)/20*1e3:.3f} ms per call")
