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
