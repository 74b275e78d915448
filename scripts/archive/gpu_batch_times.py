import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from trackdlo_amd import binding as B, synth
P = synth.LAUNCH_PARAMS
F, N, M = 32, 50000, 50
pr = B.make_params(P['beta'], P['lambda_'], P['lle_weight'], P['mu'], 50, 0.0, False)
scenes = [synth.scene(N, M, config=2, frame=f) for f in range(F)]
ctx = B.Context(max_frames=F, max_points=N, max_nodes=M)
for f in range(F): ctx.set_cloud(f, scenes[f][0])
Ys = [s[1] for s in scenes]
for _ in range(3): ctx.cpd_lle_batch(Ys, [0.0] * F, pr)
for _ in range(8):
    t = time.perf_counter(); r = ctx.cpd_lle_batch(Ys, [0.0] * F, pr); dt = time.perf_counter() - t
    s = r['stats'][0]
    print(f"python wall {dt*1e3:.3f} ms | C host {s['host_ms']:.3f} | GPU total (ev0-ev3) {s['total_ms']:.3f} | loop {s['loop_ms']:.3f}")
