"""32-frame batch (C3 per GPU): EM it/s against the number of E-step workgroups per frame."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from trackdlo_amd import binding as B, synth
P = synth.LAUNCH_PARAMS
F, N, M = 32, 50000, 50
pr = B.make_params(P['beta'], P['lambda_'], P['lle_weight'], P['mu'], 50, 0.0, False)
scenes = [synth.scene(N, M, config=2, frame=f) for f in range(F)]
for blocks in (0, 196, 131, 98, 66, 49, 0):
    ctx = B.Context(max_frames=F, max_points=N, max_nodes=M, estep_blocks=blocks)
    for f in range(F): ctx.set_cloud(f, scenes[f][0])
    Ys = [s[1] for s in scenes]
    for _ in range(3): ctx.cpd_lle_batch(Ys, [0.0] * F, pr)
    t = time.perf_counter()
    for _ in range(10): r = ctx.cpd_lle_batch(Ys, [0.0] * F, pr)
    dt = (time.perf_counter() - t) / 10
    print(f"estep_blocks={blocks}: {dt*1e3:.3f} ms per batch call, {F*50/dt:.0f} EM it/s, loop {r['stats'][0]['loop_ms']:.3f} ms", flush=True)
    ctx.close()
