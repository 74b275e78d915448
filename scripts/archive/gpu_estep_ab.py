"""E-step A/B on one box: E-step kernel time (back to back and in situ) at C2 / one C3 batch / C4 for the tree's library or TDLO_LIBRARY."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from trackdlo_amd import binding as B, synth
P = synth.LAUNCH_PARAMS
tag = os.environ.get("TDLO_LIBRARY", "tree").split("libtrackdlo_")[-1]
pr = B.make_params(P['beta'], P['lambda_'], P['lle_weight'], P['mu'], 50, 0.0, False)
for N, F in ((50000, 1), (2000000, 1), (50000, 32)):
    ctx = B.Context(max_frames=F, max_points=N, max_nodes=50)
    Ys = []
    for f in range(F):
        X, Y0, _ = synth.scene(N, 50, config=2 if N == 50000 else 4, frame=f)
        ctx.set_cloud(f, X); Ys.append(Y0)
    if F == 1:
        g = ctx.cpd_lle_resident(0, Ys[0], 0.0, pr); g = ctx.cpd_lle_resident(0, Ys[0], 0.0, pr)
        lm = g['loop_ms']
    else:
        g = ctx.cpd_lle_batch(Ys, [0.0] * F, pr); g = ctx.cpd_lle_batch(Ys, [0.0] * F, pr)
        lm = g['stats'][0]['loop_ms']
    e, m, it, name = ctx.profile_iteration(200)
    print(f"[{tag}] N={N} F={F}: loop {lm:.3f} ms ({lm/50*1e3:.2f} us/iter)  in-situ estep {e:.2f} us mstep {m:.2f} us iter {it:.2f} us  b2b estep {ctx.profile_kernel(0, 200):.2f} us", flush=True)
    ctx.close()
