"""E-step back-to-back time at N = 2 000 000 (and a 32-frame batch) for TDLO_ESTEP_BLOCKS / tile-row variants (TDLO_LIBRARY)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from trackdlo_amd import binding as B, synth
P = synth.LAUNCH_PARAMS
tag = os.environ.get("TDLO_LIBRARY", "tree").split("libtrackdlo_")[-1] + " blocks=" + os.environ.get("TDLO_ESTEP_BLOCKS", "default")
pr = B.make_params(P['beta'], P['lambda_'], P['lle_weight'], P['mu'], 50, 0.0, False)
N = 2000000
ctx = B.Context(max_frames=1, max_points=N, max_nodes=50)
X, Y0, _ = synth.scene(N, 50, config=4)
ctx.set_cloud(0, X)
g = ctx.cpd_lle_resident(0, Y0, 0.0, pr)
g = ctx.cpd_lle_resident(0, Y0, 0.0, pr)
g = ctx.cpd_lle_resident(0, Y0, 0.0, pr)
print(f"[{tag}] N={N}: loop {g['loop_ms']:.3f} ms  b2b estep {ctx.profile_kernel(0, 200):.2f} us", flush=True)
