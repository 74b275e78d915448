"""E-step at large N (C4 shard sizes): one registration, then the E-step launched back to back -- meant to run under
rocprofv3 --pmc (scripts/gpu_estep_pmc.sh) so that the SQ counters of k_estep can be read per launch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from trackdlo_amd import binding as B, synth
P = synth.LAUNCH_PARAMS
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2000000
M = int(sys.argv[2]) if len(sys.argv) > 2 else 50
blocks = int(sys.argv[3]) if len(sys.argv) > 3 else 0
ctx = B.Context(max_points=N, max_nodes=M, estep_blocks=blocks)
X, Y0, _ = synth.scene(N, M, config=4)
pr = B.make_params(P['beta'], P['lambda_'], P['lle_weight'], P['mu'], 20, 0.0, False)
ctx.set_cloud(0, X)
g = ctx.cpd_lle_resident(0, Y0, 0.0, pr)
us = ctx.profile_kernel(0, 20)
print(f"N={N} M={M} blocks={blocks} loop {g['loop_ms']/20*1e3:.1f} us/iter  estep {us:.1f} us  ({12*N/us/1e3:.0f} GB/s algorithmic)", flush=True)
ctx.close()
