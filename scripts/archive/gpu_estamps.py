"""E-step wave timeline at C2 from a -DTDLO_ESTEP_STAMPS build (scripts/tmp/libtrackdlo_stamps.so): shader clocks of workgroup 0, wave 0."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from trackdlo_amd import binding as B, synth
B.load_library(os.path.join(os.path.dirname(os.path.abspath(__file__)), "tmp", "libtrackdlo_stamps.so"))
B._lib = B.load_library(os.path.join(os.path.dirname(os.path.abspath(__file__)), "tmp", "libtrackdlo_stamps.so"))
P = synth.LAUNCH_PARAMS
ctx = B.Context(max_points=1 << 16)
X, Y0, _ = synth.scene(50000, 50, config=2)
pr = B.make_params(P['beta'], P['lambda_'], P['lle_weight'], P['mu'], 50, 0.0, False)
g = ctx.cpd_lle(X, Y0, 0.0, pr)
g = ctx.cpd_lle_resident(0, Y0, 0.0, pr)
st = ctx.debug_stamps(64).astype(np.int64)
print('loop_ms', g['loop_ms'])
print('estep block 0 wave 0 stamps (clocks): start, loads+barrier, pass1, second node+window, pass2, column sums, block barrier, end', (st[40:48] - st[40]).tolist())
print('   of pass1: candidate range known after', int(st[48] - st[41]), 'clocks, candidates loop', int(st[42] - st[48]))
