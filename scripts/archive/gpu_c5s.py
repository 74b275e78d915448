import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from trackdlo_amd import binding as B, synth
P = synth.LAUNCH_PARAMS
N, M = 200000, 300
ctx = B.Context(max_points=N, max_nodes=M)
X, Y0, _ = synth.scene(N, M, config=5)
pr = B.make_params(P['beta'], P['lambda_'], P['lle_weight'], P['mu'], 5, 0.0, False, precision=B.PREC_F32)
ctx.set_cloud(0, X)
g = ctx.cpd_lle_resident(0, Y0, 0.0, pr)
st = ctx.debug_stamps(64).astype(np.int64)
print('phase b per panel', st[16:36].tolist())
print('stamps', (st[:6] - st[0]).tolist(), 'phase a total', st[8], 'phase b total', st[9])
print('mstep_us', ctx.profile_kernel(2, 5))
