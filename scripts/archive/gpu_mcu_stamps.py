"""Real-time-clock stamps (100 MHz) of k_mstep_mcu; needs a library built with EXTRA=-DTDLO_MCU_STAMPS."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from trackdlo_amd import binding as B, synth
P = synth.LAUNCH_PARAMS
N, M = int(os.environ.get('N', 200000)), int(os.environ.get('M', 300))
ctx = B.Context(max_points=N, max_nodes=M)
X, Y0, _ = synth.scene(N, M, config=5)
pr = B.make_params(P['beta'], P['lambda_'], P['lle_weight'], P['mu'], 10, 0.0, False, precision=B.PREC_F64)
ctx.set_cloud(0, X)
for _ in range(3): g = ctx.cpd_lle_resident(0, Y0, 0.0, pr)
st = ctx.debug_stamps(64).astype(np.int64)
nrb = (M + 15) // 16
us = lambda a: np.round((a - st[0]) / 100.0, 2).tolist()
print('start, sums, assembled, last arriver, end (us):', us(st[[0, 1, 2, 3, 4]]), 'first round, all rounds of thread 0:', us(st[[5, 6]]))
print('owner starts :', us(st[8:8 + nrb]))
print('owner reduced:', us(st[28:28 + nrb]))
print('reduce us    :', np.round((st[28:28 + nrb] - st[8:8 + nrb]) / 100.0, 2).tolist())
print('reduced -> next owner starts us:', np.round((st[9:8 + nrb] - st[28:27 + nrb]) / 100.0, 2).tolist())
for pb, o in ((1, 48), (8, 56)):
    print(f'panel {pb}: reduced -> write-back barrier -> flag stored -> next owner sees flag -> next owner starts (us rel. reduced):',
          np.round((np.array([st[o], st[o + 1], st[o + 2], st[8 + pb + 1]]) - st[28 + pb]) / 100.0, 2).tolist())
