import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from trackdlo_amd import binding as B, synth
# the phase stamps exist only in an instrumented build: bash scripts/build_variant.sh stamps -DTDLO_ESTEP_STAMPS -DTDLO_CHAIN_STAMPS
_v = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tmp", "libtrackdlo_stamps.so")
if not os.environ.get("TDLO_LIBRARY") and os.path.exists(_v):
    B._lib = B.load_library(_v)
P = synth.LAUNCH_PARAMS
ctx = B.Context(max_points=1 << 16)
X, Y0, _ = synth.scene(50000, 50, config=2)
pr = B.make_params(P['beta'], P['lambda_'], P['lle_weight'], P['mu'], 50, 0.0, False)
g = ctx.cpd_lle(X, Y0, 0.0, pr)
g = ctx.cpd_lle_resident(0, Y0, 0.0, pr)
print('loop_ms', g['loop_ms'], 'total', g['total_ms'], 'host', g['host_ms'])
st = ctx.debug_stamps(64).astype(np.int64)
print('timeline (10 ns ticks rel. to previous M-step end): E first-block start %d, E last-block end %d, M start %d, M end %d' % (st[34]-st[36], st[35]-st[36], st[37]-st[36], st[38]-st[36]))
print('estep block 0 wave 0 stamps (cycles): start, loads+barrier, pass1, second node+window, pass2, column sums, block barrier, end', (st[40:48] - st[40]).tolist())
print('stamps', st[:8] - st[0])
print('wave load-done stamps', (st[8:12] - st[0]).tolist())
print('estep', ctx.profile_kernel(0, 300), 'mstep', ctx.profile_kernel(2, 300))
