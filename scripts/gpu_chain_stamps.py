"""Phase stamps (shader clocks) of the chain M-step for several chain lengths; TDLO_ALT_LIB=path loads another build of the library."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from trackdlo_amd import binding as B, synth
alt = os.environ.get("TDLO_ALT_LIB")
if alt:
    B.load_library(alt); B._lib = B.load_library(alt)
P = synth.LAUNCH_PARAMS
for M in (30, 50, 64, 100, 200, 300, 512):
    N = 20000
    ctx = B.Context(max_points=N, max_nodes=M)
    X, Y0, _ = synth.scene(N, M, config=2)
    pr = B.make_params(P['beta'], P['lambda_'], P['lle_weight'], P['mu'], 30, 0.0, False)
    g = ctx.cpd_lle(X, Y0, 0.0, pr)
    g = ctx.cpd_lle_resident(0, Y0, 0.0, pr)
    st = ctx.debug_stamps(64).astype(np.int64)
    d = st[:8] - st[0]
    print(f"M={M:4d} stamps {d.tolist()}  fetch {d[1]} rec {d[3]-d[1]} fwd {d[4]-d[3]} gains {d[5]-d[4]} bwd {d[6]-d[5]} final {d[7]-d[6]}  mstep_us {ctx.profile_kernel(2, 200):.2f}")
    ctx.close()
