"""The E-step's sums [P1 | R | Q | N] of the first iterations through the split interface: dump OUT.npz (for comparing two builds)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from trackdlo_amd import binding as B, synth, nsplit
P = synth.LAUNCH_PARAMS
N, M = 5000, 20
X, Y0, _ = synth.scene(N, M, config=2)
ctx = B.Context(max_points=N, max_nodes=M)
pr = B.make_params(P['beta'], P['lambda_'], P['lle_weight'], P['mu'], 5, 0.0, False)
sh = nsplit.HipShard(ctx, X)
init = sh.begin(Y0, 0.0, pr, None, None, None)
sh.set_global(init[0], init[1])
res = {}
for it in range(3):
    s = sh.estep(None)
    res[f"s{it}"] = s
    print(it, "P1", s[:5], "Rx", s[M:M + 3], "Q", s[4 * M], "N", s[4 * M + 1], flush=True)
    sh.mstep(s)
np.savez(sys.argv[1], **res)
