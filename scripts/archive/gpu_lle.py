import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import ref_cpu as R
from trackdlo_amd import binding as B, synth
ctx = B.Context(max_points=1 << 16, max_nodes=160)
for M in [int(x) for x in os.environ.get("MS", "46,47,48").split(",")]:
    for seed in (0, 1):
        X, Y0, _ = synth.scene(6000, M, config=5 + seed, frame=seed, noise=0.004)
        L = R.calc_lle_weights(Y0, 6); H = (np.eye(M) - L).T @ (np.eye(M) - L)
        kw = dict(beta=3.0, lambda_=1.0, lle_weight=10.0, mu=0.1, max_iter=1, tol=0.0, include_lle=True, alpha=0.0, k_vis=0.0, visibility_threshold=0.008)
        o = R.cpd_lle(X, Y0, 2e-5, H=H, **kw)
        res = []
        for prec in (1, 0):
            g = ctx.cpd_lle(X, Y0, 2e-5, B.make_params(precision=prec, **kw), H=H, check=False)
            res.append('%.1e' % np.abs(g['Y'] - o['Y']).max())
        print(M, seed, res, flush=True)
