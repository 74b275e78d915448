import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import ref_cpu as R
from trackdlo_amd import binding as B, synth
ctx = B.Context(max_points=1 << 16, max_nodes=160)
for M in (64, 60, 50, 30, 70, 100):
    for lam in (1.0, 500.0, 50000.0):
        for beta in (0.35, 0.6, 3.0):
            X, Y0, _ = synth.scene(6811, M, config=77, frame=17, noise=0.002)
            kw = dict(beta=beta, lambda_=lam, lle_weight=10.0, mu=0.3, max_iter=3, tol=0.0, include_lle=False, alpha=0.0, k_vis=0.0, visibility_threshold=0.008)
            o = R.cpd_lle(X, Y0, 0.0, **kw)
            res = []
            for prec in (1, 0):
                g = ctx.cpd_lle(X, Y0, 0.0, B.make_params(precision=prec, **kw))
                res.append('%.1e' % np.abs(g['Y'] - o['Y']).max())
            print(M, lam, beta, 'f64/f32 dY', res, flush=True)
