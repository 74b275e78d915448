import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import ref_cpu as R
from trackdlo_amd import binding as B, synth
P = synth.LAUNCH_PARAMS
ctx = B.Context(max_points=1 << 16)
X, Y0, _ = synth.scene(50000, 50, config=1)
for s2in in (0.0, 1e-4):
    for it in (1, 2, 3, 5, 10, 20, 50):
        kw = dict(beta=P['beta'], lambda_=P['lambda_'], lle_weight=P['lle_weight'], mu=P['mu'], max_iter=it, tol=0.0, include_lle=False)
        o = R.cpd_lle(X, Y0, s2in, **kw)
        g = ctx.cpd_lle(X, Y0, s2in, B.make_params(precision=0, **kw))
        print(s2in, it, 'dY %.3e' % np.abs(g['Y'] - o['Y']).max(), 'ds2 %.3e' % (abs(g['sigma2'] - o['sigma2']) / o['sigma2']), 'sigma2 %.4e' % o['sigma2'])
