import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from trackdlo_amd import binding as B, synth
P = synth.LAUNCH_PARAMS
ctx = B.Context(max_points=1 << 16)
for cfg in (1, 2):
    X, Y0, _ = synth.scene(50000, 50, config=cfg)
    pr = B.make_params(P['beta'], P['lambda_'], P['lle_weight'], P['mu'], 1, 0.0, False, precision=0)
    for mode in ('1', '0'):
        os.environ['TDLO_NOSORT'] = mode
        ctx.cpd_lle(X, Y0, 0.0, pr)
        Xs, ctr = ctx.debug_read_cloud(60000)
        want = (X - ctr[None, :]).astype(np.float32).astype(np.float64)
        a = np.array(sorted(map(tuple, np.round(Xs, 9))))
        b = np.array(sorted(map(tuple, np.round(want, 9))))
        nd = (np.abs(a - b).max(axis=1) > 0).sum() if a.shape == b.shape else -1
        # nearest-node ordering check
        d = np.linalg.norm(Xs[:, None, :] - (Y0 - ctr)[None, :, :], axis=2).argmin(axis=1)
        print('cfg', cfg, 'NOSORT', mode, 'N', len(Xs), 'rows differing from expected multiset:', nd, 'buckets nondecreasing:', bool((np.diff(d) >= 0).all()), 'viol', int((np.diff(d) < 0).sum()))
