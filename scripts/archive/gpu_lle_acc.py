"""Accuracy of the LLE M-step kernels for M > 128 against the oracle (run plain and with TDLO_MSTEP_LLE=1wg)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import ref_cpu as R
from trackdlo_amd import binding as B, synth
for M, N in ((129, 6000), (200, 8000), (300, 8000)):
    for variant in ("artificial", "lle"):
        X, Y0, _ = synth.scene(N, M, config=5, noise=0.004)
        if variant == "artificial":
            H = np.eye(M) * 0.1 + 0.01 * np.diag(np.ones(M - 1), 1) + 0.01 * np.diag(np.ones(M - 1), -1)
        else:
            L = R.calc_lle_weights(Y0, 6); H = (np.eye(M) - L).T @ (np.eye(M) - L)
        kw = dict(beta=3.0, lambda_=1.0, lle_weight=10.0, mu=0.1, max_iter=3, tol=0.0, include_lle=True, alpha=0.0, k_vis=0.0, visibility_threshold=0.008)
        o = R.cpd_lle(X, Y0, 2e-5, H=H, **kw)
        ctx = B.Context(max_points=1 << 16, max_nodes=M)
        g = ctx.cpd_lle(X, Y0, 2e-5, B.make_params(precision=1, **kw), H=H, check=False)
        print(f"M={M} H={variant}: status={g['status']} max|dY| vs oracle = {np.abs(g['Y'] - o['Y']).max():.2e}  dsigma2 rel = {abs(g['sigma2'] - o['sigma2']) / o['sigma2']:.1e}  cond(H)={np.linalg.cond(H):.1e}", flush=True)
        ctx.close()
