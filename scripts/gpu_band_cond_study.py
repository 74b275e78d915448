"""Where does the banded LLE M-step (tdlo_mstep_band.hip) leave the fp64 mode's 1e-9 m gate?  The sweep's draws with long chains, the same registration once on the
banded elimination and once on the dense pivoted one (same E-step, same sums): |dY| between the two next to the quantities the state precision's conditioning
is made of (chain length M, beta, lambda, link length h).  usage: python scripts/gpu_band_cond_study.py [n_seeds] [first_seed] [min_M]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
from trackdlo_amd import binding as B
import gpu_fuzz_band as FB
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import band_numpy as bn
from numpy_shard import NumpyShard
import types


def main(n, s0, min_m):
    ctx = B.Context(device=0, max_points=1 << 14, max_nodes=512)
    rows = []
    for seed in range(s0, s0 + n):
        rng = np.random.default_rng(88000 + seed)
        M = int(rng.choice([rng.integers(4, 65), rng.integers(65, 200), rng.integers(200, 513)], p=[0.7, 0.2, 0.1]))
        if M < min_m: continue
        X, Y0, H, kw, pri, s2 = FB.draw(seed)
        if np.abs(H).max() > 1e7: continue
        g = ctx.cpd_lle(X, Y0, s2, FB.params(kw), priors=pri, H=H, check=False)
        name = ctx.profile_iteration(1)[3] if g["rc"] == 0 and g["iters"] > 0 else "-"
        prev = B.mstep_lle_dense(True)
        try:
            gd = ctx.cpd_lle(X, Y0, s2, FB.params(kw), priors=pri, H=H, check=False)
        finally:
            B.mstep_lle_dense(prev)
        if g["rc"] != 0 or gd["rc"] != 0: continue
        h = np.linalg.norm(np.diff(Y0, axis=0), axis=1)
        dy = float(np.abs(g["Y"] - gd["Y"]).max())
        # the estimate: entry rounding of lambda sigma2 K, times the displacement, over the weakest data term (first iteration: the largest sigma2)
        pp = types.SimpleNamespace(**kw); pp.k_vis = 0.0
        sh = NumpyShard(X); ns = sh.begin(Y0, s2, pp, pri, None, H); sh.set_global(ns[0], ns[1])
        P1 = sh.estep(None)[:M] + kw["alpha"] * sh.J
        dg, _ = bn.state_precision(sh.coord, kw["beta"])
        vmax = float(np.abs(gd["Y"] - Y0).max())
        pred = 2.2e-16 * kw["lambda_"] * sh.sigma2 * dg[:, 0].max() * vmax / P1.min()
        print("      sigma2_0 %.2e Kmax %.2e vmax %.2e P1min %.2e" % (sh.sigma2, dg[:, 0].max(), vmax, P1.min()))
        rows.append((dy, seed, M, kw["beta"], kw["lambda_"], kw["lle_weight"], h.mean(), h.min(), g["sigma2"], g["iters"], gd["iters"], name, pred))
        print("seed %4d M %3d beta %.0f lam %4.1f g %5.1f h %.2e hmin %.2e s2 %.2e it %d/%d %s  dY %.2e  pred %.2e  dY/pred %.2e" %
              (seed, M, kw["beta"], kw["lambda_"], kw["lle_weight"], h.mean(), h.min(), g["sigma2"], g["iters"], gd["iters"], name, dy, pred, dy / pred), flush=True)
    ctx.close()


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 1500, int(sys.argv[2]) if len(sys.argv) > 2 else 0, int(sys.argv[3]) if len(sys.argv) > 3 else 150)
