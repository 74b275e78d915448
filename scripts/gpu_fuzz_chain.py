"""Sweep of the chain-smoother M-step against the oracle (no LLE term): random sizes, parameters, priors, visibility weighting, carried-over sigma2,
held to the STATED tolerances; a case outside them passes only when the oracle itself is measured to be that uncertain on it
(scripts/fuzz_adjudicate.py).  usage: python scripts/gpu_fuzz_chain.py [n_cases] [first_seed]   (FUZZ_PREC=0: fp32 mode)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from trackdlo_amd import binding as B, synth
from oracle import ref_cpu
import fuzz_adjudicate as FA
NRANGE = [int(v) for v in os.environ["FUZZ_N"].split(",")] if os.environ.get("FUZZ_N") else None      # e.g. FUZZ_N=1,150: tiny clouds


def run(n, s0=0, PREC=1, verbose=True):
    tally = FA.Tally(PREC)
    bad = degen = 0
    ctx = B.Context(device=0, max_points=1 << 14, max_nodes=512)
    for seed in range(s0, s0 + n):
        rng = np.random.default_rng(77000 + seed)
        M = int(rng.choice([rng.integers(4, 65), rng.integers(65, 200), rng.integers(200, 513)], p=[0.7, 0.2, 0.1]))
        N = int(rng.integers(200, 9000)); iters = int(rng.integers(1, 12))
        if NRANGE: N = int(rng.integers(NRANGE[0], NRANGE[1] + 1))
        vis = bool(rng.integers(0, 2)) and M >= 12
        use_pri = bool(rng.integers(0, 2))
        X, Y0, v = synth.scene(N, M, config=500 + seed, frame=seed, noise=float(rng.choice([0.0005, 0.002, 0.004])), occlude=(0.35, 0.55) if vis else None,
                               outliers=int(rng.integers(0, 20)), shift=(0.0, float(rng.uniform(0, 0.008)), float(rng.uniform(-0.003, 0.003))))
        vext = synth.extend_visible(v, M, synth.geodesic_coord(Y0)) if vis else None
        kw = dict(beta=float(rng.choice([0.1, 0.35, 0.6, 3.0])), lambda_=float(rng.choice([1.0, 500.0, 50000.0])), lle_weight=10.0,
                  mu=float(rng.choice([0.05, 0.1, 0.3])), max_iter=iters, tol=float(rng.choice([0.0, 2e-4])), include_lle=False, alpha=0.0,
                  k_vis=50.0 if vis else 0.0, visibility_threshold=0.008)
        pri = None
        if use_pri:
            idx = np.sort(rng.choice(M, size=max(1, M // 4), replace=False))
            pri = np.concatenate([idx[:, None].astype(float), Y0[idx] + rng.normal(0, 0.003, size=(len(idx), 3))], axis=1)
            kw["alpha"] = float(rng.choice([1.0, 3.0]))
        s2 = float(rng.choice([0.0, 1e-4, 2e-5]))
        if len(X) == 0: continue
        try:
            o = ref_cpu.cpd_lle(X, Y0, s2, priors=pri, visible_nodes=vext, **kw)
        except ValueError:          # the oracle gives up (every point pruned, ...): the product must report an error too
            o = None
        g = ctx.cpd_lle(X, Y0, s2, B.make_params(kw["beta"], kw["lambda_"], kw["lle_weight"], kw["mu"], kw["max_iter"], kw["tol"], False, kw["alpha"], kw["k_vis"], kw["visibility_threshold"], PREC),
                        priors=pri, visible_nodes=vext, check=False)
        if o is None:
            if g["rc"] == 0: bad += 1; print("ORACLE FAILED, PRODUCT DID NOT: seed", seed, "M", M, "N", N, flush=True)
            continue
        if not (o["sigma2"] > 1e-12 and np.all(np.isfinite(o["Y"]))):
            # the ORACLE's sigma2 collapsed to zero, went negative or not-a-number (clouds of a few points on chains of hundreds of nodes): the reference
            # has no guard and continues on garbage; the product ends such a registration with TDLO_E_NUMERIC or, when rounding keeps its own
            # sigma2 a hair above zero, carries on differently -- nothing to compare, it only must not crash
            degen += 1
            if g["rc"] not in (0, -5): bad += 1; print("DEGENERATE, rc", g["rc"], "seed", seed, flush=True)
            continue
        dy = float(np.abs(g["Y"] - o["Y"]).max()); ds = abs(g["sigma2"] - o["sigma2"]) / o["sigma2"]
        same = g["rc"] == 0 and g["iters"] == o["iters"] and g["n_kept"] == o["n_kept"]
        if not FA.judge(tally, (seed, M, N, kw["beta"], kw["lambda_"]), dy, ds, same, lambda: FA.cpd_uncertainty(ref_cpu, PREC, X, Y0, s2, kw, o, priors=pri, visible_nodes=vext)):
            bad += 1
            print("MISMATCH seed", seed, "M", M, "N", N, "iters", g["iters"], o["iters"], "rc", g["rc"], "dY %.2e ds %.2e" % (dy, ds), "sigma2 %.3e / %.3e" % (g["sigma2"], o["sigma2"]), kw, flush=True)
    ctx.close()
    if verbose:
        for line in tally.notes: print("   " + line)
        print(f"chain sweep, {n} cases from seed {s0} ({('fp32', 'fp64')[PREC]} mode), {degen} with a collapsed sigma2 in the oracle: {tally.summary()}; {bad - tally.unexplained} error mismatches")
    return dict(bad=bad, degenerate=degen, **tally.as_dict())


if __name__ == "__main__":
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 200, int(sys.argv[2]) if len(sys.argv) > 2 else 0, int(os.environ.get("FUZZ_PREC", "1")))
