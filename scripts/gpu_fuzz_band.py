"""Sweep of the banded LLE M-step against the oracle (include_lle, the oracle's own H injected on both sides): random chain lengths, sizes,
parameters, priors, jittered and unevenly spaced nodes, carried-over sigma2, held to the STATED tolerances; a case outside them passes only when
the oracle itself is measured to be that uncertain on it (scripts/fuzz_adjudicate.py).  Also reported: the cases the gap test sent to the
dense kernels and the repeats after a non-positive pivot.  usage: python scripts/gpu_fuzz_band.py [n_cases] [first_seed]   (FUZZ_PREC=0: fp32 mode)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from trackdlo_amd import binding as B, synth
from oracle import ref_cpu
import fuzz_adjudicate as FA


PREC = int(os.environ.get("FUZZ_PREC", "1"))          # 1: fp64 mode (gate 1e-9 m, 1e-7); 0: fp32 mode (1e-5 m, 1e-3)
GY, GS = ((1e-5, 1e-3), (1e-9, 1e-7))[PREC]
NRANGE = [int(v) for v in os.environ["FUZZ_N"].split(",")] if os.environ.get("FUZZ_N") else None      # e.g. FUZZ_N=1,150: tiny clouds


def draw(seed):
    """The case of one seed: (X, Y0, H, kw, priors, sigma2)."""
    rng = np.random.default_rng(88000 + seed)
    M = int(rng.choice([rng.integers(4, 65), rng.integers(65, 200), rng.integers(200, 513)], p=[0.7, 0.2, 0.1]))
    N = int(rng.integers(300, 9000)); iters = int(rng.integers(1, 9))
    if NRANGE: N = int(rng.integers(NRANGE[0], NRANGE[1] + 1))
    use_pri = bool(rng.integers(0, 2))
    X, Y0, _ = synth.scene(N, M, config=600 + seed, frame=seed, noise=float(rng.choice([0.0005, 0.002, 0.004])),
                           outliers=int(rng.integers(0, 20)), shift=(0.0, float(rng.uniform(0, 0.006)), float(rng.uniform(-0.003, 0.003))))
    Y0 = Y0 + rng.normal(0, float(rng.choice([0.0, 0.0005, 0.002])), size=Y0.shape)         # nodes off the smooth centreline: LLE weights far from collinear
    L = ref_cpu.calc_lle_weights(np.asfortranarray(Y0))
    IL = np.eye(M) - L
    H = IL.T @ IL
    kw = dict(beta=float(rng.choice([1.0, 3.0, 5.0])), lambda_=float(rng.choice([0.1, 1.0, 10.0])), lle_weight=float(rng.choice([1.0, 10.0, 100.0])),
              mu=float(rng.choice([0.05, 0.1, 0.3])), max_iter=iters, tol=float(rng.choice([0.0, 2e-4])), include_lle=True, alpha=0.0,
              k_vis=0.0, visibility_threshold=0.008)
    pri = None
    if use_pri:
        idx = np.sort(rng.choice(M, size=max(1, M // 4), replace=False))
        pri = np.concatenate([idx[:, None].astype(float), Y0[idx] + rng.normal(0, 0.003, size=(len(idx), 3))], axis=1)
        kw["alpha"] = float(rng.choice([1.0, 3.0]))
    s2 = float(rng.choice([0.0, 1e-4, 2e-5]))
    return X, Y0, H, kw, pri, s2


def params(kw):
    return B.make_params(kw["beta"], kw["lambda_"], kw["lle_weight"], kw["mu"], kw["max_iter"], kw["tol"], True, kw["alpha"], 0.0, kw["visibility_threshold"], PREC)


def run(n, s0=0, verbose=True):
    tally = FA.Tally(PREC)
    bad = dense = degen = 0
    ctx = B.Context(device=0, max_points=1 << 14, max_nodes=512)
    for seed in range(s0, s0 + n):
        X, Y0, H, kw, pri, s2 = draw(seed)
        M, N = len(Y0), len(X)
        try:
            o = ref_cpu.cpd_lle(X, Y0, s2, priors=pri, H=H, **kw)
        except ValueError:           # the oracle gives up (every point pruned, ...): the product must report an error too
            o = None
        r0 = ctx.band_retries()
        g = ctx.cpd_lle(X, Y0, s2, params(kw), priors=pri, H=H, check=False)
        if o is None:
            if g["rc"] == 0: bad += 1; print("ORACLE FAILED, PRODUCT DID NOT: seed", seed, "M", M, "N", N, flush=True)
            continue
        if not (o["sigma2"] > 1e-12 and np.all(np.isfinite(o["Y"]))):      # the oracle's sigma2 collapsed (see gpu_fuzz_chain.py): nothing to compare
            degen += 1
            if g["rc"] not in (0, -5): bad += 1; print("DEGENERATE, rc", g["rc"], "seed", seed, flush=True)
            continue
        if g["rc"] != 0: print("   rc", g["rc"], ctx.lib.tdlo_last_error(ctx.h).decode(), flush=True)
        name = ctx.profile_iteration(1)[3] if g["rc"] == 0 and g["iters"] > 0 else "-"
        dense += name != "k_mstep_band"
        dy = float(np.abs(g["Y"] - o["Y"]).max()); ds = abs(g["sigma2"] - o["sigma2"]) / o["sigma2"]
        same = g["rc"] == 0 and g["iters"] == o["iters"] and g["n_kept"] == o["n_kept"]
        ok = FA.judge(tally, (seed, M, N, kw["beta"], kw["lambda_"], kw["lle_weight"]), dy, ds, same, lambda: FA.cpd_uncertainty(ref_cpu, PREC, X, Y0, s2, kw, o, priors=pri, H=H))
        if not ok:
            # whose deviation is it?  the same registration on the dense pivoted eliminations (same sums, same E-step): if THAT agrees with the oracle
            # the banded elimination's own rounding is what is left (seen on chains of ~400 nodes and more with beta = 5: profiles/r05_fuzz.log)
            prev = B.mstep_lle_dense(True)
            try:
                gd = ctx.cpd_lle(X, Y0, s2, params(kw), priors=pri, H=H, check=False)
                print(f"   seed {seed}: the dense pivoted solve on the same sums is {float(np.abs(gd['Y'] - o['Y']).max()):.2e} m from the oracle (banded: {dy:.2e} m)", flush=True)
            finally:
                B.mstep_lle_dense(prev)
        if not ok or ctx.band_retries() != r0:
            bad += not ok
            print("MISMATCH" if not ok else "REPEAT", "seed", seed, "M", M, "N", N, "iters", g["iters"], o["iters"], "rc", g["rc"], name, "dY %.2e ds %.2e" % (dy, ds), "sigma2 %.3e / %.3e" % (g["sigma2"], o["sigma2"]),
                  "H max %.1e" % np.abs(H).max(), {k: kw[k] for k in ("beta", "lambda_", "lle_weight", "alpha", "tol")}, flush=True)
    retries = ctx.band_retries()
    ctx.close()
    if verbose:
        for line in tally.notes: print("   " + line)
        print(f"band sweep, {n} cases from seed {s0} ({('fp32', 'fp64')[PREC]} mode), {degen} with a collapsed sigma2 in the oracle, {dense} on the dense kernels, {retries} repeats: {tally.summary()}; "
              f"{bad - tally.unexplained} error mismatches")
    return dict(bad=bad, degenerate=degen, dense=dense, retries=retries, **tally.as_dict())


if __name__ == "__main__":
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 200, int(sys.argv[2]) if len(sys.argv) > 2 else 0)
