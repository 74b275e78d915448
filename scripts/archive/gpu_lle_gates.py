"""Calibration aid (GPU box): for the LLE draws of tests/test_parity_gpu.py::test_randomised_configurations, the oracle's own
error (QR vs quadruple-precision solve, oracle.extended_solver) beside the device's distance to the oracle in both precisions."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import ref_cpu as oracle
from trackdlo_amd import binding as B, synth

ctx = B.Context(device=0, max_frames=1, max_points=1 << 16, max_nodes=64)
def params(kw, prec):
    return B.make_params(kw["beta"], kw["lambda_"], kw["lle_weight"], kw["mu"], kw["max_iter"], kw["tol"], kw["include_lle"], kw["alpha"], kw["k_vis"], kw["visibility_threshold"], prec)
nseed = int(sys.argv[1]) if len(sys.argv) > 1 else 192
print("seed M N it kappa | own dY ds | f64 dY ds | f32 dY ds")
for seed in range(nseed):
    rng = np.random.default_rng(9000 + seed)
    M = int(rng.integers(4, 65)) if seed % 6 else int(rng.integers(65, 140))
    N = int(rng.integers(64, 12000))
    iters = int(rng.integers(1, 9))
    prec = int(rng.integers(0, 2))
    vis = bool(rng.integers(0, 2)) and M >= 12
    use_pri = bool(rng.integers(0, 2))
    use_lle = bool(rng.integers(0, 3) == 0)
    noise = float(rng.choice([0.0005, 0.002, 0.004]))
    X, Y0, v = synth.scene(N, M, config=60 + seed, frame=seed, noise=noise, occlude=(0.35, 0.55) if vis else None,
                           outliers=int(rng.integers(0, 20)), shift=(0.0, float(rng.uniform(0, 0.008)), float(rng.uniform(-0.003, 0.003))))
    vext = synth.extend_visible(v, M, synth.geodesic_coord(Y0)) if vis else None
    kw = dict(beta=float(rng.choice([0.35, 0.6, 3.0])), lambda_=float(rng.choice([1.0, 500.0, 50000.0])), lle_weight=10.0,
              mu=float(rng.choice([0.05, 0.1, 0.3])), max_iter=iters, tol=0.0, include_lle=False, alpha=0.0,
              k_vis=50.0 if vis else 0.0, visibility_threshold=0.008)
    pri = None; H = None
    if use_pri:
        idx = np.sort(rng.choice(M, size=max(1, M // 4), replace=False))
        pri = np.concatenate([idx[:, None].astype(float), Y0[idx] + rng.normal(0, 0.003, size=(len(idx), 3))], axis=1)
        kw["alpha"] = float(rng.choice([1.0, 3.0]))
    if not use_lle:
        continue
    L = oracle.calc_lle_weights(Y0, 6); H = (np.eye(M) - L).T @ (np.eye(M) - L)
    kw.update(include_lle=True, beta=3.0, lambda_=1.0)
    s2 = float(rng.choice([0.0, 1e-4, 2e-5]))
    o = oracle.cpd_lle(X, Y0, s2, priors=pri, visible_nodes=vext, H=H, **kw)
    with oracle.extended_solver():
        e = oracle.cpd_lle(X, Y0, s2, priors=pri, visible_nodes=vext, H=H, **kw)
    G = oracle.kernel_G(Y0, kw["beta"])[1]
    s2e = max(s2, 1.5e-5)
    kappa = np.linalg.cond(np.diag(np.full(M, o["n_kept"] / M)) @ G + kw["lambda_"] * s2e * np.eye(M) + s2e * kw["lle_weight"] * H @ G)
    out = [f"{seed:3d} {M:3d} {N:5d} {iters} {kappa:8.1e} | {np.abs(o['Y']-e['Y']).max():8.1e} {abs(o['sigma2']-e['sigma2'])/e['sigma2']:8.1e}"]
    for p in (1, 0):
        g = ctx.cpd_lle(X, Y0, s2, params(kw, p), priors=pri, visible_nodes=vext, H=H, check=False)
        out.append(f"| {np.abs(g['Y']-o['Y']).max():8.1e} {abs(g['sigma2']-o['sigma2'])/o['sigma2']:8.1e} vs-quad {np.abs(g['Y']-e['Y']).max():8.1e}" + ("" if g["iters"] == o["iters"] and g["rc"] == 0 else f" rc={g['rc']} it={g['iters']}/{o['iters']}"))
    print(" ".join(out), flush=True)
