#!/usr/bin/env python3
"""Generates tests/golden/oracle_cases.npz: per-iteration dumps (P1, PX, Np, sigma2, Y) of the CPU
oracle (oracle/ref_cpu.c, the restatement of trackdlo.cpp:161-441) on small seeded scenes covering
every branch of the path: plain, carried-over sigma2, visibility weighting (k_vis), correspondence
priors (alpha), both, the LLE regulariser with an explicit H, the prune with outliers, early
convergence, and a contrived folded-tip scene that triggers the end-node gap quirk
(trackdlo.cpp:313-350).

These fixtures pin the HIP path (and the oracle against accidental edits) to committed numbers.
They are outputs of OUR restatement, not of the reference build (which cannot be produced in this
image) -- see the "parity unpinned" note in oracle/ref_cpu.h.

Run: python tests/golden/make_oracle_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import ref_cpu as R            # noqa: E402
from trackdlo_amd import synth            # noqa: E402

P = synth.LAUNCH_PARAMS


def quirk_scene(M=12, N=300, seed=5):
    """Chain whose tip folds back so that, for points near node 0, node 2 is closer than node 1."""
    rng = np.random.default_rng(seed)
    Y = np.zeros((M, 3)); Y[:, 2] = 0.6
    Y[1] = (0.02, 0.02, 0.6); Y[2] = (0.012, 0.0, 0.6)
    for k in range(3, M):
        Y[k] = (0.012 + 0.02 * (k - 2), 0.0, 0.6)
    idx = rng.integers(2, M - 1, size=N); t = rng.random(N)
    X = (1 - t)[:, None] * Y[idx] + t[:, None] * Y[idx + 1] + rng.normal(0, 0.002, (N, 3))
    X[:40] = np.array([0.003, 0.0, 0.6]) + rng.normal(0, 0.001, (40, 3))
    return np.asfortranarray(X.astype(np.float32).astype(np.float64)), np.asfortranarray(Y)


def cases():
    N, M, it = 600, 20, 6
    base = dict(beta=P["beta"], lambda_=P["lambda_"], lle_weight=P["lle_weight"], mu=P["mu"], max_iter=it, tol=0.0,
                include_lle=False, alpha=0.0, k_vis=0.0, visibility_threshold=P["visibility_threshold"])
    X, Y0, _ = synth.scene(N, M, config=80)
    coord = synth.geodesic_coord(Y0)
    out = {}
    out["plain"] = dict(X=X, Y0=Y0, sigma2=0.0, kw=dict(base))
    out["sigma2_prev"] = dict(X=X, Y0=Y0, sigma2=1e-4, kw=dict(base))
    Xo, Yo, vis = synth.scene(N, M, config=81, occlude=(0.4, 0.6))
    vext = synth.extend_visible(vis, M, synth.geodesic_coord(Yo))
    out["vis"] = dict(X=Xo, Y0=Yo, sigma2=0.0, vis=vext, kw=dict(base, k_vis=P["k_vis"]))
    idx = np.arange(0, M, 3)
    pri = np.concatenate([idx[:, None].astype(float), Y0[idx] + np.array([0.0, 0.004, 0.001])], axis=1)
    out["priors"] = dict(X=X, Y0=Y0, sigma2=0.0, priors=pri, kw=dict(base, alpha=P["alpha"]))
    prio = np.concatenate([vext[::2, None].astype(float), Yo[vext[::2]] + np.array([0.0, 0.004, 0.0])], axis=1)
    out["vis_priors"] = dict(X=Xo, Y0=Yo, sigma2=2e-5, vis=vext, priors=prio, kw=dict(base, k_vis=P["k_vis"], alpha=P["alpha"]))
    L = R.calc_lle_weights(Y0, 6)
    H = (np.eye(M) - L).T @ (np.eye(M) - L)
    out["lle"] = dict(X=X, Y0=Y0, sigma2=0.0, H=H, kw=dict(base, include_lle=True, beta=P["beta_pre_proc"], lambda_=P["lambda_pre_proc"]))
    Xout, Yout, _ = synth.scene(N, M, config=82, outliers=37)
    out["outliers"] = dict(X=Xout, Y0=Yout, sigma2=0.0, kw=dict(base))
    out["tol"] = dict(X=X, Y0=Y0, sigma2=0.0, kw=dict(base, max_iter=50, tol=P["tol"]))
    Xq, Yq = quirk_scene()
    out["quirk"] = dict(X=Xq, Y0=Yq, sigma2=1e-5, kw=dict(base, max_iter=4))
    return out


def main():
    blob = {}
    for name, c in cases().items():
        o = R.cpd_lle(c["X"], c["Y0"], c["sigma2"], priors=c.get("priors"), visible_nodes=c.get("vis"), H=c.get("H"),
                      trace=True, **c["kw"])
        blob[f"{name}__X"] = c["X"]; blob[f"{name}__Y0"] = c["Y0"]; blob[f"{name}__sigma2_in"] = np.float64(c["sigma2"])
        for k in ("priors", "vis", "H"):
            if c.get(k) is not None:
                blob[f"{name}__{k}"] = np.asarray(c[k])
        for k, v in c["kw"].items():
            blob[f"{name}__kw_{k}"] = np.float64(v)
        blob[f"{name}__Y"] = o["Y"]; blob[f"{name}__sigma2"] = np.float64(o["sigma2"])
        blob[f"{name}__iters"] = np.int64(o["iters"]); blob[f"{name}__converged"] = np.int64(o["converged"])
        blob[f"{name}__n_kept"] = np.int64(o["n_kept"]); blob[f"{name}__gap_quirk"] = np.int64(o["gap_quirk"])
        for k, v in o["trace"].items():
            blob[f"{name}__trace_{k}"] = v
        print(name, "iters", o["iters"], "conv", o["converged"], "kept", o["n_kept"], "quirk", o["gap_quirk"], "sigma2", o["sigma2"])
    np.savez_compressed(os.path.join(HERE, "oracle_cases.npz"), **blob)


if __name__ == "__main__":
    main()
