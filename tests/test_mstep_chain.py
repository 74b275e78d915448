"""The M-step without the LLE term as a smoother along the chain (csrc/tdlo_mstep_chain.hip).

CPU: the formulation (tests/chain_numpy.py, the kernel's arithmetic in numpy) against an 80-bit dense solve of the reference's
system (trackdlo.cpp:405-417) -- it is the same linear system, and the O(M) recursion loses FEWER digits than partial-pivot
LU / least squares on the dense matrix.  GPU: the kernel against the oracle over the chain lengths that exercise every branch
of its slot bookkeeping, against the dense eliminations kept as comparators, with unobserved and coincident nodes."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import chain_numpy as cn  # noqa: E402


def _system(M, spacing, beta, lam, sigma2, seed, zero_frac, npts=50000, dup=False):
    rng = np.random.default_rng(seed)
    h = spacing * np.clip(1 + 0.3 * rng.standard_normal(M - 1), 0.2, None)
    if dup:
        h[M // 3] = 0.0                       # two coincident nodes: G is singular there, the system is not
    coord = np.concatenate([[0.0], np.cumsum(h)])
    p = (npts / M) * (0.5 + rng.random(M))
    p[rng.random(M) < zero_frac] = 0.0        # nodes without a single assigned point
    B = p[:, None] * 0.01 * rng.standard_normal((M, 3))
    return coord, p, B, lam * sigma2


@pytest.mark.parametrize("M,spacing", [(2, 0.1), (3, 0.1), (4, 0.1), (5, 0.1), (6, 0.08), (7, 0.05), (9, 0.05), (30, 0.02), (50, 0.012), (51, 0.001), (300, 0.003), (512, 0.002)])
def test_chain_formulation_against_80bit_dense_solve(M, spacing):
    worst_chain, worst_dense = 0.0, 0.0
    for sigma2 in (1e-2, 1e-5, 1e-8):
        for zf in (0.0, 0.4):
            coord, p, B, c = _system(M, spacing, 0.35, 50000.0, sigma2, 7, zf, dup=(M == 30))
            Gl = cn.kernel_G(coord, 0.35, np.longdouble)
            Tl = Gl @ cn.dense_solve_longdouble(p.astype(np.longdouble)[:, None] * Gl + np.longdouble(c) * np.eye(M, dtype=np.longdouble), B)
            V = cn.chain_solve(coord, 0.35, c, p, B)
            G = cn.kernel_G(coord, 0.35)
            Td = G @ np.linalg.solve(p[:, None] * G + c * np.eye(M), B)
            worst_chain = max(worst_chain, float(np.abs(V - Tl).max()))
            worst_dense = max(worst_dense, float(np.abs(Td - Tl).max()))
    # displacements are ~1e-2 m: the recursion stays at rounding level; the dense fp64 solve is what loses digits
    assert worst_chain <= 1e-13, worst_chain
    assert worst_chain <= 10 * worst_dense + 1e-16, (worst_chain, worst_dense)


def test_chain_link_small_gap_series():
    """Q of a link keeps full relative accuracy where 1 - e^-2x (1 + 2x + 2x^2) cancels (x = s h -> 0)."""
    beta = 0.35
    s, sf2 = np.sqrt(2.0) / beta, 1.0 / (2.0 * np.sqrt(2.0) * beta)
    for h in (1e-9, 1e-6, 1e-4, 1e-3, 1e-2, 0.1, 0.24, 0.3, 1.0):
        L = cn.chain_link(beta, h)
        x = np.longdouble(s) * np.longdouble(h)
        e2 = np.exp(-2 * x)
        # 80-bit evaluation of the same positive series
        t = 2 * x
        term, sm = t ** 3 / 6, np.longdouble(0)
        for n in range(3, 80):
            sm += term
            term = term * t / (n + 1)
        q11 = np.longdouble(sf2) * e2 * sm
        q22 = np.longdouble(sf2) * np.longdouble(s) ** 2 * e2 * (4 * x + sm)
        assert abs(L[4] - q11) <= 1e-15 * q11 and abs(L[6] - q22) <= 1e-15 * q22, h     # a few ulp
    assert cn.chain_link(beta, 0.0)[:4] == [1.0, 0.0, -0.0, 1.0] and cn.chain_link(beta, 0.0)[4:] == [0.0, 0.0, 0.0]


# ---------------------------------------------------------------------------------------------- GPU
def _kw(max_iter, **over):
    from trackdlo_amd import synth
    P = synth.LAUNCH_PARAMS
    kw = dict(beta=P["beta"], lambda_=P["lambda_"], lle_weight=P["lle_weight"], mu=P["mu"], max_iter=max_iter, tol=0.0, include_lle=False,
              alpha=0.0, k_vis=0.0, visibility_threshold=P["visibility_threshold"])
    kw.update(over)
    return kw


def _params(kw, prec):
    from trackdlo_amd import binding as B
    return B.make_params(kw["beta"], kw["lambda_"], kw["lle_weight"], kw["mu"], kw["max_iter"], kw["tol"], kw["include_lle"],
                         kw["alpha"], kw["k_vis"], kw["visibility_threshold"], prec)


@pytest.mark.gpu
@pytest.mark.parametrize("M", [4, 5, 6, 7, 8, 9, 10, 11, 13, 31, 50, 51, 52, 53, 64, 65, 127, 128, 129, 191, 192, 254, 255, 256, 300, 319, 320, 511, 512])
def test_chain_mstep_against_oracle_over_chain_lengths(oracle, M):
    """fp64 mode at the stated tolerance (1e-9 m, 1e-7 in sigma2; equal iteration counts): even / odd chains (a dummy first step in
    one direction or not), one or several step slots per thread (M + 1 > 256), partial rows fetched in one or several trips (elements of the sums per thread: 1 up to 63 nodes, 2 up to 127, 3 up to 191,
    4 up to 255, 5 up to 319 -- fetched in straight-line code -- and the element-by-element form beyond)."""
    from trackdlo_amd import binding as B, synth
    assert B.mstep_dense(False) is False
    N = 3000 if M <= 128 else 6000
    X, Y0, _ = synth.scene(N, M, config=300 + M)
    kw = _kw(4)
    ctx = B.Context(device=0, max_points=N, max_nodes=M)
    try:
        g = ctx.cpd_lle(X, Y0, 0.0, _params(kw, 1))
        assert ctx.profile_iteration(1)[3] == "k_mstep_chain"
    finally:
        ctx.close()
    o = oracle.cpd_lle(X, Y0, 0.0, **kw)
    assert g["rc"] == 0 and g["iters"] == o["iters"] == 4 and g["n_kept"] == o["n_kept"]
    assert np.abs(g["Y"] - o["Y"]).max() <= 1e-9 and abs(g["sigma2"] - o["sigma2"]) <= 1e-7 * o["sigma2"]


@pytest.mark.gpu
@pytest.mark.parametrize("prec", [0, 1], ids=["f32", "f64"])
def test_chain_mstep_against_dense_eliminations(prec):
    """The chain smoother and the dense eliminations (k_mstep_fast<MFMA> up to 60 nodes, k_mstep_mcu beyond) solve the same system
    from the same sums: trajectories agree to rounding (the E-step is common; stated 1e-11 m over 12 iterations in both modes),
    with priors (alpha J) and with the visibility weighting."""
    from trackdlo_amd import binding as B, synth
    P = synth.LAUNCH_PARAMS
    cases = [(20000, 50, False, None), (20000, 50, True, None), (9000, 30, False, (0.35, 0.6)), (6000, 90, True, None), (8000, 300, False, None),
             (5000, 61, False, (0.1, 0.3))]
    for i, (N, M, pri, occ) in enumerate(cases):
        X, Y0, vis = synth.scene(N, M, config=340 + i, occlude=occ)
        kw = _kw(12, alpha=P["alpha"] if pri else 0.0, k_vis=P["k_vis"] if occ else 0.0)
        opt = {}
        if pri:
            idx = np.arange(1, M, 5)
            opt["priors"] = np.column_stack([idx, Y0[idx] + 0.002])
        if occ:
            opt["visible_nodes"] = np.asarray(vis, dtype=np.int32)
        res = []
        for dense in (False, True):
            prev = B.mstep_dense(dense)
            try:
                ctx = B.Context(device=0, max_points=N, max_nodes=M)
                try:
                    res.append(ctx.cpd_lle(X, Y0, 0.0, _params(kw, prec), **opt))
                    name = ctx.profile_iteration(1)[3]
                    assert (name == "k_mstep_chain") != dense, name
                finally:
                    ctx.close()
            finally:
                B.mstep_dense(prev)
        a, b = res
        assert a["rc"] == 0 and b["rc"] == 0 and a["iters"] == b["iters"] and a["n_kept"] == b["n_kept"]
        assert np.abs(a["Y"] - b["Y"]).max() <= 1e-11, (i, np.abs(a["Y"] - b["Y"]).max())
        assert abs(a["sigma2"] - b["sigma2"]) <= 1e-9 * b["sigma2"]


@pytest.mark.gpu
def test_chain_mstep_unobserved_and_coincident_nodes(oracle):
    """Nodes no point is assigned to (P1 = 0: a pure prediction step of the filter) and two coincident nodes (a gap h = 0: an
    identity link; G itself is singular there, c I + D G is not) -- against the oracle, fp64."""
    from trackdlo_amd import binding as B, synth
    M, N = 40, 6000
    X, Y0, _ = synth.scene(N, M, config=361, occlude=(0.3, 0.55))          # a stretch of the chain without points
    Y1 = Y0.copy()
    Y1[13] = Y1[12]                                                          # coincident nodes
    for Yin in (Y0, np.asfortranarray(Y1)):
        kw = _kw(6)
        ctx = B.Context(device=0, max_points=N, max_nodes=M)
        try:
            g = ctx.cpd_lle(X, Yin, 0.0, _params(kw, 1))
        finally:
            ctx.close()
        o = oracle.cpd_lle(X, Yin, 0.0, **kw)
        assert g["rc"] == 0 and g["iters"] == o["iters"] and g["n_kept"] == o["n_kept"]
        assert np.abs(g["Y"] - o["Y"]).max() <= 1e-9 and abs(g["sigma2"] - o["sigma2"]) <= 1e-7 * o["sigma2"]
