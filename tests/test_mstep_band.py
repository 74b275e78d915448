"""The M-step WITH the LLE term as a banded L D L^T in the chain's state (csrc/tdlo_mstep_band.hip).

CPU: the formulation (tests/band_numpy.py, the kernel's arithmetic and data flow in numpy) against an 80-bit dense solve of the
reference's system (trackdlo.cpp:396-417) with the REAL ill-conditioned H of trackdlo.cpp:236-237 -- it is the same linear system,
and the banded elimination loses fewer digits than partial-pivot LU of the dense matrix; where it stops being usable (consecutive
nodes closer than about a millimetre) and that prepare_frame's bound keeps such chains away from it.
GPU: the kernel against the oracle over the chain lengths that exercise its chunking, against the dense pivoted eliminations kept as
comparators, in batches, and the chains / H matrices it must hand to the dense kernels."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import band_numpy as bn  # noqa: E402
import chain_numpy as cn  # noqa: E402


def _lle_H(oracle, Y0):
    M = len(Y0)
    L = oracle.calc_lle_weights(np.asfortranarray(Y0))
    IL = np.eye(M) - L
    return IL.T @ IL


def _system(oracle, M, sigma2, seed, jitter=0.0, npts=5000, beta=3.0, lam=1.0, gamma=10.0, gap=None):
    from trackdlo_amd import synth
    rng = np.random.default_rng(seed)
    Y0 = synth.nodes(M).copy()
    if jitter:
        Y0 += jitter * rng.standard_normal(Y0.shape)
    if gap is not None:                              # one short gap in the middle of the chain
        i = M // 3
        d = Y0[i + 1] - Y0[i]
        Y0[i + 1:] -= d * (1 - gap / np.linalg.norm(d))
    H = _lle_H(oracle, Y0)
    coord = synth.geodesic_coord(Y0)
    p = (npts / M) * (0.5 + rng.random(M))
    B = p[:, None] * 0.005 * rng.standard_normal((M, 3)) - sigma2 * gamma * (H @ Y0)
    return coord, H, p, B, lam * sigma2, gamma * sigma2


@pytest.mark.parametrize("M", [4, 5, 7, 8, 20, 45, 50, 128, 300])
def test_band_formulation_against_80bit_dense_solve(oracle, M):
    worst_band, worst_dense = 0.0, 0.0
    for sigma2 in (1e-3, 1e-5, 1e-7):
        for jit in (0.0, 0.002):                     # jittered nodes: LLE neighbourhoods far from collinear, H entries up to 1e6
            coord, H, p, B, c, g = _system(oracle, M, sigma2, 3, jit)
            assert bn.h_is_banded(H)                 # the reference's own H reaches +-6 nodes, exactly
            Tl = bn.dense_reference(coord, 3.0, c, p, g, H, B)
            V, _ = bn.band_solve(coord, 3.0, c, p, g, H, B)
            G = cn.kernel_G(coord, 3.0)
            Td = G @ np.linalg.solve((np.diag(p) + g * H) @ G + c * np.eye(M), B)
            worst_band = max(worst_band, float(np.abs(V - Tl).max()))
            worst_dense = max(worst_dense, float(np.abs(Td - Tl).max()))
    # displacements are ~1e-2 m: the banded elimination stays within a few hundred ulp; the dense fp64 solve is what loses digits
    assert worst_band <= 2e-13, worst_band
    assert worst_band <= worst_dense, (worst_band, worst_dense)


@pytest.mark.parametrize("M", [4, 6, 7, 13, 20, 45, 128])
def test_tile_data_flow_is_the_band_elimination(oracle, M):
    """The kernel's own data structures -- column records divided by sigma2, circular 13-slot window in a 16 x 16 tile, right-hand
    sides entering one step late through the spare k-slot, step records, column-oriented back substitution -- reproduce the plain
    banded L D L^T to rounding."""
    sigma2 = 1e-5
    coord, H, p, B, c, g = _system(oracle, M, sigma2, 5, 0.002)
    rec = bn.build_records(coord, 3.0, 1.0, 10.0, bn.lle_band(H, M))
    x = bn.tile_solve(rec, p, B, sigma2, 2 * M)
    V, _ = bn.band_solve(coord, 3.0, c, p, g, H, B)
    assert np.abs(x[0::2] - V).max() <= 1e-15


@pytest.mark.parametrize("M", [19, 20, 30, 45, 50, 128, 300, 448, 512])
def test_twisted_plan_and_its_arithmetic(oracle, M):
    """BandPlan (csrc/tdlo_internal.h), restated: both directions eliminate whole chunks of 13 unknowns, the 12 unknowns between them are
    exactly the separator, the dummy unknowns of the tail direction number fewer than 13, the records fit the LDS (else one direction) -- and
    eliminating from both ends with the Schur complements merged on the separator IS the banded solve."""
    bp = bn.band_plan(M)
    nU = 2 * M
    if bp["tw"]:
        assert bp["mT"] % 13 == 0 and bp["mB"] % 13 == 0 and 0 <= bp["D"] < 13
        assert bp["mT"] + 12 + bp["mB"] == nU + bp["D"] == bp["nUp"] and abs(bp["cT"] - bp["cB"]) <= 1
        assert bp["lds_bytes"] <= 160 * 1024
    else:
        assert bp["cB"] == 0 and bp["limT"] == nU and bp["sT"] >= nU
    assert (M >= 19) == bool(bp["tw"]) or M > 448          # two directions from 19 nodes on, one again where the records would not fit twice
    if M > 128:
        return
    sigma2 = 1e-5
    coord, H, p, B, c, g = _system(oracle, M, sigma2, 9, 0.002)
    A, R = bn.assemble(coord, 3.0, c, p, g, bn.lle_band(H, M), B)
    X1, _ = bn.band_ldlt_solve(A, R)
    X2 = bn.twisted_solve(A, R, bp["mT"] if bp["tw"] else 2 * M - 12)
    assert np.abs(X1[0::2] - X2[0::2]).max() <= 1e-14


def test_short_gaps_bound_of_prepare_frame(oracle):
    """K contains Q^-1 ~ 1 / h^3: the banded form degrades as two nodes approach each other.  At the bound prepare_frame applies
    (1 mm with beta 3, lambda 1) it is still at 1e-12 m; one decade below it is not better than the dense solve any more."""
    for sigma2 in (1e-2, 1e-3, 1e-6):
        coord, H, p, B, c, g = _system(oracle, 45, sigma2, 5, gap=1e-3)
        Tl = bn.dense_reference(coord, 3.0, c, p, g, H, B)
        V, _ = bn.band_solve(coord, 3.0, c, p, g, H, B)
        assert np.abs(V - Tl).max() <= 3e-12, (sigma2, np.abs(V - Tl).max())
    coord, H, p, B, c, g = _system(oracle, 45, 1e-3, 5, gap=1e-5)
    assert np.abs(bn.band_solve(coord, 3.0, c, p, g, H, B)[0] - bn.dense_reference(coord, 3.0, c, p, g, H, B)).max() > 1e-9


# ---------------------------------------------------------------------------------------------- GPU
def _kw(max_iter, **over):
    from trackdlo_amd import synth
    P = synth.LAUNCH_PARAMS
    kw = dict(beta=P["beta_pre_proc"], lambda_=P["lambda_pre_proc"], lle_weight=P["lle_weight"], mu=P["mu"], max_iter=max_iter, tol=0.0,
              include_lle=True, alpha=0.0, k_vis=0.0, visibility_threshold=P["visibility_threshold"])
    kw.update(over)
    return kw


def _params(kw, prec):
    from trackdlo_amd import binding as B
    return B.make_params(kw["beta"], kw["lambda_"], kw["lle_weight"], kw["mu"], kw["max_iter"], kw["tol"], kw["include_lle"],
                         kw["alpha"], kw["k_vis"], kw["visibility_threshold"], prec)


@pytest.mark.gpu
@pytest.mark.parametrize("M", [4, 5, 6, 7, 8, 12, 13, 14, 19, 20, 26, 45, 50, 64, 65, 100, 128, 129, 200, 256, 257, 300, 448, 470, 477, 481, 482, 511, 512])
def test_band_mstep_against_oracle_over_chain_lengths(oracle, M):
    """fp64 mode at the stated tolerance (1e-9 m, 1e-7 in sigma2; equal iteration counts), the oracle's own H of :236-237 injected on both
    sides (the LLE weights themselves are not reproducible to the last digits: rank-3 Gram matrices): 2M unknowns in whole and partial chunks
    of 13, fewer unknowns than one window (M <= 6), one or several nodes per thread (M > 256), the largest chain the LDS holds."""
    from trackdlo_amd import binding as B, synth
    assert B.mstep_lle_dense(False) is False
    N = 3000 if M <= 128 else 6000
    X, Y0, _ = synth.scene(N, M, config=400 + M, noise=0.004)
    H = _lle_H(oracle, Y0)
    kw = _kw(4)
    ctx = B.Context(device=0, max_points=N, max_nodes=M)
    try:
        g = ctx.cpd_lle(X, Y0, 2e-5, _params(kw, 1), H=H)
        assert ctx.profile_iteration(1)[3] == "k_mstep_band"
    finally:
        ctx.close()
    o = oracle.cpd_lle(X, Y0, 2e-5, H=H, **kw)
    assert g["rc"] == 0 and g["iters"] == o["iters"] == 4 and g["n_kept"] == o["n_kept"]
    assert np.abs(g["Y"] - o["Y"]).max() <= 1e-9 and abs(g["sigma2"] - o["sigma2"]) <= 1e-7 * o["sigma2"]


@pytest.mark.gpu
@pytest.mark.parametrize("prec", [0, 1], ids=["f32", "f64"])
def test_band_mstep_against_dense_pivoted_eliminations(oracle, prec):
    """The banded L D L^T and the dense pivoted eliminations (k_mstep_fast<pivoted> up to 64 nodes, k_mstep up to 128, k_mstep_pivot_mcu
    beyond) solve the same system from the same sums: trajectories agree to the dense solves' own rounding over 10 iterations, with
    priors (alpha J) and from sigma2 = 0 (first frame)."""
    from trackdlo_amd import binding as B, synth
    cases = [(6000, 45, False, 2e-5), (6000, 50, True, 0.0), (5000, 30, False, 0.0), (6000, 100, False, 1e-4), (8000, 200, True, 2e-5)]
    for i, (N, M, pri, s2) in enumerate(cases):
        X, Y0, _ = synth.scene(N, M, config=440 + i, noise=0.003)
        H = _lle_H(oracle, Y0)
        kw = _kw(10, alpha=3.0 if pri else 0.0)
        opt = dict(H=H)
        if pri:
            idx = np.arange(1, M, 5)
            opt["priors"] = np.column_stack([idx, Y0[idx] + 0.002])
        res = []
        for dense in (False, True):
            prev = B.mstep_lle_dense(dense)
            try:
                ctx = B.Context(device=0, max_points=N, max_nodes=M)
                try:
                    res.append(ctx.cpd_lle(X, Y0, s2, _params(kw, prec), **opt))
                    name = ctx.profile_iteration(1)[3]
                    assert (name == "k_mstep_band") != dense, name
                finally:
                    ctx.close()
            finally:
                B.mstep_lle_dense(prev)
        a, b = res
        assert a["rc"] == 0 and b["rc"] == 0 and a["iters"] == b["iters"] and a["n_kept"] == b["n_kept"]
        assert np.abs(a["Y"] - b["Y"]).max() <= 2e-10, (i, np.abs(a["Y"] - b["Y"]).max())
        assert abs(a["sigma2"] - b["sigma2"]) <= 1e-8 * b["sigma2"]


@pytest.mark.gpu
def test_band_mstep_own_lle_weights_and_f32_mode(oracle):
    """Without an injected H the library forms H = (I - L)^T (I - L) itself (host, trackdlo.cpp:92-159, :236-237): the default path of
    tracking_step's pre-processing registration.  fp32 mode at its stated tolerance (1e-5 m, 1e-3)."""
    from trackdlo_amd import binding as B, synth
    X, Y0, _ = synth.scene(5000, 45, config=470, noise=0.003)
    kw = _kw(8)
    ctx = B.Context(device=0, max_points=5000, max_nodes=45)
    try:
        g = ctx.cpd_lle(X, Y0, 2e-5, _params(kw, 0))
        assert ctx.profile_iteration(1)[3] == "k_mstep_band"
    finally:
        ctx.close()
    o = oracle.cpd_lle(X, Y0, 2e-5, **kw)
    assert g["rc"] == 0 and g["iters"] == o["iters"] == 8 and g["n_kept"] == o["n_kept"]
    assert np.abs(g["Y"] - o["Y"]).max() <= 1e-5 and abs(g["sigma2"] - o["sigma2"]) <= 1e-3 * o["sigma2"]


@pytest.mark.gpu
def test_chains_and_regularisers_the_banded_solve_hands_to_the_dense_kernels(oracle):
    """Coincident nodes (K infinite), nodes half a millimetre apart, and an H_override that is not banded: prepare_frame keeps the dense
    pivoted eliminations, and the results are the oracle's."""
    from trackdlo_amd import binding as B, synth
    M, N = 40, 5000
    X, Y0, _ = synth.scene(N, M, config=480, noise=0.003)
    Ydup = Y0.copy(); Ydup[13] = Ydup[12]
    Ynear = Y0.copy(); Ynear[13] = Ynear[12] + (Ynear[13] - Ynear[12]) * 0.02
    Hfull = np.eye(M) * 0.1 + 0.01 * np.diag(np.ones(M - 1), 1) + 0.01 * np.diag(np.ones(M - 1), -1)
    Hfull[0, M - 1] = Hfull[M - 1, 0] = 0.001                               # one entry outside the band
    for Yin, H in ((np.asfortranarray(Ydup), _lle_H(oracle, Y0)), (np.asfortranarray(Ynear), _lle_H(oracle, Y0)), (Y0, Hfull)):
        kw = _kw(4)
        ctx = B.Context(device=0, max_points=N, max_nodes=M)
        try:
            g = ctx.cpd_lle(X, Yin, 2e-5, _params(kw, 1), H=H)
            assert ctx.profile_iteration(1)[3] == "k_mstep_fast<pivoted>"
        finally:
            ctx.close()
        o = oracle.cpd_lle(X, Yin, 2e-5, H=H, **kw)
        assert g["rc"] == 0 and g["iters"] == o["iters"] and g["n_kept"] == o["n_kept"]
        assert np.abs(g["Y"] - o["Y"]).max() <= 1e-9 and abs(g["sigma2"] - o["sigma2"]) <= 1e-7 * o["sigma2"]


@pytest.mark.gpu
def test_band_mstep_batch_equals_single(oracle):
    """A batch of registrations with the LLE term: one workgroup per frame, bit for bit the single calls."""
    from trackdlo_amd import binding as B, synth
    F, M, N = 8, 45, 4000
    ctx = B.Context(device=0, max_frames=F, max_points=N, max_nodes=M)
    try:
        kw = _kw(6)
        Ys, single = [], []
        for fr in range(F):
            X, Y0, _ = synth.scene(N, M, config=490, frame=fr, noise=0.003)
            ctx.set_cloud(fr, X)
            Ys.append(Y0)
        for fr in range(F):
            single.append(ctx.cpd_lle_resident(fr, Ys[fr], 2e-5, _params(kw, 0)))
        out = ctx.cpd_lle_batch(Ys, [2e-5] * F, _params(kw, 0))
        assert ctx.profile_iteration(1)[3] == "k_mstep_band"
        for fr in range(F):
            assert np.array_equal(np.asarray(out["Y"][fr]), single[fr]["Y"]) and out["stats"][fr]["sigma2"] == single[fr]["sigma2"]
    finally:
        ctx.close()


@pytest.mark.gpu
def test_band_mstep_hands_an_indefinite_system_to_the_dense_kernels(oracle):
    """The banded L D L^T takes no pivots.  An H_override that is banded and symmetric but NOT positive semi-definite (here
    (I - L)^T (I - L) - I with lle_weight 1e7: eigenvalues down to -110, condition number ~100) makes the M-step system indefinite: the reference's general solver
    (trackdlo.cpp:415) still solves it, so the call is repeated once on the dense pivoted kernels and the caller gets their result --
    the oracle's, not an error."""
    from trackdlo_amd import binding as B, synth
    assert B.mstep_lle_dense(False) is False
    M, N = 30, 3000
    X, Y0, _ = synth.scene(N, M, config=480, noise=0.003)
    L = np.zeros((M, M))
    for i in range(M):
        nb = [j for j in range(i - 3, i + 4) if j != i and 0 <= j < M]
        L[i, nb] = 1.0 / len(nb)
    IL = np.eye(M) - L
    kw = _kw(1, lle_weight=1e7)
    ctx = B.Context(device=0, max_points=N, max_nodes=M)
    try:
        # positive semi-definite: the banded solve serves it, no retry
        H = IL.T @ IL
        g = ctx.cpd_lle(X, Y0, 2e-5, _params(kw, 1), H=H)
        o = oracle.cpd_lle(X, Y0, 2e-5, H=H, **kw)
        assert g["rc"] == 0 and ctx.band_retries() == 0
        assert np.abs(g["Y"] - o["Y"]).max() <= 1e-9
        # indefinite: one retry, the dense kernels' result
        H = IL.T @ IL - np.eye(M)
        g = ctx.cpd_lle(X, Y0, 2e-5, _params(kw, 1), H=H)
        o = oracle.cpd_lle(X, Y0, 2e-5, H=H, **kw)
        assert ctx.band_retries() == 1
        assert g["rc"] == 0 and g["iters"] == o["iters"] == 1
        move = np.abs(o["Y"] - Y0).max()
        assert move > 0.01                                   # (the indefinite system throws the nodes far: that is the reference's behaviour too)
        assert np.abs(g["Y"] - o["Y"]).max() <= 1e-8 and abs(g["sigma2"] - o["sigma2"]) <= 1e-6 * o["sigma2"]
        assert g["band_retry"] == 1                          # (round 4: the call itself says so, not only the context's counter)
        # and the context goes back to the banded solve afterwards
        H = IL.T @ IL
        g = ctx.cpd_lle(X, Y0, 2e-5, _params(kw, 1), H=H)
        assert g["rc"] == 0 and g["band_retry"] == 0 and ctx.band_retries() == 1 and ctx.profile_iteration(1)[3] == "k_mstep_band"
        # the N-split driver repeats the call on the dense kernels as well -- every rank solves the same system, so every rank meets
        # the same pivot -- in the one-shot form (one rank here) and in the RCCL form (a one-rank communicator made by the library)
        H = IL.T @ IL - np.eye(M)
        plain = ctx.cpd_lle(X, Y0, 2e-5, _params(kw, 1), H=H)
        ctx.xch_bind(0, [ctx.xch_create(1, 64)])
        s1 = ctx.split_run(Y0, 2e-5, _params(kw, 1), H=H)
        ctx.xch_unbind()
        assert s1["rc"] == 0 and s1["band_retry"] == 1 and s1["iters"] == plain["iters"]
        np.testing.assert_array_equal(s1["Y"], plain["Y"])
        import os
        os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
        comm = ctx.rccl_comm_init(1, 0, B.rccl_unique_id())
        s2 = ctx.split_run(Y0, 2e-5, _params(kw, 1), comm=comm, H=H)
        assert s2["rc"] == 0 and s2["band_retry"] == 1 and np.abs(s2["Y"] - plain["Y"]).max() <= 1e-12
        assert ctx.band_retries() == 4
    finally:
        ctx.close()


@pytest.mark.gpu
def test_band_mstep_leaves_a_large_sigma2_to_the_dense_kernels_in_fp64_mode(oracle):
    """Round 5's band sweep: five of 1 500 draws 1.4 .. 5.5e-9 m from the oracle -- chains of 394 .. 508 nodes (8 .. 10 m), beta = 5, started from sigma2 = 0,
    i.e. from sigma2 = 3 .. 6 m2.  The state precision's entries (~ 3 beta^4 / h^3), rounded to fp64, do not resolve the mode's 1e-9 m there whatever solves the
    band (scripts/gpu_band_cond_study.py); the dense system does.  prepare_frame's bound (FrameDev::band_s2_max) hands such a registration over: the
    M-step that meets a sigma2 above it ends the call like a bad pivot and the call is repeated on the dense kernels; a sigma2 GIVEN above it goes there
    directly; fp32 mode (1e-5 m) keeps the banded solve."""
    import importlib.util
    from trackdlo_amd import binding as B
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "scripts"))
    spec = importlib.util.spec_from_file_location("gpu_fuzz_band", os.path.join(root, "scripts", "gpu_fuzz_band.py"))
    FB = importlib.util.module_from_spec(spec); spec.loader.exec_module(FB)
    X, Y0, H, kw, pri, s2 = FB.draw(1425)                      # 404 nodes, beta 5, lambda 10, priors, sigma2 from the data (3.7 m2)
    assert s2 == 0.0 and len(Y0) == 404 and kw["beta"] == 5.0
    o = oracle.cpd_lle(X, Y0, 0.0, priors=pri, H=H, **kw)
    ctx = B.Context(device=0, max_points=1 << 14, max_nodes=512)
    try:
        p64 = B.make_params(kw["beta"], kw["lambda_"], kw["lle_weight"], kw["mu"], kw["max_iter"], kw["tol"], True, kw["alpha"], 0.0, kw["visibility_threshold"], 1)
        p32 = B.make_params(kw["beta"], kw["lambda_"], kw["lle_weight"], kw["mu"], kw["max_iter"], kw["tol"], True, kw["alpha"], 0.0, kw["visibility_threshold"], 0)
        r0 = ctx.band_retries()
        g = ctx.cpd_lle(X, Y0, 0.0, p64, priors=pri, H=H)
        assert ctx.band_retries() == r0 + 1 and g["band_retry"] == 1
        assert g["rc"] == 0 and g["iters"] == o["iters"]
        assert np.abs(g["Y"] - o["Y"]).max() <= 1e-9 and abs(g["sigma2"] - o["sigma2"]) <= 1e-7 * o["sigma2"]
        # a sigma2 given above the bound: the dense kernels without a first attempt
        g2 = ctx.cpd_lle(X, Y0, 3.0, p64, priors=pri, H=H)
        o2 = oracle.cpd_lle(X, Y0, 3.0, priors=pri, H=H, **kw)
        assert ctx.band_retries() == r0 + 1 and ctx.profile_iteration(1)[3] != "k_mstep_band"
        assert np.abs(g2["Y"] - o2["Y"]).max() <= 1e-9
        # the reference's own scale of sigma2 on the same chain: the banded solve, inside the gate
        g3 = ctx.cpd_lle(X, Y0, 1e-4, p64, priors=pri, H=H)
        o3 = oracle.cpd_lle(X, Y0, 1e-4, priors=pri, H=H, **kw)
        assert ctx.band_retries() == r0 + 1 and ctx.profile_iteration(1)[3] == "k_mstep_band"
        assert np.abs(g3["Y"] - o3["Y"]).max() <= 1e-9
        # fp32 mode: no bound
        g4 = ctx.cpd_lle(X, Y0, 0.0, p32, priors=pri, H=H)
        assert ctx.band_retries() == r0 + 1 and g4["rc"] == 0 and np.abs(g4["Y"] - o["Y"]).max() <= 1e-5
    finally:
        ctx.close()


@pytest.mark.gpu
def test_band_sigma2_bound_in_a_batch(oracle):
    """The same hand-over inside a batch (fp64 mode): one frame of three is GIVEN a sigma2 above prepare_frame's bound, so the batch as a whole takes the dense
    pivoted kernels (a batch runs one M-step kernel); a second batch leaves sigma2 to the device on every frame -- the first banded M-step meets 3.7 m2 and the
    whole batch is repeated on the dense kernels.  Every frame inside the mode's gate against the oracle either way."""
    import importlib.util
    from trackdlo_amd import binding as B
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "scripts"))
    spec = importlib.util.spec_from_file_location("gpu_fuzz_band", os.path.join(root, "scripts", "gpu_fuzz_band.py"))
    FB = importlib.util.module_from_spec(spec); spec.loader.exec_module(FB)
    X, Y0, H, kw, pri, _ = FB.draw(1425)
    kw = dict(kw, alpha=0.0)                              # (a batch shares its priors: none here)
    M = len(Y0)
    p64 = B.make_params(kw["beta"], kw["lambda_"], kw["lle_weight"], kw["mu"], kw["max_iter"], kw["tol"], True, 0.0, 0.0, kw["visibility_threshold"], 1)
    ctx = B.Context(device=0, max_frames=3, max_points=len(X), max_nodes=M)
    try:
        for fr in range(3):
            ctx.set_cloud(fr, X)
        for s2s in ([1e-4, 3.0, 2e-4], [0.0, 0.0, 0.0]):
            r0 = ctx.band_retries()
            out = ctx.cpd_lle_batch([Y0, Y0, Y0], s2s, p64, H=H)
            assert ctx.profile_iteration(1)[3] != "k_mstep_band"
            assert ctx.band_retries() == r0 + (1 if s2s[0] == 0.0 else 0)
            for fr in range(3):
                o = oracle.cpd_lle(X, Y0, s2s[fr], H=H, **kw)
                st = out["stats"][fr]
                assert st["status"] == 0 and st["iters"] == o["iters"]
                assert np.abs(np.asarray(out["Y"][fr]) - o["Y"]).max() <= 1e-9 and abs(st["sigma2"] - o["sigma2"]) <= 1e-7 * o["sigma2"]
    finally:
        ctx.close()
