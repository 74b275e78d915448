"""CPU suite, part 1: the oracle (oracle/ref_cpu.c) against

  * golden vectors produced by IMPORTING the reference's numpy prototype utils/tracking_test.py
    (tests/golden/proto_*.npz, generator tests/golden/make_golden.py) -- pins the Euclidean E-step,
    reductions, M-step (with and without the LLE term), T, the sigma2 update and the prototype's
    geodesic membership;
  * its own committed per-iteration dumps for the C++-only branches (oracle_cases.npz);
  * closed forms / analytic cases for the kernel G, line-sphere intersection and LLE structure.
"""
import numpy as np
import pytest

from conftest import GOLDEN, case_kwargs, load_cases

PROTO_MODES = {"euclid_a": (1, 1), "euclid_b": (1, 1), "geo_a": (2, 2), "geo_b": (2, 2)}


@pytest.mark.parametrize("name", sorted(PROTO_MODES))
def test_oracle_matches_prototype_cpd(oracle, name):
    g = np.load(f"{GOLDEN}/proto_cpd.npz")
    kernel, e_mode = PROTO_MODES[name]
    kw = dict(beta=float(g[f"{name}__beta"]), lambda_=float(g[f"{name}__alpha"]), lle_weight=float(g[f"{name}__gamma"]),
              mu=float(g[f"{name}__mu"]), tol=0.0, include_lle=False, kernel=kernel, e_mode=e_mode, no_prune=1, conv_rule=1,
              den_guard=1, lle_extended=1)
    for i, mi in enumerate(g["iters"]):
        o = oracle.cpd_lle(g["X"], g["Y0"], 0.0, max_iter=int(mi), **kw)
        assert o["iters"] == mi
        np.testing.assert_allclose(o["Y"], g[f"{name}__Y"][i], rtol=0, atol=5e-11)
        assert abs(o["sigma2"] - g[f"{name}__sigma2"][i]) <= 1e-9 * g[f"{name}__sigma2"][i]
    o = oracle.cpd_lle(g["X"], g["Y0"], 2.5e-5, max_iter=5, **kw)          # use_prev_sigma2=True
    np.testing.assert_allclose(o["Y"], g[f"{name}__Y_prev"], rtol=0, atol=5e-11)
    assert abs(o["sigma2"] - g[f"{name}__sigma2_prev"]) <= 1e-9 * g[f"{name}__sigma2_prev"]


def test_oracle_matches_prototype_lle_mstep(oracle):
    g = np.load(f"{GOLDEN}/proto_lle_mstep.npz")
    for i, mi in enumerate(g["iters"]):
        o = oracle.cpd_lle(g["X"], g["Y0"], 0.0, beta=float(g["beta"]), lambda_=float(g["alpha"]), lle_weight=float(g["gamma"]),
                           mu=float(g["mu"]), max_iter=int(mi), tol=0.0, include_lle=True, H=g["H"], kernel=1, e_mode=1, no_prune=1,
                           conv_rule=1, den_guard=1)
        np.testing.assert_allclose(o["Y"], g["Y"][i], rtol=0, atol=1e-9)
        assert abs(o["sigma2"] - g["sigma2"][i]) <= 1e-9 * g["sigma2"][i]


def test_oracle_lle_weights_vs_prototype(oracle):
    g = np.load(f"{GOLDEN}/proto_lle_weights.npz")
    Y0 = g["Y0"]; M = Y0.shape[0]
    # 2-neighbour weights are well conditioned: exact comparison (extended end neighbourhoods = prototype)
    np.testing.assert_allclose(oracle.calc_lle_weights(Y0, 2, extended=True), g["W_k2"], rtol=0, atol=1e-9)
    # interior rows agree between the truncated (C++) and extended (prototype) variants
    Wt = oracle.calc_lle_weights(Y0, 2, extended=False)
    np.testing.assert_allclose(Wt[1:M - 1], g["W_k2"][1:M - 1], rtol=0, atol=1e-9)
    # k = 6: sparsity pattern of the prototype's index sets; rows sum to one
    W6e = oracle.calc_lle_weights(Y0, 6, extended=True)
    assert ((W6e != 0) <= (g["nbr6_mask"] != 0)).all()
    np.testing.assert_allclose(W6e.sum(axis=1), 1.0, atol=1e-6)
    np.testing.assert_allclose(g["W6_rowsum"], 1.0, atol=1e-6)
    # C++ variant: truncated +-3 neighbourhoods (trackdlo.cpp:92-117)
    W6 = oracle.calc_lle_weights(Y0, 6, extended=False)
    for i in range(M):
        nz = set(np.nonzero(W6[i])[0].tolist())
        lo, hi = (0, i + 3) if i - 3 < 0 else ((i - 3, M - 1) if i + 3 >= M else (i - 3, i + 3))
        assert nz <= set(range(lo, hi + 1)) - {i}
    np.testing.assert_allclose(W6.sum(axis=1), 1.0, atol=1e-6)


def test_oracle_self_consistency_with_committed_cases(oracle):
    """The committed dumps must be reproduced exactly by the oracle as built now."""
    for name, c in load_cases().items():
        o = oracle.cpd_lle(c["X"], c["Y0"], float(c["sigma2_in"]), priors=c.get("priors"), visible_nodes=c.get("vis"),
                           H=c.get("H"), trace=True, **case_kwargs(c))
        assert o["iters"] == int(c["iters"]) and o["converged"] == bool(c["converged"]) and o["n_kept"] == int(c["n_kept"])
        assert o["gap_quirk"] == int(c["gap_quirk"])
        np.testing.assert_allclose(o["Y"], c["Y"], rtol=0, atol=1e-12)
        np.testing.assert_allclose(o["trace"]["sigma2"], c["trace_sigma2"], rtol=1e-10)
        np.testing.assert_allclose(o["trace"]["P1"], c["trace_P1"], rtol=0, atol=1e-9)


def test_gap_quirk_case_fires(oracle):
    c = load_cases()["quirk"]
    assert int(c["gap_quirk"]) > 0          # the folded-tip scene exercises trackdlo.cpp:313-321 + :335-350


def test_kernel_G_closed_form(oracle):
    from trackdlo_amd import synth
    beta = 0.35
    Y = synth.nodes(50)
    coord, G = oracle.kernel_G(Y, beta, 0)
    np.testing.assert_allclose(coord, synth.geodesic_coord(Y), rtol=0, atol=1e-15)
    np.testing.assert_allclose(np.diag(G), np.sqrt(2.0) / (4 * beta), rtol=1e-14)     # G_ii = sqrt(2)/(4 beta) = 1.0102
    assert abs(G[0, 0] - 1.0101525445522108) < 1e-12
    np.testing.assert_allclose(G, G.T, rtol=0, atol=0)
    d = abs(coord[3] - coord[10])
    assert abs(G[3, 10] - np.exp(-np.sqrt(2) * d / beta) * (2 * d + np.sqrt(2) * beta) / (4 * beta ** 2)) < 1e-15
    w = np.linalg.eigvalsh(G)
    assert w.min() > 0                                                               # PSD (Matern-3/2 type)


def test_line_sphere_analytic(oracle):
    lsi = oracle.line_sphere_intersection
    # segment along x through a sphere centred at the origin: two hits at +-r when both lie on the segment
    h = lsi([-1, 0, 0], [1, 0, 0], [0, 0, 0], 0.5)
    assert h.shape == (2, 3)
    np.testing.assert_allclose(sorted(h[:, 0]), [-0.5, 0.5], atol=1e-15)
    # only the far hit lies on the segment
    h = lsi([0, 0, 0], [1, 0, 0], [0, 0, 0], 0.25)
    np.testing.assert_allclose(h, [[0.25, 0, 0]], atol=1e-15)
    # miss
    assert lsi([0, 1, 0], [1, 1, 0], [0, 0, 0], 0.5).shape == (0, 3)
    # hit beyond the end of the segment but inside the 0.1 mm slack of isBetween (utils.cpp:172-183)
    h = lsi([0, 0, 0], [1, 0, 0], [0, 0, 0], 1.00005)
    assert h.shape == (1, 3)
    assert lsi([0, 0, 0], [1, 0, 0], [0, 0, 0], 1.001).shape == (0, 3)
    # tangent line: delta == 0 exactly -> single solution
    h = lsi([-1, 0.5, 0], [1, 0.5, 0], [0, 0, 0], 0.5)
    assert h.shape[0] in (1, 2)
    np.testing.assert_allclose(h[0], [0, 0.5, 0], atol=1e-7)


def test_traverse_euclidean_respaces_at_original_arc_lengths(oracle):
    from trackdlo_amd import synth
    M = 30
    Y = synth.nodes(M)
    coord = synth.geodesic_coord(Y)
    vis = np.arange(M)
    for alignment in (0, 1):
        pairs = oracle.traverse_euclidean(coord, Y, vis, alignment)
        assert len(pairs) == M
        idx = pairs[:, 0].astype(int)
        assert (idx == (np.arange(M) if alignment == 0 else np.arange(M)[::-1])).all()
        # re-spacing an unmoved chain at its own arc lengths returns the chain itself
        np.testing.assert_allclose(pairs[:, 1:], Y[idx], atol=1e-6)
    # head-only visibility stops at the last consecutive visible node
    vis = np.arange(12)
    pairs = oracle.traverse_euclidean(coord, Y[vis], vis, 0)
    assert len(pairs) == 12 and pairs[-1, 0] == 11


def test_solver_against_numpy(oracle):
    rng = np.random.default_rng(3)
    for n in (4, 17, 50):
        A = rng.normal(size=(n, n)) + n * np.eye(n)
        Bm = rng.normal(size=(n, 3))
        np.testing.assert_allclose(oracle.solve_qrcp(A, Bm), np.linalg.solve(A, Bm), rtol=1e-10, atol=1e-12)


def test_oracle_tracking_step_runs_all_occlusion_states(oracle):
    from trackdlo_amd import synth
    P = synth.LAUNCH_PARAMS
    M = 30
    Y0 = synth.nodes(M); coord = synth.geodesic_coord(Y0)
    for occl in (None, (0.45, 0.5), (0.35, 0.65), (0.0, 0.3), (0.7, 1.0)):
        X, _, vis = synth.scene(1500, M, config=70, occlude=occl)
        vis = np.arange(M) if vis is None else vis
        vext = synth.extend_visible(vis, M, coord)
        t = oracle.Tracker(M, P["visibility_threshold"], P["beta"], P["lambda_"], P["alpha"], P["k_vis"], P["mu"], 30, P["tol"],
                           P["beta_pre_proc"], P["lambda_pre_proc"], P["lle_weight"])
        t.initialize_nodes(Y0); t.initialize_geodesic_coord(coord)
        t.tracking_step(X, vis, vext)
        Y = t.get_tracking_result()
        assert np.isfinite(Y).all() and t.get_sigma2() > 0
        assert np.abs(Y - Y0).max() < 0.03
        K = t.get_correspondence_pairs()
        assert len(K) >= 1 and (K[:, 0] >= 0).all() and (K[:, 0] < M).all()


def test_visibility_prepass_oracle(oracle):
    from trackdlo_amd import synth
    M = 30
    X, Y0, vis_true = synth.scene(4000, M, config=31, occlude=(0.4, 0.6))
    coord = synth.geodesic_coord(Y0)
    d, vis, ext = oracle.visibility_prepass(X, Y0, 0.008, 0.06, coord)
    dn = np.linalg.norm(X[:, None, :] - Y0[None, :, :], axis=2).min(axis=0)
    np.testing.assert_allclose(d, dn, rtol=0, atol=1e-14)
    np.testing.assert_array_equal(vis, np.nonzero(dn <= 0.008)[0])
    np.testing.assert_array_equal(ext, synth.extend_visible(vis, M, coord, 0.06))
    assert set(vis.tolist()) <= set(ext.tolist())


def test_reg_matches_prototype_register(oracle):
    """G6: `reg` (utils.cpp:21-82) restated; its prototype-mode switch reproduces the reference's numpy `register`
    (tracking_test.py:118-172) on golden vectors generated by importing that prototype (make_golden.py)."""
    z = np.load(f"{GOLDEN}/proto_register.npz")
    for mu in (0.05, 0.0):
        for k, it in enumerate(z["iters"]):
            Y, s2 = oracle.reg(z["X"], int(z["M"]), mu=mu, max_iter=int(it), proto=True)
            np.testing.assert_allclose(Y, z[f"mu{mu}__Y"][k], rtol=0, atol=1e-12)
            assert abs(s2 - z[f"mu{mu}__sigma2"][k]) <= 1e-12 * s2
    # the C++ variant differs only in the start (y axis, data-driven sigma2) and the iteration count
    Y, s2 = oracle.reg(z["X"], 8, mu=0.05, max_iter=0)
    assert np.array_equal(Y[:, 1], 0.1 / 8 * np.arange(8)) and np.all(Y[:, [0, 2]] == 0)
    d2 = ((Y[:, None, :] - z["X"][None, :, :]) ** 2).sum()
    assert abs(s2 - d2 / (3 * 8 * len(z["X"]))) <= 1e-13 * s2


def test_openmp_build_of_the_oracle_agrees_with_the_serial_one():
    """libref_cpu_omp.so (bench.py's all-cores timing column only, never the checker) spreads the points over threads; its
    sums over points are per-thread partial sums, so it agrees with the serial restatement to rounding, not bit for bit."""
    import numpy as np
    from oracle import ref_cpu
    from trackdlo_amd import synth
    P = synth.LAUNCH_PARAMS
    X, Y0, v = synth.scene(4000, 30, config=1, occlude=(0.4, 0.6))
    vext = synth.extend_visible(v, 30, synth.geodesic_coord(Y0))
    kw = dict(beta=P["beta"], lambda_=P["lambda_"], lle_weight=P["lle_weight"], mu=P["mu"], max_iter=8, tol=0.0, include_lle=False,
              alpha=0.0, k_vis=P["k_vis"], visibility_threshold=P["visibility_threshold"], visible_nodes=vext)
    a = ref_cpu.cpd_lle(X, Y0, 0.0, **kw)
    ref_cpu.set_threads(3)
    b = ref_cpu.cpd_lle(X, Y0, 0.0, all_cores=True, **kw)
    assert a["iters"] == b["iters"] and a["n_kept"] == b["n_kept"] and a["gap_quirk"] == b["gap_quirk"]
    assert np.abs(a["Y"] - b["Y"]).max() <= 1e-11 and abs(a["sigma2"] - b["sigma2"]) <= 1e-10 * a["sigma2"]
