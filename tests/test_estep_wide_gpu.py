"""The fp64 E-step of chains beyond 64 nodes, batches with a WIDE node window (csrc/tdlo_estep_wide.h, round 6): lane = node instead of thread = point.
Checked against the thread = point form (TDLO_ESTEP_WIDE=0) after every number of iterations, against the CPU oracle, with visibility weighting, with the
end-node gap of trackdlo.cpp:313-321 in the batch, at the widths where the form switches (128 / 129 / 320 / 321 nodes) and with a ragged last batch."""
import os
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _ctx(mode, N, M):
    from trackdlo_amd import binding as B
    if mode is not None:
        os.environ["TDLO_ESTEP_WIDE"] = str(mode)
    try:
        return B.Context(device=0, max_points=N, max_nodes=M, timing=False)
    finally:
        os.environ.pop("TDLO_ESTEP_WIDE", None)


@pytest.mark.parametrize("N,M,occ", [(20000, 300, None), (20000, 300, (0.3, 0.5)), (9973, 129, None), (9973, 128, None), (15000, 320, (0.6, 0.8)), (15000, 321, None),
                                     (15000, 500, None)],
                         ids=["300", "300vis", "129", "128", "320vis", "321", "500"])
def test_lane_per_node_form_agrees_with_thread_per_point_and_the_oracle(oracle, N, M, occ):
    from trackdlo_amd import binding as B, synth
    P = synth.LAUNCH_PARAMS
    X, Y0, vis = synth.scene(N, M, config=600 + M, occlude=occ)
    opt = {}
    if occ:
        opt["visible_nodes"] = np.asarray(synth.extend_visible(vis, M, synth.geodesic_coord(Y0)), dtype=np.int32)
    ctxs = {m: _ctx(m, N, M) for m in (0, None)}
    try:
        for iters in (1, 2, 4, 7, 30):
            pr = B.make_params(P["beta"], P["lambda_"], P["lle_weight"], P["mu"], iters, 0.0, False, P["alpha"], P["k_vis"] if occ else 0.0,
                               P["visibility_threshold"], B.PREC_F64)
            g = {m: c.cpd_lle(X, Y0, 0.0, pr, **opt) for m, c in ctxs.items()}
            a, b = g[0], g[None]
            assert a["rc"] == 0 and b["rc"] == 0 and a["iters"] == b["iters"] == iters
            # (the two forms add the same numbers in different orders: 1e-16 relative per sum, and sigma2 is a difference of sums a million times its size)
            assert np.abs(a["Y"] - b["Y"]).max() <= 1e-12 and abs(a["sigma2"] - b["sigma2"]) <= 1e-10 * a["sigma2"], (iters, np.abs(a["Y"] - b["Y"]).max())
        ref = oracle.cpd_lle(X, Y0, 0.0, max_iter=7, tol=0.0, include_lle=False, beta=P["beta"], lambda_=P["lambda_"], lle_weight=P["lle_weight"], mu=P["mu"],
                             alpha=P["alpha"], k_vis=P["k_vis"] if occ else 0.0, visibility_threshold=P["visibility_threshold"], **opt)
        pr = B.make_params(P["beta"], P["lambda_"], P["lle_weight"], P["mu"], 7, 0.0, False, P["alpha"], P["k_vis"] if occ else 0.0, P["visibility_threshold"], B.PREC_F64)
        b = ctxs[None].cpd_lle(X, Y0, 0.0, pr, **opt)
        assert np.abs(b["Y"] - ref["Y"]).max() <= 1e-9 and abs(b["sigma2"] - ref["sigma2"]) <= 1e-7 * ref["sigma2"]          # the mode's stated tolerance
    finally:
        for c in ctxs.values():
            c.close()


def test_end_node_gap_inside_a_wide_batch(oracle):
    """The chain's tips folded back (node 1 and node M - 2 pushed 9 cm sideways): for the points around node 0 the second node is node 2, not node 1
    (trackdlo.cpp:313-329), and the node between keeps the zero of :305, a membership of exp(0) (:332-350).  The oracle counts the (point, iteration) pairs
    that take the branch; the lane = node form reproduces geo_arg's zero there while the window is still the whole chain."""
    from trackdlo_amd import binding as B, synth
    P = synth.LAUNCH_PARAMS
    N, M = 6000, 200
    X, Y0, _ = synth.scene(N, M, config=777)
    rng = np.random.default_rng(5)
    X = np.asarray(X).copy(); Y0 = np.asarray(Y0).copy()
    Y0[1] += (0.0, 0.0, 0.09); Y0[M - 2] += (0.0, 0.0, 0.09)
    X[:400] = Y0[0] + rng.normal(0, 0.004, (400, 3))
    X[400:800] = Y0[M - 1] + rng.normal(0, 0.004, (400, 3))
    pr = B.make_params(P["beta"], P["lambda_"], P["lle_weight"], P["mu"], 5, 0.0, False, precision=B.PREC_F64)
    ref = oracle.cpd_lle(X, Y0, 0.0, max_iter=5, tol=0.0, include_lle=False, beta=P["beta"], lambda_=P["lambda_"], lle_weight=P["lle_weight"], mu=P["mu"])
    assert ref["gap_quirk"] > 1000
    for mode in (0, None):
        c = _ctx(mode, N, M)
        try:
            g = c.cpd_lle(X, Y0, 0.0, pr)
        finally:
            c.close()
        assert g["rc"] == 0 and np.abs(g["Y"] - ref["Y"]).max() <= 1e-9 and abs(g["sigma2"] - ref["sigma2"]) <= 1e-7 * ref["sigma2"], mode
