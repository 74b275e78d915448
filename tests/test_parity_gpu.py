"""GPU suite: the HIP path (through the C ABI, include/trackdlo_hip.h) against the CPU oracle.

Stated tolerances, after an equal, fixed number of EM iterations:
  TDLO_PREC_F32 (fp32 E-step, fp64 M-step):  max |dY| <= 1e-5 m,  |d sigma2| / sigma2 <= 1e-3
  TDLO_PREC_F64:                              max |dY| <= 1e-9 m,  |d sigma2| / sigma2 <= 1e-7
Iteration counts, the converged flag and the kept-point count must match exactly.
"""
import os

import numpy as np
import pytest

# TDLO_SWEEP_SCALE=k multiplies the number of seeds of the randomised sweeps (bug hunting; the committed default is 1)
_SWEEP = int(__import__("os").environ.get("TDLO_SWEEP_SCALE", "1"))
_SEQ_EXITS = {"compared": 0, "oracle_undecided": 0, "f32_iteration_count": 0}      # how the frames of the random sequences ended (see the summary test)

from conftest import case_kwargs, load_cases

pytestmark = pytest.mark.gpu

TOL = {0: (1e-5, 1e-3), 1: (1e-9, 1e-7)}


def _params(kw, prec):
    from trackdlo_amd import binding as B
    return B.make_params(kw["beta"], kw["lambda_"], kw["lle_weight"], kw["mu"], kw["max_iter"], kw["tol"], kw["include_lle"],
                         kw["alpha"], kw["k_vis"], kw["visibility_threshold"], prec)


def _measured_gate(oracle, prec, run, H, o):
    """Gate for a registration with the (ill-conditioned) LLE term: the stated tolerance, widened ONLY by what the oracle itself is
    measured to be uncertain by on this very input -- 8 x the larger of (a) its own rounding error: the faithful QR solve
    against the quadruple-precision solve of the same systems, (b) its sensitivity to the last bit of the data: H perturbed
    by +-1 ulp per entry (the reference forms H, H G and the sums in another order than any re-implementation can).
    On all but the pathological draws both are ~1e-12 m and the stated tolerance stands (scripts/archive/gpu_lle_gates.py)."""
    ty, ts = TOL[prec]
    with oracle.extended_solver():
        e = run(H)
    dy = np.abs(e["Y"] - o["Y"]).max(); ds = abs(e["sigma2"] - o["sigma2"]) / o["sigma2"]
    rng = np.random.default_rng(4242)
    for _ in range(2):
        Hp = np.asarray(H) * (1.0 + 2.220446049250313e-16 * rng.choice([-1.0, 1.0], size=np.shape(H)))
        p = run(Hp)
        if p["iters"] != o["iters"]:
            return np.inf, np.inf                 # the last bit of H moves the stopping decision: nothing can be compared
        dy = max(dy, np.abs(p["Y"] - o["Y"]).max()); ds = max(ds, abs(p["sigma2"] - o["sigma2"]) / o["sigma2"])
    return max(ty, 8.0 * dy), max(ts, 8.0 * ds)


def _check(g, o, prec):
    ty, ts = TOL[prec]
    assert g["rc"] == 0
    assert g["iters"] == o["iters"] and g["converged"] == o["converged"] and g["n_kept"] == o["n_kept"]
    assert np.abs(g["Y"] - o["Y"]).max() <= ty
    assert abs(g["sigma2"] - o["sigma2"]) <= ts * o["sigma2"]


CASES = sorted(load_cases())


@pytest.mark.parametrize("prec", [0, 1], ids=["f32", "f64"])
@pytest.mark.parametrize("name", CASES)
def test_committed_golden_cases(hip_ctx, name, prec):
    """HIP vs the committed per-branch fixtures (tests/golden/oracle_cases.npz)."""
    c = load_cases()[name]
    kw = case_kwargs(c)
    g = hip_ctx.cpd_lle(c["X"], c["Y0"], float(c["sigma2_in"]), _params(kw, prec), priors=c.get("priors"),
                        visible_nodes=c.get("vis"), H=c.get("H"))
    o = dict(Y=c["Y"], sigma2=float(c["sigma2"]), iters=int(c["iters"]), converged=bool(c["converged"]), n_kept=int(c["n_kept"]))
    _check(g, o, prec)


@pytest.mark.parametrize("prec", [0, 1], ids=["f32", "f64"])
def test_per_iteration_trajectory(hip_ctx, prec):
    """The whole trajectory is pinned, not just the end point: max_iter = 1..6 against the dumped Y / sigma2."""
    for name in ("plain", "vis_priors", "lle"):
        c = load_cases()[name]
        kw = case_kwargs(c)
        for it in range(1, int(c["iters"]) + 1):
            kw["max_iter"] = it
            g = hip_ctx.cpd_lle(c["X"], c["Y0"], float(c["sigma2_in"]), _params(kw, prec), priors=c.get("priors"),
                                visible_nodes=c.get("vis"), H=c.get("H"))
            ty, ts = TOL[prec]
            assert np.abs(g["Y"] - c["trace_Y"][it - 1]).max() <= ty
            assert abs(g["sigma2"] - c["trace_sigma2"][it - 1]) <= ts * c["trace_sigma2"][it - 1]


@pytest.mark.parametrize("prec", [0, 1], ids=["f32", "f64"])
@pytest.mark.parametrize("N,M,iters,opts", [
    (2000, 30, 20, {}),                                    # BASELINE.json configs[0] (C1)
    (2000, 30, 20, dict(vis=True)),
    (2000, 30, 20, dict(priors=True)),
    (2000, 30, 20, dict(vis=True, priors=True, sigma2=1e-4)),
    (2000, 30, 20, dict(lle=True)),
    (1999, 45, 10, {}),                                    # ragged: N not a multiple of 64, production M
    (63, 4, 5, {}),                                        # smallest legal chain, less than one wave of points
    (5000, 64, 6, {}),                                     # exactly one node tile
    (5000, 65, 6, {}),                                     # first size needing a second node tile
    (4000, 100, 8, dict(vis=True)),
    (3000, 130, 4, {}),                                    # M-step leaves LDS (M > 128)
    (3000, 61, 5, {}),                                     # first M outside the 64-column MFMA tableau (M + 3 > 64)
    (3000, 200, 3, dict(priors=True)),                     # blocked M-step in global memory, with the alpha J G term
    (2500, 80, 4, dict(lle=True)),                         # M > 64 with the LLE term: pivoted generic solve
    (6000, 300, 3, {}),                                    # BASELINE.json configs[4] node count (C5)
    (4000, 512, 2, {}),                                    # largest chain of the four-direction smoother (its slots fill a CU's LDS)
    (4000, 513, 2, {}),                                    # round 4: beyond 512 nodes -- E-step with 16 node chunks, the one-direction smoother k_mstep_chain_long
    (3000, 600, 2, dict(priors=True)),
    (5000, 700, 3, dict(vis=True)),
    (4000, 1024, 2, {}),                                   # largest supported chain
    (1500, 600, 1, dict(lle=True)),                        # beyond 512 nodes with the LLE term: the one-workgroup dense elimination (O(M^3): served, not tuned)
], ids=lambda v: str(v).replace(" ", ""))
def test_live_oracle_small(hip_ctx, oracle, N, M, iters, opts, prec):
    from trackdlo_amd import synth
    P = synth.LAUNCH_PARAMS
    vis = opts.get("vis", False)
    X, Y0, v = synth.scene(N, M, config=40 + M, occlude=(0.4, 0.6) if vis else None, outliers=5)
    vext = synth.extend_visible(v, M, synth.geodesic_coord(Y0)) if vis else None
    kw = dict(beta=P["beta"], lambda_=P["lambda_"], lle_weight=P["lle_weight"], mu=P["mu"], max_iter=iters, tol=0.0,
              include_lle=False, alpha=0.0, k_vis=P["k_vis"] if vis else 0.0, visibility_threshold=P["visibility_threshold"])
    pri = None; H = None
    if opts.get("priors"):
        idx = np.arange(0, M, 3)
        pri = np.concatenate([idx[:, None].astype(float), Y0[idx] + np.array([0, 0.004, 0.0])], axis=1)
        kw["alpha"] = P["alpha"]
    if opts.get("lle"):
        L = oracle.calc_lle_weights(Y0, 6); H = (np.eye(M) - L).T @ (np.eye(M) - L)
        kw.update(include_lle=True, beta=P["beta_pre_proc"], lambda_=P["lambda_pre_proc"])
    s2 = opts.get("sigma2", 0.0)
    o = oracle.cpd_lle(X, Y0, s2, priors=pri, visible_nodes=vext, H=H, **kw)
    g = hip_ctx.cpd_lle(X, Y0, s2, _params(kw, prec), priors=pri, visible_nodes=vext, H=H)
    _check(g, o, prec)


def test_c2_full_size_against_oracle(hip_ctx, oracle):
    """BASELINE.json configs[1]: N = 50 000, M = 50, 50 EM iterations, fp32 E-step, tolerance-checked."""
    from trackdlo_amd import synth
    P = synth.LAUNCH_PARAMS
    X, Y0, _ = synth.scene(50000, 50, config=2)
    kw = dict(beta=P["beta"], lambda_=P["lambda_"], lle_weight=P["lle_weight"], mu=P["mu"], max_iter=50, tol=0.0, include_lle=False,
              alpha=0.0, k_vis=0.0, visibility_threshold=P["visibility_threshold"])
    o = oracle.cpd_lle(X, Y0, 0.0, **kw)
    g = hip_ctx.cpd_lle(X, Y0, 0.0, _params(kw, 0))
    _check(g, o, 0)
    # production stopping rule (tol = 2e-4): same iteration count and flag
    kw["tol"] = P["tol"]
    o = oracle.cpd_lle(X, Y0, 0.0, **kw)
    g = hip_ctx.cpd_lle(X, Y0, 0.0, _params(kw, 0))
    _check(g, o, 0)
    assert g["converged"]


@pytest.mark.parametrize("prec", [0, 1], ids=["f32", "f64"])
def test_large_cloud_grouped_partials(hip_ctx, oracle, prec):
    """A cloud that needs 512+ E-step workgroups (N >= 262 144): 24-row tile at several workgroups per CU, grid-stride batches,
    block partials summed in 32 groups (k_part_reduce) before the M-step; with and without visibility weighting, and as a
    batch of two frames (bit-identical to the single calls)."""
    from trackdlo_amd import binding as B, synth
    P = synth.LAUNCH_PARAMS
    N, M = 300000, 50
    ctx = B.Context(device=0, max_frames=2, max_points=N, max_nodes=M)
    try:
        outs = []
        for f, occl in enumerate((None, (0.4, 0.6))):
            X, Y0, v = synth.scene(N, M, config=41, frame=f, occlude=occl, outliers=50)
            vext = synth.extend_visible(v, M, synth.geodesic_coord(Y0)) if occl else None
            kw = dict(beta=P["beta"], lambda_=P["lambda_"], lle_weight=P["lle_weight"], mu=P["mu"], max_iter=4, tol=0.0, include_lle=False,
                      alpha=0.0, k_vis=P["k_vis"] if occl else 0.0, visibility_threshold=P["visibility_threshold"])
            o = oracle.cpd_lle(X, Y0, 0.0, visible_nodes=vext, **kw)
            g = ctx.cpd_lle(X, Y0, 0.0, _params(kw, prec), visible_nodes=vext)
            _check(g, o, prec)
            if occl is None:
                outs.append((X, Y0, g))
        # two frames through the batch entry point: same bits as the single calls
        X, Y0, g0 = outs[0]
        X1, Y1, _ = synth.scene(N, M, config=41, frame=7, outliers=50)
        kw["k_vis"] = 0.0
        g1 = ctx.cpd_lle(X1, Y1, 0.0, _params(kw, prec))
        ctx.set_cloud(0, X); ctx.set_cloud(1, X1)
        b = ctx.cpd_lle_batch([Y0, Y1], [0.0, 0.0], _params(kw, prec))
        np.testing.assert_array_equal(b["Y"][0], g0["Y"]); np.testing.assert_array_equal(b["Y"][1], g1["Y"])
        assert b["sigma2"][0] == g0["sigma2"] and b["sigma2"][1] == g1["sigma2"]
    finally:
        ctx.close()


@pytest.mark.parametrize("prec", [0, 1], ids=["f32", "f64"])
def test_points_whose_every_membership_underflows_go_to_node_zero(hip_ctx, oracle, prec):
    """trackdlo.cpp:298-310 takes the nearest node as the argmax of exp(-d2 / (2 sigma2)): for a kept point further than 38.6 sigma from
    every node (here 6 .. 9 cm off the chain with sigma = 1 mm) the whole column is exactly zero in fp64 and the argmax is the FIRST index,
    node 0 -- and with node 2 nearer than node 1 the end-node rule of :313-321 then hands node 1 a membership of exp(0) = 1 for a point
    at the other end of the rope.  The reference's behaviour, reproduced: same nodes as the oracle at the stated tolerances, iteration by
    iteration, with the rule firing (the oracle counts it) and the nodes really thrown (so a product that took the true nearest node
    would fail here)."""
    from trackdlo_amd import synth
    P = synth.LAUNCH_PARAMS
    for M, N, far in ((20, 1500, 3), (45, 4000, 1), (100, 3000, 5)):
        X, Y0, _ = synth.scene(N, M, config=330 + M, noise=0.0008)
        rng = np.random.default_rng(3300 + M)
        idx = rng.integers(M // 2, M - 1, size=far)                           # off the far half of the chain: node 2 is nearer than node 1
        off = np.column_stack([np.zeros(far), rng.uniform(0.06, 0.09, size=far) * rng.choice([-1.0, 1.0], size=far), np.zeros(far)])
        Xf = np.vstack([X, (Y0[idx] + off).astype(np.float32).astype(np.float64)])
        kw = dict(beta=P["beta"], lambda_=P["lambda_"], lle_weight=P["lle_weight"], mu=P["mu"], max_iter=1, tol=0.0, include_lle=False,
                  alpha=0.0, k_vis=0.0, visibility_threshold=P["visibility_threshold"])
        plain = oracle.cpd_lle(X, Y0, 1e-6, **kw)
        for it in (1, 2, 3):
            kw["max_iter"] = it
            o = oracle.cpd_lle(Xf, Y0, 1e-6, **kw)
            g = hip_ctx.cpd_lle(Xf, Y0, 1e-6, _params(kw, prec))
            _check(g, o, prec)
            if it == 1:
                assert o["gap_quirk"] >= far and o["n_kept"] == plain["n_kept"] + far
                assert np.abs(o["Y"] - plain["Y"]).max() > 1e-3               # the far points did move the nodes (through node 1)


def test_a_hip_error_does_not_leak_into_the_next_call(hip_ctx):
    """HIP keeps a per-thread "last error"; the launchers read it after their own launches.  A call that ended with TDLO_E_HIP (round 3: a
    launch refused for 477-node chains) used to leave it set, and the NEXT, perfectly good call then reported it again."""
    from trackdlo_amd import binding as B, synth
    P = synth.LAUNCH_PARAMS
    X, Y0, _ = synth.scene(3000, 30, config=710)
    pr = B.make_params(P["beta"], P["lambda_"], P["lle_weight"], P["mu"], 5, 0.0, False)
    a = hip_ctx.cpd_lle(X, Y0, 0.0, pr)
    assert hip_ctx.lib.tdlo_debug_fail_hip(hip_ctx.h) == B.TDLO_E_HIP
    assert "invalid" in hip_ctx.lib.tdlo_last_error(hip_ctx.h).decode().lower()
    b = hip_ctx.cpd_lle(X, Y0, 0.0, pr)
    assert b["rc"] == 0 and np.array_equal(a["Y"], b["Y"]) and a["sigma2"] == b["sigma2"]


def test_hostile_inputs_come_back_and_leave_the_context_usable():
    """NaN / Inf / 1e30 coordinates in the cloud or the nodes, sigma2 negative / NaN / 1e300, NaN or extreme parameters, NaN priors, NaN or 1e300 H,
    coincident nodes, a single point: 124 calls (both precisions, with and without the LLE term), each of which must return -- a result with
    finite numbers or a negative TDLO_E_* code, never a hang -- and be followed by a plain registration that reproduces its bits
    (scripts/gpu_hostile_inputs.py, run in a child process under a time limit)."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "scripts", "gpu_hostile_inputs.py")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
    assert "all calls returned" in r.stdout.splitlines()[-1]


def test_early_exit_and_max_iter_flags(hip_ctx, oracle):
    from trackdlo_amd import synth
    P = synth.LAUNCH_PARAMS
    X, Y0, _ = synth.scene(3000, 30, config=7)
    kw = dict(beta=P["beta"], lambda_=P["lambda_"], lle_weight=P["lle_weight"], mu=P["mu"], max_iter=50, tol=P["tol"], include_lle=False,
              alpha=0.0, k_vis=0.0, visibility_threshold=0.01)
    o = oracle.cpd_lle(X, Y0, 0.0, **kw)
    g = hip_ctx.cpd_lle(X, Y0, 0.0, _params(kw, 0))
    assert o["converged"] and g["converged"] and g["iters"] == o["iters"] < 50
    kw["max_iter"] = 3                       # reached without convergence -> cpd_lle returns false (:433-437)
    o = oracle.cpd_lle(X, Y0, 0.0, **kw)
    g = hip_ctx.cpd_lle(X, Y0, 0.0, _params(kw, 0))
    assert (not o["converged"]) and (not g["converged"]) and g["iters"] == 3


def test_error_paths(hip_ctx):
    from trackdlo_amd import binding as B, synth
    P = synth.LAUNCH_PARAMS
    X, Y0, _ = synth.scene(500, 10, config=8)
    pr = B.make_params(P["beta"], P["lambda_"], P["lle_weight"], P["mu"], 5, 0.0, False)
    # every point pruned: nothing within 0.1 m of any node
    g = hip_ctx.cpd_lle(X + np.array([0, 0, 5.0]), Y0, 0.0, pr, check=False)
    assert g["rc"] == B.TDLO_E_EMPTY
    np.testing.assert_array_equal(g["Y"], Y0)                 # Y untouched
    # M < 4 is undefined in the reference (clamps at :313-321) -> rejected
    g = hip_ctx.cpd_lle(X, Y0[:3], 0.0, pr, check=False)
    assert g["rc"] == B.TDLO_E_INVALID
    with pytest.raises(B.TdloError):
        hip_ctx.cpd_lle(X, Y0, 0.0, pr, priors=[[99, 0, 0, 0]])
    # the context stays usable after errors
    g = hip_ctx.cpd_lle(X, Y0, 0.0, pr)
    assert g["rc"] == 0 and g["iters"] == 5


def _oracle_or_clean_error(g, o, prec):
    """The result is the oracle's at the stated tolerance, or the call says TDLO_E_NUMERIC -- never finite garbage with rc = 0."""
    from trackdlo_amd import binding as B
    if g["rc"] == B.TDLO_E_NUMERIC:
        return "error"
    assert g["rc"] == 0 and g["iters"] == o["iters"] and g["n_kept"] == o["n_kept"]
    ty, ts = (1e-5, 1e-3) if prec == 0 else (1e-9, 1e-7)
    assert np.abs(g["Y"] - o["Y"]).max() <= ty and abs(g["sigma2"] - o["sigma2"]) <= ts * o["sigma2"]
    return "ok"


@pytest.mark.parametrize("prec", [0, 1], ids=["f32", "f64"])
def test_sums_beyond_the_assumed_extent_are_right_or_refused(oracle, prec):
    """The E-step's 64-bit fixed-point sums are scaled from an ASSUMED extent of the scene (twice the chain's length); nothing in the reference
    bounds the nodes by it (trackdlo.cpp:240-260: a correspondence prior pulls a node wherever it says; :386-389 have no range at all).  (i) a prior
    2 m off the rope with alpha = 3, (ii) a registration started 9 cm beside the cloud with lambda = 1 (nodes free to fly), (iii) mu = 0 far from the
    cloud (c = 0: a point whose memberships all underflow divides 0 by 0): the oracle's result, or TDLO_E_NUMERIC -- the range check of every
    converted value (FrameDev::acc_lim) is what stands between a wrapped-around integer and a result that merely looks fine."""
    from trackdlo_amd import binding as B, synth
    P = synth.LAUNCH_PARAMS
    M, N = 30, 4000
    X, Y0, _ = synth.scene(N, M, config=810)
    base = dict(beta=P["beta"], lambda_=P["lambda_"], lle_weight=P["lle_weight"], mu=P["mu"], max_iter=12, tol=0.0, include_lle=False,
                alpha=0.0, k_vis=0.0, visibility_threshold=P["visibility_threshold"])
    outcomes = []
    ctx = B.Context(device=0, max_points=N, max_nodes=M)
    try:
        # (i)
        kw = dict(base, alpha=3.0)
        pri = np.array([[7, Y0[7, 0], Y0[7, 1] + 2.0, Y0[7, 2]], [20, Y0[20, 0] - 1.5, Y0[20, 1], Y0[20, 2] + 1.0]])
        g = ctx.cpd_lle(X, Y0, 0.0, _params(kw, prec), priors=pri, check=False)
        outcomes.append(_oracle_or_clean_error(g, oracle.cpd_lle(X, Y0, 0.0, priors=pri, **kw), prec))
        # (ii)
        kw = dict(base, lambda_=1.0, beta=0.1)
        Yoff = np.asfortranarray(Y0 + np.array([0.0, 0.09, 0.0]))
        g = ctx.cpd_lle(X, Yoff, 0.0, _params(kw, prec), check=False)
        outcomes.append(_oracle_or_clean_error(g, oracle.cpd_lle(X, Yoff, 0.0, **kw), prec))
        # (iii)
        kw = dict(base, mu=0.0)
        g = ctx.cpd_lle(X, Yoff, 1e-6, _params(kw, prec), check=False)
        o = oracle.cpd_lle(X, Yoff, 1e-6, **kw)
        if np.isfinite(o["Y"]).all() and np.isfinite(o["sigma2"]):
            outcomes.append(_oracle_or_clean_error(g, o, prec))
        else:                                   # the reference itself produces NaN here (0 / 0 at :301): an error is the only right answer
            assert g["rc"] == B.TDLO_E_NUMERIC
            outcomes.append("error")
        # the context stays usable
        g = ctx.cpd_lle(X, Y0, 0.0, _params(base, prec))
        assert g["rc"] == 0 and g["iters"] == 12
    finally:
        ctx.close()
    print("outcomes:", outcomes)


def test_a_contribution_beyond_the_fixed_point_range_is_refused():
    """Forces the range check itself: a cloud of a few points and a prior a kilometre away -- after the first M-step the node sits ~1 km
    from its points, far beyond anything the accumulators' exponents were chosen for (R ~ 1e3 m against an assumed extent of ~1 m is still
    representable; 1e9 m is not).  The call must not return rc = 0 with a finite-looking result that differs from the oracle's."""
    from trackdlo_amd import binding as B, synth
    P = synth.LAUNCH_PARAMS
    M, N = 12, 600
    X, Y0, _ = synth.scene(N, M, config=811)
    kw = dict(beta=P["beta"], lambda_=1.0, lle_weight=P["lle_weight"], mu=P["mu"], max_iter=6, tol=0.0, include_lle=False,
              alpha=1e12, k_vis=0.0, visibility_threshold=P["visibility_threshold"])
    pri = np.array([[5, Y0[5, 0] + 3e4, Y0[5, 1], Y0[5, 2]]])
    ctx = B.Context(device=0, max_points=N, max_nodes=M)
    try:
        g = ctx.cpd_lle(X, Y0, 0.0, _params(kw, 1), priors=pri, check=False)
    finally:
        ctx.close()
    if g["rc"] == 0:        # then it must be right
        import oracle.ref_cpu as ref
        o = ref.cpd_lle(X, Y0, 0.0, priors=pri, **kw)
        assert np.abs(g["Y"] - o["Y"]).max() <= 1e-6 * max(1.0, np.abs(o["Y"]).max())
    else:
        assert g["rc"] == B.TDLO_E_NUMERIC


def test_batch_with_an_empty_frame():
    """A batch (stream groups, merged copies) in which one frame loses every point to the prune: the call reports
    TDLO_E_EMPTY, the other frames are registered as if alone, and the context stays usable."""
    import ctypes as C
    from trackdlo_amd import binding as B, synth
    P = synth.LAUNCH_PARAMS
    F, M = 9, 20
    ctx = B.Context(device=0, max_frames=F, max_points=1 << 13, max_nodes=32)
    try:
        pr = B.make_params(P["beta"], P["lambda_"], P["lle_weight"], P["mu"], 6, 0.0, False)
        Ys, single = [], []
        for f in range(F):
            X, Y0, _ = synth.scene(1500 + 100 * f, M, config=55, frame=f)
            if f == 6: X = X + np.array([0, 0, 5.0])
            ctx.set_cloud(f, X); Ys.append(Y0)
        for f in range(F):
            single.append(ctx.cpd_lle_resident(f, Ys[f], 0.0, pr, check=False))
        assert single[6]["rc"] == B.TDLO_E_EMPTY
        Yb = np.ascontiguousarray(np.asarray(Ys).transpose(0, 2, 1)); s2 = np.zeros(F); st = (B.Stats * F)()
        rc = ctx.lib.tdlo_cpd_lle_batch(ctx.h, F, B._ptr(Yb), M, B._ptr(s2), C.byref(pr), None, 0, None, 0, None, C.cast(st, C.c_void_p))
        assert rc == B.TDLO_E_EMPTY
        for f in range(F):
            if f == 6:
                assert st[f].status == B.TDLO_E_EMPTY
                np.testing.assert_array_equal(Yb[f].T, Ys[f])                                # untouched
            else:
                assert st[f].status == 0 and st[f].iters == 6
                np.testing.assert_array_equal(Yb[f].T, single[f]["Y"])
        ctx.set_cloud(6, synth.scene(2100, M, config=55, frame=6)[0])
        out = ctx.cpd_lle_batch(Ys, [0.0] * F, pr)
        assert all(s["status"] == 0 and s["iters"] == 6 for s in out["stats"])
    finally:
        ctx.close()


def test_bitwise_repeatable(hip_ctx):
    from trackdlo_amd import binding as B, synth
    P = synth.LAUNCH_PARAMS
    X, Y0, v = synth.scene(20000, 50, config=9, occlude=(0.3, 0.5))
    vext = synth.extend_visible(v, 50, synth.geodesic_coord(Y0))
    pr = B.make_params(P["beta"], P["lambda_"], P["lle_weight"], P["mu"], 15, 0.0, False, 0.0, P["k_vis"], P["visibility_threshold"])
    a = hip_ctx.cpd_lle(X, Y0, 0.0, pr, visible_nodes=vext)
    b = hip_ctx.cpd_lle(X, Y0, 0.0, pr, visible_nodes=vext)
    np.testing.assert_array_equal(a["Y"], b["Y"])
    assert a["sigma2"] == b["sigma2"]


@pytest.mark.parametrize("prec", [0, 1], ids=["f32", "f64"])
def test_full_size_invariances(hip_ctx, prec):
    """Size-independent properties at N = 200 000 (oracle would take minutes): the registration is
    invariant under a permutation of the cloud and equivariant under a rigid translation."""
    from trackdlo_amd import binding as B, synth
    P = synth.LAUNCH_PARAMS
    N, M = 200000, 50
    X, Y0, _ = synth.scene(N, M, config=4)
    pr = B.make_params(P["beta"], P["lambda_"], P["lle_weight"], P["mu"], 10, 0.0, False, precision=prec)
    a = hip_ctx.cpd_lle(X, Y0, 0.0, pr)
    perm = np.random.default_rng(1).permutation(N)
    b = hip_ctx.cpd_lle(X[perm], Y0, 0.0, pr)
    ty, ts = TOL[prec]
    assert np.abs(a["Y"] - b["Y"]).max() <= ty * 0.1
    assert abs(a["sigma2"] - b["sigma2"]) <= ts * a["sigma2"]
    sh = np.array([0.25, -0.125, 0.5])        # exactly representable shift
    c = hip_ctx.cpd_lle(X + sh, Y0 + sh, 0.0, pr)
    assert np.abs((c["Y"] - sh) - a["Y"]).max() <= ty
    assert abs(c["sigma2"] - a["sigma2"]) <= ts * a["sigma2"]
    assert a["n_kept"] == N and a["iters"] == 10


def test_batch_equals_single(hip_ctx):
    """BASELINE.json configs[2] per GPU: frames registered concurrently give exactly the single-frame results."""
    from trackdlo_amd import binding as B, synth
    P = synth.LAUNCH_PARAMS
    F, N, M = 6, 8000, 50
    pr = B.make_params(P["beta"], P["lambda_"], P["lle_weight"], P["mu"], 12, 0.0, False)
    Xs, Ys = [], []
    for f in range(F):
        X, Y0, _ = synth.scene(N - 37 * f, M, config=3, frame=f)       # ragged frame sizes
        Xs.append(X); Ys.append(Y0)
    single = [hip_ctx.cpd_lle(Xs[f], Ys[f], 0.0, pr) for f in range(F)]
    for f in range(F):
        hip_ctx.set_cloud(f, Xs[f])
    out = hip_ctx.cpd_lle_batch(Ys, [0.0] * F, pr)
    for f in range(F):
        np.testing.assert_array_equal(out["Y"][f], single[f]["Y"])
        assert out["sigma2"][f] == single[f]["sigma2"]
        assert out["stats"][f]["iters"] == 12 and out["stats"][f]["n_kept"] == single[f]["n_kept"]


@pytest.mark.parametrize("F,tol,vis_on", [(8, 0.0, False), (13, 0.0, True), (16, 2e-4, False), (20, 2e-4, True), (9, 2e-4, False)])
def test_batch_stream_groups_equal_single_calls(F, tol, vis_on):
    """Batches of 8+ frames run as 2 or 4 groups of frames on as many streams (run_frames): fixed iteration counts and the
    early-exit polling path (frames stop after different numbers of iterations), with and without visibility weighting --
    every frame must equal, bit for bit, the same frame registered on its own."""
    from trackdlo_amd import binding as B, synth
    P = synth.LAUNCH_PARAMS
    M = 32
    rng = np.random.default_rng(4200 + F)
    ctx = B.Context(device=0, max_frames=F, max_points=1 << 14, max_nodes=64)
    try:
        kw = dict(beta=P["beta"], lambda_=P["lambda_"], lle_weight=P["lle_weight"], mu=P["mu"], max_iter=40 if tol else 7, tol=tol, include_lle=False,
                  alpha=0.0, k_vis=P["k_vis"] if vis_on else 0.0, visibility_threshold=P["visibility_threshold"])
        params = _params(kw, 0)
        Ys, s2s, single, vext = [], [], [], None
        for f in range(F):
            X, Y0, v = synth.scene(int(rng.integers(300, 12000)), M, config=130 + F, frame=f, occlude=(0.4, 0.6) if vis_on else None)
            if vis_on: vext = synth.extend_visible(v, M, synth.geodesic_coord(Y0))
            ctx.set_cloud(f, X)
            Ys.append(Y0); s2s.append(float(rng.choice([0.0, 1e-4, 1e-5])))
        for f in range(F):
            single.append(ctx.cpd_lle_resident(f, Ys[f], s2s[f], params, visible_nodes=vext))
        out = ctx.cpd_lle_batch(Ys, s2s, params, visible_nodes=vext)
        for f in range(F):
            assert np.array_equal(out["Y"][f], single[f]["Y"]) and out["sigma2"][f] == single[f]["sigma2"]
            assert out["stats"][f]["iters"] == single[f]["iters"] and out["stats"][f]["converged"] == single[f]["converged"]
        if tol:
            assert len({out["stats"][f]["iters"] for f in range(F)}) > 1          # the frames did stop at different iterations
    finally:
        ctx.close()


def test_nsplit_single_rank_equals_plain(hip_ctx):
    """The N-split entry points with one rank (identity collectives) reproduce the plain call."""
    from trackdlo_amd import binding as B, nsplit, synth
    P = synth.LAUNCH_PARAMS

    class Identity:
        def all_reduce_sum(self, a): return np.array(a, dtype=np.float64)
        def all_reduce_min(self, a): return np.array(a, dtype=np.float64)

    for vis in (False, True):
        X, Y0, v = synth.scene(6000, 40, config=6, occlude=(0.4, 0.6) if vis else None, outliers=11)
        vext = synth.extend_visible(v, 40, synth.geodesic_coord(Y0)) if vis else None
        pr = B.make_params(P["beta"], P["lambda_"], P["lle_weight"], P["mu"], 10, 0.0, False, 0.0, P["k_vis"] if vis else 0.0,
                           P["visibility_threshold"])
        a = hip_ctx.cpd_lle(X, Y0, 0.0, pr, visible_nodes=vext)
        b = nsplit.cpd_lle_nsplit(nsplit.HipShard(hip_ctx, X), Identity(), Y0, 0.0, pr, visible_nodes=vext)
        assert np.abs(a["Y"] - b["Y"]).max() <= 1e-12
        assert abs(a["sigma2"] - b["sigma2"]) <= 1e-12 * a["sigma2"]
        assert b["iters"] == 10 and b["n_kept"] == a["n_kept"]


def test_nsplit_two_shards_on_one_gpu(hip_ctx, oracle):
    """Two shards (two contexts on the same GPU) exchanging sums through the N-split driver equal the oracle."""
    import threading, queue
    from trackdlo_amd import binding as B, nsplit, synth
    P = synth.LAUNCH_PARAMS
    X, Y0, v = synth.scene(9000, 40, config=12, occlude=(0.4, 0.6))
    vext = synth.extend_visible(v, 40, synth.geodesic_coord(Y0))
    pr = B.make_params(P["beta"], P["lambda_"], P["lle_weight"], P["mu"], 8, 0.0, False, 0.0, P["k_vis"], P["visibility_threshold"])

    class Pair:      # in-process 2-rank "communicator"
        def __init__(self):
            self.b = threading.Barrier(2); self.slots = [None, None]
        def comm(self, rank):
            outer = self
            class C_:
                def _x(self, a, op):
                    outer.slots[rank] = np.array(a, dtype=np.float64); outer.b.wait()
                    r = op(outer.slots[0], outer.slots[1]); outer.b.wait(); return r
                def all_reduce_sum(self, a): return self._x(a, np.add)
                def all_reduce_min(self, a): return self._x(a, np.minimum)
            return C_()

    pair = Pair(); res = queue.Queue()
    ctxs = [hip_ctx, B.Context(device=0, max_points=1 << 14, max_nodes=64)]
    n = X.shape[0]

    def work(r):
        try:
            out = nsplit.cpd_lle_nsplit(nsplit.HipShard(ctxs[r], X[r * n // 2:(r + 1) * n // 2]), pair.comm(r), Y0, 0.0, pr,
                                        visible_nodes=vext)
            res.put((r, out))
        except Exception as e:      # pragma: no cover
            res.put((r, e)); pair.b.abort()

    th = [threading.Thread(target=work, args=(r,)) for r in range(2)]
    [t.start() for t in th]; [t.join() for t in th]
    outs = dict(res.get() for _ in range(2))
    ctxs[1].close()
    o = oracle.cpd_lle(X, Y0, 0.0, beta=P["beta"], lambda_=P["lambda_"], lle_weight=P["lle_weight"], mu=P["mu"], max_iter=8, tol=0.0,
                       include_lle=False, k_vis=P["k_vis"], visibility_threshold=P["visibility_threshold"], visible_nodes=vext)
    for r in range(2):
        assert not isinstance(outs[r], Exception), outs[r]
        assert np.abs(outs[r]["Y"] - o["Y"]).max() <= 1e-5
        assert abs(outs[r]["sigma2"] - o["sigma2"]) <= 1e-3 * o["sigma2"]
    np.testing.assert_array_equal(outs[0]["Y"], outs[1]["Y"])


def test_nsplit_device_exchange_single_rank_rccl(hip_ctx):
    """The device-resident N-split exchange (tdlo_split_*_enqueue + RCCL all-reduce of the bound device buffers, ordered on
    the context's stream, no host synchronisation inside an iteration) with a one-rank RCCL group reproduces the plain call,
    with fixed iteration counts and with the production stopping rule (device-side flag polled every few iterations)."""
    import tempfile
    import torch
    import torch.distributed as dist
    from trackdlo_amd import binding as B, nsplit, synth
    P = synth.LAUNCH_PARAMS
    os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")       # one-rank group on a box without network: bootstrap over loopback
    torch.cuda.set_device(0)
    store = os.path.join(tempfile.mkdtemp(prefix="tdlo_pg_"), "store")      # file rendezvous: no TCP store, no host-name lookups
    dist.init_process_group("nccl", init_method="file://" + store, rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        for vis, tol, prec in ((False, 0.0, B.PREC_F32), (True, 0.0, B.PREC_F32), (True, 2e-4, B.PREC_F32), (True, 0.0, B.PREC_F64)):
            X, Y0, v = synth.scene(6000, 40, config=6, occlude=(0.4, 0.6) if vis else None, outliers=11)
            vext = synth.extend_visible(v, 40, synth.geodesic_coord(Y0)) if vis else None
            pr = B.make_params(P["beta"], P["lambda_"], P["lle_weight"], P["mu"], 10 if tol == 0 else 50, tol, False, 0.0,
                               P["k_vis"] if vis else 0.0, P["visibility_threshold"], precision=prec)
            a = hip_ctx.cpd_lle(X, Y0, 0.0, pr, visible_nodes=vext)
            xch = nsplit.TorchDeviceExchange(40, "cuda:0", stream_ptr=hip_ctx.stream_ptr())
            b = nsplit.cpd_lle_nsplit_device(nsplit.HipDeviceShard(hip_ctx, X, xch), xch, nsplit.TorchComm("cuda:0"), Y0, 0.0, pr,
                                             visible_nodes=vext)
            assert np.abs(a["Y"] - b["Y"]).max() <= 1e-12
            assert abs(a["sigma2"] - b["sigma2"]) <= 1e-12 * a["sigma2"]
            assert b["iters"] == a["iters"] and b["converged"] == a["converged"] and b["n_kept"] == a["n_kept"]
    finally:
        dist.destroy_process_group()


def test_nsplit_device_exchange_two_shards_on_one_gpu(hip_ctx, oracle):
    """Two shards (two contexts, two streams on the same GPU) driven by cpd_lle_nsplit_device; the collective is replaced by an
    in-process exchange that reduces the two bound DEVICE buffers with torch ops (what RCCL does between two GPUs)."""
    import threading, queue
    import torch
    from trackdlo_amd import binding as B, nsplit, synth
    P = synth.LAUNCH_PARAMS
    M = 40
    X, Y0, v = synth.scene(9000, M, config=12, occlude=(0.4, 0.6))
    vext = synth.extend_visible(v, M, synth.geodesic_coord(Y0))
    ctxs = [hip_ctx, B.Context(device=0, max_points=1 << 14, max_nodes=64)]
    n = X.shape[0]

    class PairExchange:
        def __init__(self):
            self.b = threading.Barrier(2)
            self.bufs = [torch.zeros(5 * M + 2, dtype=torch.float64, device="cuda:0") for _ in range(2)]
        def view(self, rank):
            outer = self
            class V:
                dmin = outer.bufs[rank][:M]; sums = outer.bufs[rank][M:]
                def _x(self, lo, hi, op):
                    ctxs[rank].synchronize(); outer.b.wait()
                    r = op(outer.bufs[0][lo:hi], outer.bufs[1][lo:hi]); torch.cuda.synchronize(); outer.b.wait()
                    outer.bufs[rank][lo:hi].copy_(r); torch.cuda.synchronize()
                def all_reduce_min_dmin(self): self._x(0, M, torch.minimum)
                def all_reduce_sum_sums(self): self._x(M, 5 * M + 2, torch.add)
            return V()

    class PairInit:
        def __init__(self):
            self.b = threading.Barrier(2); self.slots = [None, None]
        def comm(self, rank):
            outer = self
            class C_:
                def all_reduce_sum(self, a):
                    outer.slots[rank] = np.array(a, dtype=np.float64); outer.b.wait()
                    r = outer.slots[0] + outer.slots[1]; outer.b.wait(); return r
            return C_()

    for tol, max_iter in ((0.0, 8), (2e-4, 50)):
        pr = B.make_params(P["beta"], P["lambda_"], P["lle_weight"], P["mu"], max_iter, tol, False, 0.0, P["k_vis"], P["visibility_threshold"])
        pair = PairExchange(); init = PairInit(); res = queue.Queue()

        def work(r):
            try:
                xch = pair.view(r)
                out = nsplit.cpd_lle_nsplit_device(nsplit.HipDeviceShard(ctxs[r], X[r * n // 2:(r + 1) * n // 2], xch), xch, init.comm(r),
                                                   Y0, 0.0, pr, visible_nodes=vext)
                res.put((r, out))
            except Exception as e:      # pragma: no cover
                res.put((r, e)); pair.b.abort(); init.b.abort()

        th = [threading.Thread(target=work, args=(r,)) for r in range(2)]
        [t.start() for t in th]; [t.join() for t in th]
        outs = dict(res.get() for _ in range(2))
        o = oracle.cpd_lle(X, Y0, 0.0, beta=P["beta"], lambda_=P["lambda_"], lle_weight=P["lle_weight"], mu=P["mu"], max_iter=max_iter, tol=tol,
                           include_lle=False, k_vis=P["k_vis"], visibility_threshold=P["visibility_threshold"], visible_nodes=vext)
        for r in range(2):
            assert not isinstance(outs[r], Exception), outs[r]
            assert outs[r]["iters"] == o["iters"] and outs[r]["converged"] == o["converged"]
            assert np.abs(outs[r]["Y"] - o["Y"]).max() <= 1e-5
            assert abs(outs[r]["sigma2"] - o["sigma2"]) <= 1e-3 * o["sigma2"]
        np.testing.assert_array_equal(outs[0]["Y"], outs[1]["Y"])
        assert outs[0]["n_kept"] + outs[1]["n_kept"] == o["n_kept"]
    ctxs[1].close()


@pytest.mark.parametrize("occl", [None, (0.45, 0.5), (0.35, 0.65), (0.0, 0.3), (0.7, 1.0), (0.0, 0.2, 0.8, 1.0)],
                         ids=["all", "minor", "mid", "head", "tail", "both-ends"])
def test_tracking_step_against_oracle(hip_ctx, oracle, occl):
    """trackdlo::tracking_step (trackdlo.cpp:900-999) through the tracker object, all five occlusion states.
    The pre-processing registration's LLE matrix is injected on both sides (its weights are ill-conditioned)."""
    _tracking_step_case(hip_ctx, oracle, occl, 30, 3000)


@pytest.mark.parametrize("occl", [None, (0.45, 0.55), (0.0, 0.2)], ids=["all", "mid", "head"])
def test_tracking_step_long_chain(hip_ctx, oracle, occl):
    """The same on a chain of 160 nodes: the pre-processing registration (LLE term) runs on k_mstep_pivot_mcu when more than
    128 nodes are visible (k_mstep below), the main registration on k_mstep_mcu."""
    _tracking_step_case(hip_ctx, oracle, occl, 160, 9000)


def _tracking_step_case(hip_ctx, oracle, occl, M, N):
    from trackdlo_amd import binding as B, synth
    P = synth.LAUNCH_PARAMS
    Y0 = synth.nodes(M); coord = synth.geodesic_coord(Y0)
    if occl is None:
        X, _, _ = synth.scene(N, M, config=20); vis = np.arange(M)
    elif len(occl) == 2:
        X, _, vis = synth.scene(N, M, config=20, occlude=occl)
    else:
        X, _, _ = synth.scene(N, M, config=20)
        s = np.linspace(0, 1, M)
        vis = np.nonzero((s > occl[1]) & (s < occl[2]))[0].astype(np.int32)
        d = np.linalg.norm(X[:, None, :] - Y0[None, vis, :], axis=2).min(axis=1)
        X = np.asfortranarray(X[d < 0.012])
    vext = synth.extend_visible(vis, M, coord)
    Lg = oracle.calc_lle_weights(Y0[vext], 6)
    Hpre = (np.eye(len(vext)) - Lg).T @ (np.eye(len(vext)) - Lg)
    args = (M, P["visibility_threshold"], P["beta"], P["lambda_"], P["alpha"], P["k_vis"], P["mu"], 30, P["tol"], P["beta_pre_proc"],
            P["lambda_pre_proc"], P["lle_weight"])
    ref = oracle.Tracker(*args); ref.initialize_nodes(Y0); ref.initialize_geodesic_coord(coord)
    trk = B.trackdlo(*args, ctx=hip_ctx); trk.initialize_nodes(Y0); trk.initialize_geodesic_coord(coord)
    for step in range(2):        # two consecutive frames: state (Y_, sigma2_) carries over
        ref.tracking_step(X, vis, vext, H_pre=Hpre)
        trk.tracking_step(X, vis, vext, None, 0, 0, H_pre=Hpre)
        assert trk.last_stats[0]["iters"] == ref.stats_pre.iters and trk.last_stats[1]["iters"] == ref.stats_main.iters
        np.testing.assert_allclose(trk.get_guide_nodes(), ref.get_guide_nodes(), rtol=0, atol=1e-5)
        kp, kr = trk.get_correspondence_pairs(), ref.get_correspondence_pairs()
        assert kp.shape == kr.shape
        np.testing.assert_allclose(kp, kr, rtol=0, atol=1e-5)
        np.testing.assert_allclose(trk.get_tracking_result(), ref.get_tracking_result(), rtol=0, atol=1e-5)
        assert abs(trk.get_sigma2() - ref.get_sigma2()) <= 1e-3 * ref.get_sigma2()


def test_cpp_drop_in_class():
    """include/trackdlo_shim.hpp (`class trackdlo`, reference signatures) used like trackdlo_node.cpp:131-143/:366-369,
    compared with the oracle inside the C++ program (tests/cpp/shim_test.cpp, built by __graft_entry__.build())."""
    import os, subprocess
    from conftest import ROOT
    exe = os.path.join(ROOT, "tests", "cpp", "shim_test")
    assert os.path.exists(exe), "run __graft_entry__.build() first"
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "OK" in r.stdout


@pytest.mark.parametrize("occl", [None, (0.45, 0.5), (0.3, 0.6), (0.0, 0.25)], ids=["all", "minor", "mid", "head"])
def test_visibility_prepass(hip_ctx, oracle, occl):
    """the node<->cloud distance pre-pass of trackdlo_node.cpp:257-277 and the gap fill of :345-360."""
    from trackdlo_amd import synth
    P = synth.LAUNCH_PARAMS
    M = 45
    X, Y0, _ = synth.scene(30000, M, config=30, occlude=occl, outliers=7)
    coord = synth.geodesic_coord(Y0)
    hip_ctx.set_cloud(0, X)
    d, vis, ext = hip_ctx.visibility_prepass(0, Y0, P["visibility_threshold"], 0.06, coord)
    do, viso, exto = oracle.visibility_prepass(X, Y0, P["visibility_threshold"], 0.06, coord)
    np.testing.assert_allclose(d, do, rtol=0, atol=1e-12)
    np.testing.assert_array_equal(vis, viso)
    np.testing.assert_array_equal(ext, exto)
    assert len(vis) >= 1 and (occl is not None or len(vis) >= M - 2)
    np.testing.assert_array_equal(ext, synth.extend_visible(vis, M, coord, 0.06))


def test_tracking_sequence_in_evaluation_units(hip_ctx, oracle):
    """A short synthetic sequence tracked by the HIP tracker and by the oracle tracker, compared in the reference's
    own evaluation unit: evaluator::compute_error (evaluator.cpp:333-341), metres of mean node-to-curve distance."""
    from trackdlo_amd import binding as B, synth
    P = synth.LAUNCH_PARAMS
    M = 40
    Y0 = synth.nodes(M); coord = synth.geodesic_coord(Y0)
    args = (M, P["visibility_threshold"], P["beta"], P["lambda_"], P["alpha"], P["k_vis"], P["mu"], 30, P["tol"], P["beta_pre_proc"],
            P["lambda_pre_proc"], P["lle_weight"])
    ref = oracle.Tracker(*args); ref.initialize_nodes(Y0); ref.initialize_geodesic_coord(coord)
    trk = B.trackdlo(*args, ctx=hip_ctx); trk.initialize_nodes(Y0); trk.initialize_geodesic_coord(coord)
    Lg = oracle.calc_lle_weights(Y0, 6)
    Hpre = (np.eye(M) - Lg).T @ (np.eye(M) - Lg)
    vis = np.arange(M)
    for frame in range(5):
        shift = np.array([0.0, 0.004 * (frame + 1), 0.002 * frame])
        X, _, _ = synth.scene(4000, M, config=21, frame=frame, shift=shift)
        truth = Y0 + shift
        ref.tracking_step(X, vis, vis, H_pre=Hpre)
        trk.tracking_step(X, vis, vis, None, 0, 0, H_pre=Hpre)
        Yg, Yr = trk.get_tracking_result(), ref.get_tracking_result()
        assert B.compute_error(Yg, Yr) <= 1e-5                      # HIP vs oracle, in evaluation units
        assert abs(B.compute_error(Yg, truth) - oracle.compute_error(Yr, truth)) <= 1e-5
        assert B.compute_error(Yg, truth) < 0.004                   # and it actually tracks the moving rope


@pytest.mark.gpu
@pytest.mark.parametrize("M,leaf,zero,shape", [(50, 0.008, 0, None), (30, 0.02, 7, None), (45, 0.004, 0, None), (50, 0.008, 0, (120, 161)),
                                                (50, 0.0015, 0, None)])
def test_depth_to_cloud_bit_exact(hip_ctx, oracle, M, leaf, zero, shape):
    """SURVEY 8(f) row 2 (trackdlo_node.cpp:195-241): masked back-projection + voxel-grid centroids on the device are
    BIT-EXACT against the oracle (float work in the same order): same count, same order, same doubles.  Cases: the launch
    file's 8 mm leaf, a coarse leaf with invalid-depth pixels (back-projected to the origin, as the reference does), a fine
    leaf, an odd image size (ragged last blocks), a leaf that needs three radix passes."""
    from trackdlo_amd import synth
    kw = dict(rows=shape[0], cols=shape[1]) if shape else {}
    depth, mask, cam, _ = synth.depth_scene(M, config=9, frame=M, zero_depth_pixels=zero, **kw)
    Xo, nraw_o = oracle.depth_to_cloud(depth, mask, cam["fx"], cam["fy"], cam["cx"], cam["cy"], leaf)
    Xg, n, nraw = hip_ctx.depth_to_cloud(0, depth, mask, cam["fx"], cam["fy"], cam["cx"], cam["cy"], leaf)
    assert nraw == nraw_o and n == Xo.shape[0]
    assert np.array_equal(Xg, Xo)
    # idempotent on repeat (workspace reuse), and the resident cloud is what came back
    Xg2, n2, _ = hip_ctx.depth_to_cloud(0, depth, mask, cam["fx"], cam["fy"], cam["cx"], cam["cy"], leaf)
    assert n2 == n and np.array_equal(Xg2, Xg)


@pytest.mark.gpu
def test_depth_to_cloud_edge_cases(hip_ctx, oracle):
    from trackdlo_amd import synth, binding as B
    depth, mask, cam, _ = synth.depth_scene(30, config=9)
    args = (cam["fx"], cam["fy"], cam["cx"], cam["cy"])
    X, n, nraw = hip_ctx.depth_to_cloud(0, depth, np.zeros_like(mask), *args, 0.008)          # nothing segmented
    assert n == 0 and nraw == 0 and X.shape == (0, 3)
    Xo, _ = oracle.depth_to_cloud(depth, mask, *args, 1e-5)                               # cell count overflows int32: pass-through
    X, n, nraw = hip_ctx.depth_to_cloud(0, depth, mask, *args, 1e-5)
    assert n == nraw == Xo.shape[0] and np.array_equal(X, Xo)
    one = np.zeros_like(mask); one[100, 200] = 255                                        # a single pixel
    Xo, _ = oracle.depth_to_cloud(depth, one, *args, 0.008)
    X, n, _ = hip_ctx.depth_to_cloud(0, depth, one, *args, 0.008)
    assert n == 1 and np.array_equal(X, Xo)
    full = np.full_like(mask, 255)                                                        # every pixel (background wall included)
    Xo, _ = oracle.depth_to_cloud(depth, full, *args, 0.02)
    X, n, nraw = hip_ctx.depth_to_cloud(0, depth, full, *args, 0.02)
    assert nraw == depth.size and np.array_equal(X, Xo)
    with pytest.raises(B.TdloError):
        hip_ctx.depth_to_cloud(0, depth, mask, *args, 0.0)


@pytest.mark.gpu
def test_frame_born_on_device_matches_host_path(hip_ctx, oracle):
    """depth image -> cloud (device) -> visibility pre-pass -> tracking_step with X = NULL gives exactly the nodes of the
    host path (oracle cloud uploaded through tracking_step(X))."""
    from trackdlo_amd import synth, binding as B
    M = 30            # 0.58 m of rope: fits the 640-pixel image at 0.6 m
    P = synth.LAUNCH_PARAMS
    depth, mask, cam, Y0 = synth.depth_scene(M, config=9, frame=3)
    args = (cam["fx"], cam["fy"], cam["cx"], cam["cy"])
    coord = synth.geodesic_coord(Y0)
    def mk():
        t = B.trackdlo(M, P["visibility_threshold"], P["beta"], P["lambda_"], P["alpha"], P["k_vis"], P["mu"], 50, P["tol"],
                       P["beta_pre_proc"], P["lambda_pre_proc"], P["lle_weight"], ctx=hip_ctx)
        t.initialize_nodes(Y0); t.initialize_geodesic_coord(coord)
        return t
    Xo, _ = oracle.depth_to_cloud(depth, mask, *args, 0.008)
    vis = np.arange(M, dtype=np.int32)
    a = mk(); a.tracking_step(Xo, vis, vis)
    b = mk()
    _, n, _ = hip_ctx.depth_to_cloud(0, depth, mask, *args, 0.008, fetch=False)
    assert n == Xo.shape[0]
    b.tracking_step(None, vis, vis)
    assert np.array_equal(a.get_tracking_result(), b.get_tracking_result())
    assert np.abs(a.get_tracking_result() - Y0).max() < 0.02


@pytest.mark.gpu
@pytest.mark.parametrize("N,M,mu,iters", [(400, 8, 0.05, 20), (5000, 30, 0.05, 10), (50000, 50, 0.1, 5), (777, 5, 0.0, 50), (300, 8, 0.05, 0)])
def test_reg_matches_oracle(hip_ctx, oracle, N, M, mu, iters):
    """reg (utils.cpp:21-82) on the device against the oracle, fp64: centroids within 1e-9 m,
    sigma2 within 1e-9 relative (summation order is the only difference)."""
    from trackdlo_amd import synth
    X, _, _ = synth.scene(N, max(M, 4), config=11, frame=N)
    X = X - np.array([0.0, 0.0, 0.6])
    Yo, so = oracle.reg(X, M, mu=mu, max_iter=iters)
    Yg, sg = hip_ctx.reg(X, M, mu=mu, max_iter=iters)
    assert np.all(np.isfinite(Yo))
    np.testing.assert_allclose(Yg, Yo, rtol=0, atol=1e-9)
    assert abs(sg - so) <= 1e-9 * so
    # resident-cloud form gives the same bits
    Yr, sr = hip_ctx.reg(None, M, mu=mu, max_iter=iters)
    assert np.array_equal(Yr, Yg) and sr == sg


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(64 * _SWEEP))
def test_randomised_configurations(hip_ctx, oracle, seed):
    """Seeded sweep over sizes, parameters and branches (visibility weighting, priors, LLE, carried-over sigma2, noise,
    clutter, both precisions): every draw must meet the stated tolerances against the oracle after a fixed number of
    iterations.  Draws stay in the regime the path is specified for (smooth chain, cloud on the chain)."""
    from trackdlo_amd import synth
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
    if use_lle:
        L = oracle.calc_lle_weights(Y0, 6); H = (np.eye(M) - L).T @ (np.eye(M) - L)
        kw.update(include_lle=True, beta=3.0, lambda_=1.0)
    s2 = float(rng.choice([0.0, 1e-4, 2e-5]))
    o = oracle.cpd_lle(X, Y0, s2, priors=pri, visible_nodes=vext, H=H, **kw)
    g = hip_ctx.cpd_lle(X, Y0, s2, _params(kw, prec), priors=pri, visible_nodes=vext, H=H)
    if not use_lle:
        _check(g, o, prec)
        return
    # LLE system A = lambda s2 I + (diag(P1) + s2 gamma H) G: the weights behind H come from rank-deficient local Gram
    # matrices (entries of H reach 1e10 on a few draws, cond(A) 1e13).  The gate is the stated tolerance unless
    # the ORACLE ITSELF is measurably less certain on this draw: its own rounding error (QR against the quadruple-precision
    # solve of the same system) and its sensitivity to the last bit of H (tests/test_solver_error.py explains both).
    ty, ts = _measured_gate(oracle, prec, lambda H_: oracle.cpd_lle(X, Y0, s2, priors=pri, visible_nodes=vext, H=H_, **kw), H, o)
    assert g["rc"] == 0 and g["iters"] == o["iters"] and g["n_kept"] == o["n_kept"] and np.all(np.isfinite(g["Y"]))
    assert np.abs(g["Y"] - o["Y"]).max() <= ty
    assert abs(g["sigma2"] - o["sigma2"]) <= ts * o["sigma2"]


@pytest.mark.gpu
@pytest.mark.parametrize("n_visible", [4, 5, 6, 7])
def test_tracking_step_with_few_visible_nodes(hip_ctx, oracle, n_visible):
    """Only a handful of nodes visible: the pre-processing registration runs on a 4..7-node chain, where the reference's
    LLE neighbour selection indexes out of bounds (trackdlo.cpp:92-117).  Product and oracle clip the neighbourhood; the
    step must run and agree (same H is injected so that the ill-conditioned weights do not enter the comparison)."""
    from trackdlo_amd import synth, binding as B
    M = 30
    P = synth.LAUNCH_PARAMS
    X, Y0, _ = synth.scene(3000, M, config=31, occlude=(n_visible / (M - 1) + 0.01, 1.0))
    coord = synth.geodesic_coord(Y0)
    vis = np.arange(n_visible, dtype=np.int32)
    Yg = Y0[vis]
    L = oracle.calc_lle_weights(Yg, 6); H = (np.eye(n_visible) - L).T @ (np.eye(n_visible) - L)
    t = B.trackdlo(M, P["visibility_threshold"], P["beta"], P["lambda_"], P["alpha"], P["k_vis"], P["mu"], 50, P["tol"],
                   P["beta_pre_proc"], P["lambda_pre_proc"], P["lle_weight"], ctx=hip_ctx, precision=B.PREC_F64)
    t.initialize_nodes(Y0); t.initialize_geodesic_coord(coord)
    o = oracle.Tracker(M, P["visibility_threshold"], P["beta"], P["lambda_"], P["alpha"], P["k_vis"], P["mu"], 50, P["tol"],
                       P["beta_pre_proc"], P["lambda_pre_proc"], P["lle_weight"])
    o.initialize_nodes(Y0); o.initialize_geodesic_coord(coord)
    t.tracking_step(X, vis, vis, H_pre=H)
    o.tracking_step(X, vis, vis, H_pre=H)
    np.testing.assert_allclose(t.get_tracking_result(), o.get_tracking_result(), rtol=0, atol=1e-6)
    # and without an injected H (weights computed inside the product): must simply run and stay finite
    t2 = B.trackdlo(M, P["visibility_threshold"], P["beta"], P["lambda_"], P["alpha"], P["k_vis"], P["mu"], 50, P["tol"],
                    P["beta_pre_proc"], P["lambda_pre_proc"], P["lle_weight"], ctx=hip_ctx)
    t2.initialize_nodes(Y0); t2.initialize_geodesic_coord(coord)
    t2.tracking_step(X, vis, vis)
    assert np.all(np.isfinite(t2.get_tracking_result()))


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(16 * _SWEEP))
def test_randomised_tracking_sequences(hip_ctx, oracle, seed):
    """Seeded sweep over tracking_step (trackdlo.cpp:900-999): random chain length, cloud size, a random occluded interval
    per frame (head, tail, middle, none, or only a few nodes left visible), small inter-frame motion, four frames with the
    tracker state carried over.  The visible sets come from the product's own visibility pre-pass; the LLE regulariser of
    the pre-processing registration is injected in both (its weights are ill-conditioned by construction).  Per frame: the
    same occlusion branch (number of priors), guide nodes, priors, nodes and sigma2 within the fp32-mode tolerances."""
    from trackdlo_amd import synth, binding as B
    rng = np.random.default_rng(31000 + seed)
    P = synth.LAUNCH_PARAMS
    M = int(rng.integers(12, 56))
    N = int(rng.integers(800, 6000))
    Y0 = synth.nodes(M); coord = synth.geodesic_coord(Y0)
    args = (M, P["visibility_threshold"], P["beta"], P["lambda_"], P["alpha"], P["k_vis"], P["mu"], 30, P["tol"], P["beta_pre_proc"],
            P["lambda_pre_proc"], P["lle_weight"])
    ref = oracle.Tracker(*args); ref.initialize_nodes(Y0); ref.initialize_geodesic_coord(coord)
    trk = B.trackdlo(*args, ctx=hip_ctx); trk.initialize_nodes(Y0); trk.initialize_geodesic_coord(coord)
    for frame in range(4):
        kind = int(rng.integers(0, 5))
        occl = None
        if kind == 1: occl = (0.0, float(rng.uniform(0.1, 0.4)))
        elif kind == 2: occl = (float(rng.uniform(0.6, 0.9)), 1.0)
        elif kind == 3:
            a = float(rng.uniform(0.2, 0.6)); occl = (a, a + float(rng.uniform(0.05, 0.3)))
        elif kind == 4: occl = (float(rng.uniform(0.15, 0.3)), 1.0)                       # only the head stays visible
        X, _, _ = synth.scene(N, M, config=70 + seed, frame=frame, occlude=occl, noise=0.0015,
                              shift=(0.0, 0.002 * (frame + 1), 0.0))
        Ycur = ref.get_tracking_result()
        hip_ctx.set_cloud(0, X)
        _, vis, vext = hip_ctx.visibility_prepass(0, Ycur, P["visibility_threshold"], 0.06, coord)
        if len(vis) < 4:
            continue                                                                       # the reference needs a chain to register
        Lg = oracle.calc_lle_weights(Ycur[vext], 6)
        Hpre = (np.eye(len(vext)) - Lg).T @ (np.eye(len(vext)) - Lg)
        Yprev, s2prev = ref.get_tracking_result(), ref.get_sigma2()
        snap = B.trackdlo(*args, ctx=hip_ctx); snap.copy_state_from(trk)              # the product's state before this frame
        try:
            ref.tracking_step(X, vis, vext, H_pre=Hpre)
        except Exception:
            with pytest.raises(B.TdloError):                                                # traverse_euclidean would read out of bounds:
                trk.tracking_step(X, vis, vext, None, 0, 0, H_pre=Hpre)                     # the product reports it as an error too
            break
        trk.tracking_step(X, vis, vext, None, 0, 0, H_pre=Hpre)

        # What the oracle itself is uncertain by on this frame (pre-processing registration with the LLE term, H from
        # rank-deficient local Gram matrices): its own rounding error (QR against the quadruple-precision solve) and its
        # sensitivity to the last bit of H -- measured by re-running the frame from the same state.  Gates are the stated
        # fp32-mode tolerances unless 8 x that measurement is larger.
        def rerun(Hx, solver):
            t2 = oracle.Tracker(*args); t2.initialize_nodes(Yprev); t2.initialize_geodesic_coord(coord); t2.set_sigma2(s2prev)
            oracle.set_solver(solver)
            try:
                t2.tracking_step(X, vis, vext, H_pre=Hx)
            finally:
                oracle.set_solver(0)
            return t2
        prng = np.random.default_rng(777 + seed)
        alts = [rerun(Hpre, 1)] + [rerun(Hpre * (1.0 + 2.220446049250313e-16 * prng.choice([-1.0, 1.0], size=Hpre.shape)), 0) for _ in range(2)]
        if any(a.stats_pre.iters != ref.stats_pre.iters or a.stats_main.iters != ref.stats_main.iters or
               a.get_correspondence_pairs().shape != ref.get_correspondence_pairs().shape for a in alts):
            # the last bit of H moves a stopping decision of the ORACLE: this frame pins nothing; carry on from the oracle's state
            assert np.all(np.isfinite(trk.get_tracking_result())) and np.isfinite(trk.get_sigma2())
            _SEQ_EXITS["oracle_undecided"] += 1
            break
        unc_y = max(np.abs(a.get_tracking_result() - ref.get_tracking_result()).max() for a in alts)
        unc_g = max(max(np.abs(a.get_guide_nodes() - ref.get_guide_nodes()).max(), np.abs(a.get_correspondence_pairs() - ref.get_correspondence_pairs()).max()) for a in alts)
        unc_s = max(abs(a.get_sigma2() - ref.get_sigma2()) / ref.get_sigma2() for a in alts)

        if trk.last_stats[0]["iters"] != ref.stats_pre.iters or trk.last_stats[1]["iters"] != ref.stats_main.iters:
            # The stopping rule (:424) thresholds a rounded quantity: with the fp32 E-step the criterion can pass tol one
            # iteration apart (2 of 160 sequences).  That it IS the fp32 rounding is shown, not assumed: the same frame from the
            # same state in TDLO_PREC_F64 must reproduce the oracle's iteration counts and meet the fp64 gates; the sequence
            # then continues from that state.
            t64 = B.trackdlo(*args, ctx=hip_ctx); t64.copy_state_from(snap); t64.set_precision(B.PREC_F64)
            t64.tracking_step(X, vis, vext, None, 0, 0, H_pre=Hpre)
            assert t64.last_stats[0]["iters"] == ref.stats_pre.iters and t64.last_stats[1]["iters"] == ref.stats_main.iters
            np.testing.assert_allclose(t64.get_tracking_result(), ref.get_tracking_result(), rtol=0, atol=max(1e-9, 8 * unc_y))
            assert trk.last_stats[0]["converged"] and trk.last_stats[1]["converged"]
            trk.copy_state_from(t64); trk.set_precision(B.PREC_F32)
            _SEQ_EXITS["f32_iteration_count"] += 1
            continue
        _SEQ_EXITS["compared"] += 1
        kp, kr = trk.get_correspondence_pairs(), ref.get_correspondence_pairs()
        assert kp.shape == kr.shape
        tol_pre = max(1e-5, 8 * unc_g)                    # outputs of the pre-processing registration: guide nodes, priors
        np.testing.assert_allclose(kp, kr, rtol=0, atol=tol_pre)
        np.testing.assert_allclose(trk.get_guide_nodes(), ref.get_guide_nodes(), rtol=0, atol=tol_pre)
        np.testing.assert_allclose(trk.get_tracking_result(), ref.get_tracking_result(), rtol=0, atol=max(1e-5, 8 * unc_y))
        assert abs(trk.get_sigma2() - ref.get_sigma2()) <= max(1e-3, 8 * unc_s) * ref.get_sigma2()


@pytest.mark.gpu
def test_random_sequences_rarely_leave_through_the_escape_hatches():
    """the two exits of the random-sequence test that compare nothing at the fp32 gates -- a frame on which the last bit of H
    moves the ORACLE's own stopping decision, and a frame whose fp32-mode iteration count differs from the oracle's by one (then re-run and
    gated in fp64) -- must stay rare: a regression that pushes many frames through them would otherwise go unseen.  Counted over the
    sequences of this session (16 sequences x up to 4 frames in the default sweep)."""
    total = sum(_SEQ_EXITS.values())
    if total == 0:
        pytest.skip("the random sequences did not run in this session")
    print("random-sequence frames:", dict(_SEQ_EXITS))
    assert _SEQ_EXITS["oracle_undecided"] <= max(2, total // 10), dict(_SEQ_EXITS)
    assert _SEQ_EXITS["f32_iteration_count"] <= max(2, total // 10), dict(_SEQ_EXITS)
    assert _SEQ_EXITS["compared"] >= (3 * total) // 4, dict(_SEQ_EXITS)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(12 * _SWEEP))
def test_randomised_depth_images(hip_ctx, oracle, seed):
    """Random image sizes, masks (rope, scattered salt pixels, blobs), depth noise, invalid depths and leaf sizes: the
    device voxel grid must reproduce the oracle bit for bit (count, order, values)."""
    from trackdlo_amd import synth
    rng = np.random.default_rng(52000 + seed)
    rows = int(rng.integers(17, 300)); cols = int(rng.integers(33, 400))
    depth, mask, cam, _ = synth.depth_scene(int(rng.integers(10, 40)), config=80 + seed, frame=seed, rows=rows, cols=cols,
                                            samples=60000, zero_depth_pixels=int(rng.integers(0, 4)))
    depth = depth.astype(np.int64) + rng.integers(-3, 4, size=depth.shape)                 # sensor noise in millimetres
    depth = np.clip(depth, 0, 65535).astype(np.uint16)
    salt = rng.random(mask.shape) < float(rng.choice([0.0, 0.001, 0.02]))                   # isolated false positives
    mask = np.where(salt, 255, mask).astype(np.uint8)
    if rng.integers(0, 2):
        r0, c0 = int(rng.integers(0, rows - 8)), int(rng.integers(0, cols - 8)); mask[r0:r0 + 8, c0:c0 + 8] = 255   # a blob
    leaf = float(rng.choice([0.003, 0.008, 0.02, 0.05, 0.3]))
    args = (cam["fx"], cam["fy"], cam["cx"], cam["cy"])
    Xo, nraw_o = oracle.depth_to_cloud(depth, mask, *args, leaf)
    Xg, n, nraw = hip_ctx.depth_to_cloud(0, depth, mask, *args, leaf)
    assert nraw == nraw_o == np.count_nonzero(mask) and n == Xo.shape[0]
    assert np.array_equal(Xg, Xo)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(6 * _SWEEP))
def test_randomised_batches(hip_ctx, oracle, seed):
    """Random batches: 2..8 frames of different sizes registered concurrently must equal, bit for bit, the same frames
    registered one at a time (both precisions, with and without priors / visibility weighting)."""
    from trackdlo_amd import synth
    rng = np.random.default_rng(77000 + seed)
    P = synth.LAUNCH_PARAMS
    F = int(rng.integers(2, 9)); M = int(rng.integers(8, 61)); prec = int(rng.integers(0, 2)); iters = int(rng.integers(2, 9))
    vis_on = bool(rng.integers(0, 2)) and M >= 12
    kw = dict(beta=P["beta"], lambda_=P["lambda_"], lle_weight=P["lle_weight"], mu=P["mu"], max_iter=iters, tol=0.0, include_lle=False,
              alpha=0.0, k_vis=P["k_vis"] if vis_on else 0.0, visibility_threshold=P["visibility_threshold"])
    params = _params(kw, prec)
    Ys, s2s, single = [], [], []
    vext = None
    for f in range(F):
        X, Y0, v = synth.scene(int(rng.integers(200, 9000)), M, config=90 + seed, frame=f, occlude=(0.4, 0.6) if vis_on else None)
        if vis_on: vext = synth.extend_visible(v, M, synth.geodesic_coord(Y0))
        hip_ctx.set_cloud(f, X)
        Ys.append(Y0); s2s.append(float(rng.choice([0.0, 1e-4])))
    for f in range(F):
        single.append(hip_ctx.cpd_lle_resident(f, Ys[f], s2s[f], params, visible_nodes=vext))
    out = hip_ctx.cpd_lle_batch(Ys, s2s, params, visible_nodes=vext)
    for f in range(F):
        assert np.array_equal(out["Y"][f], single[f]["Y"]) and out["sigma2"][f] == single[f]["sigma2"]
        assert out["stats"][f]["iters"] == single[f]["iters"]


# ---- multi-CU M-step (k_mstep_mcu, 60 < M <= 512 without the LLE term): one workgroup per 16 rows of the tableau,
# ---- pivot rows handed over between workgroups inside the launch

@pytest.mark.parametrize("M,F,tol", [(100, 5, 0.0), (200, 12, 2e-4), (300, 24, 0.0), (61, 9, 0.0), (512, 10, 0.0)])
def test_multi_cu_mstep_batches_equal_single_calls(M, F, tol):
    """Frames registered concurrently (grid.y = frame; 8+ frames on 2 / 4 stream groups; 24 frames x 19 row blocks are more
    workgroups than the GPU has CUs, so row blocks wait for workgroups that are not resident yet) must equal, bit for bit,
    the same frames registered one at a time, and repeated calls on a slot must repeat (generation counter of the flags)."""
    from trackdlo_amd import binding as B, synth
    P = synth.LAUNCH_PARAMS
    rng = np.random.default_rng(9100 + M + F)
    ctx = B.Context(device=0, max_frames=F, max_points=1 << 14, max_nodes=M)
    try:
        pr = B.make_params(P["beta"], P["lambda_"], P["lle_weight"], P["mu"], 30 if tol else 4, tol, False)
        Ys, s2s, single = [], [], []
        for f in range(F):
            X, Y0, _ = synth.scene(int(rng.integers(1500, 9000)), M, config=140 + F, frame=f)
            ctx.set_cloud(f, X)
            Ys.append(Y0); s2s.append(float(rng.choice([0.0, 1e-4])))
        for f in range(F):
            single.append(ctx.cpd_lle_resident(f, Ys[f], s2s[f], pr))
        for rep in range(2):
            out = ctx.cpd_lle_batch(Ys, s2s, pr)
            for f in range(F):
                assert out["stats"][f]["status"] == 0
                assert np.array_equal(out["Y"][f], single[f]["Y"]) and out["sigma2"][f] == single[f]["sigma2"]
                assert out["stats"][f]["iters"] == single[f]["iters"] and out["stats"][f]["converged"] == single[f]["converged"]
        again = ctx.cpd_lle_resident(0, Ys[0], s2s[0], pr)
        assert np.array_equal(again["Y"], single[0]["Y"]) and again["sigma2"] == single[0]["sigma2"]
    finally:
        ctx.close()


_ONE_WG_SCRIPT = r"""
import sys, numpy as np
sys.path.insert(0, sys.argv[1])
from trackdlo_amd import binding as B, synth
P = synth.LAUNCH_PARAMS
res = {}
for i, (N, M, prec, iters, pri, lle) in enumerate(eval(sys.argv[3])):
    ctx = B.Context(max_points=N, max_nodes=M)
    X, Y0, _ = synth.scene(N, M, config=150 + i)
    pr = B.make_params(P['beta'], P['lambda_'], P['lle_weight'], P['mu'], iters, 0.0, False, P['alpha'] if pri else 0.0, precision=prec)
    kw = {}
    if lle:     # the pre-processing registration of tracking_step (launch values), well-conditioned stand-in for H
        pr = B.make_params(3.0, 1.0, 10.0, 0.1, iters, 0.0, True, P['alpha'] if pri else 0.0, precision=prec)
        kw['H'] = np.eye(M) * 0.1 + 0.01 * np.diag(np.ones(M - 1), 1) + 0.01 * np.diag(np.ones(M - 1), -1)
    if pri:
        idx = np.arange(0, M, 7)
        kw['priors'] = np.column_stack([idx, Y0[idx] + 0.003])
    g = ctx.cpd_lle(X, Y0, 0.0, pr, **kw)
    res[f'Y{i}'] = g['Y']; res[f's{i}'] = np.array([g['sigma2'], g['iters'], g['status']])
    ctx.close()
np.savez(sys.argv[2], **res)
"""


def test_multi_cu_mstep_matches_one_workgroup_kernel(tmp_path):
    """k_mstep_mcu against k_mstep_big (TDLO_MSTEP_BIG=1wg, the whole elimination in one workgroup): the eliminations perform
    the same operations in the same order; the block partials and G W are summed in a different (fixed) order, so the
    results agree to rounding (observed <= 7e-15 m after 30 iterations at M = 300)."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cases = [(3000, 61, 0, 5, False, False), (3000, 130, 1, 4, False, False), (3000, 200, 0, 3, True, False), (6000, 300, 1, 8, False, False),
             (9000, 512, 0, 3, False, False), (70000, 90, 0, 6, False, False),
             # with the LLE term beyond 128 nodes: k_mstep_pivot_mcu against the one-workgroup k_mstep (TDLO_MSTEP_LLE=1wg)
             (4000, 129, 1, 3, False, True), (4000, 200, 0, 2, True, True), (5000, 300, 1, 2, False, True)]
    outs = []
    for mode in ("mcu", "1wg"):
        env = dict(os.environ)
        env.pop("TDLO_MSTEP_BIG", None)
        env.pop("TDLO_MSTEP_LLE", None)
        env["TDLO_MSTEP"] = "dense"              # the dense eliminations (comparators since round 2: the product path without LLE is the chain smoother)
        if mode == "1wg": env["TDLO_MSTEP_BIG"] = "1wg"; env["TDLO_MSTEP_LLE"] = "1wg"
        out = tmp_path / f"{mode}.npz"
        r = subprocess.run([sys.executable, "-c", _ONE_WG_SCRIPT, root, str(out), repr(cases)], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(np.load(out))
    a, b = outs
    for i in range(len(cases)):
        assert a[f"s{i}"][2] == 0 and b[f"s{i}"][2] == 0 and a[f"s{i}"][1] == b[f"s{i}"][1]
        # the pivoted eliminations of the LLE system (c = lambda sigma2 ~ 1e-5 against entries of order 100) order their row
        # operations differently; both are Gaussian elimination with back substitution (backward stable): stated fp64 tolerance
        ty, ts = (1e-9, 1e-7) if cases[i][5] else (1e-12, 1e-10)
        assert np.abs(a[f"Y{i}"] - b[f"Y{i}"]).max() <= ty
        assert abs(a[f"s{i}"][0] - b[f"s{i}"][0]) <= ts * b[f"s{i}"][0]


def test_multi_cu_mstep_nsplit_and_oracle_at_c5_nodes(hip_ctx, oracle):
    """M = 300 (BASELINE.json configs[4]): fp64 against the oracle, and the N-split interface (sums exported by the
    one-workgroup kernel, solved by the multi-CU kernel from the reduced sums) against the plain call."""
    from trackdlo_amd import binding as B, nsplit, synth
    P = synth.LAUNCH_PARAMS

    class Identity:
        def all_reduce_sum(self, a): return np.array(a, dtype=np.float64)
        def all_reduce_min(self, a): return np.array(a, dtype=np.float64)

    X, Y0, _ = synth.scene(8000, 300, config=8)
    kw = dict(beta=P["beta"], lambda_=P["lambda_"], lle_weight=P["lle_weight"], mu=P["mu"], max_iter=6, tol=0.0, include_lle=False,
              alpha=0.0, k_vis=0.0, visibility_threshold=P["visibility_threshold"])
    pr = _params(kw, 1)
    a = hip_ctx.cpd_lle(X, Y0, 0.0, pr)
    o = oracle.cpd_lle(X, Y0, 0.0, **kw)
    _check(a, o, 1)
    b = nsplit.cpd_lle_nsplit(nsplit.HipShard(hip_ctx, X), Identity(), Y0, 0.0, pr)
    assert np.abs(a["Y"] - b["Y"]).max() <= 1e-12 and abs(a["sigma2"] - b["sigma2"]) <= 1e-10 * a["sigma2"]
    assert b["iters"] == 6 and b["n_kept"] == a["n_kept"]


def test_multi_cu_mstep_hand_offs_under_uneven_load():
    from trackdlo_amd import binding as B
    prev_dense = B.mstep_dense(True)           # the multi-CU dense elimination (comparator since round 2: the product path is the chain smoother)
    try:
        _hand_offs_under_uneven_load()
    finally:
        B.mstep_dense(prev_dense)


def _hand_offs_under_uneven_load():
    """The in-launch hand-offs of k_mstep_mcu (write-through pivot rows + flag, L1-bypassing loads) must not depend on
    timing or placement: batches of M = 130 / 300 frames are registered while a second context (its own stream, driven from
    another thread) keeps the GPU busy with the E-steps of a 1 000 000-point cloud, and every frame of every repetition
    must equal, bit for bit, the same frame registered alone on an idle GPU."""
    import threading
    from trackdlo_amd import binding as B, synth
    P = synth.LAUNCH_PARAMS
    stop, running = threading.Event(), threading.Event()
    errors, calls = [], [0]

    def background():
        try:
            ctx = B.Context(device=0, max_frames=1, max_points=1 << 20, max_nodes=64)
            X, Y0, _ = synth.scene(1000000, 50, config=160)
            pr = B.make_params(P["beta"], P["lambda_"], P["lle_weight"], P["mu"], 20, 0.0, False)
            ctx.set_cloud(0, X)
            while not stop.is_set():
                ctx.cpd_lle_resident(0, Y0, 1e-4, pr)
                calls[0] += 1
                running.set()
            ctx.close()
        except Exception as e:      # pragma: no cover
            errors.append(e)

    for M, F, lle in ((130, 16, False), (300, 20, False), (200, 8, True)):      # the last: k_mstep_pivot_mcu (a hand-off per column)
        rng = np.random.default_rng(9300 + M)
        ctx = B.Context(device=0, max_frames=F, max_points=1 << 14, max_nodes=M)
        try:
            pr = B.make_params(P["beta"], P["lambda_"], P["lle_weight"], P["mu"], 5, 0.0, False, precision=1)
            Hk = {}
            if lle:
                pr = B.make_params(3.0, 1.0, 10.0, 0.1, 3, 0.0, True, precision=1)
                Hk = dict(H=np.eye(M) * 0.1 + 0.01 * np.diag(np.ones(M - 1), 1) + 0.01 * np.diag(np.ones(M - 1), -1))
            Ys, s2s, single = [], [], []
            for f in range(F):
                X, Y0, _ = synth.scene(int(rng.integers(2000, 12000)), M, config=170 + M, frame=f)
                ctx.set_cloud(f, X)
                Ys.append(Y0); s2s.append(0.0)
            if lle: s2s = [2e-5] * F
            for f in range(F):
                single.append(ctx.cpd_lle_resident(f, Ys[f], s2s[f], pr, **Hk))
            stop.clear(); running.clear()
            th = threading.Thread(target=background)
            th.start()
            try:
                assert running.wait(timeout=120) or errors
                c0 = calls[0]
                for rep in range(40):
                    out = ctx.cpd_lle_batch(Ys, s2s, pr, **Hk)
                    for f in range(F):
                        assert out["stats"][f]["status"] == 0
                        assert np.array_equal(out["Y"][f], single[f]["Y"]) and out["sigma2"][f] == single[f]["sigma2"], (M, rep, f)
                    g = ctx.cpd_lle_resident(rep % F, Ys[rep % F], s2s[rep % F], pr, **Hk)
                    assert np.array_equal(g["Y"], single[rep % F]["Y"])
                assert calls[0] > c0 + 3        # the other context did run beside the batches
            finally:
                stop.set()
                th.join(timeout=120)
            assert not errors, errors
        finally:
            ctx.close()


@pytest.mark.parametrize("dense", [True, False], ids=["dense", "band"])
@pytest.mark.parametrize("M,F", [(129, 3), (200, 9), (300, 5)])
def test_multi_cu_pivoted_mstep_with_lle(oracle, M, F, dense):
    """k_mstep_pivot_mcu (LLE term, more than 128 nodes: 16 rows per workgroup, pivot search across the workgroups, one
    hand-off per column; back substitution by the finishing workgroup): against the oracle at the stated fp64 tolerance
    (1e-9 m / 1e-7; the oracle's own error on these systems is 1e-11 m, tests/test_solver_error.py) -- and frames registered
    concurrently / repeatedly must reproduce the single call bit for bit.  Since round 3 a banded H like this one goes to
    k_mstep_band by default: the dense kernels are forced for one half of the cases, the banded solve takes the other (same assertions)."""
    from trackdlo_amd import binding as B, synth
    rng = np.random.default_rng(9500 + M)
    prev_dense = B.mstep_lle_dense(dense)
    ctx = B.Context(device=0, max_frames=F, max_points=1 << 14, max_nodes=M)
    try:
        kw = dict(beta=3.0, lambda_=1.0, lle_weight=10.0, mu=0.1, max_iter=3, tol=0.0, include_lle=True, alpha=0.0, k_vis=0.0, visibility_threshold=0.008)
        pr = _params(kw, 1)
        H = np.eye(M) * 0.1 + 0.01 * np.diag(np.ones(M - 1), 1) + 0.01 * np.diag(np.ones(M - 1), -1)
        Ys, s2s, single = [], [], []
        for f in range(F):
            X, Y0, _ = synth.scene(int(rng.integers(2000, 7000)), M, config=180 + M, frame=f, noise=0.004)
            ctx.set_cloud(f, X)
            Ys.append(Y0); s2s.append(2e-5)
            g = ctx.cpd_lle_resident(f, Y0, 2e-5, pr, H=H)
            single.append(g)
            if f == 0:
                assert (ctx.profile_iteration(1)[3] == "k_mstep_band") != dense
                o = oracle.cpd_lle(X, Y0, 2e-5, H=H, **kw)
                assert g["status"] == 0 and g["iters"] == o["iters"] and g["n_kept"] == o["n_kept"]
                assert np.abs(g["Y"] - o["Y"]).max() <= TOL[1][0] and abs(g["sigma2"] - o["sigma2"]) <= TOL[1][1] * o["sigma2"]
        for rep in range(3):
            out = ctx.cpd_lle_batch(Ys, s2s, pr, H=H)
            for f in range(F):
                assert out["stats"][f]["status"] == 0
                assert np.array_equal(out["Y"][f], single[f]["Y"]) and out["sigma2"][f] == single[f]["sigma2"]
    finally:
        ctx.close()
        B.mstep_lle_dense(prev_dense)


_RETRY_SCRIPT = r"""
import sys, numpy as np
sys.path.insert(0, sys.argv[1])
from trackdlo_amd import binding as B, synth
P = synth.LAUNCH_PARAMS
res = {}
for i, (N, M, iters, lle, F) in enumerate(eval(sys.argv[3])):
    ctx = B.Context(max_frames=F, max_points=N, max_nodes=M)
    pr = B.make_params(P['beta'], P['lambda_'], P['lle_weight'], P['mu'], iters, 0.0, False, precision=1)
    kw = {}
    if lle:
        pr = B.make_params(3.0, 1.0, 10.0, 0.1, iters, 0.0, True, precision=1)
        kw['H'] = np.eye(M) * 0.1 + 0.01 * np.diag(np.ones(M - 1), 1) + 0.01 * np.diag(np.ones(M - 1), -1)
    Ys = []
    for f in range(F):
        X, Y0, _ = synth.scene(N, M, config=190 + i, frame=f)
        ctx.set_cloud(f, X); Ys.append(Y0)
    if F == 1:
        g = ctx.cpd_lle_resident(0, Ys[0], 2e-5 if lle else 0.0, pr, **kw)
        res[f'Y{i}'] = g['Y']; res[f's{i}'] = np.array([g['sigma2'], g['iters'], g['status'], g['mstep_retries']])
    else:
        out = ctx.cpd_lle_batch(Ys, [2e-5 if lle else 0.0] * F, pr, **kw)
        res[f'Y{i}'] = np.stack(out['Y']); res[f's{i}'] = np.array([out['sigma2'][0], out['stats'][0]['iters'], max(s['status'] for s in out['stats']),
                                                                     sum(s['mstep_retries'] for s in out['stats'])])
    ctx.close()
np.savez(sys.argv[2], **res)
"""


def test_multi_cu_mstep_redoes_an_iteration_after_a_timed_out_hand_off(tmp_path, oracle):
    """a hand-off of k_mstep_mcu / k_mstep_pivot_mcu that runs into its time limit (a workgroup not co-resident with
    the others: a plain launch gives no such guarantee) must not poison the registration.  TDLO_MCU_FORCE_TIMEOUT=k makes the
    last row block's wait in iteration k behave as timed out: the finishing workgroup then publishes nothing, re-arms the
    sync words, and the one-workgroup elimination that follows in the stream redoes the iteration from the same sums.  The
    result must carry status 0, report the retry, and agree with the undisturbed run to rounding (the two eliminations
    perform the same operations; only G W / the block partials are summed in another fixed order)."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cases = [(4000, 130, 4, False, 1), (6000, 300, 3, False, 1), (3000, 90, 4, False, 5), (4000, 200, 3, True, 1), (3000, 129, 3, True, 3)]
    outs = {}
    for mode in ("plain", "forced"):
        env = dict(os.environ)
        env.pop("TDLO_MCU_FORCE_TIMEOUT", None)
        env["TDLO_MSTEP"] = "dense"              # the multi-CU dense elimination (comparator since round 2)
        env["TDLO_MSTEP_LLE"] = "dense"          # ... and the dense pivoted one for the LLE term (round 3: the banded L D L^T is the product path)
        if mode == "forced": env["TDLO_MCU_FORCE_TIMEOUT"] = "1"          # the second iteration
        out = tmp_path / f"{mode}.npz"
        r = subprocess.run([sys.executable, "-c", _RETRY_SCRIPT, root, str(out), repr(cases)], env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        outs[mode] = np.load(out)
    a, b = outs["plain"], outs["forced"]
    for i, (N, M, iters, lle, F) in enumerate(cases):
        assert a[f"s{i}"][2] == 0 and b[f"s{i}"][2] == 0                      # status
        assert a[f"s{i}"][3] == 0 and b[f"s{i}"][3] == F                      # one redone iteration per frame, none undisturbed
        assert a[f"s{i}"][1] == b[f"s{i}"][1] == iters
        ty, ts = (1e-9, 1e-7) if lle else (1e-12, 1e-10)
        assert np.abs(a[f"Y{i}"] - b[f"Y{i}"]).max() <= ty, (i, np.abs(a[f"Y{i}"] - b[f"Y{i}"]).max())
        assert abs(a[f"s{i}"][0] - b[f"s{i}"][0]) <= ts * a[f"s{i}"][0]


def test_reg_rejects_more_centroids_than_fit_the_lds(hip_ctx):
    """tdlo_reg beyond the LDS-derived limit (890 centroids) is TDLO_E_INVALID, not an opaque launch failure."""
    from trackdlo_amd import binding as B, synth
    X, _, _ = synth.scene(4000, 30, config=3)
    Y, s2 = hip_ctx.reg(X, 890, max_iter=1)
    assert np.isfinite(s2)
    with pytest.raises(B.TdloError) as e:
        hip_ctx.reg(X, 891, max_iter=1)
    assert e.value.code == B.TDLO_E_INVALID
    assert hip_ctx.reg(X, 8, max_iter=2)[0].shape == (8, 3)                  # the context stays usable


def test_nsplit_empty_cloud_leaves_the_context_reusable(hip_ctx):
    """an N-split registration whose global kept-point count is 0 raises TDLO_E_EMPTY and ABORTS the split
    (tdlo_split_abort), so that the same shard / context can register the next frame."""
    import torch
    from trackdlo_amd import binding as B, nsplit, synth
    P = synth.LAUNCH_PARAMS

    class Identity:
        def all_reduce_sum(self, a): return np.array(a, dtype=np.float64)
        def all_reduce_min(self, a): return np.array(a, dtype=np.float64)

    class LocalExchange:
        def __init__(self, M):
            self.buf = torch.zeros(5 * M + 2, dtype=torch.float64, device="cuda:0"); self.dmin = self.buf[:M]; self.sums = self.buf[M:]
        def all_reduce_min_dmin(self): pass
        def all_reduce_sum_sums(self): pass

    M = 40
    X, Y0, _ = synth.scene(5000, M, config=6)
    pr = B.make_params(P["beta"], P["lambda_"], P["lle_weight"], P["mu"], 6, 0.0, False)
    far = X + np.array([0.0, 0.0, 5.0])
    with pytest.raises(B.TdloError) as e:
        nsplit.cpd_lle_nsplit(nsplit.HipShard(hip_ctx, far), Identity(), Y0, 0.0, pr)
    assert e.value.code == B.TDLO_E_EMPTY
    xch = LocalExchange(M)
    shard = nsplit.HipDeviceShard(hip_ctx, far, xch)
    with pytest.raises(B.TdloError):
        nsplit.cpd_lle_nsplit_device(shard, xch, Identity(), Y0, 0.0, pr)
    shard2 = nsplit.HipDeviceShard(hip_ctx, X, xch)                           # bind() would fail inside a dangling registration
    b = nsplit.cpd_lle_nsplit_device(shard2, xch, Identity(), Y0, 0.0, pr)
    a = hip_ctx.cpd_lle(X, Y0, 0.0, pr)
    assert b["iters"] == 6 and np.abs(a["Y"] - b["Y"]).max() <= 1e-12


@pytest.mark.parametrize("prec", [0, 1], ids=["f32", "f64"])
def test_results_do_not_depend_on_the_estep_launch_geometry(prec):
    """The E-step's sums are converted to 64-bit fixed point at the grain of one wave x one 64-point batch and are integers from there
    on (csrc/tdlo_devcommon.h: acc_fix): however the batches are dealt out to waves and workgroups -- 196 workgroups of one batch per
    wave, 7 workgroups of 28 -- and in whatever order the workgroups' atomics arrive, every bit of the result is the same.  With
    visibility weighting, priors, and chains of 130 and 300 nodes (fp64: their first iterations' batches take the lane = node form, tdlo_estep_wide.h, whose
    grain is the same one batch) as well."""
    from trackdlo_amd import binding as B, synth
    P = synth.LAUNCH_PARAMS
    for N, M, occ in ((50000, 50, None), (20000, 45, (0.3, 0.5)), (30000, 130, None), (20000, 300, (0.3, 0.5))):
        X, Y0, vis = synth.scene(N, M, config=410 + M, occlude=occ)
        pr = B.make_params(P["beta"], P["lambda_"], P["lle_weight"], P["mu"], 12, 0.0, False, P["alpha"], P["k_vis"] if occ else 0.0,
                           P["visibility_threshold"], prec)
        idx = np.arange(2, M, 9)
        opt = dict(priors=np.column_stack([idx, Y0[idx] + 0.002]))
        if occ:
            opt["visible_nodes"] = np.asarray(synth.extend_visible(vis, M, synth.geodesic_coord(Y0)), dtype=np.int32)
        ref = None
        for blocks in (0, 131, 64, 7):
            ctx = B.Context(device=0, max_points=N, max_nodes=M, estep_blocks=blocks)
            try:
                g = ctx.cpd_lle(X, Y0, 0.0, pr, **opt)
            finally:
                ctx.close()
            assert g["rc"] == 0 and g["iters"] == 12
            if ref is None:
                ref = g
            else:
                assert np.array_equal(g["Y"], ref["Y"]) and g["sigma2"] == ref["sigma2"], (N, M, blocks)


@pytest.mark.gpu
def test_fp64_exp2_of_the_estep(hip_ctx):
    """The fp64 E-step's own 2^x (tdlo_devcommon.h: Num<double>::exp2 -- rint, Taylor polynomial of degree 13 as three-address FMAs on
    register-resident coefficients, v_ldexp_f64 for the range) against numpy's: within 1 ulp over the arguments the memberships
    produce (exponents <= 0 down to the denormals), exact powers of two at the integers, exact zero below 2^-1075 (what the
    node window's skipping of far nodes relies on)."""
    rng = np.random.default_rng(7)
    x = np.concatenate([-rng.random(200000) * 1000.0, -rng.random(50000) * 2.0, -1022.0 - rng.random(20000) * 52.0,
                        -np.arange(0.0, 1075.0), np.array([0.0, -0.5, -1e-300, -1074.0, -1074.9, -1075.5, -1080.0, -1100.0, -5000.0])])
    y = hip_ctx.debug_exp2(x)
    with np.errstate(under="ignore"):
        ref = np.exp2(x.astype(np.longdouble))
    normal = x > -1021.5
    rel = np.abs((y[normal].astype(np.longdouble) - ref[normal]) / ref[normal]).astype(np.float64)
    assert rel.max() <= 2.3e-16, rel.max()                                  # 1 ulp of a number just above a power of two is 2.2e-16
    sub = ~normal
    assert np.all(np.abs(y[sub].astype(np.longdouble) - ref[sub]) <= 5e-324 * 1.5)          # denormals: within one unit of the last (denormal) place
    ints = -np.arange(0.0, 1075.0)
    assert np.array_equal(hip_ctx.debug_exp2(ints), np.ldexp(1.0, ints.astype(int)))
    assert np.all(hip_ctx.debug_exp2(np.array([-1075.5, -1080.0, -1100.0, -5000.0])) == 0.0)


@pytest.mark.gpu
def test_chains_beyond_512_nodes(oracle):
    """Round 4 (the reference takes any num_of_nodes, trackdlo.cpp:30-46): up to 1024 nodes.  Beyond 512 the E-step runs with 16
    node chunks and the M-step without the LLE term is the one-direction smoother k_mstep_chain_long.  Here: the stopping rule through the results
    mailbox, a batch against single calls (bit for bit), the N-split with a communicator; what is NOT carried says so (the one-shot exchange, 1025 nodes)."""
    import os
    from trackdlo_amd import synth, binding as B
    P = synth.LAUNCH_PARAMS
    M, N, F = 600, 5000, 3
    ctx = B.Context(device=0, max_frames=F, max_points=N, max_nodes=M)
    try:
        kw = dict(beta=P["beta"], lambda_=P["lambda_"], lle_weight=P["lle_weight"], mu=P["mu"], max_iter=30, tol=2e-4, include_lle=False,
                  alpha=0.0, k_vis=0.0, visibility_threshold=P["visibility_threshold"])
        Xs, Ys = [], []
        for f in range(F):
            X, Y0, _ = synth.scene(N - 300 * f, M, config=77, frame=f)
            ctx.set_cloud(f, X); Xs.append(X); Ys.append(Y0)
        g = ctx.cpd_lle_resident(0, Ys[0], 0.0, _params(kw, 1))
        o = oracle.cpd_lle(Xs[0], Ys[0], 0.0, **kw)
        assert ctx.profile_iteration(1)[3] == "k_mstep_chain_long"
        _check(g, o, 1)
        assert 1 < g["iters"] < 30                                   # (the stopping rule ended it: the early-exit polling ran)
        kw0 = dict(kw, tol=0.0, max_iter=3)
        single = [ctx.cpd_lle_resident(f, Ys[f], 0.0, _params(kw0, 0)) for f in range(F)]
        b = ctx.cpd_lle_batch(np.asarray(Ys), np.zeros(F), _params(kw0, 0))
        for f in range(F):
            np.testing.assert_array_equal(b["Y"][f], single[f]["Y"])
            assert b["sigma2"][f] == single[f]["sigma2"]
        os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
        comm = ctx.rccl_comm_init(1, 0, B.rccl_unique_id())
        s = ctx.split_run(Ys[0], 0.0, _params(kw0, 1), comm=comm)
        p = ctx.cpd_lle_resident(0, Ys[0], 0.0, _params(kw0, 1))
        assert s["rc"] == 0 and np.abs(s["Y"] - p["Y"]).max() <= 1e-12 and s["iters"] == p["iters"]
        with pytest.raises(B.TdloError):                              # the one-shot exchange's inbox serves up to 512 nodes
            ctx.xch_create(1, M)
        ctx.xch_bind(0, [ctx.xch_create(1, 512)])
        r = ctx.split_run(Ys[0], 0.0, _params(kw0, 1), check=False)
        ctx.xch_unbind()
        assert r["rc"] == B.TDLO_E_INVALID
        X1, Y1, _ = synth.scene(2000, 1025, config=78)
        r = ctx.cpd_lle(X1, Y1, 0.0, _params(kw0, 0), check=False)
        assert r["rc"] == B.TDLO_E_INVALID
        ctx.set_cloud(0, Xs[0])                                       # (the refused call had uploaded its cloud into slot 0)
        g = ctx.cpd_lle_resident(0, Ys[0], 0.0, _params(kw0, 1))     # and the context stays usable
        np.testing.assert_array_equal(g["Y"], p["Y"])
    finally:
        ctx.close()


def test_fp64_mode_repeats_without_the_boost_when_a_share_is_refused(oracle):
    """in fp64 mode the E-step's range check runs against limits that follow sigma (IterState::sh_boost); a share beyond them ended the
    registration with TDLO_E_NUMERIC although the coarse limits of every other mode would have served it.  Now run_frames repeats such a call once
    without the boost.  TDLO_TEST_BOOST_FAIL=1 treats every fp64-mode call's first attempt as such a refusal: the repeat (coarse limits) must hold the
    fp64 gate against the oracle, count one retry per call, and leave fp32-mode calls alone."""
    import os
    from trackdlo_amd import binding as B, synth
    P = synth.LAUNCH_PARAMS
    X, Y0, v = synth.scene(6000, 45, config=81, occlude=(0.4, 0.6), outliers=5)
    vext = synth.extend_visible(v, 45, synth.geodesic_coord(Y0))
    kw = dict(beta=P["beta"], lambda_=P["lambda_"], lle_weight=P["lle_weight"], mu=P["mu"], max_iter=25, tol=0.0, include_lle=False, alpha=0.0,
              k_vis=P["k_vis"], visibility_threshold=P["visibility_threshold"])
    o = oracle.cpd_lle(X, Y0, 0.0, visible_nodes=vext, **kw)
    os.environ["TDLO_TEST_BOOST_FAIL"] = "1"
    try:
        ctx = B.Context(device=0)
    finally:
        os.environ.pop("TDLO_TEST_BOOST_FAIL", None)
    try:
        g = ctx.cpd_lle(X, Y0, 0.0, _params(kw, 1), visible_nodes=vext)
        _check(g, o, 1)
        assert int(ctx.lib.tdlo_debug_route_count(ctx.h, 10)) == 1
        g32 = ctx.cpd_lle(X, Y0, 0.0, _params(kw, 0), visible_nodes=vext)
        _check(g32, o, 0)
        assert int(ctx.lib.tdlo_debug_route_count(ctx.h, 10)) == 1
    finally:
        ctx.close()


def test_spin_ahead_loop_gives_the_ordinary_loops_bits(oracle):
    """VERDICT r05 item 3, the experiment (TDLO_SPIN_AHEAD=1, off by default -- measured 45 % SLOWER than the dependent dispatches it replaces,
    profiles/r06_spin_ahead_ab.txt): one frame's fixed-length loop with the E-steps on a second stream and every kernel parked on a device word until the
    kernel it depends on has reported.  It must give the ordinary loop's bits (and therefore the oracle's nodes at the gate), call after call on two slots,
    and leave calls it does not take (early exit, fp64 mode, the LLE term) to the ordinary loop."""
    import os
    from trackdlo_amd import binding as B, synth
    P = synth.LAUNCH_PARAMS
    N, M = 20000, 50
    pairs = [synth.scene(N, M, config=2, frame=f)[:2] for f in range(2)]
    kw = dict(beta=P["beta"], lambda_=P["lambda_"], lle_weight=P["lle_weight"], mu=P["mu"], max_iter=30, tol=0.0, include_lle=False, alpha=0.0, k_vis=0.0,
              visibility_threshold=P["visibility_threshold"])
    outs = {}
    for mode in ("0", "1"):
        os.environ["TDLO_SPIN_AHEAD"] = mode
        try:
            ctx = B.Context(device=0, max_frames=2, max_points=N, max_nodes=64, timing=False)
        finally:
            os.environ.pop("TDLO_SPIN_AHEAD", None)
        try:
            ctx.set_sort_reuse(False)
            for k in (0, 1):
                ctx.set_cloud(k, pairs[k][0])
            outs[mode] = [ctx.cpd_lle_resident(k, pairs[k][1], 0.0, _params(kw, 0)) for k in (0, 1, 0, 1, 0)]
            spins = int(ctx.lib.tdlo_debug_route_count(ctx.h, 11))
            assert spins == (5 if mode == "1" else 0)
            early = ctx.cpd_lle_resident(0, pairs[0][1], 0.0, _params(dict(kw, max_iter=50, tol=P["tol"]), 0))        # early exit: the ordinary loop
            f64 = ctx.cpd_lle_resident(0, pairs[0][1], 0.0, _params(kw, 1))                                             # fp64 mode: the ordinary loop
            assert int(ctx.lib.tdlo_debug_route_count(ctx.h, 11)) == spins and early["rc"] == 0 and f64["rc"] == 0
        finally:
            ctx.close()
    for a, b in zip(outs["0"], outs["1"]):
        assert np.array_equal(a["Y"], b["Y"]) and a["sigma2"] == b["sigma2"] and a["iters"] == b["iters"] == 30
    _check(outs["1"][0], oracle.cpd_lle(pairs[0][0], pairs[0][1], 0.0, **kw), 0)


@pytest.mark.gpu
def test_mstep_told_the_wrong_parity_reads_its_sums_again(oracle):
    """The host counts the iterations it enqueues and tells every plain chain M-step its iteration's parity, so that the kernel requests that parity's
    accumulator rows alone (half the lines of the round trip every iteration waits for) without waiting for the device's counter; the kernel compares the word
    with the counter and, if they disagree, reads the sums again the ordinary way.  TDLO_TEST_PARITY_FLIP=1 makes every hint wrong: the same bits, one frame
    (two row counts: fp32 at 50 nodes, fp64 at 130) and a batch; TDLO_ACC_ROWS=2|4|8 (how many replica rows the E-step spreads over): the same bits as well."""
    import subprocess, sys, json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import os, sys, json, hashlib
sys.path.insert(0, %r)
import numpy as np
from trackdlo_amd import binding as B, synth
P = synth.LAUNCH_PARAMS
out = {}
for name, N, M, prec in (("f32", 30000, 50, 0), ("f64", 20000, 130, 1)):
    X, Y0, _ = synth.scene(N, M, config=900 + M)
    pr = B.make_params(P["beta"], P["lambda_"], P["lle_weight"], P["mu"], 9, 0.0, False, precision=prec)
    c = B.Context(max_points=N, max_nodes=M, timing=False)
    g = c.cpd_lle(X, Y0, 0.0, pr); c.close()
    out[name] = [hashlib.sha1(np.ascontiguousarray(g["Y"]).tobytes()).hexdigest(), float(g["sigma2"]), int(g["iters"])]
F, N, M = 3, 20000, 50
scenes = [synth.scene(N, M, config=910, frame=f)[:2] for f in range(F)]
c = B.Context(max_frames=F, max_points=N, max_nodes=M, timing=False)
for f, (X, _) in enumerate(scenes): c.set_cloud(f, X)
pr = B.make_params(P["beta"], P["lambda_"], P["lle_weight"], P["mu"], 9, 0.0, False)
g = c.cpd_lle_batch([y for _, y in scenes], [0.0] * F, pr); c.close()
out["batch"] = [hashlib.sha1(np.ascontiguousarray(np.asarray(g["Y"])).tobytes()).hexdigest(), [float(v) for v in g["sigma2"]]]
print("RESULT " + json.dumps(out))
''' % root
    res = {}
    for tag, env in (("plain", {}), ("flip", {"TDLO_TEST_PARITY_FLIP": "1"}), ("rows2", {"TDLO_ACC_ROWS": "2"}), ("rows4", {"TDLO_ACC_ROWS": "4"}),
                     ("rows8flip", {"TDLO_ACC_ROWS": "8", "TDLO_TEST_PARITY_FLIP": "1"})):
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=dict(os.environ, **env))
        line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
        assert r.returncode == 0 and line, (tag, r.stdout[-2000:], r.stderr[-2000:])
        res[tag] = json.loads(line[0][7:])
    for tag in res:
        assert res[tag] == res["plain"], (tag, res[tag], res["plain"])
    assert res["plain"]["f32"][2] == 9 and res["plain"]["f64"][2] == 9
