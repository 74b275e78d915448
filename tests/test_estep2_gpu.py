"""k_estep2 -- the E-step with two points per lane (trackdlo_amd/csrc/tdlo_estep2.hip; trackdlo.cpp:278-389) -- against the oracle.

By default the kernel serves clouds and batches that fill the GPU (>= 4096 x 64 points: C3's batch, C4); here TDLO_ESTEP2=1 puts it on small
and ragged inputs as well, so that every branch is compared with the oracle at sizes the oracle finishes in seconds: the cloud's last batch
with lanes that hold no point, windows wider than the membership tile (first iterations from sigma2 = 0: chunks behind the first are recomputed),
both tile heights, the visibility term, the end-node gap of :313-350 and the all-underflow column of :298-310.  fp32 mode's gate: 1e-5 m, 1e-3."""
import os

import numpy as np
import pytest

from conftest import case_kwargs, load_cases

pytestmark = pytest.mark.gpu
TOL_Y, TOL_S = 1e-5, 1e-3


def _ctx(B, mode, rows=None, **kw):
    old = {k: os.environ.get(k) for k in ("TDLO_ESTEP2", "TDLO_ESTEP2_ROWS")}
    os.environ["TDLO_ESTEP2"] = str(mode)
    if rows: os.environ["TDLO_ESTEP2_ROWS"] = str(rows)
    else: os.environ.pop("TDLO_ESTEP2_ROWS", None)
    try:
        return B.Context(device=0, **kw)
    finally:
        for k, v in old.items():
            if v is None: os.environ.pop(k, None)
            else: os.environ[k] = v


def _check(g, o):
    assert g["rc"] == 0
    assert g["iters"] == o["iters"] and g["converged"] == o["converged"] and g["n_kept"] == o["n_kept"]
    dy = float(np.abs(g["Y"] - o["Y"]).max()); ds = abs(g["sigma2"] - o["sigma2"]) / o["sigma2"]
    assert dy <= TOL_Y and ds <= TOL_S, (dy, ds)
    return dy, ds


@pytest.mark.parametrize("rows", [8, 16])
@pytest.mark.parametrize("N,M,iters,vis,s2", [
    (2000, 30, 20, False, 0.0),          # BASELINE.json configs[0]
    (1999, 45, 10, False, 0.0),          # ragged: the last batch has lanes without a point
    (64, 20, 5, False, 0.0),             # half a batch
    (129, 20, 5, False, 0.0),            # one point in the second batch
    (130, 8, 6, False, 0.0),             # the shortest chain the kernel takes
    (5000, 50, 30, True, 0.0),           # visibility weighting (:354-383)
    (7777, 64, 12, True, 1e-4),          # the longest chain it takes, sigma2 given
    (3000, 50, 40, False, 1e-6),         # sigma2 far below the data's: all-underflow columns in the first iteration (:298-310)
    (20000, 50, 25, False, 0.0),         # whole batches on one node: a node's P1 share of a batch reaches 128
], ids=lambda v: str(v))
def test_estep2_against_oracle(oracle, N, M, iters, vis, s2, rows):
    from trackdlo_amd import binding as B, synth
    P = synth.LAUNCH_PARAMS
    X, Y0, v = synth.scene(N, M, config=40 + M, occlude=(0.4, 0.6) if vis else None, outliers=5)
    vext = synth.extend_visible(v, M, synth.geodesic_coord(Y0)) if vis else None
    kw = dict(beta=P["beta"], lambda_=P["lambda_"], lle_weight=P["lle_weight"], mu=P["mu"], max_iter=iters, tol=0.0, include_lle=False, alpha=0.0,
              k_vis=P["k_vis"] if vis else 0.0, visibility_threshold=P["visibility_threshold"])
    o = oracle.cpd_lle(X, Y0, s2, visible_nodes=vext, **kw)
    ctx = _ctx(B, 1, rows)
    try:
        g = ctx.cpd_lle(X, Y0, s2, B.make_params(**kw), visible_nodes=vext)
        assert ctx.estep2_frames() == 1                                  # it WAS k_estep2
        _check(g, o)
        g2 = ctx.cpd_lle(X, Y0, s2, B.make_params(**kw), visible_nodes=vext)
        assert np.array_equal(g["Y"], g2["Y"]) and g["sigma2"] == g2["sigma2"]      # repeatable bit for bit (integer sums from one wave x one batch on)
    finally:
        ctx.close()


@pytest.mark.parametrize("name", sorted(load_cases()))
def test_estep2_committed_golden_cases(name):
    """The committed per-branch fixtures (tests/golden/oracle_cases.npz: priors, visibility, the end-node quirk, the LLE term, ...) through k_estep2."""
    from trackdlo_amd import binding as B
    c = load_cases()[name]
    kw = case_kwargs(c)
    M = c["Y0"].shape[0]
    ctx = _ctx(B, 1)
    try:
        g = ctx.cpd_lle(c["X"], c["Y0"], float(c["sigma2_in"]), B.make_params(**kw), priors=c.get("priors"), visible_nodes=c.get("vis"), H=c.get("H"))
        assert ctx.estep2_frames() == (1 if 8 <= M <= 64 else 0)
        o = dict(Y=c["Y"], sigma2=float(c["sigma2"]), iters=int(c["iters"]), converged=bool(c["converged"]), n_kept=int(c["n_kept"]))
        _check(g, o)
    finally:
        ctx.close()


def test_estep2_is_chosen_by_size_and_never_for_what_it_does_not_take():
    """Default selection (TDLO_ESTEP2 unset): k_estep2 for a cloud of at least 2048 x 64 points or a batch of that many in total, fp32 mode, chains of
    8 .. 64 nodes; k_estep for everything else -- one 50 000-point frame (C2), fp64 mode, chains beyond 64 nodes."""
    from trackdlo_amd import binding as B, synth
    P = synth.LAUNCH_PARAMS
    pr = lambda **k: B.make_params(P["beta"], P["lambda_"], P["lle_weight"], P["mu"], 2, 0.0, False, **k)
    old = os.environ.pop("TDLO_ESTEP2", None)
    try:
        ctx = B.Context(device=0, max_frames=8, max_points=262144, max_nodes=80)
        try:
            X, Y0, _ = synth.scene(50000, 50, config=2)
            ctx.cpd_lle(X, Y0, 0.0, pr()); assert ctx.estep2_frames() == 0                                # C2: one frame that cannot fill the GPU
            Xb, Yb, _ = synth.scene(131072, 50, config=4)
            ctx.cpd_lle(Xb, Yb, 0.0, pr()); assert ctx.estep2_frames() == 1                               # 2048 waves of 64 points: two per SIMD
            ctx.cpd_lle(Xb, Yb, 0.0, pr(precision=B.PREC_F64)); assert ctx.estep2_frames() == 1           # fp64 mode: k_estep
            Xl, Yl, _ = synth.scene(131072, 80, config=4)
            ctx.cpd_lle(Xl, Yl, 0.0, pr()); assert ctx.estep2_frames() == 1                               # 80 nodes: k_estep
            for f in range(8):
                ctx.set_cloud(f, synth.scene(40000, 50, config=3, frame=f)[0])
            Ys = [synth.scene(40000, 50, config=3, frame=f)[1] for f in range(8)]
            ctx.cpd_lle_batch(Ys, [0.0] * 8, pr()); assert ctx.estep2_frames() == 1 + 8                   # 8 x 625 = 5000 waves: the batch fills the GPU
            ctx.cpd_lle_batch(Ys[:3], [0.0] * 3, pr()); assert ctx.estep2_frames() == 9                   # 1875 waves: it does not
        finally:
            ctx.close()
    finally:
        if old is not None: os.environ["TDLO_ESTEP2"] = old


def test_batch_loop_in_one_launch_gives_the_launch_per_step_loops_bits():
    """The round-6 experiment k_batch_loop (TDLO_BATCH_PERSIST=1, off by default -- measured four times SLOWER at C3): a batch's whole
    fixed-length loop as one launch, tickets (iteration, frame, chunk) dealt to resident workgroups, the workgroup that completes a frame's E-step runs its
    M-step, per-frame progress counters in device memory.  It must give the bits of the launch-per-step loop on stream groups, report itself in
    tdlo_debug_route_count 12 with no fall-back (13), and leave batches it does not take (early exit, too small for k_estep2) to the ordinary loop."""
    from trackdlo_amd import binding as B, synth
    P = synth.LAUNCH_PARAMS
    F, N, M = 6, 44000, 50                      # 6 x 688 = 4128 waves of 64 points: the batch fills the GPU, k_estep2
    scenes = [synth.scene(N, M, config=3, frame=f)[:2] for f in range(F)]
    Ys = [y for _, y in scenes]
    pr = B.make_params(P["beta"], P["lambda_"], P["lle_weight"], P["mu"], 12, 0.0, False)
    pr_tol = B.make_params(P["beta"], P["lambda_"], P["lle_weight"], P["mu"], 50, P["tol"], False)
    outs = {}
    for mode in ("0", "1"):
        os.environ["TDLO_BATCH_PERSIST"] = mode
        try:
            ctx = B.Context(device=0, max_frames=F, max_points=N, max_nodes=M, timing=False)
        finally:
            os.environ.pop("TDLO_BATCH_PERSIST", None)
        try:
            ctx.set_sort_reuse(False)
            for f, (X, _) in enumerate(scenes):
                ctx.set_cloud(f, X)
            outs[mode] = [ctx.cpd_lle_batch(Ys, [0.0] * F, pr) for _ in range(2)]
            calls, fb = int(ctx.lib.tdlo_debug_route_count(ctx.h, 12)), int(ctx.lib.tdlo_debug_route_count(ctx.h, 13))
            assert (calls, fb) == ((2, 0) if mode == "1" else (0, 0)), (mode, calls, fb)
            early = ctx.cpd_lle_batch(Ys, [0.0] * F, pr_tol)                     # early exit: the ordinary loop
            small = ctx.cpd_lle_batch(Ys[:2], [0.0] * 2, pr)                     # 1376 waves: k_estep, the ordinary loop
            assert int(ctx.lib.tdlo_debug_route_count(ctx.h, 12)) == calls and all(s["status"] == 0 for s in list(early["stats"]) + list(small["stats"]))
        finally:
            ctx.close()
    for a, b in zip(outs["0"], outs["1"]):
        assert np.array_equal(np.asarray(a["Y"]), np.asarray(b["Y"])) and np.array_equal(a["sigma2"], b["sigma2"])
        assert [s["iters"] for s in a["stats"]] == [s["iters"] for s in b["stats"]] == [12] * F
