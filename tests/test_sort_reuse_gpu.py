"""GPU suite for the sorted-cloud reuse (VERDICT r03 "What's missing" 5).

The reference prunes in every call (trackdlo.cpp:177-195).  The library may skip prune + counting sort when a registration starts from the
same nodes, on the same resident cloud, in the same precision as the previous registration of that slot (tdlo_cpd_lle_resident in
include/trackdlo_hip.h; tdlo_stats.sort_reused says whether a call did).  It must (a) change no bit of any result, (b) never hit after the
cloud, the nodes or the precision changed, or after an N-split used the slot, (c) be all-or-nothing inside a batch.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _mk(B, P, prec, *, lle=False, vis=False, iters=6, tol=0.0):
    if lle:
        return B.make_params(P["beta_pre_proc"], P["lambda_pre_proc"], P["lle_weight"], P["mu"], iters, tol, True, precision=prec)
    return B.make_params(P["beta"], P["lambda_"], P["lle_weight"], P["mu"], iters, tol, False, 0.0, P["k_vis"] if vis else 0.0,
                         P["visibility_threshold"], precision=prec)


def _same(a, b):
    np.testing.assert_array_equal(a["Y"], b["Y"])
    assert a["sigma2"] == b["sigma2"] and a["iters"] == b["iters"] and a["converged"] == b["converged"] and a["n_kept"] == b["n_kept"]


@pytest.mark.parametrize("prec", [0, 1], ids=["f32", "f64"])
@pytest.mark.parametrize("mode", ["plain", "vis", "lle", "priors"])
def test_reuse_on_and_off_give_the_same_bits(mode, prec):
    from trackdlo_amd import binding as B, synth
    P = synth.LAUNCH_PARAMS
    M, N = 45, 7000
    vis_on = mode == "vis"
    X, Y0, v = synth.scene(N, M, config=61, occlude=(0.4, 0.6) if vis_on else None, outliers=7)
    vext = synth.extend_visible(v, M, synth.geodesic_coord(Y0)) if vis_on else None
    pr = _mk(B, P, prec, lle=mode == "lle", vis=vis_on)
    pri = None
    if mode == "priors":
        idx = np.arange(0, M, 4)
        pri = np.concatenate([idx[:, None].astype(float), Y0[idx] + np.array([0, 0.003, 0.0])], axis=1)
        pr.alpha = P["alpha"]
    s2 = 2e-5 if mode == "lle" else 0.0
    ctx = B.Context(device=0, max_points=N, max_nodes=64)
    try:
        ctx.set_cloud(0, X)
        assert ctx.set_sort_reuse(False) is True          # the library's default is ON
        off = [ctx.cpd_lle_resident(0, Y0, s2, pr, priors=pri, visible_nodes=vext) for _ in range(3)]
        assert [g["sort_reused"] for g in off] == [0, 0, 0]
        ctx.set_sort_reuse(True)
        on = [ctx.cpd_lle_resident(0, Y0, s2, pr, priors=pri, visible_nodes=vext) for _ in range(3)]
        # the last call with the switch off left a sort for exactly these nodes behind: all three calls may reuse it
        assert [g["sort_reused"] for g in on] == [1, 1, 1]
        for g in off[1:] + on:
            _same(off[0], g)
        # other nodes on the same cloud: prunes again; and back
        Y1 = Y0 + np.array([0.0, 1e-3, 0.0])
        g1 = ctx.cpd_lle_resident(0, Y1, s2, pr, visible_nodes=vext)
        g0 = ctx.cpd_lle_resident(0, Y0, s2, pr, priors=pri, visible_nodes=vext)
        assert g1["sort_reused"] == 0 and g0["sort_reused"] == 0
        _same(off[0], g0)
        # a different registration (other parameters) of the same nodes reuses the sort too: prune and sort do not depend on them
        pr2 = _mk(B, P, prec, lle=mode != "lle", iters=4)
        a = ctx.cpd_lle_resident(0, Y0, 2e-5, pr2)
        ctx.set_sort_reuse(False)
        b = ctx.cpd_lle_resident(0, Y0, 2e-5, pr2)
        assert a["sort_reused"] == 1 and b["sort_reused"] == 0
        _same(a, b)
    finally:
        ctx.close()


def test_a_replaced_cloud_a_precision_switch_and_an_nsplit_do_not_hit():
    from trackdlo_amd import binding as B, synth
    P = synth.LAUNCH_PARAMS
    M, N = 40, 6000
    XA, Y0, _ = synth.scene(N, M, config=62, frame=0)
    XB, _, _ = synth.scene(N, M, config=62, frame=1)          # another cloud with the SAME number of points around the same chain
    assert XA.shape == XB.shape and not np.array_equal(XA, XB)
    ctx = B.Context(device=0, max_points=N, max_nodes=64)
    ref = B.Context(device=0, max_points=N, max_nodes=64)
    ref.set_sort_reuse(False)
    try:
        p32, p64 = _mk(B, P, 0), _mk(B, P, 1)
        ctx.set_cloud(0, XA)
        a = ctx.cpd_lle_resident(0, Y0, 0.0, p32)
        assert ctx.cpd_lle_resident(0, Y0, 0.0, p32)["sort_reused"] == 1
        # (1) tdlo_set_cloud with equal N
        ctx.set_cloud(0, XB)
        b = ctx.cpd_lle_resident(0, Y0, 0.0, p32)
        ref.set_cloud(0, XB)
        assert b["sort_reused"] == 0
        _same(b, ref.cpd_lle_resident(0, Y0, 0.0, p32))
        assert not np.array_equal(a["Y"], b["Y"])
        # (2) precision switch: the sorted cloud is stored in compute precision
        c = ctx.cpd_lle_resident(0, Y0, 0.0, p64)
        assert c["sort_reused"] == 0
        _same(c, ref.cpd_lle_resident(0, Y0, 0.0, p64))
        assert ctx.cpd_lle_resident(0, Y0, 0.0, p64)["sort_reused"] == 1
        assert ctx.cpd_lle_resident(0, Y0, 0.0, p32)["sort_reused"] == 0
        # (3) an N-split on the slot (its own prune works on shard-local counts): the plain call after it must prune again
        ctx.xch_bind(0, [ctx.xch_create(1, 64)])
        s = ctx.split_run(Y0, 0.0, p32)
        ctx.xch_unbind()
        d = ctx.cpd_lle_resident(0, Y0, 0.0, p32)
        assert d["sort_reused"] == 0
        _same(d, b); _same(s, b)
        # (4) the step-wise split entry points invalidate as well
        init = np.zeros(2)
        import ctypes as C
        Yc = np.asfortranarray(Y0)
        assert ctx.lib.tdlo_split_begin(ctx.h, Yc.ctypes.data_as(C.c_void_p), M, 0.0, C.byref(p32), None, 0, None, 0, None, init.ctypes.data_as(C.c_void_p)) == 0
        assert ctx.lib.tdlo_split_abort(ctx.h) == 0
        e = ctx.cpd_lle_resident(0, Y0, 0.0, p32)
        assert e["sort_reused"] == 0
        _same(e, b)
    finally:
        ctx.close(); ref.close()


def test_a_cloud_replaced_through_depth_to_cloud_does_not_hit(oracle):
    """tdlo_depth_to_cloud makes a new resident cloud: the sort of the previous one must not serve it, whether the new cloud has the same
    number of points (the same image again: an identical cloud, pruned again all the same) or not (another image)."""
    from trackdlo_amd import binding as B, synth
    P = synth.LAUNCH_PARAMS
    M = 30
    rows, cols = 120, 160
    fx = fy = 140.0; cx, cy = cols / 2, rows / 2

    def image(shift):
        depth = np.zeros((rows, cols), dtype=np.uint16); mask = np.zeros((rows, cols), dtype=np.uint8)
        u = np.arange(10, cols - 10)
        for k in range(-2, 3):
            vrow = np.clip((rows / 2 + 20 * np.sin((u + shift) / 25.0)).astype(int) + k, 0, rows - 1)
            depth[vrow, u] = 600 + (3 * np.cos(u / 15.0)).astype(int); mask[vrow, u] = 255
        return depth, mask

    ctx = B.Context(device=0, max_points=1 << 14, max_nodes=64)
    ref = B.Context(device=0, max_points=1 << 14, max_nodes=64)
    ref.set_sort_reuse(False)
    try:
        pr = _mk(B, P, 0, iters=5)
        dA, mA = image(0.0); dB, mB = image(4.0)
        XA, nA, _ = ctx.depth_to_cloud(0, dA, mA, fx, fy, cx, cy, 0.004)
        order = np.argsort(XA[:, 0]); Y0 = XA[order][np.linspace(0, nA - 1, M).astype(int)] + np.array([0, 0.004, 0.0])
        a = ctx.cpd_lle_resident(0, Y0, 0.0, pr)
        assert ctx.cpd_lle_resident(0, Y0, 0.0, pr)["sort_reused"] == 1
        XA2, nA2, _ = ctx.depth_to_cloud(0, dA, mA, fx, fy, cx, cy, 0.004)         # the same image: equal N, equal points
        assert nA2 == nA
        a2 = ctx.cpd_lle_resident(0, Y0, 0.0, pr)
        assert a2["sort_reused"] == 0
        _same(a, a2)
        XB, nB, _ = ctx.depth_to_cloud(0, dB, mB, fx, fy, cx, cy, 0.004)           # another image
        b = ctx.cpd_lle_resident(0, Y0, 0.0, pr)
        ref.set_cloud(0, XB)
        assert b["sort_reused"] == 0
        _same(b, ref.cpd_lle_resident(0, Y0, 0.0, pr))
    finally:
        ctx.close(); ref.close()


@pytest.mark.parametrize("F", [3, 9])
def test_a_batch_reuses_only_when_every_frame_can(F):
    """prune, scan and scatter are skipped per LAUNCH.  A batch in which one frame's cloud was replaced must prune ALL its frames
    again (a frame that skipped its scan while the scatter ran would be re-scattered from stale offsets); the results are the single calls'."""
    from trackdlo_amd import binding as B, synth
    P = synth.LAUNCH_PARAMS
    M = 35
    ctx = B.Context(device=0, max_frames=F, max_points=1 << 13, max_nodes=64)
    one = B.Context(device=0, max_frames=1, max_points=1 << 13, max_nodes=64)
    one.set_sort_reuse(False)
    try:
        pr = _mk(B, P, 0, iters=5)
        Xs, Ys = [], []
        for f in range(F):
            X, Y0, _ = synth.scene(3000 + 173 * f, M, config=63, frame=f)
            ctx.set_cloud(f, X); Xs.append(X); Ys.append(Y0)
        Ys = np.asarray(Ys); s2 = np.zeros(F)

        def singles():
            out = []
            for f in range(F):
                one.set_cloud(0, Xs[f]); out.append(one.cpd_lle_resident(0, Ys[f], 0.0, pr))
            return out

        b1 = ctx.cpd_lle_batch(Ys, s2, pr)
        assert [s["sort_reused"] for s in b1["stats"]] == [0] * F
        b2 = ctx.cpd_lle_batch(Ys, s2, pr)
        assert [s["sort_reused"] for s in b2["stats"]] == [1] * F
        sg = singles()
        for f in range(F):
            np.testing.assert_array_equal(b1["Y"][f], sg[f]["Y"]); np.testing.assert_array_equal(b2["Y"][f], sg[f]["Y"])
            assert b1["sigma2"][f] == sg[f]["sigma2"] == b2["sigma2"][f]
        # one frame's cloud replaced (same size, other points): nobody reuses
        Xn, _, _ = synth.scene(Xs[1].shape[0], M, config=63, frame=100)
        ctx.set_cloud(1, Xn); Xs[1] = Xn
        b3 = ctx.cpd_lle_batch(Ys, s2, pr)
        assert [s["sort_reused"] for s in b3["stats"]] == [0] * F
        sg = singles()
        for f in range(F):
            np.testing.assert_array_equal(b3["Y"][f], sg[f]["Y"])
            assert b3["sigma2"][f] == sg[f]["sigma2"]
        # one frame's NODES changed: the same
        Ys2 = Ys.copy(); Ys2[F - 1] += np.array([0.0, 5e-4, 0.0])
        b4 = ctx.cpd_lle_batch(Ys2, s2, pr)
        assert [s["sort_reused"] for s in b4["stats"]] == [0] * F
        one.set_cloud(0, Xs[F - 1])
        np.testing.assert_array_equal(b4["Y"][F - 1], one.cpd_lle_resident(0, Ys2[F - 1], 0.0, pr)["Y"])
        np.testing.assert_array_equal(b4["Y"][0], sg[0]["Y"])
    finally:
        ctx.close(); one.close()


@pytest.mark.parametrize("prec", [0, 1], ids=["f32", "f64"])
def test_tracking_step_reuses_only_with_every_node_visible(oracle, prec):
    """trackdlo.cpp:913-927 / :998: with every node visible both registrations of a frame start from Y_ -- the second reuses the first one's
    sort; with an occluded stretch the first registration runs on the visible sub-chain and nothing is reused.  Same results either way."""
    from trackdlo_amd import binding as B, synth
    P = synth.LAUNCH_PARAMS
    M, N = 45, 5000
    out = {}
    for reuse in (True, False):
        ctx = B.Context(device=0, max_points=N, max_nodes=64)
        ctx.set_sort_reuse(reuse)
        try:
            for occl in (None, (0.4, 0.55)):
                X, Y0, v = synth.scene(N, M, config=64, occlude=occl)
                coord = synth.geodesic_coord(Y0)
                v = np.arange(M, dtype=np.int32) if v is None else v
                vext = synth.extend_visible(v, M, coord)
                trk = B.trackdlo(M, P["visibility_threshold"], P["beta"], P["lambda_"], P["alpha"], P["k_vis"], P["mu"], 50, P["tol"], P["beta_pre_proc"],
                                 P["lambda_pre_proc"], P["lle_weight"], ctx=ctx, precision=prec)
                trk.initialize_nodes(Y0); trk.initialize_geodesic_coord(coord)
                flags = []
                for _ in range(3):
                    trk.tracking_step(X, v, vext)
                    flags.append([s["sort_reused"] for s in trk.last_stats])
                all_visible = len(vext) == M
                # (2: reused, and the main registration's set-up had ridden in the pre-processing registration's prologue -- tdlo_stats.sort_reused)
                assert flags == [[0, 2 if (reuse and all_visible) else 0]] * 3, (reuse, occl, flags)
                out[(reuse, occl)] = (trk.get_tracking_result(), trk.get_sigma2())
        finally:
            ctx.close()
    for occl in (None, (0.4, 0.55)):
        np.testing.assert_array_equal(out[(True, occl)][0], out[(False, occl)][0])
        assert out[(True, occl)][1] == out[(False, occl)][1]
