"""GPU suite for depth image -> cloud -> voxel grid in one launch (k_cloud_fused, round 5; SURVEY.md 8(f) row 2, trackdlo_node.cpp:195-241).

The one-launch kernel, the multi-launch form it replaces (TDLO_CLOUD_FUSED=0) and the CPU oracle perform the same float operations in the same order:
same count, same order, same doubles.  Sizes: the 640 x 480 stream of the synthetic scenes, the reference camera's 1280 x 720
(launch/realsense_node.launch:7-12), ragged images; the cases the one-launch kernel passes on (more than 32 704 masked pixels, too many cell-index
bits, PCL's pass-through) must come back through the multi-launch form with the same bits and be counted as passed on.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _ctx(B, fused=True, **kw):
    old = os.environ.get("TDLO_CLOUD_FUSED")
    try:
        if fused:
            os.environ.pop("TDLO_CLOUD_FUSED", None)
        else:
            os.environ["TDLO_CLOUD_FUSED"] = "0"
        return B.Context(device=0, **kw)
    finally:
        if old is None:
            os.environ.pop("TDLO_CLOUD_FUSED", None)
        else:
            os.environ["TDLO_CLOUD_FUSED"] = old


def _args(cam):
    return (cam["fx"], cam["fy"], cam["cx"], cam["cy"])


@pytest.mark.parametrize("shape,M,leaf,zero,taken", [
    (None, 50, 0.008, 0, True), (None, 30, 0.005, 0, True), (None, 30, 0.02, 7, True), ((720, 1280), 50, 0.008, 0, True), ((720, 1280), 30, 0.005, 0, True),
    ((120, 161), 50, 0.008, 0, True), ((97, 4099), 30, 0.008, 0, True), ((1, 77), 30, 0.008, 0, True),
    (None, 50, 0.0015, 0, False),          # 2.7 M cells: cell-index bits + rank bits > 32 -> passed on
    ((720, 1280), 50, 0.003, 0, False),    # 15 rank bits + 19 cell-index bits
])
def test_one_launch_equals_multi_launch_and_oracle(oracle, shape, M, leaf, zero, taken):
    from trackdlo_amd import binding as B, synth
    kw = dict(rows=shape[0], cols=shape[1]) if shape else {}
    depth, mask, cam, _ = synth.depth_scene(M, config=9, frame=M, zero_depth_pixels=zero, **kw)
    if shape and shape[0] < 100:
        mask[:] = 0; mask[0, ::3] = 255; mask[-1, -1] = 255          # (the rope is not in such an image: some pixels of the wall instead)
    Xo, nraw_o = oracle.depth_to_cloud(depth, mask, *_args(cam), leaf)
    one, multi = _ctx(B), _ctx(B, fused=False)
    try:
        for rep in range(2):                                          # (the second call finds the kernel's state words as the first one left them)
            Xg, n, nraw = one.depth_to_cloud(0, depth, mask, *_args(cam), leaf)
            Xm, nm, nrawm = multi.depth_to_cloud(0, depth, mask, *_args(cam), leaf)
            assert nraw == nrawm == nraw_o and n == nm == Xo.shape[0]
            assert np.array_equal(Xg, Xo) and np.array_equal(Xm, Xo)
        assert one.cloud_route_counts() == ([2, 0] if taken else [0, 2]) and multi.cloud_route_counts() == [0, 0]
        # the resident cloud is what came back
        p = B.make_params(0.35, 50000.0, 10.0, 0.1, 1, 0.0, False)
        if n >= 4:
            Y0 = Xo[np.linspace(0, n - 1, 8).astype(int)]
            a = one.cpd_lle_resident(0, Y0, 0.0, p, check=False); b = multi.cpd_lle_resident(0, Y0, 0.0, p, check=False)
            assert a["n_kept"] == b["n_kept"] and np.array_equal(a["Y"], b["Y"])
    finally:
        one.close(); multi.close()


def test_images_in_the_pinned_buffers_are_read_in_place(oracle):
    """tdlo_image_buffers: the caller writes depth and mask into the context's pinned buffers and the kernel reads them there; frames of two sizes
    in turn (the buffers are re-made for the larger one), and a frame changed in place between two calls."""
    from trackdlo_amd import binding as B, synth
    ctx = _ctx(B)
    try:
        for shape, frame in (((480, 640), 1), ((720, 1280), 2), ((480, 640), 3), ((720, 1280), 4)):
            depth, mask, cam, _ = synth.depth_scene(40, config=9, frame=frame, rows=shape[0], cols=shape[1])
            d, m = ctx.image_buffers(*shape)
            d[:] = depth; m[:] = mask
            Xo, nraw_o = oracle.depth_to_cloud(depth, mask, *_args(cam), 0.008)
            Xg, n, nraw = ctx.depth_to_cloud(0, d, m, *_args(cam), 0.008)
            assert nraw == nraw_o and np.array_equal(Xg, Xo)
            m[: shape[0] // 2] = 0; mask[: shape[0] // 2] = 0          # the upper half of the frame loses its segmentation
            Xo, nraw_o = oracle.depth_to_cloud(depth, mask, *_args(cam), 0.008)
            Xg, n, nraw = ctx.depth_to_cloud(0, d, m, *_args(cam), 0.008)
            assert nraw == nraw_o and np.array_equal(Xg, Xo)
        assert ctx.cloud_route_counts() == [8, 0]
    finally:
        ctx.close()


@pytest.mark.parametrize("seed", range(6))
def test_random_images(oracle, seed):
    """Speckle masks over random depth: many cells with one point, cells whose points lie rows apart, invalid depth, the box spanning the
    whole frustum -- taken by the one-launch kernel whenever the bits allow, equal to the oracle either way."""
    from trackdlo_amd import binding as B, synth
    rng = np.random.default_rng(500 + seed)
    rows, cols = [(480, 640), (720, 1280), (333, 517)][seed % 3]
    cam = dict(fx=600.0 * cols / 640, fy=600.0 * cols / 640, cx=cols / 2.0 - 3.5, cy=rows / 2.0 + 1.25)
    near = rng.integers(400, 800)
    depth = (near + rng.integers(0, [30, 200, 1500][seed % 3], size=(rows, cols))).astype(np.uint16)
    mask = (rng.random((rows, cols)) < [0.02, 0.03, 0.15][seed // 3 % 3]).astype(np.uint8) * rng.integers(1, 256, size=(rows, cols)).astype(np.uint8)
    depth[rng.random((rows, cols)) < 0.001] = 0
    leaf = [0.008, 0.02, 0.05][seed % 3]
    Xo, nraw_o = oracle.depth_to_cloud(depth, mask, *_args(cam), leaf)
    ctx = _ctx(B)
    try:
        Xg, n, nraw = ctx.depth_to_cloud(0, depth, mask, *_args(cam), leaf)
        assert nraw == nraw_o and n == Xo.shape[0] and np.array_equal(Xg, Xo)
        assert sum(ctx.cloud_route_counts()) == 1
    finally:
        ctx.close()


def test_cases_the_one_launch_kernel_passes_on(oracle):
    from trackdlo_amd import binding as B, synth
    depth, mask, cam, _ = synth.depth_scene(30, config=9)
    ctx = _ctx(B)
    try:
        X, n, nraw = ctx.depth_to_cloud(0, depth, np.zeros_like(mask), *_args(cam), 0.008)      # nothing segmented: the kernel itself reports 0 points
        assert n == 0 and nraw == 0 and X.shape == (0, 3) and ctx.cloud_route_counts() == [1, 0]
        Xo, _ = oracle.depth_to_cloud(depth, mask, *_args(cam), 1e-5)                           # cell count overflows int32: PCL's pass-through
        X, n, nraw = ctx.depth_to_cloud(0, depth, mask, *_args(cam), 1e-5)
        assert n == nraw == Xo.shape[0] and np.array_equal(X, Xo) and ctx.cloud_route_counts() == [1, 1]
        full = np.full_like(mask, 255)                                                          # every pixel: 307 200 > 32 704
        Xo, _ = oracle.depth_to_cloud(depth, full, *_args(cam), 0.02)
        X, n, nraw = ctx.depth_to_cloud(0, depth, full, *_args(cam), 0.02)
        assert nraw == depth.size and np.array_equal(X, Xo) and ctx.cloud_route_counts() == [1, 2]
        exact = np.zeros_like(mask); exact.reshape(-1)[np.arange(32704) * 9] = 1                # exactly the kernel's limit, then one more
        for extra in (0, 1):
            if extra:
                exact.reshape(-1)[5] = 1
            Xo, nr = oracle.depth_to_cloud(depth, exact, *_args(cam), 0.05)
            X, n, nraw = ctx.depth_to_cloud(0, depth, exact, *_args(cam), 0.05)
            assert nraw == nr == 32704 + extra and np.array_equal(X, Xo)
        assert ctx.cloud_route_counts() == [2, 3]
        one = np.zeros_like(mask); one[100, 200] = 255                                          # a single pixel
        Xo, _ = oracle.depth_to_cloud(depth, one, *_args(cam), 0.008)
        X, n, _ = ctx.depth_to_cloud(0, depth, one, *_args(cam), 0.008)
        assert n == 1 and np.array_equal(X, Xo) and ctx.cloud_route_counts() == [3, 3]
    finally:
        ctx.close()


def test_one_finishing_workgroup_and_the_team_give_the_same_bits(oracle):
    """The one-launch kernel in its two forms: phase B by the last EIGHT workgroups as a team (k_cloud_team, the default) and by the one workgroup with
    the last ticket (k_cloud_fused, TDLO_CLOUD_TEAM=0).  Same words, same stable order, same float sums: the same doubles, at both image sizes and on a
    speckle mask (every point its own cell: the team's slices end inside runs of length one)."""
    from trackdlo_amd import binding as B, synth
    old = os.environ.get("TDLO_CLOUD_TEAM")
    try:
        os.environ.pop("TDLO_CLOUD_TEAM", None)
        team = _ctx(B)
        os.environ["TDLO_CLOUD_TEAM"] = "0"
        one = _ctx(B)
    finally:
        if old is None:
            os.environ.pop("TDLO_CLOUD_TEAM", None)
        else:
            os.environ["TDLO_CLOUD_TEAM"] = old
    try:
        rng = np.random.default_rng(77)
        for shape, leaf in (((480, 640), 0.008), ((720, 1280), 0.008), ((720, 1280), 0.05), ((90, 100), 0.02)):
            depth, mask, cam, _ = synth.depth_scene(40, config=9, frame=shape[0], rows=shape[0], cols=shape[1])
            if shape[0] < 100:
                mask[:] = (rng.random(shape) < 0.3) * 255
            for speckle in (False, True):
                if speckle:
                    mask = ((rng.random(shape) < 0.02) * 255).astype(np.uint8)
                    depth = (500 + rng.integers(0, 300, size=shape)).astype(np.uint16)
                Xo, nraw_o = oracle.depth_to_cloud(depth, mask, *_args(cam), leaf)
                Xt, nt, nrawt = team.depth_to_cloud(0, depth, mask, *_args(cam), leaf)
                X1, n1, nraw1 = one.depth_to_cloud(0, depth, mask, *_args(cam), leaf)
                assert nrawt == nraw1 == nraw_o and nt == n1 == Xo.shape[0]
                assert np.array_equal(Xt, Xo) and np.array_equal(X1, Xo)
        assert team.cloud_route_counts()[0] >= 6 and one.cloud_route_counts()[0] >= 6
    finally:
        team.close(); one.close()


def test_a_team_that_loses_a_member_gives_the_launch_up(oracle):
    """Every wait inside the team is bounded: with a member that never arrives (TDLO_CLOUD_TEAM_FORCE_TIMEOUT=2: the process's second team launch) the
    others give the launch up after 2 s, the last one reports that, the host initialises the state words again and runs the multi-launch form -- the
    caller gets the same cloud, only late; the launches before and after are served by the team."""
    import subprocess, sys, textwrap
    code = textwrap.dedent("""
        import os, sys, time
        sys.path.insert(0, os.getcwd())
        import numpy as np
        from oracle import ref_cpu
        from trackdlo_amd import binding as B, synth
        depth, mask, cam, _ = synth.depth_scene(40, config=9, frame=5)
        a = (cam["fx"], cam["fy"], cam["cx"], cam["cy"])
        Xo, _ = ref_cpu.depth_to_cloud(depth, mask, *a, 0.008)
        ctx = B.Context(device=0)
        for k in range(4):
            t0 = time.perf_counter(); X, n, _ = ctx.depth_to_cloud(0, depth, mask, *a, 0.008); dt = time.perf_counter() - t0
            assert np.array_equal(X, Xo), k
            assert (dt > 1.9) == (k == 1), (k, dt)
        assert ctx.cloud_route_counts() == [3, 1], ctx.cloud_route_counts()
        print("OK")
    """)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, TDLO_CLOUD_TEAM_FORCE_TIMEOUT="2")
    env.pop("TDLO_CLOUD_TEAM", None); env.pop("TDLO_CLOUD_FUSED", None)
    r = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "OK" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])


@pytest.mark.parametrize("shape,M,leaf,rides", [
    (None, 30, 0.008, True), (None, 45, 0.008, True), (None, 64, 0.005, True), ((720, 1280), 50, 0.008, True), ((120, 161), 30, 0.008, True),
    (None, 65, 0.008, False),              # more than 64 nodes: the two steps one behind the other
    (None, 50, 0.0015, False),             # the one-launch kernel passes the frame on (cell-index bits): multi-launch form, then the pre-pass
])
def test_frame_cloud_and_visibility_prepass_in_one_launch(oracle, shape, M, leaf, rides):
    """tdlo_depth_to_cloud_visibility (trackdlo_node.cpp:195-277, :345-360): the visibility pre-pass of the tracker's nodes rides in the depth -> cloud team
    kernel -- every team member takes the minima over the centroids it has just formed -- and gives the bits of tdlo_depth_to_cloud followed by
    tdlo_visibility_prepass (which are the oracle's distances to 1e-12 m and its visible sets); the cloud left in the slot is the same one."""
    from trackdlo_amd import binding as B, synth
    kw = dict(rows=shape[0], cols=shape[1]) if shape else {}
    depth, mask, cam, Y0 = synth.depth_scene(M, config=9, frame=M, **kw)
    if shape and shape[0] < 200:
        mask[:] = 0; mask[5, ::3] = 255
    coord = synth.geodesic_coord(Y0)
    rng = np.random.default_rng(M)
    a, b = _ctx(B), _ctx(B)
    try:
        for rep in range(3):
            Y = Y0 + rng.normal(0, 0.003, size=Y0.shape) + (np.array([0.0, 0.0, 0.02 * rep]) if rep else 0.0)      # (rep 2: some nodes beyond the threshold)
            d1, v1, e1, n1, nraw1 = a.depth_to_cloud_visibility(0, depth, mask, *_args(cam), leaf, Y, 0.008, 0.06, coord)
            _, n2, nraw2 = b.depth_to_cloud(0, depth, mask, *_args(cam), leaf, fetch=False)
            d2, v2, e2 = b.visibility_prepass(0, Y, 0.008, 0.06, coord)
            assert n1 == n2 and nraw1 == nraw2
            assert np.array_equal(d1, d2) and np.array_equal(v1, v2) and np.array_equal(e1, e2)
            if n1 >= 4:                    # the cloud left in the slot is the same one: a registration on it gives the same bits
                p = B.make_params(0.35, 50000.0, 10.0, 0.1, 2, 0.0, False)
                ra = a.cpd_lle_resident(0, Y0[:8], 0.0, p, check=False); rb = b.cpd_lle_resident(0, Y0[:8], 0.0, p, check=False)
                assert ra["rc"] == rb["rc"] and np.array_equal(ra["Y"], rb["Y"]) and ra["n_kept"] == rb["n_kept"]
        assert a.cloud_vis_rides() == (3 if rides else 0) and b.cloud_vis_rides() == 0
        # ... and against the oracle: the cloud bit for bit, the distances to rounding, the sets exactly
        Xo, _ = oracle.depth_to_cloud(depth, mask, *_args(cam), leaf)
        do, viso, exto = oracle.visibility_prepass(np.asfortranarray(Xo), Y, 0.008, 0.06, coord)
        np.testing.assert_allclose(d1, do, rtol=0, atol=1e-12)
        assert np.array_equal(v1, viso) and np.array_equal(e1, exto)
        # the ordinary pre-pass on the same context afterwards finds its state armed
        d3, v3, e3 = a.visibility_prepass(0, Y, 0.008, 0.06, coord)
        assert np.array_equal(d3, d1)
    finally:
        a.close(); b.close()


def test_frame_prepass_survives_a_team_that_gives_up(monkeypatch):
    """The team kernel's launch is abandoned (test hook: a member never arrives): the frame comes back through the multi-launch form and the pre-pass runs
    behind it -- the same numbers, and the pre-pass's device state is armed again for the next frame, which rides."""
    import subprocess, sys, textwrap
    code = textwrap.dedent('''
        import numpy as np
        from trackdlo_amd import binding as B, synth
        depth, mask, cam, Y0 = synth.depth_scene(30, config=9, frame=3)
        coord = synth.geodesic_coord(Y0)
        args = (cam["fx"], cam["fy"], cam["cx"], cam["cy"])
        a = B.Context(device=0); b = B.Context(device=0)
        out = []
        for rep in range(3):
            d1, v1, e1, n1, _ = a.depth_to_cloud_visibility(0, depth, mask, *args, 0.008, Y0, 0.008, 0.06, coord)
            out.append((d1, v1, e1, n1))
        _, n2, _ = b.depth_to_cloud(0, depth, mask, *args, 0.008, fetch=False)
        d2, v2, e2 = b.visibility_prepass(0, Y0, 0.008, 0.06, coord)
        for d1, v1, e1, n1 in out:
            assert n1 == n2 and np.array_equal(d1, d2) and np.array_equal(v1, v2) and np.array_equal(e1, e2)
        print("rides", a.cloud_vis_rides(), "routes", a.cloud_route_counts())
    ''')
    env = dict(os.environ, TDLO_CLOUD_TEAM_FORCE_TIMEOUT="2", PYTHONPATH=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "rides 2 routes [2, 1]" in r.stdout, r.stdout


def test_the_callback_in_one_call_equals_its_three_steps():
    """tdlo_tracker_frame_from_depth (trackdlo_node.cpp:195-369): images in, nodes out -- the bits of tdlo_depth_to_cloud, tdlo_visibility_prepass and
    tdlo_tracker_tracking_step(X = NULL) called one after the other, over a short sequence (the rope drifts, one frame hides a stretch of it); a frame
    whose mask is empty, or that leaves no node visible, is TDLO_E_EMPTY and leaves the tracker as it was."""
    from trackdlo_amd import binding as B, synth
    P = synth.LAUNCH_PARAMS
    M = 30
    a, b = _ctx(B), _ctx(B)
    try:
        depth0, mask0, cam, Y0 = synth.depth_scene(M, config=9, frame=0)
        coord = synth.geodesic_coord(Y0)
        targs = (M, P["visibility_threshold"], P["beta"], P["lambda_"], P["alpha"], P["k_vis"], P["mu"], 30, P["tol"], P["beta_pre_proc"], P["lambda_pre_proc"], P["lle_weight"])
        ta, tb = B.trackdlo(*targs, ctx=a), B.trackdlo(*targs, ctx=b)
        for t in (ta, tb):
            t.initialize_nodes(Y0); t.initialize_geodesic_coord(coord)
        da, ma = a.image_buffers(*depth0.shape)
        for fr in range(5):
            depth, mask, _, _ = synth.depth_scene(M, config=9, frame=fr)
            if fr == 3:
                mask = mask.copy(); mask[:, 300:340] = 0                 # a stretch of the rope hidden: some nodes beyond the threshold
            da[:] = depth; ma[:] = mask
            va, ea, na, nra = ta.frame_from_depth(da, ma, *_args(cam), 0.008, 0.06)
            _, nb, nrb = b.depth_to_cloud(0, depth, mask, *_args(cam), 0.008, fetch=False)
            _, vb, eb = b.visibility_prepass(0, tb.get_tracking_result(), P["visibility_threshold"], 0.06, coord)
            tb.tracking_step(None, vb, eb)
            assert na == nb and nra == nrb and np.array_equal(va, vb) and np.array_equal(ea, eb)
            assert np.array_equal(ta.get_tracking_result(), tb.get_tracking_result()) and ta.get_sigma2() == tb.get_sigma2()
            assert [s["iters"] for s in ta.last_stats] == [s["iters"] for s in tb.last_stats]
            if fr == 3:
                assert len(va) < M
        assert a.cloud_vis_rides() == 5
        Yk, sk = ta.get_tracking_result().copy(), ta.get_sigma2()
        with pytest.raises(B.TdloError) as e1:
            ta.frame_from_depth(depth0, np.zeros_like(mask0), *_args(cam), 0.008, 0.06)
        assert e1.value.code == B.TDLO_E_EMPTY
        far = depth0.copy(); far[mask0 > 0] += 400                          # the rope 0.4 m further away: a cloud, but no node near it
        with pytest.raises(B.TdloError) as e2:
            ta.frame_from_depth(far, mask0, *_args(cam), 0.008, 0.06)
        assert e2.value.code == B.TDLO_E_EMPTY
        assert np.array_equal(ta.get_tracking_result(), Yk) and ta.get_sigma2() == sk
    finally:
        a.close(); b.close()


def _crossing_rope(M, samples=300000, radius=0.005, seed=3):
    """A rope that crosses itself in the image: the inner loop of a limacon r = 0.1 + 0.2 cos(theta) passes the origin twice, 6 mm apart in depth --
    the farther branch is within the visibility threshold (8 mm) of the nearer branch's points there, so only the painter test can tell them apart.
    Returns (depth, mask, cam, Y0 [M x 3], proj 3 x 4)."""
    from trackdlo_amd import synth
    cam = dict(synth.CAMERA)
    def centre(th):
        r = 0.1 + 0.2 * np.cos(th)
        return np.stack([r * np.cos(th) - 0.12, r * np.sin(th), 0.55 + 0.003 * th], axis=1)
    th = np.linspace(0.1, 2 * np.pi - 0.1, 4000)
    c = centre(th)
    arc = np.concatenate([[0.0], np.cumsum(np.linalg.norm(np.diff(c, axis=0), axis=1))])
    Y0 = np.stack([np.interp(np.linspace(0, arc[-1], M), arc, c[:, k]) for k in range(3)], axis=1)
    rng = np.random.default_rng(seed)
    pc = centre(rng.uniform(0.1, 2 * np.pi - 0.1, samples))
    ang = rng.random(samples) * 2 * np.pi; rad = radius * np.sqrt(rng.random(samples))
    p = pc + np.stack([rad * np.cos(ang), rad * np.sin(ang), np.zeros(samples)], axis=1)
    u = np.rint(p[:, 0] * cam["fx"] / p[:, 2] + cam["cx"]).astype(np.int64); v = np.rint(p[:, 1] * cam["fy"] / p[:, 2] + cam["cy"]).astype(np.int64)
    ok = (u >= 0) & (u < cam["cols"]) & (v >= 0) & (v < cam["rows"])
    zmm = np.clip(np.rint(p[ok, 2] * 1000.0), 1, 65535).astype(np.int64)
    depth = np.full(cam["rows"] * cam["cols"], 65535, dtype=np.int64)
    np.minimum.at(depth, v[ok] * cam["cols"] + u[ok], zmm)
    mask = (depth != 65535).astype(np.uint8) * 255
    depth[depth == 65535] = 1500
    proj = np.array([[cam["fx"], 0, cam["cx"], 0], [0, cam["fy"], cam["cy"], 0], [0, 0, 1.0, 0]])
    return depth.astype(np.uint16).reshape(cam["rows"], cam["cols"]), mask.reshape(cam["rows"], cam["cols"]), cam, Y0, proj


def test_a_rope_that_crosses_itself(oracle):
    """VERDICT r05 missing 3 / item 6b: tdlo_tracker_frame_from_depth forms its visible sets from the distance test and the gap fill alone
    (trackdlo_node.cpp:257-277, :345-360) -- the callback's painter test (:279-343) is NOT in it by default.  On a rope that crosses itself in the image
    the two differ: the default sets hold nodes of the branch that lies under the nearer one.  The two ways to the callback's sets: (a) the caller
    applies the test itself (tdlo_self_occlusion_visible + tdlo_extend_visible_nodes on the pre-pass's node distances) and passes its sets to
    tracking_step; (b) tdlo_tracker_set_self_occlusion(proj, width): frame_from_depth applies it.  Both give the same sets and the same nodes, bit for bit;
    the sets are the oracle's (its literal restatement of :279-343 on the same distances)."""
    from trackdlo_amd import binding as B, synth
    P = synth.LAUNCH_PARAMS
    M, width = 40, 40          # (a 2 cm rope half a metre from the camera: 40 pixels)
    depth, mask, cam, Y0, proj = _crossing_rope(M)
    coord = synth.geodesic_coord(Y0)
    targs = (M, P["visibility_threshold"], P["beta"], P["lambda_"], P["alpha"], P["k_vis"], P["mu"], 30, P["tol"], P["beta_pre_proc"], P["lambda_pre_proc"], P["lle_weight"])
    ctxs = [_ctx(B) for _ in range(3)]
    try:
        trk = [B.trackdlo(*targs, ctx=c) for c in ctxs]
        for t in trk:
            t.initialize_nodes(Y0); t.initialize_geodesic_coord(coord)
        # default: distance test + gap fill
        v0, e0, n0, _ = trk[0].frame_from_depth(depth, mask, *_args(cam), 0.008, 0.06)
        # (a) the caller's own self-occlusion test between the pre-pass and tracking_step
        _, n1, _ = ctxs[1].depth_to_cloud(0, depth, mask, *_args(cam), 0.008, fetch=False)
        dist, vplain, _ = ctxs[1].visibility_prepass(0, Y0, P["visibility_threshold"], 0.06, coord)
        v1 = B.self_occlusion_visible(Y0, proj, width, dist, P["visibility_threshold"])
        e1 = B.extend_visible_nodes(v1, coord, 0.06, M)
        trk[1].tracking_step(None, v1, e1)
        # (b) the tracker told to apply it
        trk[2].set_self_occlusion(proj, width)
        v2, e2, n2, _ = trk[2].frame_from_depth(depth, mask, *_args(cam), 0.008, 0.06)
        assert n0 == n1 == n2 and np.array_equal(v0, vplain)
        assert np.array_equal(v1, oracle.self_occlusion(np.asfortranarray(Y0), proj, width, dist, P["visibility_threshold"]))
        assert np.array_equal(v1, v2) and np.array_equal(e1, e2)
        assert np.array_equal(trk[1].get_tracking_result(), trk[2].get_tracking_result()) and trk[1].get_sigma2() == trk[2].get_sigma2()
        hidden = sorted(set(v0) - set(v1))
        assert len(hidden) >= 1 and set(v1) < set(v0), (v0, v1)          # the default sets are LARGER: nodes under the nearer branch
        # ... and they are where the rope crosses itself: the hidden nodes' pixels lie within the rope's width of a non-adjacent, nearer edge
        uv = (proj @ np.concatenate([Y0, np.ones((M, 1))], axis=1).T).T
        px = np.trunc(uv[:, :2] / uv[:, 2:3])
        for h in hidden:
            d = np.linalg.norm(px - px[h], axis=1)
            far_along = np.abs(np.arange(M) - h) > 3
            assert (d[far_along] <= width).any(), h
        trk[2].set_self_occlusion(None)                                   # off again: the default sets
        for t in (trk[0], trk[2]):
            t.initialize_nodes(Y0)
        va, ea, _, _ = trk[0].frame_from_depth(depth, mask, *_args(cam), 0.008, 0.06)
        vb, eb, _, _ = trk[2].frame_from_depth(depth, mask, *_args(cam), 0.008, 0.06)
        assert np.array_equal(va, vb) and np.array_equal(ea, eb) and np.array_equal(va, v0)
    finally:
        for c in ctxs:
            c.close()
