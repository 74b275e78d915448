"""GPU suite, BASELINE.json configs at FULL size: every configuration the benchmark quotes goes through the C ABI against
the CPU oracle with the stated tolerances (tests/test_parity_gpu.py header), not only through scaled-down stand-ins.

  C3  32 frames x N = 50 000 points, M = 50 (one GPU's share of the 256-frame batch): tdlo_cpd_lle_batch, every frame
      against the oracle and bit-equal to its single call
  C4  N = 2 000 000 points, M = 50: the unsplit call, and the eight 250 000-point shards of the N-split protocol
      (device-resident exchange, trackdlo_amd/nsplit.py) on one GPU, both against ONE oracle run of the whole cloud
  C5  N = 200 000 points, M = 300: fp64 everywhere at 1e-9 m / 1e-7, and the default fp32 E-step at 1e-5 m / 1e-3
The oracle affords these sizes for a few iterations (about 1 s per iteration at C5, 1.7 s at C4 on one host core);
the iteration counts are fixed (tol = 0) so that both sides execute exactly the same number.
Reference loop: trackdlo/src/trackdlo.cpp:275-438.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL = {0: (1e-5, 1e-3), 1: (1e-9, 1e-7)}


def _kw(P, max_iter, **over):
    kw = dict(beta=P["beta"], lambda_=P["lambda_"], lle_weight=P["lle_weight"], mu=P["mu"], max_iter=max_iter, tol=0.0, include_lle=False,
              alpha=0.0, k_vis=0.0, visibility_threshold=P["visibility_threshold"])
    kw.update(over)
    return kw


def _params(kw, prec):
    from trackdlo_amd import binding as B
    return B.make_params(kw["beta"], kw["lambda_"], kw["lle_weight"], kw["mu"], kw["max_iter"], kw["tol"], kw["include_lle"],
                         kw["alpha"], kw["k_vis"], kw["visibility_threshold"], prec)


def _check(g, o, prec):
    ty, ts = TOL[prec]
    assert g["iters"] == o["iters"] and g["converged"] == o["converged"] and g["n_kept"] == o["n_kept"]
    dy = float(np.abs(g["Y"] - o["Y"]).max()); ds = abs(g["sigma2"] - o["sigma2"]) / o["sigma2"]
    assert dy <= ty and ds <= ts, (dy, ds)
    return dy, ds


def test_c5_full_size_fp64_and_fp32(oracle):
    """BASELINE.json configs[4]: N = 200 000, M = 300.  Three iterations from sigma2 = 0 (the wide first-iteration windows:
    every E-step chunk path and the multi-CU elimination at its headline size), fp64 at 1e-9 m / 1e-7 and fp32 at 1e-5 / 1e-3."""
    from trackdlo_amd import binding as B, synth
    P = synth.LAUNCH_PARAMS
    N, M = 200000, 300
    X, Y0, _ = synth.scene(N, M, config=5)
    kw = _kw(P, 3)
    o = oracle.cpd_lle(X, Y0, 0.0, **kw)
    assert o["n_kept"] == N
    ctx = B.Context(device=0, max_frames=1, max_points=N, max_nodes=M)
    try:
        ctx.set_cloud(0, X)
        for prec in (1, 0):
            g = ctx.cpd_lle_resident(0, Y0, 0.0, _params(kw, prec))
            assert g["rc"] == 0 and g["mstep_retries"] == 0
            _check(g, o, prec)
        # steady state (sigma2 of a tracker that has converged: millimetre windows), visibility weighting on
        _, _, vis = synth.scene(1000, M, config=5, occlude=(0.45, 0.5))
        vext = synth.extend_visible(vis, M, synth.geodesic_coord(Y0))
        kw2 = _kw(P, 2, k_vis=P["k_vis"])
        o2 = oracle.cpd_lle(X, Y0, 2e-5, visible_nodes=vext, **kw2)
        g2 = ctx.cpd_lle_resident(0, Y0, 2e-5, _params(kw2, 1), visible_nodes=vext)
        _check(g2, o2, 1)
    finally:
        ctx.close()


def test_c4_full_size_unsplit_and_eight_shards(oracle):
    """BASELINE.json configs[3]: ONE frame of N = 2 000 000 points, M = 50.  (i) the unsplit call; (ii) the cloud as eight
    contiguous 250 000-point shards -- eight contexts on this one GPU standing in for the eight ranks, driven by
    nsplit.cpd_lle_nsplit_device with the exchange buffers resident on the device and the all-reduce replaced by an
    in-process reduction of the eight bound device buffers (what RCCL does between eight GPUs).  Both against the same
    oracle run (3 iterations, visibility weighting on: the MIN exchange of trackdlo.cpp:278-296 runs too)."""
    import queue
    import threading
    import torch
    from trackdlo_amd import binding as B, nsplit, synth
    P = synth.LAUNCH_PARAMS
    N, M, R = 2000000, 50, 8
    X, Y0, _ = synth.scene(N, M, config=4)
    _, _, vis = synth.scene(1000, M, config=4, occlude=(0.4, 0.46))
    vext = synth.extend_visible(vis, M, synth.geodesic_coord(Y0))
    kw = _kw(P, 3, k_vis=P["k_vis"])
    o = oracle.cpd_lle(X, Y0, 0.0, visible_nodes=vext, **kw)
    assert o["n_kept"] == N
    pr = _params(kw, 0)
    ctx = B.Context(device=0, max_frames=1, max_points=N, max_nodes=M)
    try:
        g = ctx.cpd_lle(X, Y0, 0.0, pr, visible_nodes=vext)
        assert g["rc"] == 0
        _check(g, o, 0)
        g64 = ctx.cpd_lle_resident(0, Y0, 0.0, _params(kw, 1), visible_nodes=vext)
        _check(g64, o, 1)
    finally:
        ctx.close()

    ctxs = [B.Context(device=0, max_frames=1, max_points=N // R, max_nodes=M) for _ in range(R)]

    class Exchange:
        def __init__(self):
            self.b = threading.Barrier(R)
            self.bufs = [torch.zeros(5 * M + 2, dtype=torch.float64, device="cuda:0") for _ in range(R)]

        def view(self, rank):
            outer = self

            class V:
                dmin = outer.bufs[rank][:M]; sums = outer.bufs[rank][M:]

                def _x(self, lo, hi, op):
                    ctxs[rank].synchronize(); outer.b.wait()
                    r = outer.bufs[0][lo:hi].clone()
                    for q in range(1, R):                              # rank order: every "rank" forms the same bits
                        r = op(r, outer.bufs[q][lo:hi])
                    torch.cuda.synchronize(); outer.b.wait()
                    outer.bufs[rank][lo:hi].copy_(r); torch.cuda.synchronize()

                def all_reduce_min_dmin(self): self._x(0, M, torch.minimum)
                def all_reduce_sum_sums(self): self._x(M, 5 * M + 2, torch.add)
            return V()

    class Init:
        def __init__(self):
            self.b = threading.Barrier(R); self.slots = [None] * R

        def comm(self, rank):
            outer = self

            class C_:
                def all_reduce_sum(self, a):
                    outer.slots[rank] = np.array(a, dtype=np.float64); outer.b.wait()
                    r = sum(outer.slots[1:], outer.slots[0].copy()); outer.b.wait(); return r
            return C_()

    xchg, init, res = Exchange(), Init(), queue.Queue()

    def work(r):
        try:
            xch = xchg.view(r)
            shard = nsplit.HipDeviceShard(ctxs[r], X[r * N // R:(r + 1) * N // R], xch)
            res.put((r, nsplit.cpd_lle_nsplit_device(shard, xch, init.comm(r), Y0, 0.0, pr, visible_nodes=vext)))
        except Exception as e:      # pragma: no cover
            res.put((r, e)); xchg.b.abort(); init.b.abort()

    th = [threading.Thread(target=work, args=(r,)) for r in range(R)]
    [t.start() for t in th]; [t.join() for t in th]
    outs = dict(res.get() for _ in range(R))
    for c in ctxs:
        c.close()
    for r in range(R):
        assert not isinstance(outs[r], Exception), outs[r]
        assert outs[r]["iters"] == o["iters"] and outs[r]["n_kept_global"] == o["n_kept"]
        assert np.abs(outs[r]["Y"] - o["Y"]).max() <= TOL[0][0] and abs(outs[r]["sigma2"] - o["sigma2"]) <= TOL[0][1] * o["sigma2"]
        np.testing.assert_array_equal(outs[r]["Y"], outs[0]["Y"])           # the replicated M-step: same bits on every rank
    assert sum(outs[r]["n_kept"] for r in range(R)) == N


def _ctx_with_estep2(B, mode, **kw):
    """A context whose E-step kernel choice is pinned (TDLO_ESTEP2 is read when the context is made): 0 = k_estep (one point per lane) everywhere,
    1 = k_estep2 (two points per lane) wherever it is eligible, None = by size (the default)."""
    import os
    old = os.environ.get("TDLO_ESTEP2")
    if mode is None: os.environ.pop("TDLO_ESTEP2", None)
    else: os.environ["TDLO_ESTEP2"] = str(mode)
    try:
        return B.Context(**kw)
    finally:
        if old is None: os.environ.pop("TDLO_ESTEP2", None)
        else: os.environ["TDLO_ESTEP2"] = old


# two fp32 E-step forms on the same input (k_estep: one wave x 64 points is the grain of the fp32 tile sums; k_estep2: one wave x 128 points): both
# are held to the oracle at the mode's gate (1e-5 m, 1e-3); between them 1e-7 m / 1e-5 -- a hundredth of the gate, ten times what is observed
ROUTES_APART = (1e-7, 1e-5)


def test_c3_full_size_batch_of_32_frames(oracle):
    """BASELINE.json configs[2], one GPU's share: 32 independent frames of N = 50 000 points, M = 50, registered as ONE
    tdlo_cpd_lle_batch call (stream groups).  The batch fills the GPU, so its E-step is k_estep2 (two points per lane): every frame against the
    oracle (5 iterations from sigma2 = 0); bit for bit against its own single call ON THE SAME KERNEL (a context with TDLO_ESTEP2=1) and, with
    TDLO_ESTEP2=0, the batch on k_estep bit for bit against the single call on k_estep (round 1-5's invariant, kept per kernel); the default
    single call (one 50 000-point frame cannot fill the GPU: k_estep) within ROUTES_APART of the batch."""
    from trackdlo_amd import binding as B, synth
    P = synth.LAUNCH_PARAMS
    F, N, M = 32, 50000, 50
    kw = _kw(P, 5)
    pr = _params(kw, 0)
    kw2 = dict(kw, max_iter=50, tol=P["tol"])
    pr2 = _params(kw2, 0)
    Xs, Ys = [], []
    for f in range(F):
        X, Y0, _ = synth.scene(N, M, config=3, frame=f)
        Xs.append(X); Ys.append(Y0)
    ctx = B.Context(device=0, max_frames=F, max_points=N, max_nodes=M)
    try:
        for f in range(F):
            ctx.set_cloud(f, Xs[f])
        out = ctx.cpd_lle_batch(Ys, [0.0] * F, pr)
        again = ctx.cpd_lle_batch(Ys, [0.0] * F, pr)
        assert np.array_equal(out["Y"], again["Y"]) and np.array_equal(out["sigma2"], again["sigma2"])      # repeatable bit for bit
        worst = (0.0, 0.0)
        for f in range(F):
            single = ctx.cpd_lle_resident(f, Ys[f], 0.0, pr)
            assert np.abs(out["Y"][f] - single["Y"]).max() <= ROUTES_APART[0] and abs(out["sigma2"][f] - single["sigma2"]) <= ROUTES_APART[1] * single["sigma2"]
            o = oracle.cpd_lle(Xs[f], Ys[f], 0.0, **kw)
            st = out["stats"][f]
            g = dict(Y=out["Y"][f], sigma2=out["sigma2"][f], iters=st["iters"], converged=bool(st["converged"]), n_kept=st["n_kept"])
            dy, ds = _check(g, o, 0)
            worst = (max(worst[0], dy), max(worst[1], ds))
        # the production stopping rule on the same batch: frames stop at their own iteration
        out2 = ctx.cpd_lle_batch(Ys, [0.0] * F, pr2)
        for f in (0, 7, 19, 31):
            o = oracle.cpd_lle(Xs[f], Ys[f], 0.0, **kw2)
            # the criterion of trackdlo.cpp:424 is compared with tol in fp32-E-step arithmetic here and in fp64 there: a frame
            # whose criterion passes within rounding of tol may stop one iteration apart; then only the
            # flags are compared
            assert abs(out2["stats"][f]["iters"] - o["iters"]) <= 1 and bool(out2["stats"][f]["converged"]) == o["converged"]
            if out2["stats"][f]["iters"] == o["iters"]:
                assert np.abs(out2["Y"][f] - o["Y"]).max() <= TOL[0][0]
    finally:
        ctx.close()
    # batch == single, bit for bit, on each kernel
    for mode in (1, 0):
        ctx = _ctx_with_estep2(B, mode, device=0, max_frames=F, max_points=N, max_nodes=M)
        try:
            for f in range(F):
                ctx.set_cloud(f, Xs[f])
            b1 = ctx.cpd_lle_batch(Ys, [0.0] * F, pr)
            b2 = ctx.cpd_lle_batch(Ys, [0.0] * F, pr2)
            if mode == 1:
                assert np.array_equal(b1["Y"], out["Y"]) and np.array_equal(b2["Y"], out2["Y"])      # (the default batch IS the k_estep2 batch)
            for f in (0, 7, 19, 31):
                s1 = ctx.cpd_lle_resident(f, Ys[f], 0.0, pr)
                assert np.array_equal(b1["Y"][f], s1["Y"]) and b1["sigma2"][f] == s1["sigma2"], (mode, f)
                s2 = ctx.cpd_lle_resident(f, Ys[f], 0.0, pr2)
                assert np.array_equal(b2["Y"][f], s2["Y"]) and b2["stats"][f]["iters"] == s2["iters"], (mode, f)
        finally:
            ctx.close()
