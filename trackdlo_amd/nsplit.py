"""One frame, cloud split over several GPUs (BASELINE.json configs[3]).

Each rank holds a contiguous shard of the cloud.  Nodes, the kernel G and the M x M solve are
replicated; per EM iteration the ranks exchange
  * (visibility weighting only) the per-node minimum squared distance  -> all-reduce MIN, M values
  * the packed sums [P1 (M) | PX (3M) | Q | N_kept]                     -> all-reduce SUM, 4M+2 values
and then run the identical M-step redundantly (deterministic, no broadcast).  Once per call the
kept-point count and the sigma2-initialisation sum are all-reduced.  Pt1 is never reduced: it stays
sharded with the points.  The messages are ~1.6 kB, i.e. latency-bound; xGMI bandwidth is irrelevant.

Two drivers:
  * cpd_lle_nsplit_device -- the form the 8-GPU run uses.  The two exchange buffers are device memory (a torch tensor),
    the shard's kernels are enqueued on the context's stream (tdlo_split_*_enqueue) and RCCL reduces the buffers in
    place, ordered on the same stream: no host synchronisation inside an iteration; the device-side stopping flag is
    polled every few iterations.
  * cpd_lle_nsplit -- the same protocol through host buffers (numpy); kept for callers without torch and as the
    cross-check of the device form.

`shard` is anything with the methods used below (HipShard / HipDeviceShard wrap the C ABI; tests/numpy_shard.py
restates them on the CPU); `comm` needs all_reduce_sum / all_reduce_min (TorchComm, TorchDeviceExchange:
torch.distributed, backend "nccl" == RCCL on ROCm, "gloo" in the CPU tests).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import binding as B


class TorchComm:
    def __init__(self, device=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.device = torch, dist, device

    def _reduce(self, a, op):
        t = self.torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64))
        if self.device is not None:
            t = t.to(self.device)
        self.dist.all_reduce(t, op=op)
        return t.cpu().numpy()

    def all_reduce_sum(self, a):
        return self._reduce(a, self.dist.ReduceOp.SUM)

    def all_reduce_min(self, a):
        return self._reduce(a, self.dist.ReduceOp.MIN)


class HipShard:
    """This rank's shard on its GPU, through the tdlo_split_* entry points of the C ABI."""

    def __init__(self, ctx: B.Context, X_shard):
        self.ctx = ctx
        ctx.set_cloud(0, X_shard)
        self.M = None

    def begin(self, Y, sigma2, params, priors, visible_nodes, H):
        Y = B._f64(Y); self.M = Y.shape[0]
        pri, K, vis, nv, Hm = B.Context._opt(priors, visible_nodes, H)
        init = np.zeros(2)
        self.ctx._chk(self.ctx.lib.tdlo_split_begin(self.ctx.h, B._ptr(Y), self.M, float(sigma2), C.byref(params), B._ptr(pri), K,
                                                    B._ptr(vis), nv, B._ptr(Hm), B._ptr(init)))
        return init

    def set_global(self, n_kept, sum_d2):
        self.ctx._chk(self.ctx.lib.tdlo_split_set_global(self.ctx.h, float(n_kept), float(sum_d2)))

    def dmin(self):
        out = np.zeros(self.M)
        self.ctx._chk(self.ctx.lib.tdlo_split_dmin(self.ctx.h, B._ptr(out)))
        return out

    def estep(self, dmin_global):
        out = np.zeros(4 * self.M + 2)
        d = np.ascontiguousarray(dmin_global, dtype=np.float64) if dmin_global is not None else None
        self.ctx._chk(self.ctx.lib.tdlo_split_estep(self.ctx.h, B._ptr(d), B._ptr(out)))
        return out

    def mstep(self, sums_global):
        s = np.ascontiguousarray(sums_global, dtype=np.float64)
        done = C.c_int(0)
        self.ctx._chk(self.ctx.lib.tdlo_split_mstep(self.ctx.h, B._ptr(s), C.byref(done)))
        return bool(done.value)

    def end(self):
        Y = np.zeros((self.M, 3), order="F")
        s2 = C.c_double(0); st = B.Stats()
        rc = self.ctx.lib.tdlo_split_end(self.ctx.h, B._ptr(Y), C.byref(s2), C.byref(st))
        if rc:
            raise B.TdloError(rc, "split registration failed")
        return dict(Y=Y, sigma2=s2.value, iters=st.iters, converged=bool(st.converged), n_kept=st.n_kept)

    def abort(self):
        """Leaves the registration without results (tdlo_split_abort): the context and the shard stay usable."""
        self.ctx.lib.tdlo_split_abort(self.ctx.h)


def cpd_lle_nsplit(shard, comm, Y, sigma2, params, priors=None, visible_nodes=None, H=None):
    """trackdlo::cpd_lle (trackdlo.cpp:161-441) with the N points sharded over the ranks of `comm`."""
    M = np.asarray(Y).shape[0]
    n_vis = 0 if visible_nodes is None else len(visible_nodes)
    vis_branch = (n_vis != M and n_vis != 0 and params.k_vis != 0)          # trackdlo.cpp:358
    init = comm.all_reduce_sum(shard.begin(Y, sigma2, params, priors, visible_nodes, H))
    if init[0] <= 0:
        shard.abort()                        # same decision on every rank: the sum is global
        raise B.TdloError(B.TDLO_E_EMPTY, "every point was pruned")
    shard.set_global(init[0], init[1])
    for _ in range(params.max_iter):
        dmin = comm.all_reduce_min(shard.dmin()) if vis_branch else None
        sums = comm.all_reduce_sum(shard.estep(dmin))
        if shard.mstep(sums):
            break
    out = shard.end()
    out["n_kept_global"] = int(init[0])
    return out


# ---- device-resident exchange ------------------------------------------------------------------------------------------
class TorchDeviceExchange:
    """The exchange buffers [dmin (M) | sums (4M+2)] as ONE torch tensor on the shard's device, all-reduced in place by
    torch.distributed on the context's own stream (wrapped as a torch ExternalStream, so the collective is ordered after
    the kernels that filled the buffer and before the ones that read it, without a host synchronisation)."""

    def __init__(self, M, device, stream_ptr=None, group=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.group = torch, dist, group
        self.buf = torch.zeros(5 * M + 2, dtype=torch.float64, device=device)
        self.dmin = self.buf[:M]
        self.sums = self.buf[M:]
        self.stream = None
        if stream_ptr is not None and self.buf.is_cuda:
            self.stream = torch.cuda.ExternalStream(stream_ptr, device=self.buf.device)

    def _on_stream(self):
        import contextlib
        return self.torch.cuda.stream(self.stream) if self.stream is not None else contextlib.nullcontext()

    def all_reduce_min_dmin(self):
        with self._on_stream():
            self.dist.all_reduce(self.dmin, op=self.dist.ReduceOp.MIN, group=self.group)

    def all_reduce_sum_sums(self):
        with self._on_stream():
            self.dist.all_reduce(self.sums, op=self.dist.ReduceOp.SUM, group=self.group)


class HipDeviceShard(HipShard):
    """This rank's shard with the exchange buffers bound to device memory (tdlo_split_bind_exchange)."""

    def __init__(self, ctx: B.Context, X_shard, xch):
        super().__init__(ctx, X_shard)
        self.xch = xch
        self.bind()

    def bind(self):
        """(Re-)binds the exchange buffers; end() unbinds them, so a shard that registers again calls this first."""
        self.ctx._chk(self.ctx.lib.tdlo_split_bind_exchange(self.ctx.h, C.c_void_p(self.xch.dmin.data_ptr()), C.c_void_p(self.xch.sums.data_ptr())))

    def dmin_enqueue(self):
        self.ctx._chk(self.ctx.lib.tdlo_split_dmin_enqueue(self.ctx.h))

    def estep_enqueue(self):
        self.ctx._chk(self.ctx.lib.tdlo_split_estep_enqueue(self.ctx.h))

    def mstep_enqueue(self):
        self.ctx._chk(self.ctx.lib.tdlo_split_mstep_enqueue(self.ctx.h))

    def poll(self):
        done = C.c_int(0); it = C.c_int(0)
        self.ctx._chk(self.ctx.lib.tdlo_split_poll(self.ctx.h, C.byref(done), C.byref(it)))
        return bool(done.value), it.value

    def end(self):
        out = super().end()
        self.ctx.lib.tdlo_split_bind_exchange(self.ctx.h, None, None)
        return out


def poll_points(max_iter):
    """Iterations after which the stopping flag is read: 1, 2, 4, 8, then every 4 (a tracker in steady state converges
    within a couple of iterations; later the cost of a poll -- one stream synchronisation -- is spread over 4)."""
    pts = set()
    k = 1
    while k < 8 and k < max_iter:
        pts.add(k); k *= 2
    pts.update(range(8, max_iter, 4))
    return pts


def cpd_lle_nsplit_device(shard, xch, comm_init, Y, sigma2, params, priors=None, visible_nodes=None, H=None):
    """trackdlo::cpd_lle (trackdlo.cpp:161-441), N points sharded over the ranks, exchange buffers resident on the device.
    `comm_init` (all_reduce_sum on a 2-element numpy array) carries the once-per-call kept-point count and sigma2
    initialisation sum; everything inside the loop goes through `xch`."""
    M = np.asarray(Y).shape[0]
    n_vis = 0 if visible_nodes is None else len(visible_nodes)
    vis_branch = (n_vis != M and n_vis != 0 and params.k_vis != 0)          # trackdlo.cpp:358
    init = comm_init.all_reduce_sum(shard.begin(Y, sigma2, params, priors, visible_nodes, H))
    if init[0] <= 0:
        shard.abort()
        raise B.TdloError(B.TDLO_E_EMPTY, "every point was pruned")
    shard.set_global(init[0], init[1])
    polls = poll_points(params.max_iter) if params.tol > 0 else set()
    for it in range(1, params.max_iter + 1):
        if vis_branch:
            shard.dmin_enqueue()
            xch.all_reduce_min_dmin()
        shard.estep_enqueue()
        xch.all_reduce_sum_sums()
        shard.mstep_enqueue()
        if it in polls and shard.poll()[0]:          # same flag on every rank: the loops leave together
            break
    out = shard.end()
    out["n_kept_global"] = int(init[0])
    return out
