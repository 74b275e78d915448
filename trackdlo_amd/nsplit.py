"""One frame, cloud split over several GPUs (BASELINE.json configs[3]).

Each rank holds a contiguous shard of the cloud.  Nodes, the kernel G and the M x M solve are
replicated; per EM iteration the ranks exchange
  * (visibility weighting only) the per-node minimum squared distance  -> all-reduce MIN, M values
  * the packed sums [P1 (M) | PX (3M) | Q | N_kept]                     -> all-reduce SUM, 4M+2 values
and then run the identical M-step redundantly (deterministic, no broadcast).  Once per call the
kept-point count and the sigma2-initialisation sum are all-reduced.  Pt1 is never reduced: it stays
sharded with the points.  The messages are ~1.6 kB, i.e. latency-bound; xGMI bandwidth is irrelevant.

`backend` is anything with the tdlo_split_* methods (trackdlo_amd.nsplit.HipShard wraps the C ABI);
`comm` needs all_reduce_sum / all_reduce_min on numpy float64 arrays (TorchComm wraps
torch.distributed: backend "nccl" == RCCL on ROCm, "gloo" in the CPU tests).
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


def cpd_lle_nsplit(shard, comm, Y, sigma2, params, priors=None, visible_nodes=None, H=None):
    """trackdlo::cpd_lle (trackdlo.cpp:161-441) with the N points sharded over the ranks of `comm`."""
    M = np.asarray(Y).shape[0]
    n_vis = 0 if visible_nodes is None else len(visible_nodes)
    vis_branch = (n_vis != M and n_vis != 0 and params.k_vis != 0)          # trackdlo.cpp:358
    init = comm.all_reduce_sum(shard.begin(Y, sigma2, params, priors, visible_nodes, H))
    if init[0] <= 0:
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
