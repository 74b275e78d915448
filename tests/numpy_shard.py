"""A numpy stand-in for one rank's shard in the N-split driver (tests only).

It restates, for small inputs, the per-shard halves of one EM iteration of trackdlo.cpp:275-438 so
that trackdlo_amd.nsplit.cpd_lle_nsplit (the host logic: what is all-reduced, in which order) can be
exercised under gloo on a box without a GPU.  It is NOT a product path.
"""
import numpy as np


class NumpyShard:
    def __init__(self, X_shard):
        self.X0 = np.asarray(X_shard, dtype=np.float64)

    def begin(self, Y, sigma2, params, priors, visible_nodes, H):
        p = params
        self.p = p
        self.Y0 = np.array(Y, dtype=np.float64); self.Y = self.Y0.copy()
        M = self.M = self.Y0.shape[0]
        d = np.linalg.norm(self.X0[:, None, :] - self.Y0[None, :, :], axis=2)
        self.X = self.X0[d.min(axis=1) < 0.1]                                   # :177-195
        seg = np.linalg.norm(np.diff(self.Y0, axis=0), axis=1)
        self.coord = np.concatenate([[0.0], np.cumsum(seg)])
        dd = np.abs(self.coord[:, None] - self.coord[None, :])
        b = p.beta
        self.G = 1 / (2 * b * 2 * b) * np.exp(-np.sqrt(2) * dd / b) * (2 * dd + np.sqrt(2) * b)   # :233
        self.J = np.zeros(M); self.Yext = self.Y0.copy()
        self.K = 0 if priors is None else len(priors)
        if self.K:
            for r in np.asarray(priors).reshape(-1, 4):
                self.J[int(r[0])] = 1.0; self.Yext[int(r[0])] = r[1:]
        self.H = None if H is None else np.asarray(H)
        nv = 0 if visible_nodes is None else len(visible_nodes)
        self.vis_branch = (nv != M and nv != 0 and p.k_vis != 0)
        self.sigma2 = float(sigma2)
        self.it = 0; self.converged = True; self.done = False
        d2 = ((self.X[:, None, :] - self.Y0[None, :, :]) ** 2).sum()
        return np.array([float(len(self.X)), d2])

    def set_global(self, n, s):
        self.Ng = float(n)
        if self.sigma2 == 0:
            self.sigma2 = s / (3.0 * self.M * n)                               # :271-273

    def dmin(self):
        if len(self.X) == 0:
            return np.full(self.M, 1e300)
        return (((self.X[:, None, :] - self.Y[None, :, :]) ** 2).sum(axis=2)).min(axis=0)

    def estep(self, dmin_sq):
        M, p, X, Y, s2 = self.M, self.p, self.X, self.Y, self.sigma2
        sums = np.zeros(4 * M + 2)
        if len(X) == 0:
            return sums
        d2 = ((X[:, None, :] - Y[None, :, :]) ** 2).sum(axis=2)                # n x m
        a = d2.argmin(axis=1)
        c1 = np.where(a == 0, 2, a - 1); c2 = np.where(a == M - 1, M - 3, a + 1)
        n = np.arange(len(X))
        b = np.where(np.sqrt(d2[n, c1]) < np.sqrt(d2[n, c2]), c1, c2)
        lo = np.minimum(a, b); hi = np.maximum(a, b)
        dlo = np.sqrt(d2[n, lo]); dhi = np.sqrt(d2[n, hi])
        m = np.arange(M)[None, :]
        t = np.where(m <= lo[:, None], self.coord[lo][:, None] - self.coord[None, :] + dlo[:, None],
                     np.where(m >= hi[:, None], self.coord[None, :] - self.coord[hi][:, None] + dhi[:, None], 0.0))
        P = np.exp(-0.5 * t * t / s2)
        c = (2 * np.pi * s2) ** 1.5 * p.mu / (1 - p.mu)
        if self.vis_branch:
            d = np.sqrt(dmin_sq); d = np.where(d <= p.visibility_threshold, 0.0, d)
            v = np.exp(-p.k_vis * d); v = v / v.sum()
            P = P * v[None, :]; c = c / self.Ng                               # :378
        else:
            c = c * M / self.Ng                                               # :300
        P = P / (P.sum(axis=1, keepdims=True) + c)
        sums[:M] = P.sum(axis=0)
        sums[M:4 * M] = (P.T @ X).T.reshape(-1)                               # column-major M x 3
        sums[4 * M] = (P.sum(axis=1) * (X ** 2).sum(axis=1)).sum()            # tr(X^T diag(Pt1) X)
        sums[4 * M + 1] = len(X)
        return sums

    def mstep(self, sums):
        M, p, s2 = self.M, self.p, self.sigma2
        P1 = sums[:M]; PX = sums[M:4 * M].reshape(3, M).T; trX = sums[4 * M]
        A = P1[:, None] * self.G + p.lambda_ * s2 * np.eye(M)
        Bm = PX - P1[:, None] * self.Y0
        if p.include_lle:
            A = A + s2 * p.lle_weight * self.H @ self.G; Bm = Bm - s2 * p.lle_weight * self.H @ self.Y0
        if self.K:
            A = A + p.alpha * self.J[:, None] * self.G; Bm = Bm + p.alpha * (self.Yext - self.Y0)
        W = np.linalg.solve(A, Bm)
        T = self.Y0 + self.G @ W
        self.sigma2 = (trX - 2 * np.trace(PX.T @ T) + np.trace(T.T @ (P1[:, None] * T))) / (P1.sum() * 3)
        crit = np.linalg.norm(self.Y - T, axis=1).sum() / M
        self.Y = T; self.it += 1
        if crit < p.tol:
            self.done = True
        elif self.it >= p.max_iter:
            self.converged = False; self.done = True
        return self.done

    def end(self):
        return dict(Y=self.Y, sigma2=self.sigma2, iters=self.it, converged=self.converged, n_kept=len(self.X))

    def abort(self):
        self.aborted = True


class NumpyDeviceShard(NumpyShard):
    """The same arithmetic behind the enqueue-style interface of trackdlo_amd.nsplit.HipDeviceShard: the exchange buffers are
    the torch tensors of a TorchDeviceExchange (CPU tensors under gloo), filled / consumed in place."""

    def __init__(self, X_shard, xch):
        super().__init__(X_shard)
        self.xch = xch

    def dmin_enqueue(self):
        if not self.done:
            self.xch.dmin.numpy()[:] = self.dmin()

    def estep_enqueue(self):
        if not self.done:                                   # kernels of a finished registration are no-ops
            d = self.xch.dmin.numpy().copy() if self.vis_branch else None
            self.xch.sums.numpy()[:] = self.estep(d)

    def mstep_enqueue(self):
        if not self.done:
            self.mstep(self.xch.sums.numpy().copy())

    def poll(self):
        return self.done, self.it
