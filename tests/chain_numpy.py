"""Host-side restatement (numpy, fp64) of the arithmetic of csrc/tdlo_mstep_chain.hip -- test infrastructure.

The M-step of trackdlo.cpp:405-417 without the LLE term, (c I + D G) W = B, T = Y0 + G W, solved for V = G W through the
state-space form of the kernel G of :233 (Matern-3/2 in the chain coordinate): Kalman filters from four ends (two from the chain's
ends, two from the middle node with its state as the unknown), a small solve for the junction states, Rauch-Tung-Striebel smoothing
back along every direction.  Same operations as the kernel, one step at a time."""
import numpy as np


def _tail_series(t, n0=3, terms=48):
    term = t ** n0
    for k in range(2, n0 + 1):
        term /= k
    s = term
    for n in range(n0 + 1, terms):
        term = term * t / n
        s += term
    return s


def chain_link(beta, h):
    """{Phi11, Phi12, Phi21, Phi22, Q11, Q12, Q22} of a gap h (chain_link in csrc/tdlo_devcommon.h)."""
    s = np.sqrt(2.0) / beta
    sf2 = 1.0 / (2.0 * np.sqrt(2.0) * beta)
    x = s * h
    e = np.exp(-x)
    e2 = e * e
    if x < 1.0:
        sm = _tail_series(2.0 * x)
        u11, u22 = e2 * sm, e2 * (4.0 * x + sm)
    else:
        u11, u22 = 1.0 - e2 * (1.0 + 2.0 * x + 2.0 * x * x), 1.0 - e2 * (1.0 - 2.0 * x + 2.0 * x * x)
    return [e * (1.0 + x), e * h, -s * s * h * e, e * (1.0 - x), sf2 * u11, 2.0 * sf2 * s ** 3 * h * h * e2, sf2 * s * s * u22]


def _inv2(a, b, d):
    det = a * d - b * b
    return d / det, -b / det, a / det


def carve(M):
    """Junction nodes and steps per direction (ChainCarve in csrc/tdlo_mstep_chain.hip)."""
    j2 = (M - 1) // 2
    j1 = j2 // 2
    j3 = (j2 + M) // 2
    n = [j1 + 1, j2 - j1, j3 - j2 + 1, M - j3]
    return j1, j2, j3, n, max(n)


def slot_info(M, dr, k):
    """(node, link index -- 0 = identity --, observed) of step k of direction dr; leading dummy steps observe nothing."""
    j1, j2, j3, n, nQ = carve(M)
    kk = k - (nQ - n[dr])
    if kk < 0:
        return 0, 0, False
    if dr == 0:
        return kk, (kk if kk > 0 else 0), True
    if dr == 1:
        node = j2 - 1 - kk
        return node, node + 1, node != j1
    if dr == 2:
        node = j2 + kk
        return node, (node if kk > 0 else 0), node != j3 and not (kk == 0 and j1 == j2)
    node = M - 1 - kk
    return node, (node + 1 if kk > 0 else 0), True


def chain_solve(coord, beta, c, pobs, B):
    """V (M x 3) with (c G^-1 + diag(pobs)) V = B, i.e. V = G W for (c I + diag(pobs) G) W = B.

    Four directions: 0 filters nodes 0 .. j1 from the stationary prior, 3 nodes M-1 .. j3 of the reversed process; the inner
    directions 2 (nodes j2 .. j3) and 1 (nodes j2-1 .. j1, reversed process) start at the middle node from its exact but unknown
    state x: covariance 0, means affine in x (columns 0..2 the coordinates g, 3 and 4 the spike columns F), and the likelihood of
    their data as a function of x is summed along the way (acc)."""
    M = len(coord)
    s = np.sqrt(2.0) / beta
    sf2 = 1.0 / (2.0 * np.sqrt(2.0) * beta)
    pinf0, pinf1 = sf2, s * s * sf2
    cp0, cp1 = c * (1.0 / pinf0), c * (1.0 / pinf1)
    links = [None] + [chain_link(beta, coord[i] - coord[i - 1]) for i in range(1, M)]
    ident = [1.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0]
    j1, j2, j3, n, nQ = carve(M)
    sg = np.array([1.0, 1.0, 1.0, -1.0, -1.0])
    runs = []
    for dr in range(4):
        inner = dr in (1, 2)
        a, b, d = (0.0, 0.0, 0.0) if inner else (pinf0 / c, 0.0, pinf1 / c)
        m = np.zeros((2, 5)); m[0, 3] = 1.0; m[1, 4] = 1.0
        acc = np.zeros((2, 5))                  # [:, :3] = eta, [:, 3:] = J of -1/2 x^T J x + x^T eta
        rec, post, mean = [], [], []
        for k in range(nQ):
            node, li, obs = slot_info(M, dr, k)
            L = ident if li == 0 else links[li]
            p = pobs[node] if obs else 0.0
            bb = np.r_[B[node] if obs else np.zeros(3), 0.0, 0.0]
            f11, f12, f21, f22 = L[:4]
            q11, q12, q22 = L[4] / c, L[5] / c, L[6] / c
            t1, t2, t3, t4 = f11 * a + f12 * b, f11 * b + f12 * d, f21 * a + f22 * b, f21 * b + f22 * d
            pa, pb, pd = t1 * f11 + t2 * f12 + q11, t1 * f21 + t2 * f22 + q12, t3 * f21 + t4 * f22 + q22
            g = 1.0 / (1.0 + p * pa)
            a, b = pa * g, pb * g
            d = pd - p * pb * b
            pm0, pm1 = f11 * m[0] + f12 * m[1], f21 * m[0] + f22 * m[1]
            innov = bb - p * pm0
            w = (sg * g) * innov
            acc[0] += w * pm0[3]; acc[1] += w * pm0[4]
            m = np.stack([pm0 + a * innov, pm1 + b * innov])
            rec.append(L); post.append((a, b, d)); mean.append(m.copy())
        runs.append((rec, post, mean, acc))

    def half(outer, inner):
        """message of one half to x, in the half's frame (left: reversed process, right: process): the outer posterior changes frame"""
        (_, postO, meanO, _), (_, postI, meanI, accI) = runs[outer], runs[inner]
        aO, bO, dO = postO[-1]; bO = -bO
        mO = meanO[-1][:, :3].copy(); mO[1] = -mO[1]
        ia, ib, id_ = _inv2(aO, bO, dO)
        La, Lb, Ld = ia - cp0, ib, id_ - cp1                 # Lam = P^-1 - Pinf^-1: the outer data as a likelihood
        xi = np.stack([ia * mO[0] + ib * mO[1], ib * mO[0] + id_ * mO[1]])
        pa, pb, pd = postI[-1]
        F, g = meanI[-1][:, 3:], meanI[-1][:, :3]
        n00, n01, n10, n11 = 1 + La * pa + Lb * pb, La * pb + Lb * pd, Lb * pa + Ld * pb, 1 + Lb * pb + Ld * pd
        r = 1.0 / (n00 * n11 - n01 * n10)
        A = np.array([[n11 * r, -n01 * r], [-n10 * r, n00 * r]])          # (I + Lam Pc)^-1
        Lam = np.array([[La, Lb], [Lb, Ld]])
        W = A @ Lam
        w = A @ xi
        J = F.T @ W @ F + accI[:, 3:]
        e = F.T @ (w - W @ g) + accI[:, :3]
        return J, e, (Lam, xi, A, F, g, np.array([[pa, pb], [pb, pd]]))

    JL, eL, auxL = half(0, 1)
    JR, eR, auxR = half(3, 2)
    Rm = np.diag([1.0, -1.0])
    Jt = np.diag([cp0, cp1]) + Rm @ JL @ Rm + JR
    et = Rm @ eL + eR
    ka, kb, kd = _inv2(Jt[0, 0], Jt[0, 1], Jt[1, 1])
    x2 = np.stack([ka * et[0] + kb * et[1], kb * et[0] + kd * et[1]])       # state at j2, frame of the process

    def back(aux, xs):
        Lam, xi, A, F, g, Pc = aux
        u = F @ xs + g
        return u + Pc @ (A @ (xi - Lam @ u))

    yL, yR = back(auxL, Rm @ x2), back(auxR, x2)            # junction states at j1 (reversed frame) and j3
    ends = [Rm @ yL, yL, yR, Rm @ yR]                       # smoothed state at the last slot of each direction, in its own frame
    starts = [None, Rm @ x2, x2, None]
    V = np.zeros((M, 3))
    for dr in range(4):
        rec, post, mean, _ = runs[dr]
        x = ends[dr].copy()
        for k in range(nQ - 1, -1, -1):
            if k < nQ - 1:
                a, b, d = post[k]
                h11, h12, h21, h22 = rec[k + 1][:4]
                q11, q12, q22 = rec[k + 1][4] / c, rec[k + 1][5] / c, rec[k + 1][6] / c
                t1, t2, t3, t4 = h11 * a + h12 * b, h11 * b + h12 * d, h21 * a + h22 * b, h21 * b + h22 * d
                pa, pb, pd = t1 * h11 + t2 * h12 + q11, t1 * h21 + t2 * h22 + q12, t3 * h21 + t4 * h22 + q22
                det = pa * pd - pb * pb
                if det > 0.0:
                    ia, ib, id_ = pd / det, -pb / det, pa / det
                else:                                   # covariance still exactly zero (inner direction): state = filtered mean
                    ia = ib = id_ = 0.0
                Cm = np.array([[t1 * ia + t3 * ib, t1 * ib + t3 * id_], [t2 * ia + t4 * ib, t2 * ib + t4 * id_]])
                Ph = np.array([[h11, h12], [h21, h22]])
                ek = mean[k] - Cm @ (Ph @ mean[k])      # 5 columns: e of the coordinates, E of the spike columns
                e3 = ek[:, :3] + (ek[:, 3:] @ starts[dr] if dr in (1, 2) else 0.0)
                x = e3 + Cm @ x
            node, li, obs = slot_info(M, dr, k)
            if obs:
                V[node] = x[0]
    return V


def kernel_G(coord, beta, dtype=np.float64):
    """G of trackdlo.cpp:233."""
    c = np.asarray(coord, dtype=dtype)
    d = np.abs(c[:, None] - c[None, :])
    r2 = np.sqrt(dtype(2))
    return 1 / (2 * dtype(beta) * 2 * dtype(beta)) * np.exp(-r2 * d / dtype(beta)) * (2 * d + r2 * dtype(beta))


def dense_solve_longdouble(A, B):
    """Gaussian elimination with partial pivoting in 80-bit arithmetic (the yardstick)."""
    ld = np.longdouble
    A = A.astype(ld).copy(); B = B.astype(ld).copy(); n = len(A)
    for k in range(n):
        p = k + int(np.argmax(np.abs(A[k:, k])))
        if p != k:
            A[[k, p]] = A[[p, k]]; B[[k, p]] = B[[p, k]]
        f = A[k + 1:, k] / A[k, k]
        A[k + 1:] -= f[:, None] * A[k]; B[k + 1:] -= f[:, None] * B[k]
    X = np.zeros_like(B)
    for k in range(n - 1, -1, -1):
        X[k] = (B[k] - A[k, k + 1:] @ X[k + 1:]) / A[k, k]
    return X
