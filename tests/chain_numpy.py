"""Host-side restatement (numpy, fp64) of the arithmetic of csrc/tdlo_mstep_chain.hip -- test infrastructure.

The M-step of trackdlo.cpp:405-417 without the LLE term, (c I + D G) W = B, T = Y0 + G W, solved for V = G W through the
state-space form of the kernel G of :233 (Matern-3/2 in the chain coordinate): a two-ended Kalman filter, the two Gaussians
fused at the middle node, Rauch-Tung-Striebel smoothing outwards.  Same operations as the kernel, one step at a time."""
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


def chain_solve(coord, beta, c, pobs, B):
    """V (M x 3) with (c G^-1 + diag(pobs)) V = B, i.e. V = G W for (c I + diag(pobs) G) W = B."""
    M = len(coord)
    s = np.sqrt(2.0) / beta
    sf2 = 1.0 / (2.0 * np.sqrt(2.0) * beta)
    pinf0, pinf1 = sf2, s * s * sf2
    links = [None] + [chain_link(beta, coord[i] - coord[i - 1]) for i in range(1, M)]
    ident = [1.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0]
    jn = (M - 1) // 2
    nB = M - jn
    sh = nB - (jn + 1)
    V = np.zeros((M, 3))
    runs = {}
    for dr in (0, 1):
        a, b, d = pinf0 / c, 0.0, pinf1 / c
        m = np.zeros((2, 3))
        rec, post, mean = [], [], []
        for k in range(nB):
            real = dr == 1 or k >= sh
            node = (M - 1 - k if dr else k - sh) if real else 0
            li = (M - k if k > 0 else 0) if dr else (k - sh if k > sh else 0)
            L = ident if li == 0 else links[li]
            obs = real and not (dr == 1 and k == nB - 1)
            p = pobs[node] if obs else 0.0
            bb = B[node] if obs else np.zeros(3)
            f11, f12, f21, f22 = L[:4]
            q11, q12, q22 = L[4] / c, L[5] / c, L[6] / c
            pm = np.array([[f11, f12], [f21, f22]]) @ m
            t1, t2, t3, t4 = f11 * a + f12 * b, f11 * b + f12 * d, f21 * a + f22 * b, f21 * b + f22 * d
            pa, pb, pd = t1 * f11 + t2 * f12 + q11, t1 * f21 + t2 * f22 + q12, t3 * f21 + t4 * f22 + q22
            g = 1.0 / (1.0 + p * pa)
            a, b = pa * g, pb * g
            d = pd - p * pb * b
            innov = bb - p * pm[0]
            m = pm + np.outer([a, b], innov)
            rec.append(L); post.append((a, b, d)); mean.append(m.copy())
        runs[dr] = (rec, post, mean)
    (_, postA, meanA), (_, postB, meanB) = runs[0], runs[1]
    aA, bA, dA = postA[-1]
    aB, bB, dB = postB[-1]
    bB = -bB
    mA = meanA[-1]
    mB = meanB[-1].copy(); mB[1] = -mB[1]
    ia, ib, id_ = _inv2(aA, bA, dA)
    ja, jb, jd = _inv2(aB, bB, dB)
    e0 = ia * mA[0] + ib * mA[1] + ja * mB[0] + jb * mB[1]
    e1 = ib * mA[0] + id_ * mA[1] + jb * mB[0] + jd * mB[1]
    ka, kb, kd = _inv2(ia + ja - c / pinf0, ib + jb, id_ + jd - c / pinf1)
    xs = np.stack([ka * e0 + kb * e1, kb * e0 + kd * e1])
    V[jn] = xs[0]
    for dr in (0, 1):
        rec, post, mean = runs[dr]
        x = xs.copy()
        if dr:
            x[1] = -x[1]
        for k in range(nB - 2, -1, -1):
            a, b, d = post[k]
            h11, h12, h21, h22 = rec[k + 1][:4]
            q11, q12, q22 = rec[k + 1][4] / c, rec[k + 1][5] / c, rec[k + 1][6] / c
            t1, t2, t3, t4 = h11 * a + h12 * b, h11 * b + h12 * d, h21 * a + h22 * b, h21 * b + h22 * d
            pa, pb, pd = t1 * h11 + t2 * h12 + q11, t1 * h21 + t2 * h22 + q12, t3 * h21 + t4 * h22 + q22
            ia, ib, id_ = _inv2(pa, pb, pd)
            Cm = np.array([[t1 * ia + t3 * ib, t1 * ib + t3 * id_], [t2 * ia + t4 * ib, t2 * ib + t4 * id_]])
            Ph = np.array([[h11, h12], [h21, h22]])
            x = (mean[k] - Cm @ (Ph @ mean[k])) + Cm @ x
            real = dr == 1 or k >= sh
            if real:
                V[M - 1 - k if dr else k - sh] = x[0]
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
