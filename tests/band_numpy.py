"""Host-side restatement (numpy, fp64) of the arithmetic of csrc/tdlo_mstep_band.hip -- test infrastructure.

The M-step of trackdlo.cpp:392-417 WITH the LLE term (include_lle, the pre-processing registration of tracking_step, :925-927),
    (c I + (D + g H) G) W = B,   T = Y0 + G W,    c = lambda sigma2,  g = sigma2 lle_weight,  D = diag(P1) + alpha J,
    B = PX - P1 Y0 - g H Y0 (+ alpha (Y_ext - Y0)),   H = (I - L)^T (I - L)  (:236-237; L has +-3 chain neighbours per row),
solved for V = G W: (c G^-1 + D + g H) V = B.  G (:233) is the Matern-3/2 covariance over the chain coordinate, Markov in the
state x_i = (f_i, f'_i) (tests/chain_numpy.py): c G^-1 is the Schur complement of c K on the f components, K the block-tridiagonal
joint precision of the states,
    K = e_0 Pinf^-1 e_0^T + sum_i [-Phi_i^T; I] Q_i^-1 [-Phi_i, I]      (link i between nodes i - 1 and i).
With E the selector of the f components the system becomes
    (c K + E^T (D + g H) E) x = E^T B,    V = E x:
2M unknowns ordered (f_0, f'_0, f_1, f'_1, ...), symmetric positive definite, BANDED with half-bandwidth 12 (H reaches 6 nodes).
Solved by L D L^T without pivoting inside the band, one rank-1 update of the 13 x 13 window per unknown (the kernel: one
v_mfma_f64_16x16x4 per unknown, the three right-hand sides riding as columns 13..15 of the same tile), then the back substitution.
"""
import numpy as np

import chain_numpy as cn

HB = 12          # half-bandwidth in unknowns
WIN = HB + 1


def lle_band(H, M):
    """The 7 diagonals H[i, i + d], d = 0..6, of the LLE regulariser (zero beyond: rows of I - L reach +-3 nodes)."""
    hb = np.zeros((M, 7))
    for d in range(min(7, M)):
        hb[: M - d, d] = np.diagonal(H, d)
    return hb


def h_is_banded(H, tol=0.0):
    M = len(H)
    i, j = np.indices((M, M))
    return bool(np.all(np.abs(H[np.abs(i - j) > 6]) <= tol))


def link_precision(beta, h):
    """(Q^-1 [3: 11, 12, 22], Phi [4]) of a gap h > 0."""
    L = cn.chain_link(beta, h)
    q11, q12, q22 = L[4], L[5], L[6]
    det = q11 * q22 - q12 * q12
    return (q22 / det, -q12 / det, q11 / det), L[:4]


def state_precision(coord, beta):
    """K as (diagonal blocks [M][3: ff, fp, pp], off-diagonal blocks [M][4]: block i couples node i (rows) with node i - 1 (columns),
    order (f f, f p, p f, p p) = K[2i + a, 2(i-1) + b])."""
    M = len(coord)
    s = np.sqrt(2.0) / beta
    sf2 = 1.0 / (2.0 * np.sqrt(2.0) * beta)
    dg = np.zeros((M, 3)); off = np.zeros((M, 4))
    dg[0, 0] += 1.0 / sf2; dg[0, 2] += 1.0 / (s * s * sf2)
    for i in range(1, M):
        (a, b, d), (f11, f12, f21, f22) = link_precision(beta, coord[i] - coord[i - 1])
        # Q^-1 into node i
        dg[i, 0] += a; dg[i, 1] += b; dg[i, 2] += d
        # Phi^T Q^-1 Phi into node i - 1
        t11, t12 = a * f11 + b * f21, a * f12 + b * f22            # (Q^-1 Phi) rows
        t21, t22 = b * f11 + d * f21, b * f12 + d * f22
        dg[i - 1, 0] += f11 * t11 + f21 * t21
        dg[i - 1, 1] += f11 * t12 + f21 * t22
        dg[i - 1, 2] += f12 * t12 + f22 * t22
        off[i] = [-t11, -t12, -t21, -t22]                          # -Q^-1 Phi
    return dg, off


def assemble(coord, beta, c, dobs, g, hb, B):
    """Dense 2M x 2M matrix (only the band is non-zero) and 2M x 3 right-hand side of the state-space system."""
    M = len(coord)
    n = 2 * M
    dg, off = state_precision(coord, beta)
    A = np.zeros((n, n)); R = np.zeros((n, 3))
    for i in range(M):
        A[2 * i, 2 * i] = c * dg[i, 0] + dobs[i] + g * hb[i, 0]
        A[2 * i, 2 * i + 1] = A[2 * i + 1, 2 * i] = c * dg[i, 1]
        A[2 * i + 1, 2 * i + 1] = c * dg[i, 2]
        if i > 0:
            blk = c * off[i].reshape(2, 2)
            A[2 * i: 2 * i + 2, 2 * i - 2: 2 * i] += blk
            A[2 * i - 2: 2 * i, 2 * i: 2 * i + 2] += blk.T
        for d in range(1, 7):
            if i + d < M:
                A[2 * i, 2 * (i + d)] += g * hb[i, d]
                A[2 * (i + d), 2 * i] += g * hb[i, d]
        R[2 * i] = B[i]
    return A, R


def band_ldlt_solve(A, R):
    """L D L^T without pivoting, right-looking, one rank-1 update of the trailing window per unknown, the right-hand sides eliminated
    along (exactly the kernel's order of operations); then L^T x = D^-1 y column by column from the last unknown."""
    A = A.copy(); Y = R.copy()
    n = len(A)
    Lm = np.zeros((n, WIN))        # Lm[k, j] = l_{k + j, k}, j = 1..12
    rd = np.zeros(n)
    for k in range(n):
        r = 1.0 / A[k, k]
        rd[k] = r
        hi = min(n, k + WIN)
        u = A[k, k + 1: hi].copy()
        l = u * r
        Lm[k, 1: hi - k] = l
        A[k + 1: hi, k + 1: hi] -= np.outer(l, u)
        Y[k + 1: hi] -= np.outer(l, Y[k])
    X = Y * rd[:, None]
    for k in range(n - 1, -1, -1):
        lo = max(0, k - HB)
        for i in range(lo, k):
            X[i] -= Lm[i, k - i] * X[k]
    return X, rd


def band_solve(coord, beta, c, dobs, g, H, B):
    """V (M x 3) with (c G^-1 + diag(dobs) + g H) V = B."""
    M = len(coord)
    hb = lle_band(H, M)
    A, R = assemble(coord, beta, c, dobs, g, hb, B)
    X, rd = band_ldlt_solve(A, R)
    return X[0::2], rd


def dense_reference(coord, beta, c, dobs, g, H, B):
    """T - Y0 = G W of the reference's dense system (c I + (D + g H) G) W = B in 80-bit arithmetic."""
    ld = np.longdouble
    Gl = cn.kernel_G(coord, beta, ld)
    Hl = H.astype(ld)
    A = (np.diag(dobs.astype(ld)) + ld(g) * Hl) @ Gl + ld(c) * np.eye(len(coord), dtype=ld)
    return Gl @ cn.dense_solve_longdouble(A, B)


# ------------------------------------------------------------------------------------------------------------------------------
# The kernel's data flow, step by step (csrc/tdlo_mstep_band.hip): the system is divided by sigma2,
#     (lambda K + gamma H + D / sigma2) x = B / sigma2,
# so that everything but the diagonal D / sigma2 and the right-hand side is fixed for the whole registration: k_setup writes one RECORD of
# 16 doubles per unknown n -- column n of (lambda K + gamma H), rows n-12 .. n, at the position of the row's SLOT (row mod 13) --
# and the M-step only adds D_a / sigma2 to the diagonal entry of the even records and puts the right-hand side into the three spare
# positions.  The 13 x 13 window of the elimination lives in a 16 x 16 tile (the accumulator of one v_mfma_f64_16x16x4): row / column
# slot = unknown mod 13, columns 13..15 = the right-hand sides.  Tile element (row slot q, column c) sits in lane c + 16 (q % 4),
# register q / 4, so a record is laid out [q % 4][q / 4]: the lane that owns column n of the tile reads its four registers as 32
# consecutive bytes.  Record position of slot q: (q % 4) * 4 + q / 4; right-hand side d at position 4 d + 7 (slots 13..15).
NS = 13


def rec_pos(q):
    return (q % 4) * 4 + q // 4


def build_records(coord, beta, lam, gamma, hb, n_pad=NS):
    """rec[n][16] for n = 0 .. 2M - 1 (+ n_pad identity records: unknowns that do not exist)."""
    M = len(coord)
    n = 2 * M
    dg, off = state_precision(coord, beta)

    def entry(i, j):                      # (lambda K + gamma H)[i, j], i <= j
        a, b, ti, tj = i >> 1, j >> 1, i & 1, j & 1
        v = 0.0
        if a == b:
            v += lam * (dg[a, 0] if (ti, tj) == (0, 0) else (dg[a, 2] if (ti, tj) == (1, 1) else dg[a, 1]))
        elif b == a + 1:
            v += lam * off[b][tj * 2 + ti]
        if ti == 0 and tj == 0 and b - a <= 6:
            v += gamma * hb[a, b - a]
        return v

    rec = np.zeros((n + n_pad, 16))
    for j in range(n + n_pad):
        for q in range(NS):
            i = j - ((j - q) % NS)
            if i < 0:
                continue
            if j >= n:
                rec[j, rec_pos(q)] = 1.0 if i == j else 0.0
            else:
                rec[j, rec_pos(q)] = entry(i, j)
    return rec


def tile_solve(rec, dobs, B, sigma2, n_unknowns):
    """The kernel's elimination and back substitution on its own data structures; returns x (n_unknowns x 3)."""
    rec = rec.copy()
    n = n_unknowns
    rs = 1.0 / sigma2
    for a in range(n // 2):                                  # the M-step's part of the records
        rec[2 * a, rec_pos((2 * a) % NS)] += dobs[a] * rs
        for d in range(3):
            rec[2 * a, 4 * d + 7] = B[a, d] * rs
    C = np.zeros((16, 16))                                   # the tile
    col = lambda j: np.array([rec[j, rec_pos(q)] for q in range(NS)])
    rhs = lambda j: np.array([rec[j, 4 * d + 7] for d in range(3)])
    for j in range(NS):                                      # the first window: columns 0..12, their right-hand sides
        C[:NS, j % NS] = col(j)
        C[j % NS, 13:] = rhs(j)
    lrec = np.zeros((n, 16))
    pending = None                                           # (slot, rhs) entering through the spare k-slot of the next MFMA
    for k in range(n):
        p = k % NS
        t = C[p, :].copy()                                   # pivot row: 13 column slots + 3 right-hand sides
        nr = -1.0 / t[p]
        a = t * nr
        lrec[k] = a
        C += np.outer(a, t)                                  # the MFMA: rank-1 update of the whole tile (rows 13..15: garbage, never read)
        if pending is not None:                              # ... and the spare k-slot: e_slot x rhs
            C[pending[0], 13:] += pending[1]
        j = k + NS                                           # column k + 13 enters the slot pivot k leaves
        C[:NS, p] = col(j)                                   # (explicit set: C = C * colmask + newcol)
        pending = (p, rhs(j))
    # back substitution, column oriented: acc[slot] = z_i - sum of the known terms; x_k is final when the walk reaches k
    x = np.zeros((n, 3))
    acc = np.zeros((NS, 3))
    for k in range(n - 1, -1, -1):
        p = k % NS
        acc[p] = -lrec[k, 13:]                               # z_k = y_k / d_k (the record holds -z_k)
        for i in range(k + 1, min(n, k + NS)):
            acc[p] += lrec[k, i % NS] * x[i]
        x[k] = acc[p]
    return x


# ------------------------------------------------------------------------------------------------------------------------------
# The twisted plan (BandPlan in csrc/tdlo_internal.h): two directions of elimination that meet at a 12-unknown separator.
def band_plan(M, lds_limit=160 * 1024 - 1024):
    """(tw, cT, cB, D, mT, mB, nUp, limT, sT, sB, nRecT, nRecB) exactly as the kernels compute them."""
    nU = 2 * M
    tw = 1 if nU >= 38 else 0
    while True:
        if not tw:
            cT, cB, D = (nU + NS - 1) // NS, 0, 0
        else:
            q = nU - 12; ch = (q + NS - 1) // NS
            D = NS * ch - q; cT = (ch + 1) // 2; cB = ch - cT
        mT, mB, nUp = NS * cT, NS * cB, nU + D
        limT = mT + 12 if tw else nU
        sT = NS * (cT + tw); sB = NS * (cB + 1) if tw else 0
        nRecT = sT + 15; nRecB = sB + 15 if tw else 0
        lds = (((4 * M + 2 + 1) & ~1) + 32 + (2 if tw else 1) * (16 * 28 + 64 + 16 * 28) + 16 * (nRecT + nRecB) + (512 if tw else 0)) * 8
        if not tw or lds <= lds_limit:
            break
        tw = 0
    return dict(tw=tw, cT=cT, cB=cB, D=D, mT=mT, mB=mB, nUp=nUp, limT=limT, sT=sT, sB=sB, nRecT=nRecT, nRecB=nRecB, lds_bytes=lds)


def twisted_solve(A, R, mT):
    """L D L^T of the banded SPD system from both ends: unknowns 0 .. mT-1 are eliminated forwards, unknowns n-1 .. mT+12 backwards; both
    leave their Schur complements on the 12 unknowns in between (half-bandwidth 12: nothing else couples the two sides), those are
    solved, and each side is back-substituted on its own -- the arithmetic of the two waves of k_mstep_band."""
    A = A.copy(); Y = R.copy()
    n = len(A)
    sep = list(range(mT, mT + HB))
    top = list(range(0, mT)); bot = list(range(n - 1, mT + HB - 1, -1))
    rec = {}
    for order in (top, bot):
        step = 1 if order is top else -1
        for k in order:
            r = 1.0 / A[k, k]
            nb = [k + step * j for j in range(1, WIN) if 0 <= k + step * j < n]
            nb = [i for i in nb if (i > k if step > 0 else i < k)]
            l = A[k, nb] * r
            rec[k] = (nb, l.copy(), r)
            A[np.ix_(nb, nb)] -= np.outer(l, A[k, nb])
            Y[nb] -= np.outer(l, Y[k])
    X = np.zeros_like(Y)
    X[sep] = np.linalg.solve(A[np.ix_(sep, sep)], Y[sep])
    for order in (top[::-1], bot[::-1]):
        for k in order:
            nb, l, r = rec[k]
            X[k] = Y[k] * r - l @ X[nb]
    return X
