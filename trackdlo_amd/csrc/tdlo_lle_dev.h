// tdlo_lle_dev.h -- the LLE regulariser H = (I - L)^T (I - L) of trackdlo.cpp:236-237 (L = calc_LLE_weights with k = 6, :119-159) through its 13
// diagonals, formed ON THE DEVICE from a chain of up to 256 nodes: bit for bit what the host's lle_regulariser_band (tdlo_host.cpp) gives for the
// same nodes.  tracking_step's main registration ends with this (k_mstep_chain, FrameDev::lle_next): the pre-processing registration of the NEXT
// frame starts from exactly these nodes when every node is visible, and its set-up then finds H on the device instead of waiting for the host
// (6 of the 8.6 us of host work in front of a frame's first launch).
//
// The weights of a chain's LLE are "dominated by rounding noise" (the Gram matrix of up to 6 differences in 3-D has rank <= 3, tdlo_host.cpp): a
// different rounding anywhere gives a different H.  So this is the host's algorithm operation by operation -- same order, no fused
// multiply-adds (contract off; IEEE division), the same partial-pivot LU with the same comparisons -- with every system embedded in a 6 x 6 one
// (rows / columns beyond the node's n neighbours are never touched: each use is predicated on the index, not multiplied by zero, so not even a
// NaN travels differently), thread = node, everything in registers.
#pragma once

namespace tdlo {

// Ab: 7 M doubles of LDS (row k of I - L for the columns k - 3 .. k + 3); all MB threads of the workgroup call this (two barriers inside).
template <int MB>
__device__ __forceinline__ void lle_band_device(const double *__restrict__ Y, int M, double *__restrict__ Hb_out, double *Ab, int t) {
#pragma clang fp contract(off)
    for (int i = t; i < M; i += MB) {
        // chain_neighbours(3, M, i): trackdlo.cpp:92-117 (one side clipped by the if / else-if, then both)
        int first = i - 3, last = i + 3;
        if (i - 3 < 0) first = 0;
        else if (i + 3 >= M) last = M - 1;
        if (last > M - 1) last = M - 1;
        if (first < 0) first = 0;
        const int n = last - first;                           // neighbours (the node itself left out): 0 .. 6
        double yi[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) yi[d] = __hip_atomic_load(Y + d * M + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int nb[6];
        double df[6][3];
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            const int j = first + r + ((first + r >= i) ? 1 : 0);
            nb[r] = r < n ? j : i;
#pragma unroll
            for (int d = 0; d < 3; ++d) df[r][d] = yi[d] - __hip_atomic_load(Y + d * M + nb[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        double a[6][6], det = 1.0;
        auto factor = [&](bool regularised) __attribute__((always_inline)) {
#pragma clang fp contract(off)
            // local Gram matrix (:128-134): one triangle, mirrored (tdlo_host.cpp lle_node_weights); :139-144 adds 1e-5 to the diagonal
#pragma unroll
            for (int r = 0; r < 6; ++r)
#pragma unroll
                for (int s = r; s < 6; ++s) {
                    double acc = 0;
#pragma unroll
                    for (int d = 0; d < 3; ++d) acc += df[r][d] * df[s][d];
                    if (regularised && r == s) acc += 0.00001;
                    a[r][s] = acc; a[s][r] = acc;
                }
            det = 1.0;
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                const bool kin = k < n;
                int p = k;
                double best = fabs(a[k][k]);
#pragma unroll
                for (int r = k + 1; r < 6; ++r) {
                    const double v = fabs(a[r][k]);
                    const bool better = r < n && v > best;
                    p = better ? r : p; best = better ? v : best;
                }
#pragma unroll
                for (int r = k + 1; r < 6; ++r) {
                    const bool sw = kin && p == r;
#pragma unroll
                    for (int c = 0; c < 6; ++c) { const double x = a[k][c], y = a[r][c]; a[k][c] = sw ? y : x; a[r][c] = sw ? x : y; }
                }
                det = (kin && p != k) ? -det : det;
                const double piv = a[k][k];
                det = kin ? det * piv : det;
                const bool act = kin && !(piv == 0.0);
#pragma unroll
                for (int r = k + 1; r < 6; ++r) {
                    const bool ar = act && r < n;
                    const double l = a[r][k] / piv;
                    a[r][k] = ar ? l : a[r][k];
#pragma unroll
                    for (int c = k + 1; c < 6; ++c) { const double v = a[r][c] - l * a[k][c]; a[r][c] = ar ? v : a[r][c]; }
                }
            }
        };
        factor(false);
        if (det == 0.0) factor(true);
        // w = G^-1 1 / (1^T G^-1 1) (:146-150): G v = 1 by the factors (the right-hand side is all ones: its row permutation is itself)
        double y[6], v[6];
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            double s = 1.0;
#pragma unroll
            for (int c = 0; c < r; ++c) s = s - a[r][c] * y[c];
            y[r] = s;
        }
#pragma unroll
        for (int r = 5; r >= 0; --r) {
            double s = y[r];
#pragma unroll
            for (int c = r + 1; c < 6; ++c) { const double q = s - a[r][c] * v[c]; s = c < n ? q : s; }
            v[r] = s / a[r][r];
        }
        double tot = 0;
#pragma unroll
        for (int r = 0; r < 6; ++r) { const double q = tot + v[r]; tot = r < n ? q : tot; }
        // row i of I - L
#pragma unroll
        for (int c = 0; c < 7; ++c) Ab[7 * i + c] = c == 3 ? 1.0 : 0.0;
#pragma unroll
        for (int r = 0; r < 6; ++r) if (r < n) Ab[7 * i + (nb[r] - i + 3)] = 0.0 - v[r] / tot;
    }
    __syncthreads();
    for (int e = t; e < 13 * M; e += MB) {
        const int i = e / 13, u = e - 13 * i, j = i - 6 + u;
        double s = 0;
        if (j >= 0 && j < M) {
            const int hi = i > j ? i : j, lo = i < j ? i : j;
            const int k0 = hi - 3 > 0 ? hi - 3 : 0, k1 = lo + 3 < M - 1 ? lo + 3 : M - 1;
            for (int k = k0; k <= k1; ++k) s += Ab[7 * k + (i - k + 3)] * Ab[7 * k + (j - k + 3)];
        }
        Hb_out[e] = s;
    }
    __syncthreads();
}

}  // namespace tdlo
