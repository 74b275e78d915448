// tdlo_mstep_generic.h -- the generic pivoted M-step (trackdlo.cpp:392-437) as a device function, shared by the kernel
// k_mstep (tdlo_device.hip) and by the multi-CU eliminations of tdlo_mstep_big.hip, whose finishing workgroup redoes an
// iteration with it when one of their inter-workgroup hand-offs ran into its time limit.
#pragma once
#include "tdlo_devcommon.h"

namespace tdlo {
__device__ __forceinline__ double readlane_f64_g(double v, int src_lane) {     // src_lane wave-uniform
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), src_lane), __builtin_amdgcn_readlane(__double2loint(v), src_lane));
}


template <int NW> __device__ __forceinline__ double block_sum_n(double v, double *scratch) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
    __syncthreads();
    double r = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) r += scratch[w];
    return r;
}


// ------------------------------------------------------------------------------------------------
// M-step, trackdlo.cpp:392-437.  One workgroup per frame.
// ------------------------------------------------------------------------------------------------
// NT threads: 256 with the tableau in LDS; 1024 with the tableau in global memory (more element updates in flight: M = 300
// 7.1 -> 5.4 ms; what is left is one CU's path to the L2, 1.45 MB per column)
template <typename T, bool LDSA, int NT>
__device__ __forceinline__ void mstep_generic_body(const FrameDev &f, int from_sums, char *smem) {
    IterState *st = f.st;
    const int M = f.M, t = threadIdx.x, lane = t & 63;
    const int nS = 4 * M + 1;
    const int ld = M | 1;                     // odd leading dimension
    double *S = (double *)smem;               // nS (+pad)
    double *W = S + ((nS + 1) & ~1);          // 3M
    double *Tn = W + 3 * M;                   // 3M
    double *scratch = Tn + 3 * M;             // 16
    int *piv = (int *)(scratch + 16);          // M (rounded to a multiple of 4 ints)
    int *used = piv + ((M + 3) & ~3);         // M (rounded)
    double *Alds = (double *)(used + ((M + 3) & ~3));
    double *A = LDSA ? Alds : f.Ascr;         // ld x (M+3)

    // ---- 1. the E-step's sums (fixed-point accumulators, kAccRows replica rows)
    if (from_sums != 1) {
        const int itn = st->it;
        const long long *rows = acc_rows(f, itn);
        for (int e = t; e < nS; e += NT) S[e] = acc_read(f, rows, e);
        acc_clear_other<NT>(f, itn, t);
    } else {
        for (int e = t; e < nS; e += NT) S[e] = f.sums[e];
    }
    __syncthreads();
    if (from_sums == 2) {       // split mode, export only: publish local sums and stop
        for (int e = t; e < nS; e += NT) f.sums[e] = S[e];
        if (t == 0) f.sums[nS] = (double)st->N;
        return;
    }

    // ---- 2. assemble [A | B] (:392-413)
    const double sigma2 = st->sigma2;
    const double c2 = f.lambda * sigma2, sg = sigma2 * f.lle_weight;
    for (int e = t; e < M * M; e += NT) {
        const int i = e % M, j = e / M;
        const double g = f.G[e];
        double a = S[i] * g + (i == j ? c2 : 0.0);
        if (f.include_lle) a += sg * f.HG[e];
        if (f.has_priors) a += f.aJ[i] * g;
        A[(size_t)j * ld + i] = a;
    }
    for (int e = t; e < 3 * M; e += NT) {
        const int i = e % M, d = e / M;
        const V4<T> *ndq = (const V4<T> *)f.nodes;
        const double yd = d == 0 ? (double)ndq[i].x : (d == 1 ? (double)ndq[i].y : (double)ndq[i].z);
        double b = S[M + e] + S[i] * (yd - f.Y0[e]);     // B = R + P1 (y - Y0), R = PX - P1 y from the E-step
        if (f.include_lle) b -= sg * f.HY0[e];
        if (f.has_priors) b += f.aYd[e];
        A[(size_t)(M + d) * ld + i] = b;
    }
    for (int i = t; i < M; i += NT) used[i] = 0;
    __syncthreads();

    // ---- 3. Gaussian elimination with partial pivoting (rows permuted implicitly) + back substitution (:415).
    // NOT Gauss-Jordan: a row stops changing once it has served as pivot row, and the right-hand sides are finished by a
    // back substitution.  Gauss-Jordan is only forward stable; its error in W is not of the form A^-1 dA W, so the product
    // G W of :417 does not damp it, and on the ill-conditioned systems of the pre-processing registration (beta = 3,
    // lambda = 1) it cost 1e-8 m per solve in T (the oracle's QR: 1e-11, tests/test_solver_error.py).  LU with back
    // substitution is backward stable and sits at 1e-12.
    int singular = 0;
    const int Mr = (M + 63) & ~63;                                   // rows rounded up to whole waves
    const int ncs = Mr <= NT ? NT / Mr : 1;                  // column slots: NT / rows
    const int ri = Mr <= NT ? t % Mr : t, cs = Mr <= NT ? t / Mr : 0, rstep = Mr <= NT ? Mr : NT;
    const int rend = cs < ncs ? M : 0;                               // threads beyond the last whole slot (e.g. 192 rows: 5 slots of 1024 threads) idle
    for (int k = 0; k < M; ++k) {
        // every wave finds the pivot row redundantly (no hand-off needed)
        double bv = -1.0; int bi = 0x7fffffff;
        for (int i = lane; i < M; i += 64) {
            if (!used[i]) {
                const double v = fabs(A[(size_t)k * ld + i]);
                if (v > bv) { bv = v; bi = i; }
            }
        }
        // across the lanes: judged by the upper 32 bits of |a| (non-negative doubles order like their bit patterns; any entry within
        // 2^-20 of the largest is as good a pivot), first lane among equals -- a DPP wave max (7 instructions) instead of six
        // dependent rounds of three ds_bpermute each on every column's critical path
        const unsigned key = bi != 0x7fffffff ? ((unsigned)__double2hiint(bv) & 0x7fffffffu) + 1u : 0u;
        const unsigned mxk = wave_minmax_u32<true>(key);
        const int pl = (int)__builtin_ctzll(__ballot(key == mxk));
        const int p = __builtin_amdgcn_readlane(bi, pl);
        bv = readlane_f64_g(bv, pl);
        const double pv = A[(size_t)k * ld + p];
        if (!(bv > 0.0)) singular = 1;
        const double rp = (bv > 0.0) ? 1.0 / pv : 0.0;
        // thread = (row, column slot): the multiplier of a row is formed once, and no element pays an integer division
        // (M is a run-time value: `e % M, e / M` per element had been most of this kernel's instructions)
        for (int i = ri; i < rend; i += rstep) {
            if (i != p && !used[i]) {
                const double l = A[(size_t)k * ld + i] * rp;
                int j = k + 1 + cs;
                for (; j + 7 * ncs < M + 3; j += 8 * ncs) {        // 8 independent element updates in flight (the tableau may be in global memory)
                    double a[8], b[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) { a[u] = A[(size_t)(j + u * ncs) * ld + i]; b[u] = A[(size_t)(j + u * ncs) * ld + p]; }
#pragma unroll
                    for (int u = 0; u < 8; ++u) A[(size_t)(j + u * ncs) * ld + i] = a[u] - l * b[u];
                }
                for (; j < M + 3; j += ncs) A[(size_t)j * ld + i] -= l * A[(size_t)j * ld + p];
            }
        }
        if (t == 0) { piv[k] = p; used[p] = 1; }          // (no thread reads used[p] in this step: the update skips i == p first)
        __syncthreads();
    }
    // back substitution, column oriented: x_k = b[piv k] / U[piv k][k]; every row that pivots an earlier column takes
    // b_i -= U[i][k] x_k.  used[] is recycled as kof (the column a row pivots); the reciprocals of the pivots are formed once,
    // in parallel (Tn is free until step 4); a thread keeps its (row, right-hand side) items for the whole loop.
    for (int k = t; k < M; k += NT) { used[piv[k]] = k; const double ukk = A[(size_t)k * ld + piv[k]]; Tn[k] = ukk != 0.0 ? 1.0 / ukk : 0.0; }
    __syncthreads();
    {
        constexpr int NI = (3 * (LDSA ? kLdsSolveMaxM : kMaxNodes) + NT - 1) / NT;        // (row, right-hand side) items per thread
        int ii[NI], kk[NI]; size_t bo[NI];
#pragma unroll
        for (int q = 0; q < NI; ++q) {
            const int e = t + q * NT;
            ii[q] = e < 3 * M ? e % M : 0;
            kk[q] = e < 3 * M ? used[ii[q]] : 0x7fffffff;      // items beyond 3 M never update
            bo[q] = (size_t)(M + (e < 3 * M ? e / M : 0)) * ld;
        }
        for (int k = M - 1; k >= 0; --k) {
            const int p = piv[k];
            const double rk = Tn[k];
#pragma unroll
            for (int q = 0; q < NI; ++q)
                if (kk[q] < k) A[bo[q] + ii[q]] -= A[(size_t)k * ld + ii[q]] * (A[bo[q] + p] * rk);
            if (t < 3) W[t * M + k] = A[(size_t)(M + t) * ld + p] * rk;
            __syncthreads();
        }
    }

    // ---- 4. T = Y0 + G W (:417)
    for (int e = t; e < 3 * M; e += NT) {
        const int i = e % M, d = e / M;
        double a = 0;
        for (int k = 0; k < M; ++k) a += f.G[(size_t)k * M + i] * W[d * M + k];
        Tn[e] = f.Y0[e] + a;
    }
    __syncthreads();

    // ---- 5. sigma2 (residual form of :418-422) and the convergence criterion (:424)
    const V4<T> *nodes = (const V4<T> *)f.nodes;
    double s_np = 0, s_dr = 0, s_pd = 0, s_cr = 0;
    for (int m = t; m < M; m += NT) {
        V4<T> q; q.x = nodes[m].x; q.y = nodes[m].y; q.z = nodes[m].z; q.w = nodes[m].w;
        const double yx = (double)q.x, yy = (double)q.y, yz = (double)q.z;    // nodes as the E-step saw them
        const double p1 = S[m];
        const double dx = Tn[m] - yx, dy = Tn[M + m] - yy, dz = Tn[2 * M + m] - yz;
        const double rx = S[M + m], ry = S[2 * M + m], rz = S[3 * M + m];
        s_np += p1;
        s_dr += dx * rx + dy * ry + dz * rz;
        s_pd += p1 * (dx * dx + dy * dy + dz * dz);
        const double ex = f.Y[m] - Tn[m], ey = f.Y[M + m] - Tn[M + m], ez = f.Y[2 * M + m] - Tn[2 * M + m];
        s_cr += ::sqrt(ex * ex + ey * ey + ez * ez);
    }
    s_np = block_sum_n<NT / 64>(s_np, scratch);
    s_dr = block_sum_n<NT / 64>(s_dr, scratch);
    s_pd = block_sum_n<NT / 64>(s_pd, scratch);
    s_cr = block_sum_n<NT / 64>(s_cr, scratch);
    const double new_sigma2 = (S[4 * M] - 2.0 * s_dr + s_pd) / (s_np * 3.0);
    const double crit = s_cr / (double)M;

    // ---- 6. publish Y, nodes, iteration state
    V4<T> *nodes_w = (V4<T> *)f.nodes;
    for (int m = t; m < M; m += NT) {
        V4<T> q; q.x = (T)Tn[m]; q.y = (T)Tn[M + m]; q.z = (T)Tn[2 * M + m]; q.w = (T)f.coord[m];
        nodes_w[m] = q;
        f.dminbits[m] = ~0ull;
    }
    for (int e = t; e < 3 * M; e += NT) {
        f.Y[e] = Tn[e];
        f.Yout[e] = Tn[e] + f.ctr[e / M];
    }
    if (t == 0) {
        const int it = st->it + 1;
        st->it = it; st->crit = crit; st->Np = s_np;
        const double Nc = st->Nc;
        const bool finite_ok = (new_sigma2 == new_sigma2) && (fabs(new_sigma2) < 1e300) && (new_sigma2 > 0) && !singular;
        if (finite_ok) set_iter_consts(f, st, new_sigma2, Nc);
        else { st->sigma2 = new_sigma2; st->status = TDLO_E_NUMERIC; st->done = 1; st->converged = 0; }
        if (crit < f.tol) st->done = 1;                                   // :424-428
        else if (it >= f.max_iter) { st->converged = 0; st->done = 1; }  // :433-437
    }
}


// retry_only: the launch that follows every k_mstep_pivot_mcu launch -- a no-op unless that kernel left the iteration to
// be redone here after a timed-out hand-off (IterState::retry_pending)
template <typename T, bool LDSA, int NT>
__global__ __launch_bounds__(NT) void k_mstep(const FrameDev *__restrict__ frames, int from_sums, int retry_only) {
    const FrameDev &f = frames[blockIdx.x];
    if (f.st->done) return;
    if (retry_only) {
        const int pending = f.st->retry_pending;
        if (!pending) return;
        __syncthreads();
        if (threadIdx.x == 0) f.st->retry_pending = 0;
    }
    extern __shared__ __attribute__((aligned(16))) char smem[];
    mstep_generic_body<T, LDSA, NT>(f, from_sums, smem);
}

}  // namespace tdlo
