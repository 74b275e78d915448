// tdlo_mstep_big.hip -- M-step (trackdlo.cpp:392-437) for 60 < M <= 512 chain nodes without the LLE term
// (BASELINE.json configs[4]: M = 300).
//
// Same mathematics as k_mstep_fast's MFMA variant in tdlo_device.hip -- Gauss-Jordan elimination without pivoting of
// [A | B], A = c I + (diag(P1) + alpha J) G, which is the elimination of the SPD matrix G + c D^-1 -- but the
// tableau (up to 512 x 528 doubles, 2 MB) cannot live in one CU's registers or LDS.  Two kernels:
//   * k_mstep_mcu (the product path, second half of this file): one workgroup per 16 rows, the row block stays in MFMA
//     accumulators, the panel's reduced pivot rows travel between workgroups inside the launch.  M = 300: 131 us.
//   * k_mstep_big (first half): the whole elimination in ONE workgroup of 16 waves with the tableau in global memory
//     (L2-resident).  It is the comparator of the tests (TDLO_MSTEP_BIG=1wg: the two eliminations perform the same
//     operations in the same order) and serves the export-only call of the N-split interface.  M = 300: 528 us.
// k_mstep_big processes panels of 16 pivot columns:
//   a. the 16 pivot rows (16 x live columns) are loaded one column per thread and reduced among themselves by 16
//      sequential row operations held in registers; the only traffic per step is the pivot column, handed over through
//      LDS (one barrier per step).  Result U (16 x Cp) -> LDS (MFMA B operand) and back to the tableau;
//   b. every other row block, every column block right of the panel:  C -= L U  with four v_mfma_f64_16x16x4 per
//      16 x 16 tile; L = the panel's columns, which are never written again (column blocks up to the panel are dead).
//      Every wave takes a contiguous row-major range of tiles.
// fp64 vector and matrix rates are equal on gfx950, so the MFMA buys data flow (no per-element index arithmetic),
// not flops.  What bounds k_mstep_big is ONE CU's memory path: every panel re-reads and re-writes the live part of the
// tableau.  Three layout decisions follow from measurements (M = 300: 0.88 -> 0.51 ms; the scalar column-at-a-time
// kernel it replaced took 12.9 ms):
//   * the tableau is tile-major in accumulator order (tix()): a tile is 2 KB of contiguous memory, 32 bytes per lane;
//   * the panel's columns are re-ordered once per panel into A-operand order (Lb): scattered 8-byte reads cost a
//     128-byte line each in the address pipeline and had been half of the run time;
//   * U lives in LDS in B-operand order; consecutive tiles of a wave share the row block, so the A operand is reused.
// Mp = M rounded up to 16 rows (identity padded), Cp = Mp + 16 columns, right-hand sides in columns Mp .. Mp + 2.
#include "tdlo_devcommon.h"
#include "tdlo_mstep_generic.h"
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <hip/hip_ext.h>

namespace tdlo {
extern thread_local hipEvent_t g_mstep_ev[2];       // tdlo_device.hip: start/stop events for the M-step dispatch (tdlo_profile_iteration)
namespace {

constexpr int kBig = 1024;

// Tableau storage: tile-major, every 16 x 16 tile in the register order of the MFMA accumulator (lane = col + 16 (row & 3),
// register = row >> 2), so a wave reads or writes a tile as 2 KB of contiguous memory, 32 bytes per lane -- with a
// row-major tableau the same tile is 16 strided 128-byte pieces fetched 8 bytes per lane, and one CU's memory path
// sustains a third of the bandwidth.
__device__ __forceinline__ size_t tix(int i, int j, int ncb) {
    return ((size_t)((i >> 4) * ncb + (j >> 4)) << 8) + (size_t)((((j & 15) + 16 * (i & 3)) << 2) + ((i & 15) >> 2));
}

__device__ __forceinline__ double block_sum16(double v, double *scratch) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
    __syncthreads();
    double a = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) a += scratch[i];
    return a;
}

template <typename T>
__device__ __forceinline__ void mstep_big_body(const FrameDev &f, const int from_sums, char *smem) {
    IterState *st = f.st;
    const int M = f.M, t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int nS = 4 * M + 1;
    const int Mp = (M + 15) & ~15, Cp = Mp + 16, nrb = Mp >> 4, ncb = Cp >> 4;
    double *S = (double *)smem;               // nS (+pad)
    double *W = S + ((nS + 1) & ~1);          // 3M (+pad)
    double *Tn = W + ((3 * M + 1) & ~1);      // 3M (+pad)
    double *scratch = Tn + ((3 * M + 1) & ~1);// 16
    double *pcol = scratch + 16;              // pivot column of the current step, double-buffered (one barrier per step)
    double *U = pcol + 272;                   // 16 x Cp reduced pivot rows (B-operand order)
    const auto Tb = TDLO_AS_GLOBAL_RW(double, f.Ascr);     // Mp x Cp tableau, row-major (global address space: no flat ops)
    const auto Gg = TDLO_AS_GLOBAL(double, f.G);
    const auto Lb = Tb + (size_t)Mp * Cp;                   // panel columns in MFMA A-operand order: [rb][k-step][lane], negated

#define BSTAMP(i) do { if (t == 0) f.dbg[i] = __builtin_amdgcn_s_memtime(); } while (0)
    BSTAMP(0);
    // ---- 1. the E-step's sums (fixed-point accumulators, kAccRows replica rows)
    if (from_sums != 1) {
        const int itn = st->it;
        const long long *rows = acc_rows(f, itn);
        for (int e = t; e < nS; e += kBig) S[e] = acc_read(f, rows, e);
        acc_clear_other<kBig>(f, itn, t);
    } else {
        for (int e = t; e < nS; e += kBig) S[e] = f.sums[e];
    }
    __syncthreads();
    if (from_sums == 2) {       // split mode, export only
        for (int e = t; e < nS; e += kBig) f.sums[e] = S[e];
        if (t == 0) f.sums[nS] = (double)st->N;
        return;
    }

    BSTAMP(1);
    // ---- 2. tableau [A | B] (:392-413), identity padded; B = R + P1 (y - Y0) (+ alpha (Y_ext - Y0)), R from the E-step
    const double sigma2 = st->sigma2;
    const double c2 = f.lambda * sigma2;
    const int pri = f.has_priors;
    const V4<T> *ndq = (const V4<T> *)f.nodes;
#pragma unroll 8
    for (int e = t; e < Mp * Cp; e += kBig) {            // e runs in storage order: coalesced stores
        const int tile = e >> 8, rbt = tile / ncb, cbt = tile - rbt * ncb, ln = (e >> 2) & 63;
        const int i = 16 * rbt + 4 * (e & 3) + (ln >> 4), j = 16 * cbt + (ln & 15);
        double v = 0.0;
        if (i < M) {
            if (j < M) v = (S[i] + (pri ? f.aJ[i] : 0.0)) * Gg[(size_t)i * M + j] + (i == j ? c2 : 0.0);      // G symmetric
            else if (j >= Mp && j < Mp + 3) {
                const int d = j - Mp, q = d * M + i;
                const double yd = d == 0 ? (double)ndq[i].x : (d == 1 ? (double)ndq[i].y : (double)ndq[i].z);
                v = S[M + q] + S[i] * (yd - f.Y0[q]);
                if (pri) v += f.aYd[q];
            }
        } else if (i == j) v = 1.0;
        Tb[e] = v;
    }
    __syncthreads();

    BSTAMP(2);
    // ---- 3. blocked Gauss-Jordan, 16 pivot columns per panel
    int singular = 0;
    const int cL = lane & 15, gL = lane >> 4;
    for (int pb = 0; pb < nrb; ++pb) {
        const int k0 = pb << 4;
        // a. pivot rows: thread = column (k0 <= col < Cp), 16 entries in registers
        const int col = k0 + t;
        const bool act = col < Cp;
        double u[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) u[r] = 0.0;
        const auto ucol = Tb + (((size_t)(pb * ncb + ((act ? col : k0) >> 4))) << 8) + (size_t)((col & 15) << 2);   // + 64 g + r
        if (act) {
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4)
#pragma unroll
                for (int r = 0; r < 4; ++r) u[4 * r + g4] = ucol[64 * g4 + r];
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            double *pc = pcol + (j & 1) * 16;
            if (t == j) {
#pragma unroll
                for (int r = 0; r < 16; ++r) pc[r] = u[r];
            }
            __syncthreads();
            const double pv = pc[j];
            {
                const int e = (__double2hiint(pv) >> 20) & 0x7ff;
                if (e == 0 || e == 0x7ff) singular = 1;          // zero / denormal / non-finite pivot
            }
            const double v = u[j] * fast_rcp(pv);
#pragma unroll
            for (int r = 0; r < 16; ++r) u[r] = (r == j) ? v : fma(-pc[r], v, u[r]);
        }
        if (act) {
#pragma unroll
            for (int r = 0; r < 16; ++r) U[((col >> 4) << 8) + ((r >> 2) << 6) + (col & 15) + 16 * (r & 3)] = u[r];     // B-operand order: [cb][k-step][lane]
            if (t >= 16) {
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4)
#pragma unroll
                    for (int r = 0; r < 4; ++r) ucol[64 * g4 + r] = u[4 * r + g4];
            }
        }
        // a'. the panel's columns, re-ordered once per panel into the A-operand layout (lane = row + 16 (col & 3) of k-step
        //     col >> 2): the trailing update then fetches them with one coalesced load per k-step instead of 64 scattered
        //     8-byte reads per tile, which is what bounded this kernel (the address pipeline pays per 128-byte line)
        for (int rbx = w; rbx < nrb; rbx += 16) {
            if (rbx == pb) continue;
            const auto src = Tb + (((size_t)(rbx * ncb + pb)) << 8) + (size_t)(lane << 2);
            double v4[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v4[r] = src[r];          // elements (row 4 r + gL, column cL) of tile (rbx, pb)
#pragma unroll
            for (int r = 0; r < 4; ++r) Lb[(size_t)rbx * 256 + (cL >> 2) * 64 + (4 * r + gL) + 16 * (cL & 3)] = -v4[r];
        }
        __syncthreads();
        // b. trailing update: tiles (rb != pb, cb > pb)
        const int nlive = ncb - pb - 1, ntile = (nrb - 1) * nlive;
        // Every wave takes a CONTIGUOUS range of tiles in row-major order, TB at a time (all loads first, then the
        // MFMAs, then the stores): consecutive tiles share the row block, so the strided L operand (16 rows x 32 B) is
        // served by the L1 after its first use, and no tile needs an integer division.
        constexpr int TB = 4;
        const int chunk = (ntile + 15) >> 4;
        const int tbeg = w * chunk, tend = (tbeg + chunk) < ntile ? (tbeg + chunk) : ntile;
        int ri = tbeg / (nlive > 0 ? nlive : 1), ci = tbeg - ri * nlive;
        for (int tile0 = tbeg; tile0 < tend; tile0 += TB) {
            double a[TB][4], b[TB][4];
            mfma_d4 C[TB];
            __attribute__((address_space(1))) double *Ct[TB];
            int rb_prev = -1;
#pragma unroll
            for (int q = 0; q < TB; ++q) {
                const bool on = tile0 + q < tend;
                const int rq = on ? ri : 0, cq = on ? ci : 0;      // clamp: loads stay in bounds, result discarded
                const int rb = rq < pb ? rq : rq + 1, cb = pb + 1 + cq;
                const auto Lt = Lb + (size_t)rb * 256 + lane;
                Ct[q] = Tb + (((size_t)(rb * ncb + cb)) << 8) + (size_t)(lane << 2);
                if (q == 0 || rb != rb_prev) {                     // wave-uniform: consecutive tiles mostly share the row block
#pragma unroll
                    for (int s4 = 0; s4 < 4; ++s4) a[q][s4] = Lt[64 * s4];
                } else {
#pragma unroll
                    for (int s4 = 0; s4 < 4; ++s4) a[q][s4] = a[q > 0 ? q - 1 : 0][s4];
                }
                rb_prev = rb;
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) b[q][s4] = U[(cb << 8) + (s4 << 6) + lane];
#pragma unroll
                for (int r = 0; r < 4; ++r) C[q][r] = Ct[q][r];
                if (++ci == nlive) { ci = 0; ++ri; }
            }
#pragma unroll
            for (int q = 0; q < TB; ++q) {
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) C[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q][s4], b[q][s4], C[q], 0, 0, 0);
            }
#pragma unroll
            for (int q = 0; q < TB; ++q) {
                if (tile0 + q < tend) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) Ct[q][r] = C[q][r];
                }
            }
        }
        __syncthreads();
    }
    BSTAMP(3);
    for (int e = t; e < 3 * M; e += kBig) { const int i = e % M, d = e / M; W[e] = Tb[tix(i, Mp + d, ncb)]; }
    singular = __syncthreads_or(singular);

    // ---- 4. T = Y0 + G W (:417): thread = (node, half of the k range)
    {
        const int i = t & 511, h = t >> 9;
        double v0 = 0, v1 = 0, v2 = 0;
        if (i < M) {
            const int kh = (M + 1) >> 1, kb = h * kh, ke = (kb + kh) < M ? (kb + kh) : M;
            for (int k = kb; k < ke; ++k) { const double gk = Gg[(size_t)k * M + i]; v0 += gk * W[k]; v1 += gk * W[M + k]; v2 += gk * W[2 * M + k]; }
        }
        double *tmp = U;                      // free now: 6 x 512 doubles <= 16 x Cp
        if (i < M) { tmp[(h * 3 + 0) * 512 + i] = v0; tmp[(h * 3 + 1) * 512 + i] = v1; tmp[(h * 3 + 2) * 512 + i] = v2; }
        __syncthreads();
        for (int e = t; e < 3 * M; e += kBig) { const int m = e % M, d = e / M; Tn[e] = f.Y0[e] + (tmp[d * 512 + m] + tmp[(3 + d) * 512 + m]); }
    }
    __syncthreads();

    BSTAMP(4);
    // ---- 5. sigma2 (residual form of :418-422) and the convergence criterion (:424)
    double s_np = 0, s_dr = 0, s_pd = 0, s_cr = 0;
    for (int m = t; m < M; m += kBig) {
        const double yx = (double)ndq[m].x, yy = (double)ndq[m].y, yz = (double)ndq[m].z;    // nodes as the E-step saw them
        const double p1 = S[m];
        const double dx = Tn[m] - yx, dy = Tn[M + m] - yy, dz = Tn[2 * M + m] - yz;
        s_np += p1;
        s_dr += dx * S[M + m] + dy * S[2 * M + m] + dz * S[3 * M + m];
        s_pd += p1 * (dx * dx + dy * dy + dz * dz);
        const double ex = f.Y[m] - Tn[m], ey = f.Y[M + m] - Tn[M + m], ez = f.Y[2 * M + m] - Tn[2 * M + m];
        s_cr += ::sqrt(ex * ex + ey * ey + ez * ez);
    }
    s_np = block_sum16(s_np, scratch);
    s_dr = block_sum16(s_dr, scratch);
    s_pd = block_sum16(s_pd, scratch);
    s_cr = block_sum16(s_cr, scratch);
    const double new_sigma2 = (S[4 * M] - 2.0 * s_dr + s_pd) / (s_np * 3.0);
    const double crit = s_cr / (double)M;

    // ---- 6. publish Y, nodes, iteration state
    V4<T> *nodes_w = (V4<T> *)f.nodes;
    for (int m = t; m < M; m += kBig) {
        V4<T> q; q.x = (T)Tn[m]; q.y = (T)Tn[M + m]; q.z = (T)Tn[2 * M + m]; q.w = (T)f.coord[m];
        nodes_w[m] = q;
        f.dminbits[m] = ~0ull;
    }
    for (int e = t; e < 3 * M; e += kBig) {
        f.Y[e] = Tn[e];
        f.Yout[e] = Tn[e] + f.ctr[e / M];
    }
    BSTAMP(5);
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

// retry_only: the launch that follows every k_mstep_mcu launch -- a no-op unless that kernel's finishing workgroup found
// a timed-out hand-off and left the iteration to be redone here (IterState::retry_pending)
template <typename T>
__global__ __launch_bounds__(kBig) void k_mstep_big(const FrameDev *__restrict__ frames, int from_sums, int retry_only) {
    const FrameDev &f = frames[blockIdx.x];
    if (f.st->done) return;
    if (retry_only) {
        const int pending = f.st->retry_pending;
        if (!pending) return;
        __syncthreads();
        if (threadIdx.x == 0) f.st->retry_pending = 0;
    }
    extern __shared__ __attribute__((aligned(16))) char smem[];
    mstep_big_body<T>(f, from_sums, smem);
}

// ---------------------------------------------------------------------------------------------------------------------
// k_mstep_mcu: the same elimination spread over the row blocks' own CUs (one workgroup per 16 rows of the tableau).
//
// Gauss-Jordan updates EVERY row block in every panel, so the row blocks never exchange anything but the panel's 16
// reduced pivot rows U.  Workgroup rb keeps its 16 x Cp row block in MFMA accumulators from assembly to the end (the
// tableau never touches memory); per panel pb
//   * the owner (rb == pb) turns its tiles over through LDS (accumulator order -> thread = column), reduces the 16
//     pivot rows by the row operations of k_mstep_big in the same order (same bits), every live wave on its own,
//     takes the result back into its accumulators and publishes U (<= 40 KB at M = 300) write-through (16-byte sc1
//     buffer stores), drains, and one lane stores the panel's flag (agent scope, relaxed);
//   * everybody else negates its tile (rb, pb) into A-operand order in LDS, one lane polls the flag (relaxed, s_sleep,
//     bounded by the real-time clock), barrier, then C -= L U with four MFMAs per live tile
//     and U read straight from memory in B-operand order (coalesced 512-byte rows, sc1 loads).
// Critical path per panel = reduce (2.8 us) + publish (1) + flag hop (0.6) + one row block's update (1); the trailing
// update of the whole tableau (0.28 of k_mstep_big's 0.51 ms at M = 300) runs beside it on the other CUs.  Every row block contributes
// G[:, its rows] W[its rows]; the last workgroup to arrive (agent-scope ticket, one acquire) adds the shares in a fixed
// order and finishes the iteration (T = Y0 + G W, sigma2, stopping rule, publish).
// Hand-off protocol and its costs: /opt/skills guide "Inter-workgroup communication" (sc1 payload + drained flag,
// relaxed poll + one acquire).  Flags carry (generation, panel): the generation counter lives in the slot's sync words
// and is advanced by the finishing workgroup, so no memset is needed between the iterations of a call.
// sync words (unsigned, zeroed once when the slot is created): [0] generation, [1] arrivals, [2] singular, [16 + pb] flags.
typedef __attribute__((address_space(1))) unsigned gu32;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define TDLO_RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT
constexpr int kTS = 272;            // LDS stride of one 16 x 16 tile of the row block (256 + 16: thread = column reads spread over the banks)
constexpr unsigned long long kSpinTicks = 5000000ull;     // 50 ms of the 100 MHz real-time clock: a hand-off takes microseconds

template <typename T>
__global__ __launch_bounds__(kBig) void k_mstep_mcu(const FrameDev *__restrict__ frames, int from_sums_in) {
    const int from_sums = from_sums_in;
    const FrameDev &f = frames[blockIdx.y];
    IterState *st = f.st;
    if (st->done) return;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int M = f.M, t = threadIdx.x, lane = t & 63, w = t >> 6, rb = blockIdx.x;
    const int nS = 4 * M + 1;
    const int Mp = (M + 15) & ~15, Cp = Mp + 16, nrb = Mp >> 4, ncb = Cp >> 4;
    double *S = (double *)smem;               // nS (+pad)
    double *Sown = S + ((nS + 1) & ~1);       // 64 (3M reserved): P1, PX of the own 16 nodes
    double *Tn = Sown + ((3 * M + 1) & ~1);   // 3M (+pad)
    double *scratch = Tn + ((3 * M + 1) & ~1);// 16
    double *Abuf = scratch + 16;                // 2 x 256: -tile (rb, pb) in A-operand order, double-buffered over panels
    double *Ul = Abuf + 512;                  // ncb x kTS: the row block in B-operand order (= accumulator registers, lane-minor)
    int *flg = (int *)(Ul + (ncb * kTS > 6 * 512 ? ncb * kTS : 6 * 512));   // [0] last arriver, [1] a spin ran into its time limit
    const auto Gg = TDLO_AS_GLOBAL(double, f.G);
    const auto Ub = TDLO_AS_GLOBAL_RW(double, f.Ascr);              // [pb][cb][s4][lane]: published pivot rows
    const auto Tp = Ub + (size_t)Mp * Cp + (size_t)Mp * 16;         // [rb][d][M]: the row blocks' contributions G[:, rows] W[rows] to G W
    gu32 *sync = (gu32 *)(uintptr_t)f.sync;
    const unsigned gen = sync[0];                                    // written by the previous launch's finishing workgroup
    const int cL = lane & 15, gL = lane >> 4;
    if (t < 2) flg[t] = 0;
#ifdef TDLO_MCU_STAMPS
#define MSTAMP(i) do { if (t == 0) f.dbg[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define MSTAMP(i) do { } while (0)
#endif
    if (rb == 0) MSTAMP(0);

    // ---- 1. the E-step's accumulators -> the 4 x 16 sums this row block needs (P1, R of its 16 nodes), lane = (quantity, node).
    //         The sums go to f.sums for the
    //         workgroup that finishes the iteration (every workgroup reading all 256 rows cost 35 us at N >= 64 000).
    {
        const int ii = lane & 15, kk = lane >> 4, irow = 16 * rb + ii;
        const bool valid = irow < M;
        const int e = kk * M + (valid ? irow : 0);
        auto sums_g = (__attribute__((address_space(1))) unsigned long long *)(uintptr_t)f.sums;
        if (from_sums != 1) {
            const int itn = st->it;
            const long long *rows = acc_rows(f, itn);
            if (w == 0) {
                const double a = valid ? acc_read(f, rows, e) : 0.0;
                Sown[lane] = a;
                if (valid) __hip_atomic_store(sums_g + e, (unsigned long long)__double_as_longlong(a), TDLO_RLX_AGENT);
            }
            if (rb == 0) {                     // Q = sum P |x - y|^2, only needed for sigma2; the other parity's rows are cleared for the next E-step
                if (t == 0) __hip_atomic_store(sums_g + 4 * M, (unsigned long long)__double_as_longlong(acc_read(f, rows, 4 * M)), TDLO_RLX_AGENT);
                acc_clear_other<kBig>(f, itn, t);
            }
        } else if (w == 0) Sown[lane] = valid ? f.sums[e] : 0.0;
    }
    __syncthreads();
    if (rb == 0) MSTAMP(1);
    // ---- 2. this workgroup's rows of [A | B] (:392-413), straight into the accumulators: wave w holds column blocks w, w + 16, w + 32
    const double sigma2 = st->sigma2;
    const double c2 = f.lambda * sigma2;
    const int pri = f.has_priors;
    const V4<T> *ndq = (const V4<T> *)f.nodes;
    mfma_d4 C[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const int cb = w + 16 * q;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = 16 * rb + 4 * r + gL, j = 16 * cb + cL;
            double v = 0.0;
            if (cb < ncb) {
                if (i < M) {
                    if (j < M) v = (Sown[4 * r + gL] + (pri ? f.aJ[i] : 0.0)) * Gg[(size_t)i * M + j] + (i == j ? c2 : 0.0);      // G symmetric
                    else if (j >= Mp && j < Mp + 3) {
                        const int d = j - Mp, qq = d * M + i;
                        const double yd = d == 0 ? (double)ndq[i].x : (d == 1 ? (double)ndq[i].y : (double)ndq[i].z);
                        v = Sown[16 * (1 + d) + 4 * r + gL] + Sown[4 * r + gL] * (yd - f.Y0[qq]);
                        if (pri) v += f.aYd[qq];
                    }
                } else if (i == j) v = 1.0;
            }
            C[q][r] = v;
        }
    }

    if (rb == 0) MSTAMP(2);
    // ---- 3. blocked Gauss-Jordan, 16 pivot columns per panel
    int singular = 0, timed_out = 0;
    const size_t utile = (size_t)ncb << 8;                            // doubles per published panel
    // buffer descriptor of the publish area (wave-uniform by construction: readfirstlane of the two address halves)
    const unsigned long long ub_addr = (unsigned long long)(uintptr_t)f.Ascr;
    const unsigned ub_lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)ub_addr), ub_hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(ub_addr >> 32));
    const __amdgpu_buffer_rsrc_t ub_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(uintptr_t)(((unsigned long long)ub_hi << 32) | ub_lo), 0,
                                                                             (int)((size_t)Mp * Cp * sizeof(double)), 0x00020000);
    for (int pb = 0; pb < nrb; ++pb) {
        const int k0 = pb << 4;
        const unsigned epoch = gen * 64u + (unsigned)pb + 1u;
        if (rb == pb) {
            // a. owner: row block -> LDS -> thread = column
            MSTAMP(8 + pb);
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const int cb = w + 16 * q;
                if (cb < ncb) {                   // dead tiles (cb < pb) too: the accumulators are reloaded wholesale below, so
#pragma unroll                            // that they do not occupy registers during the reduction
                    for (int r = 0; r < 4; ++r) Ul[cb * kTS + (r << 6) + lane] = C[q][r];
                }
            }
            __syncthreads();
            // Every wave with live columns runs the 16 steps on its own: besides its 64 columns it carries a copy of the
            // 16 pivot columns (lanes 0..15 of pk), whose j-th column after j steps is the multiplier column of step j and
            // reaches the FMAs as wave-uniform operands (v_readlane).  No barrier and no LDS traffic inside the 16 steps
            // (one barrier + 16 broadcast LDS reads per step and wave had made this phase 8.2 us; the operations on every
            // element are those of k_mstep_big, in the same order).
            const int col = k0 + t;
            const bool act = col < Cp;
            const bool wlive = k0 + 64 * w < Cp;
            double u[16];
            if (wlive) {
                double pk[16];
                const double *ucol = Ul + ((act ? col : k0) >> 4) * kTS + (col & 15);        // + 64 (r >> 2) + 16 (r & 3)
                const double *pkcol = Ul + pb * kTS + cL;
#pragma unroll
                for (int r = 0; r < 16; ++r) { u[r] = ucol[((r >> 2) << 6) + 16 * (r & 3)]; pk[r] = pkcol[((r >> 2) << 6) + 16 * (r & 3)]; }
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    double pc[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const long long b = __double_as_longlong(pk[r]);
                        const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, j), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)((unsigned long long)b >> 32), j);
                        pc[r] = __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
                    }
                    const double pv = pc[j];
                    singular |= (int)((unsigned)(((__double2hiint(pv) >> 20) & 0x7ff) - 1) >= 0x7feu);     // zero / denormal / non-finite pivot
                    const double ri = fast_rcp(pv);
                    const double v = u[j] * ri, vk = pk[j] * ri;
#pragma unroll
                    for (int r = 0; r < 16; ++r) { u[r] = (r == j) ? v : fma(-pc[r], v, u[r]); pk[r] = (r == j) ? vk : fma(-pc[r], vk, pk[r]); }
                    // keep the two chains in step: left alone, the scheduler runs the 16 steps on pk first and parks all 256
                    // multipliers in spill lanes (512 v_writelane + 512 v_readlane)
                    asm volatile("" : "+v"(u[0]), "+v"(u[1]), "+v"(u[2]), "+v"(u[3]), "+v"(u[4]), "+v"(u[5]), "+v"(u[6]), "+v"(u[7]),
                                      "+v"(u[8]), "+v"(u[9]), "+v"(u[10]), "+v"(u[11]), "+v"(u[12]), "+v"(u[13]), "+v"(u[14]), "+v"(u[15]));
                }
            }
            MSTAMP(28 + pb);
            __syncthreads();                      // every wave has fetched its copy of the pivot columns
            if (wlive && act) {
                double *ucol = Ul + (col >> 4) * kTS + (col & 15);
#pragma unroll
                for (int r = 0; r < 16; ++r) ucol[((r >> 2) << 6) + 16 * (r & 3)] = u[r];
            }
            __syncthreads();
            // b. out to the other row blocks (write-through, 16 bytes per lane), then back into the accumulators
            if (pb == 1) MSTAMP(48); if (pb == 8) MSTAMP(56);
            const int n2 = (ncb - pb - 1) << 7;                       // 16-byte pieces of the live tiles
            for (int e = t; e < n2; e += kBig) {
                const int cb = pb + 1 + (e >> 7), off = (e & 127) << 1;
                const u32x4 v = *(const u32x4 *)(Ul + cb * kTS + off);
                __builtin_amdgcn_raw_buffer_store_b128(v, ub_rsrc, (int)(((size_t)pb * utile + ((size_t)cb << 8) + off) * sizeof(double)), 0, /*sc1*/ 16);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // every storing wave drains
            __syncthreads();
            if (t == 0) __hip_atomic_store(sync + 16 + pb, epoch, TDLO_RLX_AGENT);
            if (pb == 1) MSTAMP(49); if (pb == 8) MSTAMP(57);
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const int cb = w + 16 * q;
#pragma unroll
                for (int r = 0; r < 4; ++r) C[q][r] = cb < ncb ? Ul[cb * kTS + (r << 6) + lane] : 0.0;
            }
        } else {
            // c. everybody else: -L in A-operand order, wait for U, C -= L U
            double *Ab = Abuf + (pb & 1) * 256;
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                if (w + 16 * q == pb) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) Ab[(cL >> 2) * 64 + (4 * r + gL) + 16 * (cL & 3)] = -C[q][r];
                }
            }
            if (t == 0 && !flg[1]) {                  // after one time-out this workgroup waits no more: the iteration is redone anyway
                const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
                if (f.force_timeout_it == st->it && rb == nrb - 1) flg[1] = 1;      // test hook (TDLO_MCU_FORCE_TIMEOUT)
                else while (__hip_atomic_load(sync + 16 + pb, TDLO_RLX_AGENT) != epoch) {
                    __builtin_amdgcn_s_sleep(1);
                    if (__builtin_amdgcn_s_memrealtime() - t0 > kSpinTicks) { flg[1] = 1; break; }
                }
            }
            if (rb == pb + 1) { if (pb == 1) MSTAMP(50); if (pb == 8) MSTAMP(58); }
            __syncthreads();
            double a[4];
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) a[s4] = Ab[64 * s4 + lane];
            // U was stored write-through, so loads that bypass this CU's L1 (sc1) see it without an acquire (1.7 us per panel)
            const auto Up = (__attribute__((address_space(1))) unsigned long long *)(Ub + (size_t)pb * utile + lane);
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const int cb = w + 16 * q;
                if (cb > pb && cb < ncb) {
                    double b[4];
#pragma unroll
                    for (int s4 = 0; s4 < 4; ++s4) b[s4] = __longlong_as_double((long long)__hip_atomic_load(Up + ((size_t)cb << 8) + (s4 << 6), TDLO_RLX_AGENT));
#pragma unroll
                    for (int s4 = 0; s4 < 4; ++s4) C[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s4], b[s4], C[q], 0, 0, 0);
                }
            }
        }
    }
    timed_out = flg[1];

    // ---- 4. this row block's share of G W (:417): G[:, its rows] W[its rows], M x 3, handed to the finishing workgroup
    {
        double *Wown = Abuf;                  // [d][16]
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            if (w + 16 * q == nrb && cL < 3) {
#pragma unroll
                for (int r = 0; r < 4; ++r) Wown[cL * 16 + 4 * r + gL] = C[q][r];
            }
        }
        __syncthreads();
        const int i = t & 511, h = t >> 9;
        double v0 = 0, v1 = 0, v2 = 0;
        if (i < M) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int kk = 8 * h + k, row = 16 * rb + kk;
                if (row < M) { const double gk = Gg[(size_t)row * M + i]; v0 += gk * Wown[kk]; v1 += gk * Wown[16 + kk]; v2 += gk * Wown[32 + kk]; }
            }
        }
        double *tmp = Ul;
        if (h == 1 && i < M) { tmp[i] = v0; tmp[512 + i] = v1; tmp[1024 + i] = v2; }
        __syncthreads();
        if (h == 0 && i < M) {
            auto tp = (__attribute__((address_space(1))) unsigned long long *)(Tp + (size_t)rb * 3 * M);
            __hip_atomic_store(tp + i, (unsigned long long)__double_as_longlong(v0 + tmp[i]), TDLO_RLX_AGENT);
            __hip_atomic_store(tp + M + i, (unsigned long long)__double_as_longlong(v1 + tmp[512 + i]), TDLO_RLX_AGENT);
            __hip_atomic_store(tp + 2 * M + i, (unsigned long long)__double_as_longlong(v2 + tmp[1024 + i]), TDLO_RLX_AGENT);
        }
    }
    singular = __syncthreads_or(singular) ? 1 : 0;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t == 0) {
        if (singular | timed_out) __hip_atomic_fetch_or(sync + 2, (unsigned)singular | (timed_out ? 2u : 0u), TDLO_RLX_AGENT);
        const unsigned old = __hip_atomic_fetch_add(sync + 1, 1u, TDLO_RLX_AGENT);
        flg[0] = (old == (unsigned)nrb - 1u);
        if (flg[0]) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    if (!flg[0]) return;
    MSTAMP(3);
    {
        const unsigned fl = __hip_atomic_load(sync + 2, TDLO_RLX_AGENT);
        singular = (int)(fl & 1u);
        if (fl & 2u) {
            // A hand-off ran into its time limit (a workgroup that was not scheduled beside the others: a plain launch gives
            // no residency guarantee): the row blocks hold garbage.  Nothing is published; the sync words are re-armed and
            // the iteration is redone from the same sums by the one-workgroup elimination that follows in the stream
            // (k_mstep_big, retry_only).
            if (t == 0) {
                __hip_atomic_store(sync + 1, 0u, TDLO_RLX_AGENT);
                __hip_atomic_store(sync + 2, 0u, TDLO_RLX_AGENT);
                __hip_atomic_store(sync + 0, gen + 1u, TDLO_RLX_AGENT);
                st->retries += 1; st->retry_pending = 1;
            }
            return;
        }
    }
    // ---- 5. the finishing workgroup: all sums, T = Y0 + sum over the row blocks' shares (fixed order)
    for (int e = t; e < nS; e += kBig) S[e] = f.sums[e];
    for (int e = t; e < 3 * M; e += kBig) {
        double a = 0;
        for (int r0 = 0; r0 < nrb; r0 += 8) {             // 8 loads in flight, row-block order kept
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = Tp[(size_t)(r0 + u < nrb ? r0 + u : nrb - 1) * 3 * M + e];
#pragma unroll
            for (int u = 0; u < 8; ++u) if (r0 + u < nrb) a += v[u];
        }
        Tn[e] = f.Y0[e] + a;
    }
    __syncthreads();

    // ---- 6. sigma2 (residual form of :418-422) and the convergence criterion (:424)
    double s_np = 0, s_dr = 0, s_pd = 0, s_cr = 0;
    for (int m = t; m < M; m += kBig) {
        const double yx = (double)ndq[m].x, yy = (double)ndq[m].y, yz = (double)ndq[m].z;    // nodes as the E-step saw them
        const double p1 = S[m];
        const double dx = Tn[m] - yx, dy = Tn[M + m] - yy, dz = Tn[2 * M + m] - yz;
        s_np += p1;
        s_dr += dx * S[M + m] + dy * S[2 * M + m] + dz * S[3 * M + m];
        s_pd += p1 * (dx * dx + dy * dy + dz * dz);
        const double ex = f.Y[m] - Tn[m], ey = f.Y[M + m] - Tn[M + m], ez = f.Y[2 * M + m] - Tn[2 * M + m];
        s_cr += ::sqrt(ex * ex + ey * ey + ez * ez);
    }
    s_np = block_sum16(s_np, scratch);
    s_dr = block_sum16(s_dr, scratch);
    s_pd = block_sum16(s_pd, scratch);
    s_cr = block_sum16(s_cr, scratch);
    const double new_sigma2 = (S[4 * M] - 2.0 * s_dr + s_pd) / (s_np * 3.0);
    const double crit = s_cr / (double)M;

    // ---- 7. publish Y, nodes, iteration state; re-arm the sync words for the next launch
    V4<T> *nodes_w = (V4<T> *)f.nodes;
    for (int m = t; m < M; m += kBig) {
        V4<T> q; q.x = (T)Tn[m]; q.y = (T)Tn[M + m]; q.z = (T)Tn[2 * M + m]; q.w = (T)f.coord[m];
        nodes_w[m] = q;
        f.dminbits[m] = ~0ull;
    }
    for (int e = t; e < 3 * M; e += kBig) {
        f.Y[e] = Tn[e];
        f.Yout[e] = Tn[e] + f.ctr[e / M];
    }
    MSTAMP(4);
    if (t == 0) {
        __hip_atomic_store(sync + 1, 0u, TDLO_RLX_AGENT);
        __hip_atomic_store(sync + 2, 0u, TDLO_RLX_AGENT);
        __hip_atomic_store(sync + 0, gen + 1u, TDLO_RLX_AGENT);
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

// ---------------------------------------------------------------------------------------------------------------------
// k_mstep_pivot_mcu: the M-step WITH the LLE term (A = c I + (D + sigma2 gamma H + alpha J) G, no SPD structure: partial
// pivoting, trackdlo.cpp:396-415) for more than 128 nodes, where the one-workgroup kernel k_mstep keeps the tableau in
// global memory and is bound by one CU's path to the L2 (M = 300: 5.5 ms).  One workgroup (256 threads) per 16 rows; the
// rows stay in LDS for the whole elimination.  Gauss-Jordan with partial pivoting, one column at a time, ONE hand-off per
// column: every workgroup publishes its best pivot candidate's live entries (<= M + 3 values, write-through), drains, and
// raises its flag, which carries the candidate's row and a 32-bit key of |a_pk|; one wave per workgroup waits for all
// flags and takes the largest key (ties: lowest row) -- every workgroup arrives at the same winner -- and the winner's row is
// read (L1-bypassing loads), normalised and applied to the own 16 rows.  Candidate slots and flags are double-buffered by
// column parity: nobody can publish column k + 2 before everybody has consumed column k + 1, hence finished reading k.
// The pivot rows are not moved (used[] / kof[]: row -> column it pivots); at the end row piv(k) holds W_k, every workgroup
// contributes G[:, its pivot columns] W[...] and the last one to arrive finishes the iteration as in k_mstep_mcu.
// sync words: [0] generation, [1] arrivals, [2] singular, 64-bit flags from word 128: [32 (k & 1) + rb].
constexpr int kPT = 256;

template <typename T>
__global__ __launch_bounds__(kPT) void k_mstep_pivot_mcu(const FrameDev *__restrict__ frames, int from_sums_in) {
    const int from_sums = from_sums_in;
    const FrameDev &f = frames[blockIdx.y];
    IterState *st = f.st;
    if (st->done) return;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int M = f.M, t = threadIdx.x, lane = t & 63, w = t >> 6, rb = blockIdx.x;
    const int nS = 4 * M + 1, Mp = (M + 15) & ~15, nrb = Mp >> 4, Cp = Mp + 16;
    const int NC = M + 3, ld = NC | 1;                 // row stride in LDS (odd)
    const int slotsz = (NC + 2 + 1) & ~1;              // published candidate: {|a|, row} + NC values
    double *S = (double *)smem;                        // nS (+pad): all sums, finishing workgroup only
    double *Tn = S + ((nS + 1) & ~1);                  // 3M (+pad)
    double *scratch = Tn + ((3 * M + 1) & ~1);         // 16
    double *Sown = scratch + 16;                       // 64: P1, PX of the own 16 nodes
    double *prow = Sown + 64;                          // NC (+pad): normalised pivot row of the current column
    double *R = prow + ((NC + 1) & ~1);                // 16 x ld: the own rows of [A | B]
    int *ib = (int *)(R + (16 * ld > 1536 ? 16 * ld : 1536));
    int *used = ib, *kof = ib + 16, *flg = ib + 32;    // flg: [0] last arriver, [1] a spin ran into its time limit
    const auto Gg = TDLO_AS_GLOBAL(double, f.G);
    const auto Cb = (__attribute__((address_space(1))) unsigned long long *)(uintptr_t)f.Ascr;      // [parity][rb][slotsz]
    const auto Ur = TDLO_AS_GLOBAL_RW(double, f.Ascr) + (size_t)Mp * Cp + (size_t)Mp * 16 + (size_t)nrb * 3 * Mp;   // [M][NC]: the factor's rows, by pivot column
    gu32 *sync = (gu32 *)(uintptr_t)f.sync;
    // flags: one 64-bit word per (column parity, workgroup) = {column + 1 : 16 | candidate row : 16 | key of |a| : 32}; the key is
    // the upper half of the fp64 pattern (monotone for non-negative values).  The flag IS the candidate's header (no second
    // dependent load); the finishing workgroup zeroes all flags, so a tag only has to be unique within a launch.
    const auto flag64 = (__attribute__((address_space(1))) unsigned long long *)(sync + 128);
    const unsigned gen = sync[0];
    if (t < 4) flg[t] = 0;
    if (t < 16) { used[t] = (16 * rb + t < M) ? 0 : 1; kof[t] = -1; }      // padding rows never pivot

    // ---- 1. the E-step's accumulators -> the own 4 x 16 sums
    {
        const int ii = lane & 15, kk = lane >> 4, irow = 16 * rb + ii;
        const bool valid = irow < M;
        const int e = kk * M + (valid ? irow : 0);
        auto sums_g = (__attribute__((address_space(1))) unsigned long long *)(uintptr_t)f.sums;
        if (from_sums != 1) {
            const int itn = st->it;
            const long long *rows = acc_rows(f, itn);
            if (w == 0) {
                const double a = valid ? acc_read(f, rows, e) : 0.0;
                Sown[lane] = a;
                if (valid) __hip_atomic_store(sums_g + e, (unsigned long long)__double_as_longlong(a), TDLO_RLX_AGENT);
            }
            if (rb == 0) {
                if (t == 0) __hip_atomic_store(sums_g + 4 * M, (unsigned long long)__double_as_longlong(acc_read(f, rows, 4 * M)), TDLO_RLX_AGENT);
                acc_clear_other<kPT>(f, itn, t);
            }
        } else if (w == 0) Sown[lane] = valid ? f.sums[e] : 0.0;
    }
    __syncthreads();

    // ---- 2. own rows of [A | B] (:392-413)
    {
        const double sigma2 = st->sigma2;
        const double c2 = f.lambda * sigma2, sg = sigma2 * f.lle_weight;
        const int lle = f.include_lle, pri = f.has_priors;
        const V4<T> *ndq = (const V4<T> *)f.nodes;
        for (int e = t; e < 16 * NC; e += kPT) {
            const int r = e & 15, j = e >> 4, i = 16 * rb + r;
            double v = 0.0;
            if (i < M) {
                if (j < M) {
                    const size_t ge = (size_t)j * M + i;              // G symmetric; H G as stored by k_setup
                    v = (Sown[r] + (pri ? f.aJ[i] : 0.0)) * Gg[ge] + (i == j ? c2 : 0.0);
                    if (lle) v += sg * f.HG[ge];
                } else {
                    const int d = j - M, q = d * M + i;
                    const double yd = d == 0 ? (double)ndq[i].x : (d == 1 ? (double)ndq[i].y : (double)ndq[i].z);
                    v = Sown[16 * (1 + d) + r] + Sown[r] * (yd - f.Y0[q]);
                    if (lle) v -= sg * f.HY0[q];
                    if (pri) v += f.aYd[q];
                }
            }
            R[r * ld + j] = v;
        }
    }
    __syncthreads();

    // ---- 3. Gauss-Jordan with partial pivoting across the workgroups, one hand-off per column
    int singular = 0;
    for (int k = 0; k < M; ++k) {
        const unsigned long long tag = (unsigned long long)(k + 1);
        const int par = k & 1;
        // a. local candidate: every wave on its own (wave-uniform result, no LDS hand-off, no barrier)
        int lr; double lav;
        {
            double av = -1.0; int ri = 0x7fffffff;
            if (lane < 16 && !used[lane]) { av = fabs(R[lane * ld + k]); ri = lane; }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const double ov = __shfl_xor(av, o); const int oi = __shfl_xor(ri, o);
                if (ov > av || (ov == av && oi < ri)) { av = ov; ri = oi; }
            }
            lr = av >= 0.0 ? ri : -1; lav = av;
        }
        // b. publish the entries k .. NC-1 of that row, then the flag {column, row, key of |a|}
        {
            const auto slot = Cb + ((size_t)par * nrb + rb) * slotsz;
            if (lr >= 0)
                for (int j = k + t; j < NC; j += kPT) __hip_atomic_store(slot + 2 + j, (unsigned long long)__double_as_longlong(R[lr * ld + j]), TDLO_RLX_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (t == 0) {
                const unsigned key = lr >= 0 ? (unsigned)__double2hiint(lav) : 0u;
                __hip_atomic_store(flag64 + 32 * par + rb, (tag << 48) | ((unsigned long long)(16 * rb + (lr >= 0 ? lr : 0)) << 32) | key, TDLO_RLX_AGENT);
            }
        }
        // c. all candidates in: the winner (every wave polls and decides on its own; the words are final once all tags match)
        int p; bool okp;
        {
            unsigned long long word = 0;
            if (f.force_timeout_it == st->it && rb == nrb - 1 && lane == 0) flg[1] = 1;       // test hook (TDLO_MCU_FORCE_TIMEOUT)
            if (!flg[1]) {
                const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
                for (;;) {
                    if (lane < nrb) word = __hip_atomic_load(flag64 + 32 * par + lane, TDLO_RLX_AGENT);
                    if (__all(lane >= nrb || (word >> 48) == tag)) break;
                    __builtin_amdgcn_s_sleep(1);
                    if (__builtin_amdgcn_s_memrealtime() - t0 > kSpinTicks) { if (lane == 0) flg[1] = 1; break; }
                }
            }
            unsigned key = 0; int ri = 0x7fffffff;
            if (lane < nrb && (word >> 48) == tag) { key = (unsigned)word; ri = (int)((word >> 32) & 0xffffu); }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const unsigned ok_ = (unsigned)__shfl_xor((int)key, o); const int oi = __shfl_xor(ri, o);
                if (ok_ > key || (ok_ == key && oi < ri)) { key = ok_; ri = oi; }
            }
            p = ri; okp = key != 0u && p < M;
        }
        if (!okp) singular = 1;
        // d. the winner's row, normalised
        {
            const auto wsl = Cb + ((size_t)par * nrb + (okp ? (p >> 4) : 0)) * slotsz + 2;
            const double pv = __longlong_as_double((long long)__hip_atomic_load(wsl + k, TDLO_RLX_AGENT));
            const double rp = okp ? 1.0 / pv : 0.0;
            for (int j = k + t; j < NC; j += kPT) prow[j] = __longlong_as_double((long long)__hip_atomic_load(wsl + j, TDLO_RLX_AGENT)) * rp;
        }
        __syncthreads();
        // e. own rows
        if (okp) {
            const int r = t & 15, cs = t >> 4;
            if (16 * rb + r == p) {
                for (int j = k + cs; j < NC; j += 16) R[r * ld + j] = prow[j];
                if (cs == 0) { used[r] = 1; kof[r] = k; }
            } else if (!used[r]) {                    // a row that has served as pivot row is final (Gaussian elimination, not
                const double l = R[r * ld + k];       // Gauss-Jordan: see the back substitution of step 5)
                for (int j = k + 1 + cs; j < NC; j += 16) R[r * ld + j] = fma(-l, prow[j], R[r * ld + j]);
            }
        }
        __syncthreads();
    }
    const int timed_out = flg[1];

    // ---- 4. the own pivot rows (unit diagonal, entries right of it, right-hand sides) go to the finishing workgroup,
    //         stored by the column they pivot: row kof of the upper-triangular factor
    {
        const int r = t & 15, cs = t >> 4, kc = kof[r];
        if (kc >= 0) {
            auto ur = (__attribute__((address_space(1))) unsigned long long *)(Ur + (size_t)kc * NC);
            for (int j = kc + cs; j < NC; j += 16) __hip_atomic_store(ur + j, (unsigned long long)__double_as_longlong(R[r * ld + j]), TDLO_RLX_AGENT);
        }
    }
    singular = __syncthreads_or(singular) ? 1 : 0;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t == 0) {
        if (singular | timed_out) __hip_atomic_fetch_or(sync + 2, (unsigned)singular | (timed_out ? 2u : 0u), TDLO_RLX_AGENT);
        const unsigned old = __hip_atomic_fetch_add(sync + 1, 1u, TDLO_RLX_AGENT);
        flg[0] = (old == (unsigned)nrb - 1u);
        if (flg[0]) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    if (!flg[0]) return;
    {
        const unsigned fl = __hip_atomic_load(sync + 2, TDLO_RLX_AGENT);
        singular = (int)(fl & 1u);
        if (fl & 2u) {
            // time-out in a hand-off (see k_mstep_mcu): nothing is published, sync words and flags are re-armed, the generic
            // pivoted elimination that follows in the stream (k_mstep, retry_only) redoes the iteration
            if (t < 64) __hip_atomic_store(flag64 + t, 0ull, TDLO_RLX_AGENT);
            if (t == 0) {
                __hip_atomic_store(sync + 1, 0u, TDLO_RLX_AGENT);
                __hip_atomic_store(sync + 2, 0u, TDLO_RLX_AGENT);
                __hip_atomic_store(sync + 0, gen + 1u, TDLO_RLX_AGENT);
                st->retries += 1; st->retry_pending = 1;
            }
            return;
        }
    }

    // ---- 5. the finishing workgroup: back substitution on the gathered factor (Gaussian elimination is backward stable;
    //         Gauss-Jordan is only forward stable and its error in W is not damped by the product G W of :417 -- on the
    //         ill-conditioned systems of the pre-processing registration that cost ~1e-8 m per solve in T, against 1e-11 for the
    //         oracle's QR, tests/test_solver_error.py).  Row oriented, eight rows per round: the dot products with the part
    //         of W that is already known are reduced over the workgroup in one go, the 8 x 8 triangle is finished by three
    //         threads (one per right-hand side).  W lives in the dead row area R.
    {
        double *Wl = R;                                // [d][M]
        double *part = S;                              // [4 waves][8 rows][3]: S is filled afterwards
        double *tri = part + 96;                       // [8][8]
        double *rhs = tri + 64;                        // [8][3]
        const int kb0 = ((M - 1) >> 3) << 3;
        // thread = column kb + t + 256 q of the block's eight rows (q < 3: M + 3 <= 515 columns); the entries of the NEXT block
        // are requested before this block's reduction starts (the gathered rows were stored write-through, so every round
        // would otherwise begin with a round trip to memory)
        constexpr int QM = 3;
        double en[8][QM];
        auto fetch = [&](int kbx) {
#pragma unroll
            for (int q = 0; q < QM; ++q) {
                const int j = kbx + t + kPT * q;
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const int kk = kbx + r;
                    const bool on = j < NC && kk < M && j >= kk;
                    en[r][q] = on ? Ur[(size_t)(kk < M ? kk : 0) * NC + (j < NC ? j : 0)] : 0.0;
                }
            }
        };
        fetch(kb0);
        for (int kb = kb0; kb >= 0; kb -= 8) {
            double e[8][QM];
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int q = 0; q < QM; ++q) e[r][q] = en[r][q];
            if (kb >= 8) fetch(kb - 8);
            double acc[8][3];
#pragma unroll
            for (int r = 0; r < 8; ++r) { acc[r][0] = 0; acc[r][1] = 0; acc[r][2] = 0; }
#pragma unroll
            for (int q = 0; q < QM; ++q) {
                const int j = kb + t + kPT * q;
                if (j < NC) {
                    if (j >= M) {                                     // the right-hand sides (first: the top block's columns run into them) ...
#pragma unroll
                        for (int r = 0; r < 8; ++r) rhs[r * 3 + (j - M)] = e[r][q];
                    } else if (j < kb + 8) {                          // ... the triangle (unit diagonal) ...
#pragma unroll
                        for (int r = 0; r < 8; ++r) tri[r * 8 + (j - kb)] = e[r][q];
                    } else {                                          // ... the part of W that is already known
                        const double w0 = Wl[j], w1 = Wl[M + j], w2 = Wl[2 * M + j];
#pragma unroll
                        for (int r = 0; r < 8; ++r) { acc[r][0] += e[r][q] * w0; acc[r][1] += e[r][q] * w1; acc[r][2] += e[r][q] * w2; }
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int d = 0; d < 3; ++d) { const double v = wave_sum(acc[r][d]); if (lane == 0) part[(w * 8 + r) * 3 + d] = v; }
            __syncthreads();
            if (t < 3) {
                double x[8];
#pragma unroll
                for (int r = 7; r >= 0; --r) {
                    const int kk = kb + r;
                    double v = 0.0;
                    if (kk < M) {
                        v = rhs[r * 3 + t] - (((part[r * 3 + t] + part[(8 + r) * 3 + t]) + part[(16 + r) * 3 + t]) + part[(24 + r) * 3 + t]);
#pragma unroll
                        for (int c = 7; c > r; --c) if (kb + c < M) v -= tri[r * 8 + c] * x[c];
                        Wl[t * M + kk] = v;
                    }
                    x[r] = v;
                }
            }
            __syncthreads();
        }
        // T = Y0 + G W (:417): wave = quarter of the k range, lane = node (+ 64 c); the quarters are added in a fixed order
        double *tq = R + 3 * M;                        // [4 quarters][3][M], behind W in the dead row area (16 (M + 3) doubles)
        const int kq = (M + 3) >> 2, kbeg = w * kq, kend = (kbeg + kq) < M ? (kbeg + kq) : M;
        for (int i = lane; i < M; i += 64) {
            double v0 = 0, v1 = 0, v2 = 0;
            for (int k = kbeg; k < kend; k += 8) {
                double g[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) g[u] = Gg[(size_t)((k + u) < kend ? (k + u) : (kend - 1)) * M + i];
#pragma unroll
                for (int u = 0; u < 8; ++u) if (k + u < kend) { v0 += g[u] * Wl[k + u]; v1 += g[u] * Wl[M + k + u]; v2 += g[u] * Wl[2 * M + k + u]; }
            }
            tq[(w * 3 + 0) * M + i] = v0; tq[(w * 3 + 1) * M + i] = v1; tq[(w * 3 + 2) * M + i] = v2;
        }
        __syncthreads();
        for (int e = t; e < 3 * M; e += kPT) Tn[e] = f.Y0[e] + (((tq[e] + tq[3 * M + e]) + tq[6 * M + e]) + tq[9 * M + e]);
        for (int e = t; e < nS; e += kPT) S[e] = f.sums[e];
    }
    __syncthreads();
    const V4<T> *ndq = (const V4<T> *)f.nodes;
    double s_np = 0, s_dr = 0, s_pd = 0, s_cr = 0;
    for (int m = t; m < M; m += kPT) {
        const double yx = (double)ndq[m].x, yy = (double)ndq[m].y, yz = (double)ndq[m].z;    // nodes as the E-step saw them
        const double p1 = S[m];
        const double dx = Tn[m] - yx, dy = Tn[M + m] - yy, dz = Tn[2 * M + m] - yz;
        s_np += p1;
        s_dr += dx * S[M + m] + dy * S[2 * M + m] + dz * S[3 * M + m];
        s_pd += p1 * (dx * dx + dy * dy + dz * dz);
        const double ex = f.Y[m] - Tn[m], ey = f.Y[M + m] - Tn[M + m], ez = f.Y[2 * M + m] - Tn[2 * M + m];
        s_cr += ::sqrt(ex * ex + ey * ey + ez * ez);
    }
    s_np = block_sum(s_np, scratch);
    s_dr = block_sum(s_dr, scratch);
    s_pd = block_sum(s_pd, scratch);
    s_cr = block_sum(s_cr, scratch);
    const double new_sigma2 = (S[4 * M] - 2.0 * s_dr + s_pd) / (s_np * 3.0);
    const double crit = s_cr / (double)M;
    V4<T> *nodes_w = (V4<T> *)f.nodes;
    for (int m = t; m < M; m += kPT) {
        V4<T> q; q.x = (T)Tn[m]; q.y = (T)Tn[M + m]; q.z = (T)Tn[2 * M + m]; q.w = (T)f.coord[m];
        nodes_w[m] = q;
        f.dminbits[m] = ~0ull;
    }
    for (int e = t; e < 3 * M; e += kPT) {
        f.Y[e] = Tn[e];
        f.Yout[e] = Tn[e] + f.ctr[e / M];
    }
    if (t < 64) __hip_atomic_store(flag64 + t, 0ull, TDLO_RLX_AGENT);      // every workgroup has arrived: nobody polls any more
    if (t == 0) {
        __hip_atomic_store(sync + 1, 0u, TDLO_RLX_AGENT);
        __hip_atomic_store(sync + 2, 0u, TDLO_RLX_AGENT);
        __hip_atomic_store(sync + 0, gen + 1u, TDLO_RLX_AGENT);
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

size_t pivot_mcu_lds_bytes(int M) {
    const int nS = 4 * M + 1, NC = M + 3, ld = NC | 1;
    const size_t rr = (size_t)(16 * ld > 1536 ? 16 * ld : 1536);
    const size_t d = (size_t)((nS + 1) & ~1) + (size_t)((3 * M + 1) & ~1) + 16 + 64 + (size_t)((NC + 1) & ~1) + rr;
    return d * sizeof(double) + 40 * sizeof(int) + 2 * sizeof(double) + 16;
}

size_t mcu_lds_bytes(int M) {
    const int nS = 4 * M + 1, Mp = (M + 15) & ~15, Cp = Mp + 16, ncb = Cp >> 4;
    size_t ul = (size_t)ncb * kTS;
    if (ul < 6 * 512) ul = 6 * 512;
    const size_t d = (size_t)((nS + 1) & ~1) + 2 * (size_t)((3 * M + 1) & ~1) + 16 + 512 + ul + 2;
    return d * sizeof(double);
}

size_t big_lds_bytes(int M) {
    const int nS = 4 * M + 1, Mp = (M + 15) & ~15, Cp = Mp + 16;
    size_t d = (size_t)((nS + 1) & ~1) + 2 * (size_t)((3 * M + 1) & ~1) + 16 + 272 + (size_t)16 * (Cp + 16);
    if (d < (size_t)((nS + 1) & ~1) + 2 * (size_t)((3 * M + 1) & ~1) + 288 + 6 * 512) d = (size_t)((nS + 1) & ~1) + 2 * (size_t)((3 * M + 1) & ~1) + 288 + 6 * 512;
    return d * sizeof(double);
}

}  // namespace

// tableau (k_mstep_big) or published pivot rows (k_mstep_mcu) | panel columns | the row blocks' shares of G W
// k_mstep_pivot_mcu: candidate slots (front) ... | the gathered upper-triangular factor M x (M + 3) (behind everything else)
size_t mstep_big_scratch_doubles(int M) { const size_t Mp = ((size_t)M + 15) & ~(size_t)15; return Mp * (Mp + 16) + Mp * 16 + (Mp / 16) * 3 * Mp + Mp * (Mp + 4); }

// TDLO_MSTEP_BIG=1wg keeps the whole elimination in one workgroup (k_mstep_big, the comparator of the tests and of
// scripts/gpu_c5.py); the export-only form of the N-split interface (from_sums == 2) has no elimination and stays there.
static bool mcu_enabled() {
    static const int on = [] { const char *e = getenv("TDLO_MSTEP_BIG"); return (e && e[0] == '1') ? 0 : 1; }();
    return on != 0;
}

// M-step with the LLE term beyond kLdsSolveMaxM nodes: rows over the CUs, pivot search across the workgroups.
// TDLO_MSTEP_LLE=1wg keeps the one-workgroup kernel k_mstep (comparator).
bool mstep_pivot_mcu_enabled() {
    static const int on = [] { const char *e = getenv("TDLO_MSTEP_LLE"); return (e && strstr(e, "1wg")) ? 0 : 1; }();
    return on != 0;
}

hipError_t launch_mstep_pivot_mcu(const FrameDev *fd, const FrameDev *fh, int F, int from_sums, bool f64, hipStream_t s) {
    const int M = fh[0].M;
    const size_t lds = pivot_mcu_lds_bytes(M);
    const dim3 grid((unsigned)((M + 15) >> 4), (unsigned)F);
    hipError_t e;
    if (f64) {
        e = hipFuncSetAttribute((const void *)k_mstep_pivot_mcu<double>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((k_mstep_pivot_mcu<double>), grid, dim3(kPT), lds, s, fd, from_sums);
    } else {
        e = hipFuncSetAttribute((const void *)k_mstep_pivot_mcu<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((k_mstep_pivot_mcu<float>), grid, dim3(kPT), lds, s, fd, from_sums);
    }
    return hipGetLastError();
}

hipError_t launch_mstep_big(const FrameDev *fd, const FrameDev *fh, int F, int from_sums, bool f64, hipStream_t s) {
    const int M = fh[0].M;
    hipError_t e;
    if (from_sums != 2 && mcu_enabled()) {
        const size_t lds = mcu_lds_bytes(M);
        const dim3 grid((unsigned)((M + 15) >> 4), (unsigned)F);
        if (f64) {
            e = hipFuncSetAttribute((const void *)k_mstep_mcu<double>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
            if (g_mstep_ev[0]) hipExtLaunchKernelGGL((k_mstep_mcu<double>), grid, dim3(kBig), lds, s, g_mstep_ev[0], g_mstep_ev[1], 0, fd, from_sums);
            else hipLaunchKernelGGL((k_mstep_mcu<double>), grid, dim3(kBig), lds, s, fd, from_sums);
            e = hipFuncSetAttribute((const void *)k_mstep_big<double>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)big_lds_bytes(M));
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL((k_mstep_big<double>), dim3(F), dim3(kBig), big_lds_bytes(M), s, fd, from_sums, 1);
        } else {
            e = hipFuncSetAttribute((const void *)k_mstep_mcu<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
            if (g_mstep_ev[0]) hipExtLaunchKernelGGL((k_mstep_mcu<float>), grid, dim3(kBig), lds, s, g_mstep_ev[0], g_mstep_ev[1], 0, fd, from_sums);
            else hipLaunchKernelGGL((k_mstep_mcu<float>), grid, dim3(kBig), lds, s, fd, from_sums);
            e = hipFuncSetAttribute((const void *)k_mstep_big<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)big_lds_bytes(M));
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL((k_mstep_big<float>), dim3(F), dim3(kBig), big_lds_bytes(M), s, fd, from_sums, 1);
        }
        return hipGetLastError();
    }
    const size_t lds = big_lds_bytes(M);
    if (f64) {
        e = hipFuncSetAttribute((const void *)k_mstep_big<double>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((k_mstep_big<double>), dim3(F), dim3(kBig), lds, s, fd, from_sums, 0);
    } else {
        e = hipFuncSetAttribute((const void *)k_mstep_big<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((k_mstep_big<float>), dim3(F), dim3(kBig), lds, s, fd, from_sums, 0);
    }
    return hipGetLastError();
}

}  // namespace tdlo
