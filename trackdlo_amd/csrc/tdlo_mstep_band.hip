// tdlo_mstep_band.hip -- M-step (trackdlo.cpp:392-437) WITH the LLE term as a banded L D L^T along the chain: O(M), any M <= 512.
//
// What the reference solves each iteration (:396-415, include_lle == true -- the pre-processing registration of every
// tracking_step, :925-927):
//     (c I + (D + g H) G) W = B,   T = Y0 + G W,     c = lambda sigma2,  g = lle_weight sigma2,  D = diag(P1) + alpha J,
//     B = PX - P1 Y0 - g H Y0 (+ alpha (Y_ext - Y0)),     H = (I - L)^T (I - L)  (:236-237),
// a dense M x M system without structure the dense kernels could use (k_mstep_fast<pivoted>: Gaussian elimination with
// partial pivoting, M dependent column steps on an M x M tableau: 36 us at M = 50, 1.5 ms at M = 300).  With V = G W it reads
//     (c G^-1 + D + g H) V = B.
// G (:233) is the Matern-3/2 covariance over the chain coordinate, Markov in the state x_i = (f_i, f'_i) (tdlo_mstep_chain.hip):
// c G^-1 is the Schur complement, on the f components, of c K, K the block-tridiagonal joint precision of the states,
//     K = e_0 Pinf^-1 e_0^T + sum_i [-Phi_i^T; I] Q_i^-1 [-Phi_i, I]           (link i between nodes i - 1 and i),
// and L has +-3 chain neighbours per row (:92-117), so H reaches 6 nodes.  With E the selector of the f components
//     (c K + E^T (D + g H) E) x = E^T B,    V = E x
// is symmetric positive definite and BANDED: 2M unknowns (f_0, f'_0, f_1, f'_1, ...), half-bandwidth 12.  It is the same linear
// system, so the result is the reference's up to rounding -- and the rounding is smaller: against an 80-bit solve of the dense
// system (tests/test_mstep_band.py: chains of 4 .. 512 nodes, sigma2 1e-3 .. 1e-7, the real ill-conditioned H with entries up to
// 1e6) the banded form stays at 1e-16 .. 1e-13 m where partial-pivot LU of the dense matrix leaves 1e-13 .. 5e-10 m.  The price:
// K contains Q^-1 ~ 1 / h^3, so nodes closer than about a millimetre make it ill-conditioned (coincident nodes: infinite);
// such chains, and an H_override that is not banded, keep the dense pivoted kernels (prepare_frame decides, FrameDev::lle_band).
// The same entries limit the form at the other end of the scale: rounded to fp64 they come back in the solution as ~ eps sigma2 K |x| / P1, 1e-13 m
// at the reference's sigma2 but 1e-9 .. 1e-8 m at sigma2 = 3 .. 6 m2 with beta = 5 (a registration started from sigma2 = 0 on a chain of 8 .. 10 m) --
// the fp64 mode's tolerance is 1e-9 m, so there the M-step ends the call like a bad pivot and the host repeats it on the dense kernels
// (FrameDev::band_s2_max, prepare_frame; scripts/gpu_band_cond_study.py).
//
// Everything but D and B is fixed for the whole registration after division by sigma2,
//     (lambda K + lle_weight H + D / sigma2) x = E^T B / sigma2:
// k_setup writes one RECORD of 16 doubles per unknown n -- column n of (lambda K + lle_weight H), rows n - 12 .. n, each at the
// position of its row's SLOT (row mod 13); unknowns that do not exist are identity records -- and the M-step adds D_a / sigma2 to the
// diagonal entry of the even records and puts the right-hand side into the three spare positions.
//
// The elimination is L D L^T without pivoting inside the band (stable: the matrix is SPD), right-looking: one rank-1 update of the
// 13 x 13 window per unknown.  The window lives in the accumulator of ONE v_mfma_f64_16x16x4: row / column slot = unknown mod 13 (a
// circular window: nothing is ever shifted), columns 13..15 = the three right-hand sides, which are eliminated along.  Tile
// element (row slot q, column c) sits in lane c + 16 (q % 4), register q / 4: the pivot row k is the register (k % 13) / 4 of lane
// group (k % 13) % 4 -- already in the MFMA's B-operand layout for k-slot (k % 13) % 4, and, the window being symmetric, also in the
// A-operand layout of the pivot column.  One step:
//     t = C[r] masked to the pivot's lane group  (+ the right-hand side of the row that entered last, in another k-slot)
//     a = t * (-1 / d_k)                         (+ the unit vector of that row's slot in the same k-slot)
//     C += a t^T                                 one MFMA: the rank-1 update of window and right-hand sides, and the entering row's b
//     C[.][slot k] = column k + 13               the slot of the eliminated unknown is SET to the entering column (4 FMAs with a 0/1 mask)
// `a` (= -l_ik in the slots, -y_k / d_k in lanes 13..15) is the step's record for the back substitution and overwrites the dead
// column record k in LDS.  Only the entries of a row at or right of its diagonal (in the order of the unknowns) are ever read: the pivot
// row's 13 slots are exactly those; what the update leaves in a slot's row after its elimination is never looked at again.
// Back substitution L^T x = D^-1 y column by column from the last unknown: lane = (slot, right-hand side); when x_k is final it is
// broadcast inside its row of 16 lanes (DPP) and every waiting row takes acc -= l_ki x_k, the l_ki gathered from the records.
//
// What bounds it: like the chain smoother a chain of dependent instructions on ONE wave (8 cycles per dependent instruction, a
// v_mfma_f64_16x16x4 64 + ...): the kernel is designed by the length of the step's dependency chain and its instruction count.
#include "tdlo_devcommon.h"
#include <hip/hip_ext.h>
#include <type_traits>
#include <atomic>
#include <cstdlib>
#include <cstring>

namespace tdlo {
extern thread_local hipEvent_t g_mstep_ev[2];       // tdlo_device.hip: start/stop events for the M-step dispatch (tdlo_profile_iteration)
namespace {

constexpr int kBB = 256;               // workgroup size
constexpr int kNS = kBandSlots;        // slots of the circular window = half-bandwidth + 1
typedef double dbl2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ double rl_f64(double v, int src_lane) {     // src_lane wave-uniform
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src_lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src_lane);
    return __hiloint2double(hi, lo);
}
// the value of lane L of the own row of 16 lanes (DPP row_newbcast, two 32-bit halves)
template <int L> __device__ __forceinline__ double row_bcast(double v) {
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), 0x150 + L, 0xf, 0xf, false);     // (every lane has a source: no `old` operand to set up)
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), 0x150 + L, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
// -1 / d: v_rcp_f64 and one third-order correction (e = 1 - d r0, r = r0 (1 + e + e^2): the seed's ~2^-23 becomes 2^-69)
__device__ __forceinline__ double neg_rcp(double d) {
    const double r0 = __builtin_amdgcn_rcp(d);
    const double e = fma(-d, r0, 1.0);
    const double p = fma(e, e, e);
    return fma(-r0, p, -r0);
}

}  // namespace

size_t band_record_doubles(int M) { const BandPlan bp(M); return (size_t)16 * (bp.nRecT + bp.nRecB); }
size_t mstep_band_lds_bytes(int M) { return BandPlan(M).lds_doubles(M) * sizeof(double); }

template <typename T, bool SINGLE, bool XCH>
__global__ __launch_bounds__(kBB) void k_mstep_band(const FrameDev *__restrict__ frames, const FrameDev f0, int from_sums) {
    constexpr int MB = kBB;
    if (!SINGLE) __builtin_amdgcn_s_setprio(3);      // (a batch: the other stream groups' E-steps share the SIMD -- tdlo_mstep_chain.hip)
    const FrameDev &f = SINGLE ? f0 : frames[blockIdx.x];
    IterState *st = f.st;
    const int M = f.M, t = threadIdx.x, lane = t & 63;
    const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
    const int nS = 4 * M + 1;
    const BandPlan bp(M);
    const int tw = bp.tw;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // LDS (BandPlan::lds_doubles): sums | scratch | direction 0: dump, zeros, records | direction 1: the same | merge tiles
    double *S = (double *)smem, *red = S + ((4 * M + 2 + 1) & ~1);
    constexpr int kDump = 16 * 28 + 64, kZero = 16 * 28;
    double *dump0 = red + 32, *zero0 = dump0 + kDump, *rec0 = zero0 + kZero;
    double *dump1 = rec0 + 16 * (size_t)bp.nRecT, *zero1 = dump1 + kDump, *rec1 = zero1 + kZero;
    double *tiles = tw ? rec1 + 16 * (size_t)bp.nRecB : dump1;
#ifdef TDLO_CHAIN_STAMPS
#define BSTAMP(i) do { if (t == 0) f.dbg[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define BSTAMP(i) do { } while (0)
#endif
    BSTAMP(0);
    const auto stg = TDLO_AS_GLOBAL(IterState, st);
    const int done = stg->done;
    const double sigma2 = stg->sigma2;
    const int pri = f.has_priors;
    const double ctr0 = f.ctr[0], ctr1 = f.ctr[1], ctr2 = f.ctr[2];
    const auto ndg = TDLO_AS_GLOBAL(V4<T>, f.nodes);
    const auto Yg = TDLO_AS_GLOBAL(double, f.Y);
    const auto Y0g = TDLO_AS_GLOBAL(double, f.Y0);
    const auto aJg = TDLO_AS_GLOBAL(double, f.aJ);
    const auto aYg = TDLO_AS_GLOBAL(double, f.aYd);
    const auto HYg = TDLO_AS_GLOBAL(double, f.HY0);
    const auto bandg = TDLO_AS_GLOBAL(dbl2, f.band);

    // ---- 1. everything that comes from memory is requested up front: the E-step's sums (both parities: no load waits for the
    //         iteration counter), this thread's node, the first column records
    struct NodeQ { double y[3], y0[3], yp[3], ay[3], hy[3], aj, w; };
    auto load_node = [&](int m) __attribute__((always_inline)) {
        NodeQ q;
        const int mc = m < M ? m : M - 1;
        q.y[0] = (double)ndg[mc].x; q.y[1] = (double)ndg[mc].y; q.y[2] = (double)ndg[mc].z; q.w = (double)ndg[mc].w;
#pragma unroll
        for (int d = 0; d < 3; ++d) { q.y0[d] = Y0g[d * M + mc]; q.yp[d] = Yg[d * M + mc]; q.ay[d] = pri ? aYg[d * M + mc] : 0.0; q.hy[d] = HYg[d * M + mc]; }
        q.aj = pri ? aJg[mc] : 0.0;
        return q;
    };
    // where unknown 2m (the f component of node m) lives: direction 0 record 2m up to limT, else direction 1 record nUp - 1 - 2m
    auto rec_of_node = [&](int m) __attribute__((always_inline)) -> double * {
        const int u = 2 * m;
        return u < bp.limT ? rec0 + 16 * (size_t)u : rec1 + 16 * (size_t)(bp.nUp - 1 - u);
    };
    auto slot_of_node = [&](int m) __attribute__((always_inline)) { const int u = 2 * m; return (u < bp.limT ? u : bp.nUp - 1 - u) % kNS; };
    const int itn = stg->it;
    double sq[9];
    sq[0] = acc_read_both(f, t < nS ? t : nS - 1, itn);
#pragma unroll
    for (int u = 1; u < 9; ++u) sq[u] = 0.0;
    constexpr int RQ = 5;                               // column records through registers: 5 x 16 bytes per thread cover 50 nodes
    const int nT2 = 8 * bp.nRecT, nR2 = 8 * (bp.nRecT + bp.nRecB);      // dbl2 elements of direction 0's records / of all records
    dbl2 rq[RQ];
#pragma unroll
    for (int u = 0; u < RQ; ++u) { const int i = t + u * MB; rq[u] = bandg[i < nR2 ? i : nR2 - 1]; }
    if (nS > MB && from_sums != 1) {
#pragma unroll
        for (int u = 1; u < 9; ++u) { const int i = t + u * MB; if (i < nS) sq[u] = acc_read_both(f, i, itn); }
    }
    const NodeQ q0 = load_node(t);
    const double rs = fast_rcp(sigma2);
    const double gam = f.lle_weight;
    const double Nc = stg->Nc;
    const double kc = f.mu / (1.0 - f.mu) * (f.vis_branch ? 1.0 / Nc : (double)M / Nc);
    if (done) {
        if (XCH && from_sums == 3) xch_post_error(f, st, t);
        if (!XCH && t < 64 && stg->status != 0) host_publish(f, st, lane, false);      // a registration that ended on an error somewhere else (E-step, setup)
        return;
    }
    BSTAMP(1);
    if (from_sums != 1) {
#pragma unroll
        for (int u = 0; u < 9; ++u) { const int i = t + u * MB; if (i < nS) S[i] = sq[u]; }
    } else {
        const auto sums = TDLO_AS_GLOBAL(double, f.sums);
        for (int i = t; i < nS; i += MB) S[i] = sums[i];
    }
    {
        dbl2 *r0 = (dbl2 *)rec0, *r1 = (dbl2 *)rec1;
        auto put = [&](int i, dbl2 v) __attribute__((always_inline)) { if (i < nT2) r0[i] = v; else if (i < nR2) r1[i - nT2] = v; };
#pragma unroll
        for (int u = 0; u < RQ; ++u) put(t + u * MB, rq[u]);
        for (int i = t + RQ * MB; i < nR2; i += MB) put(i, bandg[i]);
    }
    for (int i = t; i < kZero; i += MB) { zero0[i] = 0.0; if (tw) zero1[i] = 0.0; }
    __syncthreads();
    if (SINGLE && !XCH && from_sums == 0 && f.pair_sums != nullptr && itn == 0) {      // tracking_step's main registration starts from these (FrameDev::pair_sums)
        for (int i = t; i < nS; i += MB) f.pair_sums[i] = S[i];
    }
    if (from_sums == 2) {       // split mode, export only
        acc_clear_other<MB>(f, itn, t);
        for (int i = t; i < nS; i += MB) f.sums[i] = S[i];
        if (t == 0) f.sums[nS] = (double)stg->N;
        return;
    }
    if (XCH && from_sums == 3 && (f.xch_nranks > 1 || f.xch_self)) {      // (a lone rank: its own sums are the total)
        // N-split with the one-shot exchange (see k_mstep_chain): sums to every peer's inbox, flag, wait for the R flags in the
        // own inbox, add the R contributions in rank order
        const int R = f.xch_nranks, me = f.xch_rank, Mc = f.xch_mcap, it = stg->it, par = it & 1;
        const unsigned long long tag = ((unsigned long long)f.xch_epoch << 32) | (unsigned)(it + 1);
        const size_t so = xch_off_sums(R, Mc), sl = 4 * (size_t)Mc + 2;
        for (int i = t; i < nS; i += MB) {
            const double v = S[i];
            for (int q = 0; q < R; ++q) xch_store_f64(xch_ptr(f.xch_inbox[q]) + so + ((size_t)par * R + me) * sl + i, v);
        }
        xch_release();
        __syncthreads();
        if (t < R) xch_store(xch_ptr(f.xch_inbox[t]) + xch_off_flag_sums(R) + par * R + me, tag);
        const xch_word *own = xch_ptr(f.xch_inbox[me]);
        if (t == 0) red[24] = 1.0;
        __syncthreads();
        if (t < R) { const int w_ = xch_wait_sums(own + xch_off_flag_sums(R) + par * R + t, tag); if (w_ != 1) red[24] = w_ == 0 ? 0.0 : -1.0; }      // (-1: that peer's own shard failed, kXchErrMark)
        __syncthreads();
        xch_acquire();
        if (red[24] != 1.0) { if (t == 0) { st->status = red[24] == 0.0 ? TDLO_E_EXCHANGE : TDLO_E_NUMERIC; st->done = 1; st->converged = 0; } return; }
        for (int i = t; i < nS; i += MB) {
            double a = 0;
            for (int r = 0; r < R; ++r) a += xch_load_f64(own + so + ((size_t)par * R + r) * sl + i);
            S[i] = a;
        }
        __syncthreads();
    }
    BSTAMP(2);

    // ---- 2. thread = node: D_a / sigma2 onto the diagonal of the node's f record, the right-hand side
    //         B / sigma2 = (R + P1 (y - Y0) + alpha (Y_ext - Y0)) / sigma2 - lle_weight H Y0   into its spare positions
    //         (the E-step delivers R = PX - P1 y, y = the nodes as it saw them)
    for (int m = t, r = 0; m < M; m += MB, ++r) {
        const NodeQ q = r == 0 ? q0 : load_node(m);
        const double p1 = S[m];
        double *o = rec_of_node(m);
        o[band_rec_pos(slot_of_node(m))] += (p1 + q.aj) * rs;
#pragma unroll
        for (int d = 0; d < 3; ++d) o[4 * d + 7] = fma(S[(1 + d) * M + m] + (p1 * (q.y[d] - q.y0[d]) + q.ay[d]), rs, -(gam * q.hy[d]));
    }
    __syncthreads();
    BSTAMP(3);

    // ---- 3. elimination and back substitution: wave 0 from the chain's head, wave 1 (twisted plan) from its tail -- the same code on
    //         their own records; the other waves clear the other parity's accumulator rows for the next E-step
    int bad_pivot = 0;
    const bool worker = wv == 0 || (tw && wv == 1);
    if (!worker && from_sums != 1) { if (tw) { if (wv >= 2) acc_clear_other<MB - 128>(f, itn, t - 128); } else acc_clear_other<MB - 64>(f, itn, t - 64); }
    const int c = lane & 15, gl = lane >> 4;
    double *const recD = wv == 1 ? rec1 : rec0;
    const unsigned recB = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void *)recD;
    const unsigned zeroB = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void *)(wv == 1 ? zero1 : zero0);
    const unsigned dumpB = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void *)(wv == 1 ? dump1 : dump0) + 8u * (unsigned)lane;
    auto lds_d2 = [](unsigned addr) __attribute__((always_inline)) { return *(const dbl2 *)(__attribute__((address_space(3))) const void *)(uintptr_t)addr; };
    auto lds_d1 = [](unsigned addr) __attribute__((always_inline)) { return *(const double *)(__attribute__((address_space(3))) const void *)(uintptr_t)addr; };
    auto lds_w1 = [](unsigned addr, double v) __attribute__((always_inline)) { *(double *)(__attribute__((address_space(3))) void *)(uintptr_t)addr = v; };
    mfma_d4 C = {0.0, 0.0, 0.0, 0.0};
    // column j of the window's matrix: the four registers of the lanes that own tile column j % 13 (zeros for every other lane)
    struct Col { dbl2 lo, hi; };
    auto load_col = [&](int P, unsigned chunkB, int ahead) __attribute__((always_inline)) {     // record (chunk base) + ahead
        const unsigned a = (c == P ? chunkB + 32u * (unsigned)gl : zeroB) + 128u * (unsigned)ahead;
        Col r; r.lo = lds_d2(a); r.hi = lds_d2(a + 16);
        return r;
    };
    auto set_col = [&](int P, const Col &n) __attribute__((always_inline)) {
        const double keep = c == P ? 0.0 : 1.0;
        C[0] = fma(C[0], keep, n.lo.x); C[1] = fma(C[1], keep, n.lo.y); C[2] = fma(C[2], keep, n.hi.x); C[3] = fma(C[3], keep, n.hi.y);
    };
    // Steps come in chunks of 13 with static slots.  Carried from step to step: C = the result of the last MFMA, `ncp` = the column that
    // enters the slot it freed (k + 12 into slot Pm; still to be SET), `rb` = the right-hand side of the same unknown (enters through
    // the k-slot gs = g ^ 2 of this step's MFMA, onto what the elimination left in that slot's row: a rounding-level multiple of
    // the eliminated row's y -- a perturbation of b of relative size 1e-16, as every solver commits).  The pivot row is extracted
    // from the raw MFMA result with the pending column folded in (one FMA: t = C[r] m + add, m = 0/1 mask of the pivot's lane group
    // without the pending column, add = that column's entry + the entering b).  A lone wave cannot issue anything while its
    // v_mfma_f64_16x16x4 runs (scripts/ubench/mfma64.hip: 65 clocks, and eight independent VALU instructions behind it add their full
    // 46), so a step costs the MFMA plus every other instruction: the step is written for instruction count -- the reciprocal of the
    // pivot (v_rcp_f64 + a third-order correction, wave-uniform) sits on the chain; predicting it a step ahead costs more
    // instructions than it hides.
    unsigned chunkB = recB;                                           // record kb
    Col ncp, ncq;
    double rb = 0.0, rbq = 0.0;
    auto step = [&](auto PC) __attribute__((always_inline)) {
        constexpr int P = decltype(PC)::value, g = P & 3, r = P >> 2, Pm = (P + kNS - 1) % kNS, gs = g ^ 2;
        constexpr int P1 = (P + 1) % kNS, P2 = (P + 2) % kNS, gs2 = (P2 & 3) ^ 2;
        const double ncr = r == 0 ? ncp.lo.x : (r == 1 ? ncp.lo.y : (r == 2 ? ncp.hi.x : ncp.hi.y));
        const double add2 = fma(ncr, gl == g ? 1.0 : 0.0, rb);
        const double tt = fma(C[r], (gl == g && c != Pm) ? 1.0 : 0.0, add2);
        const double d = rl_f64(tt, P + 16 * g);
        bad_pivot |= __double2hiint(d);                               // a negative pivot (the matrix is not positive definite in floating point); zero
                                                                      // and non-finite pivots end in a non-finite sigma2
        const double nr = neg_rcp(d);
        const double a = fma(tt, nr, (lane == Pm + 16 * gs) ? 1.0 : 0.0);
        set_col(Pm, ncp);
        C = __builtin_amdgcn_mfma_f64_16x16x4f64(a, tt, C, 0, 0, 0);
        // the step's record; the requests for step k + 2: column k + 14 (enters slot P1), right-hand side of row k + 14
        lds_w1((gl == g ? chunkB + 8u * (unsigned)c : dumpB) + 128u * (unsigned)P, a);
        ncp = ncq; rb = rbq;
        ncq = load_col(P1, chunkB, P + kNS + 1);
        rbq = lds_d1(((c >= 13 && gl == gs2) ? chunkB + 8u * (unsigned)(4 * (c - 13) + 7) : zeroB) + 128u * (unsigned)(P + kNS + 1));
    };
    auto chunk = [&]() __attribute__((always_inline)) {
        step(std::integral_constant<int, 0>()); step(std::integral_constant<int, 1>()); step(std::integral_constant<int, 2>());
        step(std::integral_constant<int, 3>()); step(std::integral_constant<int, 4>()); step(std::integral_constant<int, 5>());
        step(std::integral_constant<int, 6>()); step(std::integral_constant<int, 7>()); step(std::integral_constant<int, 8>());
        step(std::integral_constant<int, 9>()); step(std::integral_constant<int, 10>()); step(std::integral_constant<int, 11>());
        step(std::integral_constant<int, 12>());
        chunkB += 128u * kNS;
    };
    const int nchA = wv == 1 ? bp.cB : bp.cT;                             // chunks before the two sides meet (one direction: all of them)
    if (worker) {
        // the first window: lane (c, gl) of column c < 13 reads its four registers straight from record c (32 consecutive bytes: the rows
        // at or above the diagonal; a record's positions for rows below are zeros); the right-hand sides of rows 0 .. 12 go to the
        // lanes c >= 13 (register r of group gl: row slot 4 r + gl)
        {
            const unsigned a0 = c < kNS ? recB + 128u * (unsigned)c + 32u * (unsigned)gl : zeroB;
            const dbl2 lo = lds_d2(a0), hi = lds_d2(a0 + 16);
            const int d = c - 13;
            double v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int q = 4 * r + gl;
                v[r] = lds_d1((c >= 13 && q < kNS) ? recB + 128u * (unsigned)q + 8u * (unsigned)(4 * d + 7) : zeroB);
            }
            C[0] = lo.x + v[0]; C[1] = lo.y + v[1]; C[2] = hi.x + v[2]; C[3] = hi.y + v[3];
        }
        BSTAMP(8);
        ncp = load_col(kNS - 1, recB, kNS - 1);                           // (column 12 once more: setting it twice changes nothing)
        ncq = load_col(0, recB, kNS);                                     // column 13: pending at step 1
        rbq = lds_d1((c >= 13 && gl == ((1 & 3) ^ 2)) ? recB + 128u * kNS + 8u * (unsigned)(4 * (c - 13) + 7) : zeroB);     // b of row 13: enters at step 1
        for (int kb = 0; kb < nchA; ++kb) chunk();
    }
    if (tw) {
        // The two sides meet: each wave's window holds, in its own order, the 12 unknowns R between them -- direction 0 with R's own
        // matrix entries and right-hand sides minus its updates, direction 1 with minus its updates only (k_setup left R x R out of its
        // records).  Slot s of one direction is slot 11 - s of the other (mT and mB are whole chunks; the 13th slot is an identity
        // unknown on both sides).  A window keeps only the entries at or right of the diagonal IN ITS OWN ORDER valid, and the other
        // side's order is the reverse: element (row q, column c) of this window pairs with the other's element (row 11 - c, column
        // 11 - q) -- the transposed position, which is the valid one there (right-hand sides: row 11 - q, the same column).  Both
        // waves add the other's tile and finish R themselves (one more chunk).
        if (worker) {
            double *mine = tiles + 256 * wv;
#pragma unroll
            for (int r = 0; r < 4; ++r) mine[64 * r + lane] = C[r];
        }
        __syncthreads();
        if (worker) {
            const double *other = tiles + 256 * (1 - wv);
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const int q = 4 * r + gl;                                 // my row slot (0 .. 11)
                const int orow = c < 12 ? 11 - c : 11 - q, ocol = c < 12 ? 11 - q : c;
                const double v = other[64 * (orow >> 2) + ocol + 16 * (orow & 3)];
                C[r] += c != 12 ? v : 0.0;
            }
            chunk();
        }
    }
    BSTAMP(4);
    if (worker) {
        // back substitution L^T x = D^-1 y from the last unknown; lane = (slot c, right-hand side gl).  acc[c] = z_i - sum_k l_ki x_k of the
        // row i in slot c over the unknowns k > i already final.  Step k (slot P): x_k = acc[P], broadcast inside the row of 16 lanes
        // (DPP), every waiting row takes acc -= l_ki x_k (l_ki from the record of step i, position of slot k) -- and lane P, whose
        // "l" is the record's own pivot position a_P = -d_k / d_k = -1 up to rounding, falls back to ~0 for the row k - 13 that
        // takes the slot over (what stays is a rounding-level multiple of x_k).  Chunks of 13 steps: a chunk's 13 l-vectors and the
        // z of the rows that take its slots over are requested one chunk ahead (addresses do not depend on the walk), the z are
        // added and the x written once per chunk: inside a chunk the chain is  broadcast -> multiply-add  per step.
        wave_lds_sync();
        struct BQ { double L[kNS], zn; };
        auto bload = [&](unsigned cb) __attribute__((always_inline)) {      // chunk whose first record is cb (the 28 records in front of record 0 are zeros)
            BQ q;
            const unsigned b0 = cb + 128u * (unsigned)c, b1 = b0 - 128u * kNS;
#pragma unroll
            for (int P = 0; P < kNS; ++P) q.L[P] = lds_d1((c > P ? b1 : b0) + 8u * (unsigned)P);       // slot c < P: unknown kb + c, slot c > P: kb - 13 + c
            q.zn = lds_d1((c < kNS && gl < 3) ? b0 + 8u * (unsigned)(13 + gl) : zeroB);                // -z of unknown kb + c
            return q;
        };
        unsigned cb = chunkB - 128u * kNS;                                // record of the last chunk's first unknown
        BQ cur = bload(cb);
        double acc = -cur.zn;
        for (int kb = nchA + tw - 1; kb >= 0; --kb) {
            const BQ nxt = bload(cb - 128u * kNS);
            double xs = 0.0;
            auto bstep = [&](auto PC) __attribute__((always_inline)) {
                constexpr int P = decltype(PC)::value;
                const double xb = row_bcast<P>(acc);
                xs = fma(acc, c == P ? 1.0 : 0.0, xs);
                acc = fma(cur.L[P], xb, acc);
            };
            bstep(std::integral_constant<int, 12>()); bstep(std::integral_constant<int, 11>()); bstep(std::integral_constant<int, 10>());
            bstep(std::integral_constant<int, 9>()); bstep(std::integral_constant<int, 8>()); bstep(std::integral_constant<int, 7>());
            bstep(std::integral_constant<int, 6>()); bstep(std::integral_constant<int, 5>()); bstep(std::integral_constant<int, 4>());
            bstep(std::integral_constant<int, 3>()); bstep(std::integral_constant<int, 2>()); bstep(std::integral_constant<int, 1>());
            bstep(std::integral_constant<int, 0>());
            lds_w1((c < kNS && gl < 3) ? cb + 128u * (unsigned)c + 8u * (unsigned)(13 + gl) : dumpB, xs);
            acc -= nxt.zn;
            cur = nxt;
            cb -= 128u * kNS;
        }
        BSTAMP(5);
    }
    // (sigma2 beyond what the stored band resolves to the fp64 mode's tolerance, FrameDev::band_s2_max: the same verdict as a bad pivot -- the host
    //  repeats the call on the dense kernels.  Not in the one-shot exchange form: chains of more than 64 nodes have no dense kernel there.)
    bad_pivot = __syncthreads_or((bad_pivot < 0 || (!XCH && sigma2 > f.band_s2_max)) ? 1 : 0);
    BSTAMP(6);

    // ---- 4. T = Y0 + V, sigma2 (residual form of :418-422) and the convergence criterion (:424); publish Y and the nodes.  thread = node
    V4<T> *nodes_w = (V4<T> *)f.nodes;
    double s_np = 0, s_dr = 0, s_pd = 0, s_cr = 0;
    for (int m = t, r = 0; m < M; m += MB, ++r) {
        const NodeQ q = r == 0 ? q0 : load_node(m);
        const double *o = rec_of_node(m) + 13;
        const double p1 = S[m];
        double Td[3], cr2 = 0, dr = 0, pd2 = 0;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            Td[d] = q.y0[d] + o[d];
            const double del = Td[d] - q.y[d], ex = q.yp[d] - Td[d];
            dr = fma(del, S[(1 + d) * M + m], dr); pd2 = fma(del, del, pd2); cr2 = fma(ex, ex, cr2);
        }
        s_np += p1; s_dr += dr; s_pd += p1 * pd2; s_cr += ::sqrt(cr2);
        V4<T> w; w.x = (T)Td[0]; w.y = (T)Td[1]; w.z = (T)Td[2]; w.w = (T)q.w;      // .w = chain coordinate, unchanged
        nodes_w[m] = w;
        f.dminbits[m] = ~0ull;
#pragma unroll
        for (int d = 0; d < 3; ++d) { f.Y[d * M + m] = Td[d]; f.Yout[d * M + m] = Td[d] + (d == 0 ? ctr0 : (d == 1 ? ctr1 : ctr2)); }
    }
    s_np = wave_sum(s_np); s_dr = wave_sum(s_dr); s_pd = wave_sum(s_pd); s_cr = wave_sum(s_cr);
    if (lane == 0) { red[4 * wv] = s_np; red[4 * wv + 1] = s_dr; red[4 * wv + 2] = s_pd; red[4 * wv + 3] = s_cr; }
    __syncthreads();
    BSTAMP(7);
    int pub = 0;        // lane 0: this M-step has something to tell the host (results mailbox, FrameDev::host_prog)
    if (t == 0) {
        const double t_np = ((red[0] + red[4]) + red[8]) + red[12], t_dr = ((red[1] + red[5]) + red[9]) + red[13];
        const double t_pd = ((red[2] + red[6]) + red[10]) + red[14], t_cr = ((red[3] + red[7]) + red[11]) + red[15];
        const double new_sigma2 = (S[4 * M] - 2.0 * t_dr + t_pd) * fast_rcp(t_np * 3.0);
        const double crit = t_cr / (double)M;
        const int it = itn + 1;
        st->it = it; st->crit = crit; st->Np = t_np;
        const bool finite_ok = (new_sigma2 == new_sigma2) && (fabs(new_sigma2) < 1e300) && (new_sigma2 > 0) && (crit == crit) && !bad_pivot;
        st->sigma2 = new_sigma2;
        if (finite_ok) {        // set_iter_consts with the sigma2-independent factor formed up front
            const double tp = 2.0 * M_PI * new_sigma2, rtp = ::sqrt(tp);
            st->k2 = -1.4426950408889634 * 0.5 * fast_rcp(new_sigma2);
            st->c_norm = tp * rtp * kc;
            st->rwin32 = f.win_e32 * 1.3862943611198906 * new_sigma2; st->rwin64 = f.win_e64 * 1.3862943611198906 * new_sigma2;      // the E-step's node window (set_iter_consts)
        } else { st->status = TDLO_E_NUMERIC; st->done = 1; st->converged = 0; pub = 1; }
        if (crit < f.tol) { st->done = 1; pub = 1; }                                   // :424-428
        else if (it >= f.max_iter) { st->converged = 0; st->done = 1; pub = 1; }      // :433-437
        if (it == f.host_report_it) pub = 1;      // the host looks at the progress word after this iteration (the last one of an early-exit polling chunk)
    }
    if (!XCH && t < 64 && __builtin_amdgcn_readfirstlane(pub)) host_publish(f, st, lane, true);      // progress (and, from the M-step that finishes the registration, the results) into pinned host memory
#undef BSTAMP
}

template <typename K> static hipError_t set_lds_b(K kernel, size_t bytes) {
    if (bytes > 64 * 1024) return hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    return hipSuccess;
}

template <typename T> static hipError_t launch_mstep_band_T(const FrameDev *fd, const FrameDev *fh, int F, int from_sums, hipStream_t s) {
    const size_t lds = mstep_band_lds_bytes(fh[0].M);
    hipError_t e;
    if (from_sums == 3) {          // one frame (a shard of the split cloud), exchange inside the kernel
        if (F != 1) return hipErrorInvalidValue;
        if ((e = set_lds_b(k_mstep_band<T, true, true>, lds)) != hipSuccess) return e;
        hipLaunchKernelGGL((k_mstep_band<T, true, true>), dim3(1), dim3(kBB), lds, s, fd, fh[0], from_sums);
    } else if (F == 1) {
        if ((e = set_lds_b(k_mstep_band<T, true, false>, lds)) != hipSuccess) return e;
        if (g_mstep_ev[0]) hipExtLaunchKernelGGL((k_mstep_band<T, true, false>), dim3(1), dim3(kBB), lds, s, g_mstep_ev[0], g_mstep_ev[1], 0, fd, fh[0], from_sums);
        else hipLaunchKernelGGL((k_mstep_band<T, true, false>), dim3(1), dim3(kBB), lds, s, fd, fh[0], from_sums);
    } else {
        if ((e = set_lds_b(k_mstep_band<T, false, false>, lds)) != hipSuccess) return e;
        if (g_mstep_ev[0]) hipExtLaunchKernelGGL((k_mstep_band<T, false, false>), dim3(F), dim3(kBB), lds, s, g_mstep_ev[0], g_mstep_ev[1], 0, fd, fh[0], from_sums);
        else hipLaunchKernelGGL((k_mstep_band<T, false, false>), dim3(F), dim3(kBB), lds, s, fd, fh[0], from_sums);
    }
    return hipGetLastError();
}

hipError_t launch_mstep_band(const FrameDev *fd, const FrameDev *fh, int F, int from_sums, bool f64, hipStream_t s) {
    return f64 ? launch_mstep_band_T<double>(fd, fh, F, from_sums, s) : launch_mstep_band_T<float>(fd, fh, F, from_sums, s);
}

// Which M-step serves registrations WITH the LLE term: 0 the banded L D L^T (default, where prepare_frame finds the chain and H
// suitable), 1 the dense pivoted eliminations, kept as comparators.  Read when a frame is prepared (FrameDev::lle_band); initial
// value from TDLO_MSTEP_LLE=dense.
static std::atomic<int> g_lle_dense{[] { const char *e = getenv("TDLO_MSTEP_LLE"); return (e && (strstr(e, "dense") || strstr(e, "1wg"))) ? 1 : 0; }()};    // ("1wg": the one-workgroup dense kernel, a comparator of the multi-workgroup one)
bool mstep_band_enabled() { return g_lle_dense.load(std::memory_order_relaxed) == 0; }
int mstep_set_lle_dense(int on) { return g_lle_dense.exchange(on ? 1 : 0); }

}  // namespace tdlo
