// tdlo_mstep_chain_body.h -- the chain smoother of tdlo_mstep_chain.hip (trackdlo.cpp:392-437 without the LLE term) as a DEVICE FUNCTION, so that the kernel
// k_mstep_chain and the batches' persistent loop (tdlo_estep2.hip, k_batch_loop: the workgroup that finishes a frame's E-step runs the frame's M-step) share one body.
// The derivation, the phases and what bounds it: the head of tdlo_mstep_chain.hip.
#pragma once
#include <type_traits>
#include "tdlo_devcommon.h"
#include "tdlo_lle_dev.h"
#include <atomic>
#include <cstdlib>
#include <hip/hip_ext.h>

namespace tdlo {
namespace {

constexpr int kCB = 256;               // workgroup size
typedef double dbl2 __attribute__((ext_vector_type(2)));

// Four wave-wide sums at once by a halving butterfly: v_permlane32_swap puts the lower halves of two values side by side and their upper
// halves side by side (one add folds lanes 32 apart of BOTH values), v_permlane16_swap does the same for rows 16 apart, four DPP
// rotations finish inside the rows.  Every lane of row 0 returns sum(a), row 1 sum(c), row 2 sum(b), row 3 sum(d).
__device__ __forceinline__ double swap_add32(double x, double y) {      // lanes 0..31: x[l] + x[l + 32], lanes 32..63: y[l - 32] + y[l]
    const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(x), (unsigned)__double2loint(y), false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(x), (unsigned)__double2hiint(y), false, false);
    return __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
}
__device__ __forceinline__ double swap_add16(double x, double y) {      // rows 0, 2: x's rows (0, 1), (2, 3) folded; rows 1, 3: y's
    const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(x), (unsigned)__double2loint(y), false, false);
    const auto hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(x), (unsigned)__double2hiint(y), false, false);
    return __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
}
template <int CTRL> __device__ __forceinline__ double dpp_f64(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false), hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum4(double a, double b, double c, double d) {
    double z = swap_add16(swap_add32(a, b), swap_add32(c, d));
    z += dpp_f64<0x128>(z);      // row_ror:8
    z += dpp_f64<0x124>(z);      // row_ror:4
    z += dpp_f64<0x122>(z);      // row_ror:2
    z += dpp_f64<0x121>(z);      // row_ror:1
    return z;
}

// the value of lane L of the own row of 16 lanes (DPP row_newbcast, two 32-bit halves)
template <int L> __device__ __forceinline__ double row_bcast_f64(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x150 + L, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x150 + L, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

}  // namespace

// Step slots.  All directions have nQ steps, the shorter ones start with dummy steps (identity link, no observation), so that every
// direction reaches its junction in its LAST step and the loops need no per-direction bookkeeping.  Slot sl = dir * nQ + k.  One slot =
// kSlot doubles of LDS:
//   [ 0.. 7] record of the step  {f11 f12 | f21 f22 | q11/c q12/c | q22/c p}   (Phi and Q of the link INTO the step, observation precision)
//   [ 8..13] right-hand side     {bx - | by - | bz -}
//   [14..19] mean                {mx0 mx1 | my0 my1 | mz0 mz1}   filtered -> e_k -> smoothed
//   [20..23] posterior           {a b | d g}           g = 1 / (1 + p P^-_11)
//   [24..27] smoother gain       {C11 C12 | C21 C22}
//   [28..31] spike means         {F00 F10 | F01 F11}   inner directions: the mean's dependence on the direction's start state
constexpr int kSlot = 34;             // (272 bytes: consecutive slots start 4 banks apart, so a 128-bit access of 16 lanes -- thread = slot -- covers the
                                      //  64 banks exactly once; a stride of 256 bytes would put every lane on the same four banks)
constexpr int kAhead = 6;             // slots behind the last one that the software-pipelined loops may read
constexpr int kBehind = 9;            // slots in front of the first one that the backward walks may read (values unused)
constexpr int kDump = 32;              // doubles of the dump area (lanes that have nothing to store write there; never read for a result)
constexpr int kRed = 112;              // [0..15] wave sums, [24] exchange flag, [26, 27] zeros, [28] progress counter,
                                       // [32..71] likelihood sums of the four directions (5 columns x 2), [72..77] junction state x2,
                                       // [80..93], [96..109] junction matrices of the left / right half
constexpr int kND = 4;                 // directions
constexpr int kDirectMax = 21;         // up to this many steps per direction the backward pass is walked step by step (below)

// Four directions, nQ steps each (leading dummy steps -- identity link, nothing observed -- make all four end in their last step):
//   0: nodes 0 .. j1 of the process, from the stationary prior             2: nodes j2 .. j3 of the process, from the state AT j2
//   1: nodes j2-1 .. j1 of the reversed process, from the state AT j2      3: nodes M-1 .. j3 of the reversed process, from the prior
// j2 = the middle node, j1 and j3 the quarter points.  The junction nodes' data belong to directions 0 (j1), 2 (j2) and 3 (j3).
struct ChainCarve {
    int nSp, j1, j2, j3, n0, n1, n2, n3, nQ, nSl;      // (scalars, no array: an indexed member keeps the whole object in scratch memory)
    size_t S, red, dump, slots, total;
    __host__ __device__ explicit ChainCarve(int M) {
        nSp = 4 * M + 2;
        j2 = (M - 1) >> 1; j1 = j2 >> 1; j3 = (j2 + M) >> 1;
        n0 = j1 + 1; n1 = j2 - j1; n2 = j3 - j2 + 1; n3 = M - j3;
        const int a = n0 > n1 ? n0 : n1, b = n2 > n3 ? n2 : n3;
        nQ = a > b ? a : b;
        nSl = kND * nQ;
        size_t o = 0;
        S = o; o += (size_t)((nSp + 1) & ~1);           // [P1 | Rx | Ry | Rz | Q]
        red = o; o += kRed;
        dump = o; o += kDump;
        o += (size_t)kSlot * kBehind;
        slots = o; o += (size_t)kSlot * (nSl + kAhead);  // the loops read up to kAhead slots ahead
        total = o;
    }
};

// TRK: the extras of tracking_step's main registration (one frame, no exchange) -- late priors read from pinned host memory when no E-step has
// run yet, the launch ahead of its priors (FrameDev::spec_flag), the next frame's LLE regulariser at the end (FrameDev::lle_next).  The plain
// instantiation is the kernel of the registrations proper, unchanged.
// SPIN (round 6 experiment, FrameDev::spin_on): the launch was dispatched behind the M-step of the iteration before while THIS iteration's E-step still
// runs on another stream -- everything but the sums is requested, then the kernel waits for the E-step's workgroups to have counted themselves in.
// ROWS: how many replica rows of the accumulators the E-step in front used (FrameDev::acc_rows; the launcher instantiates 2 / 4 for the plain one-frame kernel)
// HINT: the launch carries the iteration's parity (par_hint, 0 / 1): the sums are requested from that parity's rows alone, without waiting for the device's counter
template <typename T, bool SINGLE, bool XCH, bool TRK = false, bool SPIN = false, int ROWS = kAccRows, bool HINT = false>
__device__ __forceinline__ void mstep_chain_run(const FrameDev &f, int from_sums, char *smem, int par_hint = 0) {
    constexpr int MB = kCB;
    // One wave walks a chain of dependent instructions.  In a batch the other stream groups' E-steps fill the same SIMDs with waves that always have
    // something to issue: at the default priority this wave takes its turn among them (C3: 10.0 us per M-step against 7.4 us with the GPU to itself)
    IterState *st = f.st;
    const int M = f.M, t = threadIdx.x, lane = t & 63;
    const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
    const int nS = 4 * M + 1;
    const ChainCarve cv(M);
    const int nQ = cv.nQ, nSl = cv.nSl, j1 = cv.j1, j2 = cv.j2, j3 = cv.j3;
    // leading dummy steps per direction (0 .. 3 of them), one nibble each: shifts instead of a select chain over four values,
    // which the compiler turns into a table in scratch memory
    const int padbits = (nQ - cv.n0) | ((nQ - cv.n1) << 4) | ((nQ - cv.n2) << 8) | ((nQ - cv.n3) << 12);
    double *S = (double *)smem + cv.S, *red = (double *)smem + cv.red, *dump = (double *)smem + cv.dump, *slots = (double *)smem + cv.slots;

    // phase stamps (scripts/gpu_stamps.py, tdlo_debug_stamps) only in a -DTDLO_CHAIN_STAMPS build (scripts/build_variant.sh stamps ...): each one is an
    // s_memtime behind a full lgkmcnt wait plus a store behind an exec branch, eight of them per launch
#ifdef TDLO_CHAIN_STAMPS
#define CSTAMP(i) do { if (t == 0) f.dbg[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define CSTAMP(i) do { } while (0)
#endif
    CSTAMP(0);
    if (SINGLE) {
        // (the descriptor is the kernel's argument block: ALL the pointers this kernel starts from are asked for here, in one scalar round trip with the few
        //  the sums' addresses need -- left to the compiler they are fetched in three batches, each waited for where it is first used, and the slot's
        //  requests below leave 600 clocks after the sums')
        const void *p0 = f.st, *p1 = f.chain, *p2 = f.nodes, *p3 = f.Y, *p4 = f.Y0, *p5 = f.aJ, *p6 = f.aYd;
        asm volatile("" :: "s"(p0), "s"(p1), "s"(p2), "s"(p3), "s"(p4), "s"(p5), "s"(p6));
    }
    const auto stg = TDLO_AS_GLOBAL(IterState, st);
    const int done = stg->done;
    const double sigma2 = stg->sigma2;
    const int pri = f.has_priors;
    const double ctr0 = f.ctr[0], ctr1 = f.ctr[1], ctr2 = f.ctr[2];
    const auto ndg = TDLO_AS_GLOBAL(V4<T>, f.nodes);
    const auto Yg = TDLO_AS_GLOBAL(double, f.Y);
    const auto Y0g = TDLO_AS_GLOBAL(double, f.Y0);
    // (late priors of a registration that starts from given sums: no E-step has copied them yet -- read from pinned host memory here, kept below)
    // (... or whose own first E-step ran before the priors existed: FrameDev::late_mstep)
    const bool late_src = TRK && f.late_aJ != nullptr && (from_sums == 1 || f.late_mstep != 0);
    const auto aJg = TDLO_AS_GLOBAL(double, late_src ? f.late_aJ : f.aJ);
    const auto aYg = TDLO_AS_GLOBAL(double, late_src ? f.late_aYd : f.aYd);
    const auto chg = TDLO_AS_GLOBAL(dbl2, f.chain);

    // slot -> (node, link into the step, does the step observe its node); li == 0: identity (a direction's first step from the prior or
    // at j2, dummy steps).  A junction node is observed (and written back) by one direction only.
    auto slot_dir = [&](int sl) __attribute__((always_inline)) { return (int)(sl >= nQ) + (int)(sl >= 2 * nQ) + (int)(sl >= 3 * nQ); };     // (no branches)
    auto slot_info = [&](int sl, int &node, int &li, bool &obs) __attribute__((always_inline)) {
        const int dir = slot_dir(sl), k = sl - dir * nQ;
        const int kk = k - ((padbits >> (4 * dir)) & 15);
        const bool real = kk >= 0;
        int nd, l; bool ob;
        if (dir == 0)      { nd = kk;         l = kk > 0 ? kk : 0;     ob = true; }
        else if (dir == 1) { nd = j2 - 1 - kk; l = nd + 1;             ob = nd != j1; }
        else if (dir == 2) { nd = j2 + kk;    l = kk > 0 ? nd : 0;     ob = nd != j3 && !(kk == 0 && j1 == j2); }
        else               { nd = M - 1 - kk; l = kk > 0 ? nd + 1 : 0; ob = true; }
        node = real ? nd : 0; li = real ? l : 0; obs = real && ob;
    };
    // ---- 1. everything that comes from memory is requested up front: this thread's step slot (its link, its node), the E-step's sums
    struct SlotQ { dbl2 l[4]; double y[3], y0[3], yp[3], ay[3], aj, w; int node, li; bool obs; };
    // (an M-step launched ahead of its priors requests everything BUT the priors before it waits for them: with_pri == false, filled in behind the wait)
    auto load_slot = [&](int sl, bool with_pri) __attribute__((always_inline)) {
        SlotQ q;
        slot_info(sl < nSl ? sl : 0, q.node, q.li, q.obs);
        const int lc = q.li > 0 ? q.li : 1, mc = q.node;
#pragma unroll
        for (int i = 0; i < 4; ++i) q.l[i] = chg[4 * (size_t)lc + i];
        // (li == 0, the identity link, is substituted where the record is written: a conditional overwrite here is a branch that waits
        //  for every load in flight -- the sums included -- before the node's loads below are requested)
        q.y[0] = (double)ndg[mc].x; q.y[1] = (double)ndg[mc].y; q.y[2] = (double)ndg[mc].z; q.w = (double)ndg[mc].w;
#pragma unroll
        for (int d = 0; d < 3; ++d) { q.y0[d] = Y0g[d * M + mc]; q.yp[d] = Yg[d * M + mc]; q.ay[d] = (pri && with_pri) ? aYg[d * M + mc] : 0.0; }
        q.aj = (pri && with_pri) ? aJg[mc] : 0.0;
        return q;
    };
    // the E-step's sums: kAccRows replica rows of fixed-point accumulators; both iteration parities are fetched so that no load waits
    // for the iteration counter (M <= 512: at most 9 elements per thread).  Requested before the slot: its index arithmetic
    // runs while these are in flight.
    const bool spec_wait = TRK && f.spec_flag != nullptr;      // launched ahead of its priors (FrameDev::spec_flag): the wait sits behind the requests below
    const int itn = stg->it;
    double sq[9];
    SlotQ q0;
    bool spin_lost = false;
    if (SPIN) {
        // (up to 63 nodes: one element per thread.)  The slot -- links, the node, Y0, Y -- is requested first: nothing of it comes from this iteration's
        // E-step; then the wait for that E-step's workgroups, then the sums
        q0 = load_slot(t, true);
        if (t == 0) red[29] = spin_wait_word(f.sync + kSpinWordE, f.spin_wait) ? 1.0 : 0.0;
        __syncthreads();
        spin_lost = red[29] == 0.0;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        sq[0] = acc_read_both<ROWS>(f, t < nS ? t : nS - 1, itn);
#pragma unroll
        for (int u = 1; u < 9; ++u) sq[u] = 0.0;
    } else {
    // (the first element without a branch -- index clamped, the accumulators exist in every mode: inside a conditional block the compiler sums the
    //  16 rows on the spot, i.e. waits for them BEFORE it requests the slot below: two memory round trips in a row instead of one)
    sq[0] = HINT ? acc_read_par<ROWS>(f, t < nS ? t : nS - 1, par_hint) : acc_read_both<ROWS>(f, t < nS ? t : nS - 1, itn);
#pragma unroll
    for (int u = 1; u < 9; ++u) sq[u] = 0.0;
    }
    // spin-ahead: this M-step's tag goes out when it is through, whatever way it leaves (the next E-step is parked on it)
    auto spin_report = [&]() __attribute__((always_inline)) {
        if (!SPIN) return;
        __syncthreads();                                           // (every thread's stores to the nodes and to Y have been performed: workgroup-scope release)
        if (t == 0) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); __hip_atomic_store(f.sync + kSpinWordM, f.spin_signal, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); }
    };
    if (SPIN && spin_lost) {      // the E-step never completed (2 s): the registration ends with an error, the tag still goes out
        if (t == 0) { st->status = TDLO_E_EXCHANGE; st->done = 1; st->converged = 0; }
        spin_report();
        if (t < 64) host_publish(f, st, lane, false);
        return;
    }
    if (SPIN) {
    } else if (nS <= MB) {             // up to 63 nodes: the slot's loads follow the sums' in the same basic block (nothing is waited for in between)
        q0 = load_slot(t, !spec_wait);
    } else {                    // longer chains: the further elements first (with the slot's forty registers live the compiler requests their
        // rows one by one, a round trip each).  Straight-line code per element count -- indices clamped instead of branched, this iteration's
        // parity only: inside `if (i < nS)` blocks every element's eight rows were waited for before the next element's were requested
        // (fetch at M = 300: 10 500 clocks, 7 000 of them these serial round trips).
        if (from_sums != 1) {
            const auto rows = TDLO_AS_GLOBAL(long long, f.acc) + (size_t)(HINT ? par_hint : (itn & 1)) * kAccRows * acc_stride(M);
            const int stride = acc_stride(M);
            auto more = [&](auto U0c, auto U1c) __attribute__((always_inline)) {        // elements U0 .. U1 - 1, all rows requested before any is summed
                constexpr int U0 = decltype(U0c)::value, U1 = decltype(U1c)::value;
                long long sa[U1 - U0];
                int ix[U1 - U0];
#pragma unroll
                for (int u = U0; u < U1; ++u) {
                    const int i = t + u * MB;
                    ix[u - U0] = i < nS ? i : nS - 1;
                    long long a = 0;
#pragma unroll
                    for (int r = 0; r < ROWS; ++r) a += rows[(size_t)r * stride + ix[u - U0]];
                    sa[u - U0] = a;
                }
#pragma unroll
                for (int u = U0; u < U1; ++u) sq[u] = ::ldexp((double)sa[u - U0], -acc_shift(f, ix[u - U0]));
            };
            using std::integral_constant;
            switch ((nS + MB - 1) / MB) {       // (two elements, 64 .. 127 nodes: both parities without waiting for the counter, as before; at most four
                case 2: if (t + MB < nS) sq[1] = HINT ? acc_read_par<ROWS>(f, t + MB, par_hint) : acc_read_both<ROWS>(f, t + MB, itn); break;                          //  elements at a time)
                case 3: more(integral_constant<int, 1>(), integral_constant<int, 3>()); break;
                case 4: more(integral_constant<int, 1>(), integral_constant<int, 4>()); break;
                case 5: more(integral_constant<int, 1>(), integral_constant<int, 5>()); break;
                default:        // more than 319 nodes: element by element as before (two groups of four in flight measured slower: 32 000 against 16 500 clocks at M = 512)
#pragma unroll
                    for (int u = 1; u < 9; ++u) { const int i = t + u * MB; if (i < nS) sq[u] = HINT ? acc_read_par<ROWS>(f, i, par_hint) : acc_read_both<ROWS>(f, i, itn); }
                    break;
            }
        }
        q0 = load_slot(t, !spec_wait);
    }
    // given sums (from_sums == 1: the N-split's reduced sums; tracking_step's paired registration): the first five elements per thread -- chains
    // of up to 256 nodes -- requested with everything else
    double ss[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
    if (TRK && from_sums == 1) {
        const auto sums = TDLO_AS_GLOBAL(double, f.sums);
#pragma unroll
        for (int u = 0; u < 5; ++u) { const int i = t + u * MB; ss[u] = sums[i < nS ? i : nS - 1]; }
    }
    if (spec_wait) {      // everything above is on its way (it does not depend on the priors): now the wait for the host's word
        if (f.spec_prev != nullptr) {             // (nullptr: launched on a stream of its own beside that registration -- the host's word alone decides)
            const auto pv = TDLO_AS_GLOBAL(IterState, f.spec_prev);
            if (!(pv->done != 0 && pv->status == 0)) return;
        }
        if (t == 0) {
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();       // 100 MHz
            int go = 0;
            for (;;) {
                const unsigned long long v = __hip_atomic_load(f.spec_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                if ((unsigned)(v >> 32) == f.spec_epoch && (v & 3ull) != 0ull) {
                    go = (v & 3ull) == 1ull;
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");      // the priors staged before the word was released (system scope, as xch_wait does)
                    break;
                }
                if (__builtin_amdgcn_s_memrealtime() - t0 > 200000000ull) break;
                __builtin_amdgcn_s_sleep(4);
            }
            red[31] = go ? 1.0 : 0.0;
        }
        __syncthreads();
        const bool go = red[31] != 0.0;
        __syncthreads();
        if (!go) {
            // sent away with this registration's own first E-step already behind it (FrameDev::late_mstep): the sums nobody takes are cleared, so
            // that the registration can start over the ordinary way (its first E-step adds to parity 0 again)
            if (f.late_mstep != 0 && from_sums == 0) acc_clear_other<MB>(f, 1, t);
            return;
        }
    }
    if (spec_wait && pri) {       // ... and this thread's slot takes its priors (staged by the host before it released the word)
#pragma unroll
        for (int d = 0; d < 3; ++d) q0.ay[d] = aYg[d * M + q0.node];
        q0.aj = aJg[q0.node];
    }
    double kq[4] = {0.0, 0.0, 0.0, 0.0};              // late priors, element t + u MB of [alpha J | alpha (Y_ext - Y0)]: likewise
    if (late_src && pri) {
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int i = t + u * MB, ic = i < 4 * M ? i : 4 * M - 1; kq[u] = ic < M ? aJg[ic] : aYg[ic - M]; }
    }
    const double pinf0 = chg[0].x, pinf1 = chg[0].y;  // sf2, s^2 sf2
    const double c2 = f.lambda * sigma2, rc2 = fast_rcp(c2);
    const double cp0 = c2 * chg[1].x, cp1 = c2 * chg[1].y;  // Pinf^-1 in the units of the filter (P = covariance / c); reciprocals from k_setup
    // what the kernel's last thread needs of set_iter_consts, formed while the loads are in flight: c of :300 / c' of :378 is
    // (2 pi sigma2)^(3/2) times this factor
    const double Nc = stg->Nc;
    const double kc = f.mu / (1.0 - f.mu) * (f.vis_branch ? 1.0 / Nc : (double)M / Nc);
#ifdef TDLO_TIMELINE      // wall-clock (100 MHz) begin / end of iterations 20..27, scripts/gpu_timeline.py
    if (t == 0 && itn >= 20 && itn < 28) f.dbg[4 * (itn - 20) + 2] = __builtin_amdgcn_s_memrealtime();
#endif
    if (done) {
        if (XCH && from_sums == 3) xch_post_error(f, st, t);
        spin_report();
        if (!XCH && t < 64 && stg->status != 0) host_publish(f, st, lane, false);      // a registration that ended on an error somewhere else (E-step, setup)
        return;
    }
    CSTAMP(1);
    if (HINT && from_sums != 1 && (itn & 1) != par_hint) {
        // the host's count of the iterations it has enqueued and the device's counter disagree (a registration continued by a caller that did not say so):
        // the sums are read again, the ordinary way.  Never seen in the tests' routes; kept so that the hint can only cost time, never a result.
#pragma unroll
        for (int u = 0; u < 9; ++u) { const int i = t + u * MB; if (i < nS) sq[u] = acc_read_both<ROWS>(f, i, itn); }
    }
    if (from_sums != 1) {
#pragma unroll
        for (int u = 0; u < 9; ++u) { const int i = t + u * MB; if (i < nS) S[i] = sq[u]; }
    } else {
        const auto sums = TDLO_AS_GLOBAL(double, f.sums);
        if (TRK) {
#pragma unroll
            for (int u = 0; u < 5; ++u) { const int i = t + u * MB; if (i < nS) S[i] = ss[u]; }
            for (int i = t + 5 * MB; i < nS; i += MB) S[i] = sums[i];
        } else {
            for (int i = t; i < nS; i += MB) S[i] = sums[i];
        }
    }
    if (late_src && pri) {      // the late priors to their place in device memory for the iterations that follow
        double *aJw = (double *)f.aJ, *aYw = (double *)f.aYd;
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int i = t + u * MB; if (i < M) aJw[i] = kq[u]; else if (i < 4 * M) aYw[i - M] = kq[u]; }
        for (int i = t + 4 * MB; i < 4 * M; i += MB) aYw[i - M] = aYg[i - M];
    }
    __syncthreads();
    if (from_sums == 2) {       // split mode, export only
        acc_clear_other<MB>(f, itn, t);
        for (int i = t; i < nS; i += MB) f.sums[i] = S[i];
        if (t == 0) f.sums[nS] = (double)stg->N;
        return;
    }
    if (XCH && from_sums == 3 && (f.xch_nranks > 1 || f.xch_self)) {      // (a lone rank: its own sums are the total)
        // N-split with the one-shot exchange (see k_mstep_fast): sums to every peer's inbox, flag, wait for the R flags in the
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
    CSTAMP(2);

    // ---- 2. thread = step slot: the step's record and right-hand side  B = PX - P1 Y0 (+ alpha (Y_ext - Y0)) = R + P1 (y - Y0) (+ ...)
    //         (the E-step delivers R = PX - P1 y, y = the nodes as it saw them)
    for (int sl = t, r = 0; sl < nSl; sl += MB, ++r) {
        dbl2 *o = (dbl2 *)(slots + (size_t)kSlot * sl);
        const SlotQ q = r == 0 ? q0 : load_slot(sl, true);
        const double p1 = q.obs ? S[q.node] : 0.0;
        const bool idl = q.li == 0;     // identity link: a direction's first step, dummy steps
        o[0] = dbl2{idl ? 1.0 : q.l[0].x, idl ? 0.0 : q.l[0].y}; o[1] = dbl2{idl ? 0.0 : q.l[1].x, idl ? 1.0 : q.l[1].y};
        o[2] = dbl2{idl ? 0.0 : q.l[2].x * rc2, idl ? 0.0 : q.l[2].y * rc2};
        o[3] = dbl2{idl ? 0.0 : q.l[3].x * rc2, q.obs ? p1 + q.aj : 0.0};
#pragma unroll
        for (int d = 0; d < 3; ++d) o[4 + d] = dbl2{q.obs ? S[(1 + d) * M + q.node] + (p1 * (q.y[d] - q.y0[d]) + q.ay[d]) : 0.0, 0.0};
    }
    if (t == 0) { *(int *)(red + 28) = 0; red[26] = 0.0; red[27] = 0.0; }      // progress counter of the covariance pass; the spike columns' right-hand side
    if (t >= MB - kAhead) {     // the slots the loops read ahead into: identity, nothing observed
        dbl2 *o = (dbl2 *)(slots + (size_t)kSlot * (nSl + (t - (MB - kAhead))));
        o[0] = dbl2{1.0, 0.0}; o[1] = dbl2{0.0, 1.0}; o[2] = dbl2{0.0, 0.0}; o[3] = dbl2{0.0, 0.0}; o[4] = dbl2{0.0, 0.0}; o[5] = dbl2{0.0, 0.0}; o[6] = dbl2{0.0, 0.0};
    }
    __syncthreads();
    CSTAMP(3);

    // ---- 3. forward pass, two waves in a pipeline.  A lone wave issues one instruction per ~8 cycles whatever the instruction is
    //         (scripts/ubench/lat.hip), so the pass is bound by its instruction count: wave 0 runs the covariance recursion (which
    //         does not depend on the data), wave 1 follows with the means as the posteriors appear.  Both: 16 lanes per direction.
    //         Hand-over through LDS: wave 0 stores a step's posterior and then the number of finished steps --
    //         a wave's LDS operations execute in order, the reader loads the counter before the posterior; the inline asm keeps the
    //         compiler from reordering across the two.
    constexpr int SB = kSlot * 8;
    const unsigned prog_addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void *)(red + 28);
    if (wv == 0) {
        const int dir = lane >> 4, hl = lane & 15;
        const bool inner = dir == 1 || dir == 2;
        char *const sb = (char *)slots, *const db = (char *)dump;
        char *ra = sb + (size_t)SB * dir * nQ;          // the direction's current slot (the same in every lane of a quarter)
        char *qa = hl == 0 ? ra : db;                   // the posterior's cells for lane 0 of the quarter, else the dump area
        const int qstep = hl == 0 ? SB : 0;
        double a = inner ? 0.0 : pinf0 * rc2, b = 0.0, d = inner ? 0.0 : pinf1 * rc2;      // an inner direction starts from a known state
        struct Rec { dbl2 r0, r1, r2, r3; };
        auto fetch = [&](int ahead) __attribute__((always_inline)) {       // record of the step `ahead` slots further on
            Rec r;
            const char *rp = ra + SB * ahead;
            r.r0 = *(const dbl2 *)(rp); r.r1 = *(const dbl2 *)(rp + 16); r.r2 = *(const dbl2 *)(rp + 32); r.r3 = *(const dbl2 *)(rp + 48);
            return r;
        };
        int kdone = 0;
        auto step = [&](const Rec &r, int at) __attribute__((always_inline)) {
            const double f11 = r.r0.x, f12 = r.r0.y, f21 = r.r1.x, f22 = r.r1.y, q11 = r.r2.x, q12 = r.r2.y, q22 = r.r3.x, p = r.r3.y;
            // predict: P^- = Phi P Phi^T + Q / c
            const double t1 = fma(f12, b, f11 * a), t2 = fma(f12, d, f11 * b);
            const double t3 = fma(f22, b, f21 * a), t4 = fma(f22, d, f21 * b);
            const double pa = fma(t2, f12, fma(t1, f11, q11));
            const double pb = fma(t2, f22, fma(t1, f21, q12));
            const double pd = fma(t4, f22, fma(t3, f21, q22));
            // update with observation precision p: P = P^- - P^- e1 e1^T P^- p / (1 + p P^-_11); the gain of the mean is (a, b) of the posterior
            const double g = fast_rcp(fma(p, pa, 1.0));
            const double npb = -(p * pb);
            a = pa * g; b = pb * g; d = fma(npb, b, pd);
            *(dbl2 *)(qa + qstep * at + 160) = dbl2{a, b};
            *(dbl2 *)(qa + qstep * at + 176) = dbl2{d, g};       // (g for the likelihood sums of the means wave: 1 - p a would cancel)
            ++kdone;
            asm volatile("ds_write_b32 %0, %1" :: "v"(prog_addr), "v"(kdone) : "memory");
        };
        // four steps per trip on four register sets: every record is requested two steps before its use, nothing is copied; the
        // slots behind a direction's last one are readable (look-ahead slots / the other direction)
        int k = 0;
        if (nQ >= 4) {
            Rec rA = fetch(0), rB = fetch(1);
            for (; k + 3 < nQ; k += 4) {
                const Rec rC = fetch(2), rD = fetch(3);
                step(rA, 0); step(rB, 1);
                rA = fetch(4); rB = fetch(5);
                step(rC, 2); step(rD, 3);
                ra += 4 * SB; qa += 4 * qstep;
            }
        }
        for (; k < nQ; ++k) {
            const Rec r = fetch(0);
            step(r, 0);
            ra += SB; qa += qstep;
        }
    } else if (wv == 1) {
        // lane = (direction, column): columns 0..2 the coordinates, 3 and 4 the SPIKE columns -- the mean of an inner direction is
        // affine in its unknown start state x, m_k = g_k + F_k x: the coordinate columns carry g (start 0), the spike columns F (start
        // I, nothing observed: right-hand side 0).  The same lanes sum the likelihood of the direction's data as a function of x,
        //   -1/2 x^T J x + x^T eta,  J = sum_k p g_k h_k h_k^T,  eta = sum_k g_k innov_k h_k,   h_k = row 0 of Phi F_{k-1}, g_k = 1 / (1 + p P^-_11):
        // with w = +-g_k innov (innov of a spike column is -p h) every lane adds w h_0 and w h_1 -- columns 0..2 end with eta, 3 and 4 with J.
        const int dir = lane >> 4, hl = lane & 15;
        const bool wr = hl < 5;
        char *const sb = (char *)slots, *const db = (char *)dump;
        char *ra = sb + (size_t)SB * dir * nQ;
        char *la = wr ? (hl < 3 ? ra + 16 * hl : ra + 112 + 16 * (hl - 3)) : db;    // the lane's mean cell is la + 112: [14 + 2 hl], spike columns [28 + 2 (hl - 3)]
        const int lstep = wr ? SB : 0;
        const char *ba = hl < 3 ? ra + 64 + 16 * hl : (const char *)(red + 26);     // right-hand side: the slot's, zero for the spike columns
        const int bstep = hl < 3 ? SB : 0;
        double m0 = hl == 3 ? 1.0 : 0.0, m1 = hl == 4 ? 1.0 : 0.0;
        const double sgl = hl < 3 ? 1.0 : -1.0;
        double acc0 = 0.0, acc1 = 0.0;
        int have = 0;                                   // steps whose posterior is known to be in LDS
        auto wait_for = [&](int need) __attribute__((always_inline)) {
            while (have < need) {
                int v;
                asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(prog_addr) : "memory");
                have = __builtin_amdgcn_readfirstlane(v);
                if (have < need) __builtin_amdgcn_s_sleep(1);
            }
        };
        struct MRec { dbl2 r0, r1, ab; double p, bb, g; };
        auto mfetch = [&](int ahead) __attribute__((always_inline)) {
            MRec r;
            const char *rp = ra + SB * ahead;
            r.r0 = *(const dbl2 *)(rp); r.r1 = *(const dbl2 *)(rp + 16); r.p = *(const double *)(rp + 56); r.ab = *(const dbl2 *)(rp + 160);
            r.bb = *(const double *)(ba + bstep * ahead);
            r.g = *(const double *)(rp + 184);
            return r;
        };
        auto mstep = [&](const MRec &r, int at) __attribute__((always_inline)) {
            // m^- = Phi m;  m = m^- + K (b - p m^-_0),  K = (a, b) of the step's posterior
            const double pm0 = fma(r.r0.y, m1, r.r0.x * m0), pm1 = fma(r.r1.y, m1, r.r1.x * m0);
            const double innov = fma(-r.p, pm0, r.bb);
            const double w = (sgl * r.g) * innov;
            const double h0 = row_bcast_f64<3>(pm0), h1 = row_bcast_f64<4>(pm0);
            acc0 = fma(w, h0, acc0); acc1 = fma(w, h1, acc1);
            m0 = fma(r.ab.x, innov, pm0); m1 = fma(r.ab.y, innov, pm1);
            *(dbl2 *)(la + lstep * at + 112) = dbl2{m0, m1};
        };
        int k = 0;
        for (; k + 3 < nQ; k += 2) {                    // in pairs while the covariance pass is far ahead ...
            wait_for(k + 2);
            const MRec rA = mfetch(0), rB = mfetch(1);
            mstep(rA, 0); mstep(rB, 1);
            ra += 2 * SB; la += 2 * lstep; ba += 2 * bstep;
        }
        for (; k < nQ; ++k) {                           // ... one by one at the end: the pass is over one step after the covariances
            wait_for(k + 1);
            const MRec r = mfetch(0);
            mstep(r, 0);
            ra += SB; la += lstep; ba += bstep;
        }
        if (wr) *(dbl2 *)(red + 32 + 2 * (dir * 5 + hl)) = dbl2{acc0, acc1};
    } else {
        // waves 2 and 3 have nothing to do in this phase: they clear the other parity's accumulator rows for the next E-step
        if (from_sums != 1) acc_clear_other<MB - 128>(f, itn, t - 128);
        if (wv == 2) {
            // ... and wave 2 prepares what the junction solve (4., below) needs of the covariances alone, while the means finish:
            // per half (lanes 0..31 left, 32..63 right) P_outer^-1, Lam, A = (I + Lam Pc)^-1, W = A Lam
            int have = 0;
            while (have < nQ) {
                int v;
                asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(prog_addr) : "memory");
                have = __builtin_amdgcn_readfirstlane(v);
                if (have < nQ) __builtin_amdgcn_s_sleep(4);
            }
            const int hf = lane >> 5, dO = hf ? 3 : 0, dI = hf ? 2 : 1;
            const double *sO = slots + (size_t)kSlot * (dO * nQ + nQ - 1), *sI = slots + (size_t)kSlot * (dI * nQ + nQ - 1);
            const dbl2 abO = *(const dbl2 *)(sO + 20), abI = *(const dbl2 *)(sI + 20);
            const double Oa = abO.x, Ob = -abO.y, Od = sO[22], pa = abI.x, pb = abI.y, pd = sI[22];      // outer posterior in the half's frame
            const double rO = fast_rcp(fma(Oa, Od, -(Ob * Ob)));
            const double ia = Od * rO, ib = -Ob * rO, id = Oa * rO;
            const double La = ia - cp0, Lb = ib, Ld = id - cp1;
            const double n00 = fma(La, pa, fma(Lb, pb, 1.0)), n01 = fma(La, pb, Lb * pd), n10 = fma(Lb, pa, Ld * pb), n11 = fma(Lb, pb, fma(Ld, pd, 1.0));
            const double rn = fast_rcp(fma(n00, n11, -(n01 * n10)));
            const double A00 = n11 * rn, A01 = -n01 * rn, A10 = -n10 * rn, A11 = n00 * rn;
            const double W00 = fma(A00, La, A01 * Lb), W01 = fma(A00, Lb, A01 * Ld), W11 = fma(A10, Lb, A11 * Ld);      // (W is symmetric)
            dbl2 *jo = (dbl2 *)((lane & 31) == 0 ? red + 80 + 16 * hf : dump);
            jo[0] = dbl2{ia, ib}; jo[1] = dbl2{id, La}; jo[2] = dbl2{Lb, Ld}; jo[3] = dbl2{A00, A01}; jo[4] = dbl2{A10, A11}; jo[5] = dbl2{W00, W01}; jo[6] = dbl2{W11, 0.0};
        }
    }
    __syncthreads();
    CSTAMP(4);
    // ---- 4. thread = step slot: smoother gain C_k = P_k Phi'^T (Phi' P_k Phi'^T + Q')^-1 with the link of step k + 1 (the next
    //         slot's record), and e_k = m_k - C_k Phi' m_k, so that the backward step is x_k = e_k + C_k x_{k+1}
    for (int sl = t; sl < nSl; sl += MB) {
        const int dir = slot_dir(sl), k = sl - dir * nQ;
        if (k < nQ - 1) {
            double *o = slots + (size_t)kSlot * sl;
            const dbl2 *np = (const dbl2 *)(o + kSlot);
            const double h11 = np[0].x, h12 = np[0].y, h21 = np[1].x, h22 = np[1].y, q11 = np[2].x, q12 = np[2].y, q22 = np[3].x;
            const dbl2 ab = *(const dbl2 *)(o + 20);
            const double a = ab.x, b = ab.y, d = o[22];
            const double t1 = fma(h12, b, h11 * a), t2 = fma(h12, d, h11 * b);       // (P Phi^T) column 1 = (t1, t2)
            const double t3 = fma(h22, b, h21 * a), t4 = fma(h22, d, h21 * b);       // (P Phi^T) column 2 = (t3, t4)
            const double pa = fma(t2, h12, fma(t1, h11, q11)), pb = fma(t2, h22, fma(t1, h21, q12)), pd = fma(t4, h22, fma(t3, h21, q22));
            const double det = fma(pa, pd, -(pb * pb));
            // (an inner direction starts from a known state: while its covariance is still exactly zero -- dummy steps, coincident
            // nodes -- the smoothed state IS the filtered mean, C = 0)
            const double rdet = det > 0.0 ? fast_rcp(det) : 0.0;
            const double ia = pd * rdet, ib = -pb * rdet, id = pa * rdet;
            const double C11 = fma(t1, ia, t3 * ib), C12 = fma(t1, ib, t3 * id), C21 = fma(t2, ia, t4 * ib), C22 = fma(t2, ib, t4 * id);
            const int nq = (dir == 1 || dir == 2) ? 5 : 3;      // the spike columns go through the same map: E_k = F_k - C_k Phi' F_k
#pragma unroll
            for (int q = 0; q < 5; ++q) {
                if (q < nq) {
                    dbl2 *mp = (dbl2 *)(o + (q < 3 ? 14 + 2 * q : 22 + 2 * q));
                    const dbl2 mm = *mp;
                    const double pm0 = fma(h12, mm.y, h11 * mm.x), pm1 = fma(h22, mm.y, h21 * mm.x);
                    *mp = dbl2{mm.x - fma(C11, pm0, C12 * pm1), mm.y - fma(C21, pm0, C22 * pm1)};
                }
            }
            *(dbl2 *)(o + 24) = dbl2{C11, C12};
            *(dbl2 *)(o + 26) = dbl2{C21, C22};
        }
    }
    // meanwhile wave 1 (its threads hold no slot up to 32 nodes) solves for the junction states.  Lanes 0..31: the left half (outer
    // direction 0 + inner direction 1, in the frame of the reversed process), lanes 32..63 the right half (3 + 2, frame of the process);
    // lane & 31 = coordinate.  Per half, with (m, P) the outer direction's posterior at its junction node brought into the half's
    // frame (f' changes sign), (F, g, Pc) the inner direction's last step (state there = F x + g + noise(Pc), x = state at j2):
    //   Lam = P^-1 - Pinf^-1, xi = P^-1 m            the outer data as a likelihood of the junction state (prior counted once)
    //   A = (I + Lam Pc)^-1,  W = A Lam,  w = A xi   ... seen through the inner direction's noise
    //   J = F^T W F + J_in,  eta = F^T (w - W g) + eta_in      the half's message to x (J_in, eta_in: the inner direction's own data)
    // x = (Pinf^-1 + J_left' + J_right)^-1 (eta_left' + eta_right)  (' = back in the frame of the process), then per half the junction
    // state u + Pc A (xi - Lam u), u = F x + g.  The junction states replace the four last slots' means, x goes to red[72..77].
    if (wv == 1) {
        const int hf = lane >> 5, hl = lane & 31, dd = hl < 3 ? hl : 2;
        const int dO = hf ? 3 : 0, dI = hf ? 2 : 1;
        double *sO = slots + (size_t)kSlot * (dO * nQ + nQ - 1), *sI = slots + (size_t)kSlot * (dI * nQ + nQ - 1);
        const dbl2 mO = *(const dbl2 *)(sO + 14 + 2 * dd), gI = *(const dbl2 *)(sI + 14 + 2 * dd);
        const dbl2 abI = *(const dbl2 *)(sI + 20);
        const dbl2 *jm = (const dbl2 *)(red + 80 + 16 * hf);                            // wave 2's part (phase 3)
        const dbl2 j0 = jm[0], j1v = jm[1], j2v = jm[2], j3v = jm[3], j4v = jm[4], j5v = jm[5];
        const double W11 = red[80 + 16 * hf + 12];
        const dbl2 Fc0 = *(const dbl2 *)(sI + 28), Fc1 = *(const dbl2 *)(sI + 30);      // columns of F: (F00, F10), (F01, F11)
        const dbl2 aE = *(const dbl2 *)(red + 32 + 2 * (dI * 5 + dd)), aJ0 = *(const dbl2 *)(red + 32 + 2 * (dI * 5 + 3)), aJ1 = *(const dbl2 *)(red + 32 + 2 * (dI * 5 + 4));
        const double m0 = mO.x, m1 = -mO.y;                                              // outer mean in the half's frame
        const double pa = abI.x, pb = abI.y, pd = sI[22];
        const double F00 = Fc0.x, F10 = Fc0.y, F01 = Fc1.x, F11 = Fc1.y, g0 = gI.x, g1 = gI.y;
        const double ia = j0.x, ib = j0.y, id = j1v.x, La = j1v.y, Lb = j2v.x, Ld = j2v.y, A00 = j3v.x, A01 = j3v.y, A10 = j4v.x, A11 = j4v.y;
        const double W00 = j5v.x, W01 = j5v.y, W10 = j5v.y;
        const double xi0 = fma(ia, m0, ib * m1), xi1 = fma(ib, m0, id * m1);
        const double w0 = fma(A00, xi0, A01 * xi1), w1 = fma(A10, xi0, A11 * xi1);
        const double v0 = w0 - fma(W00, g0, W01 * g1), v1 = w1 - fma(W10, g0, W11 * g1);
        const double WF00 = fma(W00, F00, W01 * F10), WF01 = fma(W00, F01, W01 * F11), WF10 = fma(W10, F00, W11 * F10), WF11 = fma(W10, F01, W11 * F11);
        // the half's message in the frame of the process: the left half's off-diagonal and second component change sign
        const double sg = hf ? 1.0 : -1.0;
        double J00 = fma(F00, WF00, F10 * WF10) + aJ0.x, J01 = sg * (fma(F00, WF01, F10 * WF11) + aJ0.y), J11 = fma(F01, WF01, F11 * WF11) + aJ1.y;
        double e0 = fma(F00, v0, F10 * v1) + aE.x, e1 = sg * (fma(F01, v0, F11 * v1) + aE.y);
        J00 = swap_add32(J00, J00); J01 = swap_add32(J01, J01); J11 = swap_add32(J11, J11); e0 = swap_add32(e0, e0); e1 = swap_add32(e1, e1);      // both halves: left + right
        J00 += cp0; J11 += cp1;
        const double rJ = fast_rcp(fma(J00, J11, -(J01 * J01)));
        const double x0 = (J11 * e0 - J01 * e1) * rJ, x1 = (J00 * e1 - J01 * e0) * rJ;      // state at j2, frame of the process
        const double s0 = x0, s1 = sg * x1;                                               // ... in the half's frame
        const double u0 = fma(F00, s0, fma(F01, s1, g0)), u1 = fma(F10, s0, fma(F11, s1, g1));
        const double r0 = xi0 - fma(La, u0, Lb * u1), r1 = xi1 - fma(Lb, u0, Ld * u1);
        const double z0 = fma(A00, r0, A01 * r1), z1 = fma(A10, r0, A11 * r1);
        const double y0 = fma(pa, z0, fma(pb, z1, u0)), y1 = fma(pb, z0, fma(pd, z1, u1));  // junction state, the half's (= the inner direction's) frame
        wave_lds_sync();                                // every lane has read the last slots' means
        if (hl < 3) {
            *(dbl2 *)(sI + 14 + 2 * hl) = dbl2{y0, y1};
            *(dbl2 *)(sO + 14 + 2 * hl) = dbl2{y0, -y1};
            if (hf) *(dbl2 *)(red + 72 + 2 * hl) = dbl2{x0, x1};
        }
    }
    __syncthreads();
    CSTAMP(5);
    // ---- 5. backward pass.  Short directions: one wave walks every direction step by step, x_k = e_k (+ E_k x) + C_k x_{k+1} (lane =
    //         (direction, coordinate); a step is ~10 instructions with two dependent FMAs, the records are requested two steps ahead) --
    //         cheaper than the strided form below while a direction has fewer than ~25 steps (its two parallel phases and two barriers cost
    //         ~2000 clocks before the first anchor moves).
    const bool direct = nQ <= kDirectMax;
    if (direct) {
        if (wv == 0) {
            const int dir = lane >> 4, hl = lane & 15;
            const bool wr = hl < 3, inner = dir == 1 || dir == 2;
            char *const sb = (char *)slots, *const db = (char *)dump;
            const char *ra = sb + (size_t)SB * (dir * nQ + nQ - 1);             // the direction's last slot
            char *la = wr ? (char *)ra + 16 * hl : db;
            const int lstep = wr ? SB : 0;
            dbl2 xs = dbl2{0.0, 0.0};                                           // the inner directions' start state, own frame
            if (inner && wr) { xs = *(const dbl2 *)(red + 72 + 2 * hl); if (dir == 1) xs.y = -xs.y; }
            const dbl2 xj = *(const dbl2 *)(la + 112);
            double x0 = xj.x, x1 = xj.y;
            struct BRec { dbl2 e, C0, C1, E0, E1; };
            // (four steps per trip on four register sets, as in the forward pass: every record is requested two steps before its use, nothing
            //  is copied; i = 1 .. 6: the slot i steps below the current one.  Slots in front of a direction's first one are readable)
            auto bfetch = [&](int i) __attribute__((always_inline)) {
                BRec r;
                const char *sp = ra - SB * i;
                r.e = *(const dbl2 *)(la - lstep * i + 112); r.C0 = *(const dbl2 *)(sp + 192); r.C1 = *(const dbl2 *)(sp + 208);
                r.E0 = *(const dbl2 *)(sp + 224); r.E1 = *(const dbl2 *)(sp + 240);
                return r;
            };
            auto bstep = [&](const BRec &r, int i) __attribute__((always_inline)) {
                const double e0 = fma(r.E0.x, xs.x, fma(r.E1.x, xs.y, r.e.x)), e1 = fma(r.E0.y, xs.x, fma(r.E1.y, xs.y, r.e.y));
                const double y0 = fma(r.C0.x, x0, fma(r.C0.y, x1, e0)), y1 = fma(r.C1.x, x0, fma(r.C1.y, x1, e1));
                x0 = y0; x1 = y1;
                *(dbl2 *)(la - lstep * i + 112) = dbl2{x0, x1};
            };
            int k = 1;
            BRec rA = bfetch(1), rB = bfetch(2);
            for (; k + 3 < nQ; k += 4) {
                const BRec rC = bfetch(3), rD = bfetch(4);
                bstep(rA, 1); bstep(rB, 2);
                rA = bfetch(5); rB = bfetch(6);
                bstep(rC, 3); bstep(rD, 4);
                ra -= 4 * SB; la -= 4 * lstep;
            }
            const int rem = nQ - k;                     // 0 .. 3 steps left
            if (rem >= 1) bstep(rA, 1);
            if (rem >= 2) bstep(rB, 2);
            if (rem >= 3) { const BRec r = bfetch(3); bstep(r, 3); }
        }
        __syncthreads();
    } else {
    // ---- backward pass in strides of four.  thread = step slot: the slot's step composed with the steps between it and the next
    //         ANCHOR above it (the slots a multiple of four below the junction): x_k = eh + Ch x_anchor.  The composites go where the
    //         records and right-hand sides were (dead by now): Ch -> [0..3], eh -> [8..13].
    for (int sl = t; sl < nSl; sl += MB) {
        const int k = sl - slot_dir(sl) * nQ;
        if (k < nQ - 1) {
            double *o = slots + (size_t)kSlot * sl;
            const int r4 = (nQ - 1 - k) & 3, j = r4 ? r4 : 4;       // steps k .. k + j - 1
            const double *top = o + (size_t)kSlot * (j - 1);
            dbl2 c0 = *(const dbl2 *)(top + 24), c1 = *(const dbl2 *)(top + 26);
            // an inner direction's e_k so far lacks the start state: e_k + E_k x, x = the state at j2 (red[72..77]) in the direction's frame
            const int dir = slot_dir(sl);
            const bool inner = dir == 1 || dir == 2;
            dbl2 xs[3];
#pragma unroll
            for (int q = 0; q < 3; ++q) { xs[q] = inner ? *(const dbl2 *)(red + 72 + 2 * q) : dbl2{0.0, 0.0}; if (dir == 1) xs[q].y = -xs[q].y; }
            auto e_of = [&](const double *s, dbl2 (&e)[3]) __attribute__((always_inline)) {
#pragma unroll
                for (int q = 0; q < 3; ++q) e[q] = *(const dbl2 *)(s + 14 + 2 * q);
                if (inner) {
                    const dbl2 E0 = *(const dbl2 *)(s + 28), E1 = *(const dbl2 *)(s + 30);
#pragma unroll
                    for (int q = 0; q < 3; ++q) e[q] = dbl2{fma(E0.x, xs[q].x, fma(E1.x, xs[q].y, e[q].x)), fma(E0.y, xs[q].x, fma(E1.y, xs[q].y, e[q].y))};
                }
            };
            dbl2 ev[3];
            e_of(top, ev);
            for (int i = j - 2; i >= 0; --i) {
                const double *s = o + (size_t)kSlot * i;
                const dbl2 d0 = *(const dbl2 *)(s + 24), d1 = *(const dbl2 *)(s + 26);
                dbl2 es[3];
                e_of(s, es);
#pragma unroll
                for (int q = 0; q < 3; ++q) ev[q] = dbl2{fma(d0.x, ev[q].x, fma(d0.y, ev[q].y, es[q].x)), fma(d1.x, ev[q].x, fma(d1.y, ev[q].y, es[q].y))};
                const dbl2 n0 = dbl2{fma(d0.x, c0.x, d0.y * c1.x), fma(d0.x, c0.y, d0.y * c1.y)};
                const dbl2 n1 = dbl2{fma(d1.x, c0.x, d1.y * c1.x), fma(d1.x, c0.y, d1.y * c1.y)};
                c0 = n0; c1 = n1;
            }
            // (written after every thread's reads of the raw cells: the composites live in other cells)
            *(dbl2 *)(o + 0) = c0; *(dbl2 *)(o + 2) = c1;
#pragma unroll
            for (int q = 0; q < 3; ++q) *(dbl2 *)(o + 8 + 2 * q) = ev[q];
        }
    }
    __syncthreads();
    // the anchors, one after the other (wave 0; lane = (direction, coordinate)): x_anchor = eh + Ch x_(anchor above), starting at the junction
    if (wv == 0) {
        const int dir = lane >> 4, hl = lane & 15;
        const bool wr = hl < 3;
        char *const sb = (char *)slots, *const db = (char *)dump;
        char *ra = sb + (size_t)SB * (dir * nQ + nQ - 1);
        char *la = wr ? ra + 16 * hl : db;
        const int lstep = wr ? SB : 0;
        const dbl2 xj = *(const dbl2 *)(la + 112);
        double xs0 = xj.x, xs1 = xj.y;
        const int na = (nQ - 1) >> 2;                   // anchors below the direction's last slot
        struct ARec { dbl2 e, C0, C1; };
        auto afetch = [&](int i) __attribute__((always_inline)) {           // the i-th anchor below the current position
            ARec r;                                                         // (slots in front of a direction's first one are readable)
            r.e = *(const dbl2 *)(la - 4 * lstep * i + 64); r.C0 = *(const dbl2 *)(ra - 4 * SB * i); r.C1 = *(const dbl2 *)(ra - 4 * SB * i + 16);
            return r;
        };
        auto astep = [&](const ARec &r, int i) __attribute__((always_inline)) {
            const double y0 = fma(r.C0.x, xs0, fma(r.C0.y, xs1, r.e.x));
            const double y1 = fma(r.C1.x, xs0, fma(r.C1.y, xs1, r.e.y));
            xs0 = y0; xs1 = y1;
            *(dbl2 *)(la - 4 * lstep * i + 112) = dbl2{xs0, xs1};
        };
        int i = 0;
        ARec rA = afetch(1), rB = afetch(2);
        for (; i + 1 < na; i += 2) {
            astep(rA, 1); astep(rB, 2);
            ra -= 8 * SB; la -= 8 * lstep;
            rA = afetch(1); rB = afetch(2);
        }
        if (i < na) astep(rA, 1);
    }
    __syncthreads();
    }
    CSTAMP(6);

    // ---- 5. T = Y0 + V, sigma2 (residual form of :418-422) and the convergence criterion (:424); publish Y and the nodes.  thread = step slot
    V4<T> *nodes_w = (V4<T> *)f.nodes;
    double s_np = 0, s_dr = 0, s_pd = 0, s_cr = 0;
    for (int sl = t, r = 0; sl < nSl; sl += MB, ++r) {
        const SlotQ q = r == 0 ? q0 : load_slot(sl, true);
        if (!q.obs) continue;
        const int m = q.node;
        const double *o = slots + (size_t)kSlot * sl;
        const double p1 = S[m];
        // smoothed state: junction and anchors hold it; every other slot is one composite step below its anchor
        const int ks = sl - slot_dir(sl) * nQ, r4 = (nQ - 1 - ks) & 3;
        double Vd[3];
        if (direct || r4 == 0) {
#pragma unroll
            for (int d = 0; d < 3; ++d) Vd[d] = o[14 + 2 * d];
        } else {
            const double *an = o + (size_t)kSlot * r4;
            const dbl2 c0 = *(const dbl2 *)(o + 0);
#pragma unroll
            for (int d = 0; d < 3; ++d) { const dbl2 xa = *(const dbl2 *)(an + 14 + 2 * d); Vd[d] = fma(c0.x, xa.x, fma(c0.y, xa.y, o[8 + 2 * d])); }
        }
        double Td[3], cr2 = 0, dr = 0, pd2 = 0;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            Td[d] = q.y0[d] + Vd[d];
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
    // the wave's four sums at once (wave_sum4: 21 instructions, no LDS round trips); rows 0..3 of the wave end up with sums 0, 2, 1, 3
    // (waves without slots contribute zeros)
    {
        const double tot = wave_sum4(s_np, s_dr, s_pd, s_cr);
        const int row = lane >> 4, which = ((row & 1) << 1) | (row >> 1);
        double *dst = (lane & 15) == 0 ? red + 4 * wv + which : dump;
        *dst = tot;
    }
    __syncthreads();
    CSTAMP(7);
    int pub = 0;        // lane 0: this M-step has something to tell the host (results mailbox, FrameDev::host_prog)
    if (t == 0) {
        const double t_np = ((red[0] + red[4]) + red[8]) + red[12], t_dr = ((red[1] + red[5]) + red[9]) + red[13];
        const double t_pd = ((red[2] + red[6]) + red[10]) + red[14], t_cr = ((red[3] + red[7]) + red[11]) + red[15];
        const double new_sigma2 = (S[4 * M] - 2.0 * t_dr + t_pd) * fast_rcp(t_np * 3.0);
        const double crit = t_cr / (double)M;
        const int it = itn + 1;
        st->it = it; st->crit = crit; st->Np = t_np;
        const bool finite_ok = (new_sigma2 == new_sigma2) && (fabs(new_sigma2) < 1e300) && (new_sigma2 > 0) && (crit == crit);
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
#ifdef TDLO_TIMELINE
        if (itn >= 20 && itn < 28) f.dbg[4 * (itn - 20) + 3] = __builtin_amdgcn_s_memrealtime();
#endif
    }
    spin_report();
    if (!XCH && t < 64 && __builtin_amdgcn_readfirstlane(pub)) host_publish(f, st, lane, true);      // progress (and, from the M-step that finishes the registration, the results) into pinned host memory
    if (TRK && f.lle_next != nullptr) {
        // the M-step that finishes the registration without an error goes on (the host has its results already) to form the LLE regulariser
        // of the nodes it leaves behind: the next frame's pre-processing registration starts from them (tdlo_lle_dev.h)
        if (t == 0) red[30] = (st->done != 0 && st->status == 0) ? 1.0 : 0.0;
        __syncthreads();
        if (red[30] != 0.0 && M <= 256) lle_band_device<MB>(f.Yout, M, f.lle_next, slots, t);
    }
#undef CSTAMP
}

}  // namespace tdlo
