// tdlo_mstep_chain.hip -- M-step (trackdlo.cpp:392-437) without the LLE term as a smoother along the chain: O(M); four directions up to 512 nodes, k_mstep_chain_long beyond.
//
// What the reference solves each iteration (:405-413, include_lle == false):
//     (c I + D G) W = B,   T = Y0 + G W,      c = lambda sigma2,  D = diag(P1) + alpha J  (diagonal, >= 0),
// with the kernel of :233
//     G_ij = 1/(4 beta^2) exp(-sqrt2 |c_i - c_j| / beta) (2 |c_i - c_j| + sqrt2 beta)
//          = sf2 (1 + s d) exp(-s d),        d = |c_i - c_j|,  s = sqrt2 / beta,  sf2 = 1 / (2 sqrt2 beta)
// over the chain coordinates c_0 <= c_1 <= ... (:219-223, a running sum, i.e. sorted).  Only T is used afterwards (:417-431).
// With V = G W the system reads (c G^-1 + D) V = B: V is the posterior mean of a Gaussian process with covariance G / c
// observed at the nodes with precisions D_i ("observation" B_i / D_i).  G is the Matern-3/2 kernel in one dimension, and the
// process with that covariance is Markov in the state x = (f, f'):
//     x_{i+1} = Phi(h_i) x_i + noise,  Phi(h) = e^{-sh} [[1 + sh, h], [-s^2 h, 1 - sh]],  Q(h) = Pinf - Phi Pinf Phi^T,
//     Pinf = sf2 diag(1, s^2),  h_i = c_{i+1} - c_i.
// The posterior mean of a Markov chain is a Kalman filter pass plus a Rauch-Tung-Striebel smoothing pass: M dependent steps
// on 2 x 2 matrices instead of the M dependent column eliminations on an M x M tableau of the dense kernels (k_mstep_fast:
// 13 panels, 10 us at M = 50; k_mstep_mcu: 19 workgroups, 131 us at M = 300).  No M x M matrix is touched at all: neither G
// (20 KB at M = 50, 2 MB at M = 512) nor the product G W.  It is the same linear system, so the result is the reference's up
// to rounding -- and the rounding is smaller: the dense system has condition number ~ P1 G / c (1e5 ... 1e8 as sigma2 falls), the
// filter's quantities are all O(1) ratios (measured against an 80-bit solve of the dense system: 1e-17 ... 1e-14 m where
// partial-pivot LU / QR of the dense system leave 1e-14 ... 1e-9 m; tests/test_mstep_chain.py).
//
// What bounds it: O(M) flops on a handful of lanes -- a chain of dependent instructions.  A lone wave on gfx950 issues one instruction
// per ~8 cycles when it depends on its predecessor (~5 when it does not), scalar ALU, fp64 and fp32 alike; a dependent v_rcp_f64
// is 20 cycles, an LDS round trip 76, a store behind an exec-mask branch 38 (scripts/ubench/lat.hip).  So the kernel is designed by
// instruction count on the critical wave, and independent instruction streams go to different waves.
//
// The chain is walked from FOUR ends at once, 16 lanes of a wave per direction (a wave64 fp64 instruction costs the same whether 3
// or 64 lanes are active, so the dependent chain is quartered for the price of a small junction solve).  With j2 the middle node
// and j1, j3 the quarter points:
//   0: nodes 0 .. j1, a Kalman filter from the stationary prior                 3: nodes M-1 .. j3, the same for the time-reversed
//   2: nodes j2 .. j3, a filter started AT j2 from the exact but unknown            process (same Phi and Q: the process is stationary
//      state x there (covariance 0)                                                 and reversible in (f, -f'))
//   1: nodes j2-1 .. j1 of the reversed process, likewise from x.
// The covariances of an inner direction (1, 2) do not depend on x; its means are affine in x, m_k = g_k + F_k x, so it carries two
// more mean columns (the SPIKE columns F, started at the identity) next to the three coordinates g, and sums the likelihood of its
// data as a function of x along the way.  At j1 (and j3) the outer direction's posterior meets the inner direction's last step:
// eliminating the junction state leaves a 2 x 2 message for x per half, x follows from the two messages and the prior, the
// junction states by back-substitution.  From there every direction is smoothed backwards from its last step exactly as before
// (Rauch-Tung-Striebel), the inner ones with x inserted.  Phases (one workgroup of four waves per frame; thread = step slot in the parallel ones):
//   1. fetch: the E-step's sums (16 short rows of fixed-point accumulators, both parities: no load waits for the iteration
//      counter), this thread's slot (link, node); the slot records are prepared while the loads are in flight;
//   2. records and right-hand sides of the step slots;
//   3. forward pass as a two-wave pipeline: wave 0 runs the covariance recursion (2 x 2 predict Phi P Phi^T + Q, one reciprocal,
//      update -- it does not depend on the data), wave 1 follows with the means (lane & 15 < 5 = coordinate / spike column) as the
//      posteriors appear, behind a progress counter in LDS; wave 2 then prepares the junction solve's covariance-only part;
//   4. gains C_k = P_k Phi^T (P^-_{k+1})^-1 and e_k = m_k - C_k Phi m_k, lane = step (the backward step is x_k = e_k + C_k x_{k+1});
//      meanwhile wave 1 solves for x and the junction states;
//   5. backward pass: short directions (<= kDirectMax steps) are walked back step by step by one wave; longer ones in strides of four:
//      every slot composes its step with those up to the next anchor above it (affine maps compose; an inner direction's e_k gets
//      its E_k x here), one wave walks the anchors, the slots between are filled in parallel;
//   6. T = Y0 + V, sigma2 (residual form), stopping rule, the next E-step's constants.
// M / 4 dependent steps instead of M / 2 from two ends (the former version): 8.8 instead of 10.3 us at M = 50, 21 instead of 36 us at M = 300.
//
// The dense kernels stay in the tree: with the LLE term (include_lle, the pre-processing registration of tracking_step) the
// system is not of this form, and TDLO_MSTEP=dense selects them as comparators for the tests.
#include "tdlo_mstep_chain_body.h"
#include <cstdlib>

namespace tdlo {
extern thread_local hipEvent_t g_mstep_ev[2];       // tdlo_device.hip: start/stop events for the M-step dispatch (tdlo_profile_iteration)

// (the body: tdlo_mstep_chain_body.h, mstep_chain_run)
template <typename T, bool SINGLE, bool XCH, bool TRK = false, bool SPIN = false, int ROWS = kAccRows, bool HINT = false>
__global__ __launch_bounds__(kCB) void k_mstep_chain(const FrameDev *__restrict__ frames, const FrameDev f0, int from_sums_in) {
    const int from_sums = from_sums_in & 0xff, par_hint = (from_sums_in >> 8) & 1;          // (HINT: bit 8 carries the iteration's parity)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // One wave walks a chain of dependent instructions.  In a batch the other stream groups' E-steps fill the same SIMDs with waves that always have
    // something to issue: at the default priority this wave takes its turn among them (C3: 10.0 us per M-step against 7.4 us with the GPU to itself)
    if (!SINGLE) __builtin_amdgcn_s_setprio(3);
    mstep_chain_run<T, SINGLE, XCH, TRK, SPIN, ROWS, HINT>(SINGLE ? f0 : frames[blockIdx.x], from_sums, smem, par_hint);
}

// ------------------------------------------------------------------------------------------------
// Chains beyond kChainLdsMaxNodes nodes (round 4: the reference takes any num_of_nodes, trackdlo.cpp:30-46).  The four-direction kernel above
// keeps a 272-byte slot per step in LDS and is at the CU's 160 KB with 512 nodes.  This one is the plain form of the same solve: ONE Kalman
// filter pass from the chain's head, one Rauch-Tung-Striebel pass back, on one wave (lane d = coordinate d; the covariances are computed
// redundantly), with the 14 doubles per node that the two passes exchange in LDS:
//     observation precision d_i = P1_i + alpha J_i and right-hand side b_i (3)   | forward pass in, smoothed V_i out (over b)
//     backward map  m^s_i = e_i + C_i m^s_{i+1}:  e_i (3 x 2), C_i (2 x 2)       | forward pass out, backward pass in
// i.e. 112 KB at 1024 nodes.  The links (Phi, Q of every gap, written by k_setup) are streamed from memory a trip of four steps ahead.  O(M)
// like the kernel above, ~0.4 ms per M-step at 1024 nodes (a lone wave issues a dependent instruction every ~8 clocks): a chain of that
// length is far from the production size, and it is served, not tuned.  Same linear system as trackdlo.cpp:405-417:  (c G^-1 + D) V = B,
// T = Y0 + V, c = lambda sigma2: V is the posterior mean of the process with covariance G / c observed at the nodes with precisions d_i.
//   filter at node i, from the prediction (m-, P-):   g = 1 / (1 + d P-_ff);  m = m- + P-[:, f] (b - d m-_f) g;  P = P- - d g P-[:, f] P-[f, :]
//   prediction over the gap to node i + 1:            m- = Phi m;  P- = Phi P Phi^T + Q / c
//   smoother:                                         C_i = P_i Phi^T (P-_{i+1})^-1;  e_i = m_i - C_i m-_{i+1}
// No division by d anywhere: nodes without points (d = 0) are pure prediction steps.
// from_sums: 0 the E-step's accumulators, 1 the reduced sums of the N-split, 2 export only; the one-shot exchange (3) is not carried.
// ------------------------------------------------------------------------------------------------
struct ChainLink { double p11, p12, p21, p22, q11, q12, q22; };

template <typename T, bool SINGLE>
__global__ __launch_bounds__(kCB) void k_mstep_chain_long(const FrameDev *__restrict__ frames, const FrameDev f0, int from_sums) {
    constexpr int MB = kCB;
    if (!SINGLE) __builtin_amdgcn_s_setprio(3);      // (see k_mstep_chain)
    const FrameDev &f = SINGLE ? f0 : frames[blockIdx.x];
    IterState *st = f.st;
    const int M = f.M, t = threadIdx.x, lane = t & 63;
    const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
    const int nS = 4 * M + 1;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double *dob = (double *)smem;                // M: d_i
    double *bv = dob + ((M + 1) & ~1);           // 3 M: b, later V (coordinate-major)
    double *eL = bv + ((3 * M + 1) & ~1);        // 6 M: e_i [coordinate][component]
    double *CL = eL + 6 * (size_t)M;             // 4 M: C_i (ff, fp, pf, pp)
    double *red = CL + 4 * (size_t)M;            // 32
    const auto stg = TDLO_AS_GLOBAL(IterState, st);
    const int done = stg->done;
    const double sigma2 = stg->sigma2;
    const int pri = f.has_priors;
    const int itn = stg->it;
    const double Nc = stg->Nc;
    const double kc = f.mu / (1.0 - f.mu) * (f.vis_branch ? 1.0 / Nc : (double)M / Nc);
    if (done) {
        if (t < 64 && stg->status != 0) host_publish(f, st, lane, false);
        return;
    }
    const auto ndg = TDLO_AS_GLOBAL(V4<T>, f.nodes);
    const auto Yg = TDLO_AS_GLOBAL(double, f.Y);
    const auto Y0g = TDLO_AS_GLOBAL(double, f.Y0);
    // the sums are not kept in LDS: the tail reads P1 / R / Q once more (this parity's accumulator rows stay until the next M-step clears them)
    auto sum_at = [&](int i) __attribute__((always_inline)) -> double { return from_sums == 1 ? f.sums[i] : acc_read_both(f, i, itn); };
    if (from_sums == 2) {       // split mode, export only
        for (int i = t; i < nS; i += MB) f.sums[i] = acc_read_both(f, i, itn);
        if (t == 0) f.sums[nS] = (double)stg->N;
        acc_clear_other<MB>(f, itn, t);
        return;
    }
    // ---- 1. d, b (thread = node)
    for (int m = t; m < M; m += MB) {
        const double p1 = sum_at(m);
        dob[m] = p1 + (pri ? f.aJ[m] : 0.0);
        const double y[3] = {(double)ndg[m].x, (double)ndg[m].y, (double)ndg[m].z};
#pragma unroll
        for (int d = 0; d < 3; ++d) bv[d * M + m] = sum_at((1 + d) * M + m) + p1 * (y[d] - Y0g[d * M + m]) + (pri ? f.aYd[d * M + m] : 0.0);
    }
    if (from_sums != 1) acc_clear_other<MB>(f, itn, t);
    __syncthreads();
    // ---- 2. the two passes on wave 0
    if (wv == 0) {
        const auto chg = TDLO_AS_GLOBAL(double, f.chain);
        const int cd = lane < 3 ? lane : 0;                              // this lane's coordinate
        const double rc = fast_rcp(f.lambda * sigma2);                  // 1 / c
        auto load_link = [&](int i) __attribute__((always_inline)) {    // the gap between nodes i - 1 and i (index clamped: what lies past the chain is loaded and not used)
            const size_t o = 8 * (size_t)(i < 1 ? 1 : (i > M - 1 ? M - 1 : i));
            ChainLink L;
            L.p11 = chg[o + 0]; L.p12 = chg[o + 1]; L.p21 = chg[o + 2]; L.p22 = chg[o + 3]; L.q11 = chg[o + 4]; L.q12 = chg[o + 5]; L.q22 = chg[o + 6];
            return L;
        };
        // prediction at node 0: the stationary prior of the process, P_inf / c
        double Pff = chg[0] * rc, Pfp = 0.0, Ppp = chg[1] * rc, mf = 0.0, mp = 0.0;
        double Fff = 0, Ffp = 0, Fpp = 0, nf = 0, np_ = 0;             // the filtered state of the node before ...
        ChainLink G = load_link(1);                                     // ... and the gap that led from it to this node (set at the end of every step)
        ChainLink Lq[4] = {load_link(1), load_link(2), load_link(3), load_link(4)};
        for (int i0 = 0; i0 < M; i0 += 4) {
            const ChainLink Ln[4] = {load_link(i0 + 5), load_link(i0 + 6), load_link(i0 + 7), load_link(i0 + 8)};      // the next trip's gaps: requested now
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + u;
                if (i < M) {
                    if (i > 0) {        // backward map of node i - 1: C = F Phi^T (P-)^-1, e = n - C m-
                        const double aff = Fff * G.p11 + Ffp * G.p12, afp = Fff * G.p21 + Ffp * G.p22;
                        const double apf = Ffp * G.p11 + Fpp * G.p12, app = Ffp * G.p21 + Fpp * G.p22;
                        const double rdet = fast_rcp(Pff * Ppp - Pfp * Pfp);
                        const double cff = (aff * Ppp - afp * Pfp) * rdet, cfp = (afp * Pff - aff * Pfp) * rdet;
                        const double cpf = (apf * Ppp - app * Pfp) * rdet, cpp = (app * Pff - apf * Pfp) * rdet;
                        if (lane < 3) { eL[6 * (size_t)(i - 1) + 2 * cd] = nf - (cff * mf + cfp * mp); eL[6 * (size_t)(i - 1) + 2 * cd + 1] = np_ - (cpf * mf + cpp * mp); }
                        if (lane == 0) { CL[4 * (size_t)(i - 1)] = cff; CL[4 * (size_t)(i - 1) + 1] = cfp; CL[4 * (size_t)(i - 1) + 2] = cpf; CL[4 * (size_t)(i - 1) + 3] = cpp; }
                    }
                    // filter at node i
                    const double d = dob[i], b = bv[cd * M + i];
                    const double g = fast_rcp(1.0 + d * Pff);
                    const double in = (b - d * mf) * g, dg = d * g;
                    nf = mf + Pff * in; np_ = mp + Pfp * in;
                    Fff = Pff - dg * Pff * Pff; Ffp = Pfp - dg * Pff * Pfp; Fpp = Ppp - dg * Pfp * Pfp;
                    // prediction over the gap to node i + 1
                    G = Lq[u];
                    mf = G.p11 * nf + G.p12 * np_; mp = G.p21 * nf + G.p22 * np_;
                    const double bff = G.p11 * Fff + G.p12 * Ffp, bfp = G.p11 * Ffp + G.p12 * Fpp;
                    const double bpf = G.p21 * Fff + G.p22 * Ffp, bpp = G.p21 * Ffp + G.p22 * Fpp;
                    Pff = bff * G.p11 + bfp * G.p12 + G.q11 * rc; Pfp = bff * G.p21 + bfp * G.p22 + G.q12 * rc; Ppp = bpf * G.p21 + bpp * G.p22 + G.q22 * rc;
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) Lq[u] = Ln[u];
        }
        // backward: the last node's filtered state is its smoothed one
        double sf = nf, sp = np_;
        if (lane < 3) bv[cd * M + (M - 1)] = sf;
        for (int i = M - 2; i >= 0; --i) {
            const double cff = CL[4 * (size_t)i], cfp = CL[4 * (size_t)i + 1], cpf = CL[4 * (size_t)i + 2], cpp = CL[4 * (size_t)i + 3];
            const double ef = eL[6 * (size_t)i + 2 * cd], ep = eL[6 * (size_t)i + 2 * cd + 1];
            const double nsf = ef + (cff * sf + cfp * sp), nsp = ep + (cpf * sf + cpp * sp);
            sf = nsf; sp = nsp;
            if (lane < 3) bv[cd * M + i] = sf;
        }
    }
    __syncthreads();
    // ---- 3. T = Y0 + V, sigma2 (residual form of :418-422) and the convergence criterion (:424); publish Y and the nodes.  thread = node
    const double ctr0 = f.ctr[0], ctr1 = f.ctr[1], ctr2 = f.ctr[2];
    V4<T> *nodes_w = (V4<T> *)f.nodes;
    double s_np = 0, s_dr = 0, s_pd = 0, s_cr = 0;
    for (int m = t; m < M; m += MB) {
        const double p1 = sum_at(m);
        const double y[3] = {(double)ndg[m].x, (double)ndg[m].y, (double)ndg[m].z};
        const T qw = ndg[m].w;
        double Td[3], cr2 = 0, dr = 0, pd2 = 0;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            Td[d] = Y0g[d * M + m] + bv[d * M + m];
            const double del = Td[d] - y[d], ex = Yg[d * M + m] - Td[d];
            dr = fma(del, sum_at((1 + d) * M + m), dr); pd2 = fma(del, del, pd2); cr2 = fma(ex, ex, cr2);
        }
        s_np += p1; s_dr += dr; s_pd += p1 * pd2; s_cr += ::sqrt(cr2);
        V4<T> w; w.x = (T)Td[0]; w.y = (T)Td[1]; w.z = (T)Td[2]; w.w = qw;      // .w = chain coordinate, unchanged
        nodes_w[m] = w;
        f.dminbits[m] = ~0ull;
#pragma unroll
        for (int d = 0; d < 3; ++d) { f.Y[d * M + m] = Td[d]; f.Yout[d * M + m] = Td[d] + (d == 0 ? ctr0 : (d == 1 ? ctr1 : ctr2)); }
    }
    s_np = wave_sum(s_np); s_dr = wave_sum(s_dr); s_pd = wave_sum(s_pd); s_cr = wave_sum(s_cr);
    if (lane == 0) { red[4 * wv] = s_np; red[4 * wv + 1] = s_dr; red[4 * wv + 2] = s_pd; red[4 * wv + 3] = s_cr; }
    __syncthreads();
    int pub = 0;
    if (t == 0) {
        const double t_np = ((red[0] + red[4]) + red[8]) + red[12], t_dr = ((red[1] + red[5]) + red[9]) + red[13];
        const double t_pd = ((red[2] + red[6]) + red[10]) + red[14], t_cr = ((red[3] + red[7]) + red[11]) + red[15];
        const double new_sigma2 = (sum_at(4 * M) - 2.0 * t_dr + t_pd) * fast_rcp(t_np * 3.0);
        const double crit = t_cr / (double)M;
        const int it = itn + 1;
        st->it = it; st->crit = crit; st->Np = t_np;
        const bool finite_ok = (new_sigma2 == new_sigma2) && (fabs(new_sigma2) < 1e300) && (new_sigma2 > 0) && (crit == crit);
        st->sigma2 = new_sigma2;
        if (finite_ok) {
            const double tp = 2.0 * M_PI * new_sigma2, rtp = ::sqrt(tp);
            st->k2 = -1.4426950408889634 * 0.5 * fast_rcp(new_sigma2);
            st->c_norm = tp * rtp * kc;
            st->rwin32 = f.win_e32 * 1.3862943611198906 * new_sigma2; st->rwin64 = f.win_e64 * 1.3862943611198906 * new_sigma2;
        } else { st->status = TDLO_E_NUMERIC; st->done = 1; st->converged = 0; pub = 1; }
        if (crit < f.tol) { st->done = 1; pub = 1; }                                   // :424-428
        else if (it >= f.max_iter) { st->converged = 0; st->done = 1; pub = 1; }      // :433-437
        if (it == f.host_report_it) pub = 1;
    }
    if (t < 64 && __builtin_amdgcn_readfirstlane(pub)) host_publish(f, st, lane, true);
}
static size_t mstep_chain_long_lds_bytes(int M) { return sizeof(double) * ((size_t)((M + 1) & ~1) + (size_t)((3 * M + 1) & ~1) + 10 * (size_t)M + 32); }

static size_t mstep_chain_lds_bytes(int M) { return ChainCarve(M).total * sizeof(double); }

template <typename K> static hipError_t set_lds_c(K kernel, size_t bytes) {
    if (bytes > 64 * 1024) return hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    return hipSuccess;
}

// The iteration a caller is about to enqueue, counted from the registration's first (the device's counter IterState::it starts at 0 in the set-up kernel and goes
// up by one per M-step): the plain chain M-steps launched by this thread then fetch that parity's accumulator rows only.  -1: not known (both parities, as before).
static thread_local int g_par_hint = -1;
void mstep_parity_hint(int iteration) {
    static const int flip = getenv("TDLO_TEST_PARITY_FLIP") ? atoi(getenv("TDLO_TEST_PARITY_FLIP")) : 0;      // (test hook: every hint wrong -- the kernel's check must catch it)
    g_par_hint = iteration < 0 ? -1 : ((iteration + flip) & 1);
}

template <typename T> static hipError_t launch_mstep_chain_T(const FrameDev *fd, const FrameDev *fh, int F, int from_sums, hipStream_t s) {
    hipError_t e;
    if (fh[0].M > kChainLdsMaxNodes) {          // long chains: one direction, compact records (k_mstep_chain_long)
        if (from_sums == 3) return hipErrorInvalidValue;      // (the one-shot exchange is not carried: tdlo_split_run asks for a communicator)
        const size_t ldsl = mstep_chain_long_lds_bytes(fh[0].M);
        if (F == 1) {
            if ((e = set_lds_c(k_mstep_chain_long<T, true>, ldsl)) != hipSuccess) return e;
            if (g_mstep_ev[0]) hipExtLaunchKernelGGL((k_mstep_chain_long<T, true>), dim3(1), dim3(kCB), ldsl, s, g_mstep_ev[0], g_mstep_ev[1], 0, fd, fh[0], from_sums);
            else hipLaunchKernelGGL((k_mstep_chain_long<T, true>), dim3(1), dim3(kCB), ldsl, s, fd, fh[0], from_sums);
        } else {
            if ((e = set_lds_c(k_mstep_chain_long<T, false>, ldsl)) != hipSuccess) return e;
            if (g_mstep_ev[0]) hipExtLaunchKernelGGL((k_mstep_chain_long<T, false>), dim3(F), dim3(kCB), ldsl, s, g_mstep_ev[0], g_mstep_ev[1], 0, fd, fh[0], from_sums);
            else hipLaunchKernelGGL((k_mstep_chain_long<T, false>), dim3(F), dim3(kCB), ldsl, s, fd, fh[0], from_sums);
        }
        return hipGetLastError();
    }
    const size_t lds = mstep_chain_lds_bytes(fh[0].M);
    if (from_sums == 3) {          // one frame (a shard of the split cloud), exchange inside the kernel
        if (F != 1) return hipErrorInvalidValue;
        if (g_par_hint >= 0) {          // (tdlo_split_run counts its iterations as run_frames does: the shard's own sums from that parity's rows alone)
            if ((e = set_lds_c(k_mstep_chain<T, true, true, false, false, kAccRows, true>, lds)) != hipSuccess) return e;
            hipLaunchKernelGGL((k_mstep_chain<T, true, true, false, false, kAccRows, true>), dim3(1), dim3(kCB), lds, s, fd, fh[0], from_sums | (g_par_hint << 8));
            return hipGetLastError();
        }
        if ((e = set_lds_c(k_mstep_chain<T, true, true>, lds)) != hipSuccess) return e;
        hipLaunchKernelGGL((k_mstep_chain<T, true, true>), dim3(1), dim3(kCB), lds, s, fd, fh[0], from_sums);
    } else if (F == 1 && (fh[0].spec_flag != nullptr || fh[0].lle_next != nullptr || ((from_sums == 1 || fh[0].late_mstep != 0) && fh[0].late_aJ != nullptr))) {
        if ((e = set_lds_c(k_mstep_chain<T, true, false, true>, lds)) != hipSuccess) return e;
        if (g_mstep_ev[0]) hipExtLaunchKernelGGL((k_mstep_chain<T, true, false, true>), dim3(1), dim3(kCB), lds, s, g_mstep_ev[0], g_mstep_ev[1], 0, fd, fh[0], from_sums);
        else hipLaunchKernelGGL((k_mstep_chain<T, true, false, true>), dim3(1), dim3(kCB), lds, s, fd, fh[0], from_sums);
    } else if (F == 1 && fh[0].spin_on != 0) {          // the spin-ahead loop's M-step (63 nodes at most: launch_iteration_spin's caller has checked)
        if (4 * fh[0].M + 1 > kCB) return hipErrorInvalidValue;
        if ((e = set_lds_c(k_mstep_chain<T, true, false, false, true>, lds)) != hipSuccess) return e;
        if (g_mstep_ev[0]) hipExtLaunchKernelGGL((k_mstep_chain<T, true, false, false, true>), dim3(1), dim3(kCB), lds, s, g_mstep_ev[0], g_mstep_ev[1], 0, fd, fh[0], from_sums);
        else hipLaunchKernelGGL((k_mstep_chain<T, true, false, false, true>), dim3(1), dim3(kCB), lds, s, fd, fh[0], from_sums);
    } else if (F == 1) {
        // (the plain one-frame kernel exists for 2, 4 and all 8 replica rows of the accumulators: FrameDev::acc_rows says how many the E-step in front used;
        //  every other variant adds up all eight -- the unused ones are zero)
        //  When the caller has said which iteration this is (mstep_parity_hint), the HINT instantiation asks for that parity's rows alone.
        const int hint = from_sums == 0 ? g_par_hint : -1, fsh = from_sums | ((hint > 0 ? 1 : 0) << 8);
#define TDLO_CHAIN1(SGL, ROWS, HINT, NWG) do { \
        if ((e = set_lds_c(k_mstep_chain<T, SGL, false, false, false, ROWS, HINT>, lds)) != hipSuccess) return e; \
        if (g_mstep_ev[0]) hipExtLaunchKernelGGL((k_mstep_chain<T, SGL, false, false, false, ROWS, HINT>), dim3(NWG), dim3(kCB), lds, s, g_mstep_ev[0], g_mstep_ev[1], 0, fd, fh[0], fsh); \
        else hipLaunchKernelGGL((k_mstep_chain<T, SGL, false, false, false, ROWS, HINT>), dim3(NWG), dim3(kCB), lds, s, fd, fh[0], fsh); } while (0)
        if (hint >= 0) { if (fh[0].acc_rows == 2) TDLO_CHAIN1(true, 2, true, 1); else if (fh[0].acc_rows == 4) TDLO_CHAIN1(true, 4, true, 1); else TDLO_CHAIN1(true, kAccRows, true, 1); }
        else { if (fh[0].acc_rows == 2) TDLO_CHAIN1(true, 2, false, 1); else if (fh[0].acc_rows == 4) TDLO_CHAIN1(true, 4, false, 1); else TDLO_CHAIN1(true, kAccRows, false, 1); }
    } else {
        const int hint = from_sums == 0 ? g_par_hint : -1, fsh = from_sums | ((hint > 0 ? 1 : 0) << 8);
        // (a batch whose frames ALL spread over two rows -- run_frames decides that for the whole batch -- gets the two-row instantiation)
        bool two = true;
        for (int i = 0; i < F; ++i) two = two && fh[i].acc_rows == 2;
        if (hint >= 0 && two) TDLO_CHAIN1(false, 2, true, F); else if (hint >= 0) TDLO_CHAIN1(false, kAccRows, true, F); else TDLO_CHAIN1(false, kAccRows, false, F);
#undef TDLO_CHAIN1
    }
    return hipGetLastError();
}

// test aid (tdlo_debug_lle_band_device): lle_band_device on its own
__global__ __launch_bounds__(kCB) void k_lle_band_debug(const double *__restrict__ Y, int M, double *__restrict__ Hb) {
    __shared__ double Ab[7 * 256];
    lle_band_device<kCB>(Y, M, Hb, Ab, threadIdx.x);
}
hipError_t launch_lle_band_debug(const double *Y, int M, double *Hb, hipStream_t s) {
    if (M < 1 || M > 256) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_lle_band_debug, dim3(1), dim3(kCB), 0, s, Y, M, Hb);
    return hipGetLastError();
}

hipError_t launch_mstep_chain(const FrameDev *fd, const FrameDev *fh, int F, int from_sums, bool f64, hipStream_t s) {
    return f64 ? launch_mstep_chain_T<double>(fd, fh, F, from_sums, s) : launch_mstep_chain_T<float>(fd, fh, F, from_sums, s);
}

// Which M-step serves registrations without the LLE term: 0 the chain smoother (default), 1 the dense eliminations
// (k_mstep_fast<MFMA> / k_mstep_mcu), kept as comparators.  Process-wide; initial value from TDLO_MSTEP=dense.
static std::atomic<int> g_mstep_dense{[] { const char *e = getenv("TDLO_MSTEP"); return (e && e[0] == 'd') ? 1 : 0; }()};
bool mstep_chain_enabled() { return g_mstep_dense.load(std::memory_order_relaxed) == 0; }
int mstep_set_dense(int on) { return g_mstep_dense.exchange(on ? 1 : 0); }

}  // namespace tdlo
