// tdlo_estep2.hip -- the E-step of trackdlo.cpp:278-389 with TWO points per lane (round 6), for clouds that fill the GPU.
//
// k_estep (tdlo_device.hip) gives every lane one point: a wave = one 64-point batch.  With the GPU full that kernel is bound by the number of
// vector instructions it issues (profiles/r05_estep_sq_counters_c4.txt: 450 per batch, 41.6 % of them fp32 arithmetic), and most of what is not
// arithmetic is paid per BATCH: the candidate range of the nearest-node search, the node window, the scalar node loads, the fold of the column
// sums and their fixed-point tail.  Here a wave takes 128 points -- lane l holds points l and 64 + l of its batch -- so that
//   * every per-batch step is shared by twice the points (one candidate range, one window, one fold + tail per node chunk),
//   * the per-point arithmetic of the two points runs in packed fp32 instructions (v_pk_add / v_pk_mul / v_pk_fma_f32: both points' distance,
//     exponent argument and running sums in one issue slot; only the compares, selects and the two v_exp_f32 stay per point),
//   * the next batch's coordinates are requested while the current one is worked on.
// What it computes is the same mathematics as k_estep (same exact candidate pruning of the nearest-node search, same window rule, same
// 64-bit fixed-point sums, range-checked); the GRAIN of the fp32 tile sums is one wave x one 128-point batch, so its sums are not the bits of
// k_estep's -- both are held to the oracle at the fp32 mode's gate (1e-5 m, 1e-3 on sigma2), and each is repeatable bit for bit run to run.
// fp32 mode, chains of 8 .. 64 nodes; everything else stays with k_estep (launch_estep_T, FrameDev::estep2).
#include "tdlo_devcommon.h"
#include "tdlo_mstep_generic.h"
#include "tdlo_mstep_chain_body.h"
#include <algorithm>
#include <hip/hip_ext.h>
#include <type_traits>

namespace tdlo {

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f2 sp(float s) { return (f2){s, s}; }
__device__ __forceinline__ f2 fma2(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }

constexpr int kP2Stride = 132;        // dwords per row of the membership tile: 128 points + 4 (rows 4 banks apart: the 16 rows x 4 dwords a ds_read_b128 serves at a time fall on 64 different banks)
constexpr int kE2Points = 128;        // points per wave and batch

// one wave-wide butterfly for four maxima of unsigned keys (see wave_min_max_max_nonneg): a -> lane 15, c -> lane 31, b -> lane 47, d -> lane 63
__device__ __forceinline__ unsigned rows4_max(unsigned a, unsigned b, unsigned c, unsigned d) {
    return rows_max_u32(fold16_max(fold32_max(a, b), fold32_max(c, d)));
}

// (TR: rows of the membership tile.  8 rows: 28 KB of LDS per workgroup and -- told so -- 86 VGPRs: FIVE workgroups per CU, five waves per SIMD; the
//  compiler's own choice was 110 VGPRs = four.  16 rows: four workgroups of 45 KB, 96 VGPRs with a handful of spills -- the comparator.)
// The kernel's body as a device function of (frame, chunk): chunk = what blockIdx.x is to k_estep2 -- the workgroup's place among the frame's nblkE workgroups.
// Shared by k_estep2 and by the batches' persistent loop (k_batch_loop below).
template <bool VIS, int TR>
__device__ __forceinline__ void estep2_chunk(const FrameDev &f, const int chunk, char *smem) {
    constexpr int NWE = 4, EB = 256;
#ifdef TDLO_ESTEP_PHASES
    unsigned long long ph_prev = __builtin_amdgcn_s_memtime(), ph_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define E2PHASE(i) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); ph_acc[i] += t_ - ph_prev; ph_prev = t_; } while (0)
#else
#define E2PHASE(i) do { } while (0)
#endif
    const auto stg = TDLO_AS_GLOBAL(IterState, f.st);
    const int M = f.M;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // LDS carve (every offset a multiple of 16 bytes)
    V4<float> *nodesL = (V4<float> *)smem;                                  // M
    float *lvL = (float *)(nodesL + M);                                     // M rounded up to 4
    float *ptsAll = lvL + ((M + 3) & ~3);                                   // NWE x 4 x 128: a wave's normalised points as four arrays w0 | wx | wy | wz
    float *tileAll = ptsAll + NWE * 4 * kE2Points;                          // NWE x TR x kP2Stride
    float *pw = ptsAll + wave * 4 * kE2Points;
    float *pb = tileAll + (size_t)wave * TR * kP2Stride;
    double *scratch = (double *)(tileAll + (size_t)NWE * TR * kP2Stride);   // 16 doubles
    long long *accL = (long long *)(scratch + 16);                          // [M][4] 64-bit sums of the workgroup (ds_add_u64: integer, any order)

    const auto xs = TDLO_AS_GLOBAL(float, f.Xs);
    const size_t ld = f.ldx;
    const int done = stg->done;
    const int N = stg->N;
    const float k2 = (float)stg->k2;
    const float cn = (float)stg->c_norm;
    const int bstride = f.nblkE * NWE;
    int batch = chunk * NWE + wave;
    // the first batch's points, the nodes for the LDS copy, and this lane's own node (lane = node in the range searches).  A batch's loads are
    // six dword loads off three uniform bases with ONE 32-bit byte offset per lane, clamped to the cloud's last point (lanes behind it are
    // given a copy of lane 0's point below): no 64-bit address arithmetic and no divergent branch per batch.
    typedef const __attribute__((address_space(1))) char gbytes;
    gbytes *bx = (gbytes *)xs, *by = (gbytes *)(xs + ld), *bz = (gbytes *)(xs + 2 * ld);
    auto load2 = [&](int b, int last, f2 &X_, f2 &Y_, f2 &Z_) {
        const int n0 = b * kE2Points + lane, n1 = n0 + 64;
        const unsigned o0 = (unsigned)(n0 < last ? n0 : last) * 4u, o1 = (unsigned)(n1 < last ? n1 : last) * 4u;
        X_.x = *(const __attribute__((address_space(1))) float *)(bx + o0); Y_.x = *(const __attribute__((address_space(1))) float *)(by + o0); Z_.x = *(const __attribute__((address_space(1))) float *)(bz + o0);
        X_.y = *(const __attribute__((address_space(1))) float *)(bx + o1); Y_.y = *(const __attribute__((address_space(1))) float *)(by + o1); Z_.y = *(const __attribute__((address_space(1))) float *)(bz + o1);
    };
    f2 X, Y, Z;
    load2(batch, f.N0 - 1, X, Y, Z);                 // (N <= N0: in bounds, whatever the prune kept)
    const auto qg = TDLO_AS_GLOBAL(V4<float>, f.nodes);
    for (int m = tid; m < M; m += EB) { V4<float> o; o.x = qg[m].x; o.y = qg[m].y; o.z = qg[m].z; o.w = qg[m].w; nodesL[m] = o; }
    V4<float> qn; qn.x = 1e18f; qn.y = 1e18f; qn.z = 1e18f; qn.w = 3e38f;      // lanes behind the chain's end: a node nothing is near to, whose coordinate no window holds
    if (lane < M) { qn.x = qg[lane].x; qn.y = qg[lane].y; qn.z = qg[lane].z; qn.w = qg[lane].w; }
    if (done) return;
    double lv_span = 0;
    if (VIS) {
        // P_vis rows, :362-372: v_m = exp(-k_vis * dmin_m) / sum, folded into the exponent as log2 v_m (k_estep's own prologue)
        double tot = 0, dmx = 0, dmn = 1e300;
        for (int m = tid; m < M; m += EB) {
            double d = ::sqrt(Num<float>::from_bits(f.dminbits[m]));
            if (d > 10000.0) d = 10000.0;                        // initial value of :282
            if (d <= f.vis_thr) d = 0;                           // :291-293
            tot += ::exp(-f.k_vis * d);
            dmx = d > dmx ? d : dmx; dmn = d < dmn ? d : dmn;
        }
        tot = block_sum_n<NWE>(tot, scratch);
        dmx = wave_max_nonneg(dmx); dmn = wave_min_nonneg(dmn);
        __syncthreads();
        if (lane == 0) { scratch[wave] = dmx; scratch[NWE + wave] = dmn; }
        __syncthreads();
#pragma unroll
        for (int w_ = 0; w_ < NWE; ++w_) { dmx = scratch[w_] > dmx ? scratch[w_] : dmx; dmn = scratch[NWE + w_] < dmn ? scratch[NWE + w_] : dmn; }
        lv_span = f.k_vis * (dmx - dmn) * 1.4426950408889634;
        if (!(lv_span > 0)) lv_span = 0;
        for (int m = tid; m < M; m += EB) {
            double d = ::sqrt(Num<float>::from_bits(f.dminbits[m]));
            if (d > 10000.0) d = 10000.0;
            if (d <= f.vis_thr) d = 0;
            lvL[m] = (float)(-f.k_vis * d * 1.4426950408889634 - ::log2(tot));
        }
    }
    for (int i = tid; i < M * 4; i += EB) accL[i] = 0;
    __syncthreads();
    E2PHASE(0);

    // running sums in 64-bit fixed point (acc_fix at the grain of one wave x one 128-point batch; integer from there on: tdlo_devcommon.h)
    long long accQ = 0;
    const int shb = stg->sh_boost;
    const double scP = acc_scale(f.acc_sh[0]), scR = acc_scale(f.acc_sh[1] + shb), scQ = acc_scale(f.acc_sh[2] + 2 * shb);
    const double limP = f.acc_lim[0], limR = f.acc_lim[1] * acc_scale(-shb), limQ = f.acc_lim[2] * acc_scale(-2 * shb);
    // (a node's share of Q -- the column sums' tail -- is one conversion for up to a batch's points: held to the conversion's own exactness bound, 2^51 units)
    const double limQn = acc_scale(51 - (f.acc_sh[2] + 2 * shb));
    bool acc_ok = true;
    // node window (k_estep: E / |k2|, widened by what the visibility weights can take from a nearest node's membership)
    const float R2win = (float)(stg->rwin32 * (1.0 + (VIS ? lv_span / f.win_e32 : 0.0)));
    const float under_thr = -1075.f / k2;          // (k2 < 0) nearest-node distances beyond it: the reference's whole column has underflowed, :298-310

    const int nbatch = (N + kE2Points - 1) >> 7;
    for (; batch < nbatch; batch += bstride) {
        // ---- the next batch's points are requested now and waited for at the end of this one
        f2 Xn, Yn, Zn;
        load2(batch + bstride, N - 1, Xn, Yn, Zn);
        const int base = batch * kE2Points;
        const bool full = base + kE2Points <= N;                       // wave-uniform
        const float cx = bcast_first(X.x), cy = bcast_first(Y.x), cz = bcast_first(Z.x);       // lane 0's first point: always one of the cloud
        bool vA = true, vB = true;
        if (!full) {
            // the cloud's last batch: lanes without a point take a copy of lane 0's -- they widen no range and no window, and their weight is zeroed below
            vA = base + lane < N; vB = base + 64 + lane < N;
            X.x = vA ? X.x : cx; Y.x = vA ? Y.x : cy; Z.x = vA ? Z.x : cz;
            X.y = vB ? X.y : cx; Y.y = vB ? Y.y : cy; Z.y = vB ? Z.y : cz;
        }
        E2PHASE(1);
        // ---- nearest node: argmin of d2, first index (:298-310), over the exact candidate range of k_estep -- ONE range for the 128 points
        const f2 ex = X - sp(cx), ey = Y - sp(cy), ez = Z - sp(cz);       // (kept: the residual coordinates of the column sums)
        const f2 r2v = fma2(ez, ez, fma2(ey, ey, ex * ex));                // |x - o|^2: the candidate range's radius, and the points' own term of Q
        int plo = 0, phi = M - 1;
        {
            const float r2 = fmaxf(r2v.x, r2v.y);
            const float ddx = qn.x - cx, ddy = qn.y - cy, ddz = qn.z - cz;
            const float Dm = ddx * ddx + ddy * ddy + ddz * ddz;               // lane = node
            float r2w, dminw;
            wave_max_min_nonneg(r2, Dm, r2w, dminw);
            float lim = (Num<float>::sqrt_fast(dminw) + 2.f * Num<float>::sqrt_fast(r2w)) * 1.0001f + 1e-30f;
            lim = lim * lim;
            const unsigned long long cand = __ballot(Dm <= lim);
            if (cand) { plo = (int)__builtin_ctzll(cand); phi = 63 - (int)__builtin_clzll(cand); }
            plo = __builtin_amdgcn_readfirstlane(plo); phi = __builtin_amdgcn_readfirstlane(phi);
        }
        f2 best = sp(Num<float>::inf());
        int aA = plo, aB = plo;
        if (phi - plo < 16) {
            // up to 16 candidates (the rule): the argmin through ONE unsigned minimum per point and node -- a squared distance is a non-negative float, its bits
            // order like an unsigned integer, and the candidate's offset in the range rides in the four lowest mantissa bits (v_and_or_b32 with the offset as
            // an inline constant, v_min_u32) instead of a compare and two selects.  First index on ties, as before; the distance that comes out has lost four
            // bits (1e-6 relative: 5e-9 m on the pair's centimetres -- the coordinates themselves resolve 3e-8 m).
            unsigned kA = 0xffffffffu, kB = 0xffffffffu;
            auto cand1 = [&](float qx, float qy, float qz, auto OFF) {
                const f2 dx = X - sp(qx), dy = Y - sp(qy), dz = Z - sp(qz);
                const f2 d2 = fma2(dz, dz, fma2(dy, dy, dx * dx));
                const unsigned ka = (__float_as_uint(d2.x) & 0xfffffff0u) | (unsigned)decltype(OFF)::value, kb = (__float_as_uint(d2.y) & 0xfffffff0u) | (unsigned)decltype(OFF)::value;
                kA = ka < kA ? ka : kA; kB = kb < kB ? kb : kB;
            };
            auto group = [&](auto G) {                  // candidates plo + 4 G .. plo + 4 G + 3: one scalar load, evaluations beyond phi skipped wave-uniformly
                constexpr int g = decltype(G)::value;
                const int m0 = plo + 4 * g;
                const Node4<float> q4 = load_node4<float>(f.nodes, m0);
                cand1(q4.v[0], q4.v[1], q4.v[2], std::integral_constant<int, 4 * g>());
                if (m0 + 1 <= phi) cand1(q4.v[4], q4.v[5], q4.v[6], std::integral_constant<int, 4 * g + 1>());
                if (m0 + 2 <= phi) cand1(q4.v[8], q4.v[9], q4.v[10], std::integral_constant<int, 4 * g + 2>());
                if (m0 + 3 <= phi) cand1(q4.v[12], q4.v[13], q4.v[14], std::integral_constant<int, 4 * g + 3>());
            };
            group(std::integral_constant<int, 0>());
            if (plo + 4 <= phi) { group(std::integral_constant<int, 1>());
                if (plo + 8 <= phi) { group(std::integral_constant<int, 2>());
                    if (plo + 12 <= phi) group(std::integral_constant<int, 3>()); } }
            aA = plo + (int)(kA & 15u); aB = plo + (int)(kB & 15u);
            best.x = __uint_as_float(kA & 0xfffffff0u); best.y = __uint_as_float(kB & 0xfffffff0u);
        } else {
            auto cand1 = [&](float qx, float qy, float qz, int m) {
                const f2 dx = X - sp(qx), dy = Y - sp(qy), dz = Z - sp(qz);
                const f2 d2 = fma2(dz, dz, fma2(dy, dy, dx * dx));
                if (d2.x < best.x) { best.x = d2.x; aA = m; }
                if (d2.y < best.y) { best.y = d2.y; aB = m; }
            };
            int m0 = plo;
            for (; m0 + 3 <= phi; m0 += 4) {
                const Node4<float> q4 = load_node4<float>(f.nodes, m0);
#pragma unroll
                for (int k = 0; k < 4; ++k) cand1(q4.v[4 * k], q4.v[4 * k + 1], q4.v[4 * k + 2], m0 + k);
            }
            if (m0 <= phi) {
                const Node4<float> q4 = load_node4<float>(f.nodes, m0);
#pragma unroll
                for (int k = 0; k < 4; ++k) if (m0 + k <= phi) cand1(q4.v[4 * k], q4.v[4 * k + 1], q4.v[4 * k + 2], m0 + k);
            }
        }
        // all-underflow columns take node 0 (:298-310; k_estep has the derivation): rare, looked at by the wave together
        if (__builtin_expect(__ballot(fmaxf(best.x, best.y) > under_thr) != 0ull, 0)) {
            const V4<float> q0 = nodesL[0];
            const f2 dx0 = X - sp(q0.x), dy0 = Y - sp(q0.y), dz0 = Z - sp(q0.z);
            const f2 d0 = dx0 * dx0 + dy0 * dy0 + dz0 * dz0;
            if (best.x * k2 < -1075.f) { aA = 0; best.x = d0.x; }
            if (best.y * k2 < -1075.f) { aB = 0; best.y = d0.y; }
        }
        E2PHASE(2);
        // ---- second node by distance (:313-329), per point
        int loA, hiA, loB, hiB;
        f2 CLO, DLO, CHI, DHI;
        {
            auto second = [&](float x, float y, float z, int a, float bst, int &lo, int &hi, float &c_lo, float &d_lo, float &c_hi, float &d_hi) {
                const int c1 = (a == 0) ? 2 : a - 1;
                const int c2 = (a == M - 1) ? M - 3 : a + 1;
                const V4<float> q1 = nodesL[c1], q2 = nodesL[c2];
                const float caw = nodesL[a].w;
                float dx = x - q1.x, dy = y - q1.y, dz = z - q1.z;
                const float s1 = dx * dx + dy * dy + dz * dz;
                dx = x - q2.x; dy = y - q2.y; dz = z - q2.z;
                const float s2 = dx * dx + dy * dy + dz * dz;
                const bool first = s1 < s2;                                   // the decision of :324 on the squares
                const float eb = Num<float>::sqrt_fast(first ? s1 : s2);
                const int b = first ? c1 : c2;
                const float cb = first ? q1.w : q2.w;
                const float ea = Num<float>::sqrt_fast(bst);
                const bool a_lo = a < b;
                lo = a_lo ? a : b; hi = a_lo ? b : a;
                d_lo = a_lo ? ea : eb; d_hi = a_lo ? eb : ea;
                c_lo = a_lo ? caw : cb; c_hi = a_lo ? cb : caw;
            };
            float cl, dl, ch, dh;
            second(X.x, Y.x, Z.x, aA, best.x, loA, hiA, cl, dl, ch, dh); CLO.x = cl; DLO.x = dl; CHI.x = ch; DHI.x = dh;
            second(X.y, Y.y, Z.y, aB, best.y, loB, hiB, cl, dl, ch, dh); CLO.y = cl; DLO.y = dl; CHI.y = ch; DHI.y = dh;
        }
        // ---- node window of this wave: the index range of the nearest pairs, the largest nearest-node distance and "some point has the end-node
        //      gap" in ONE folded butterfly; the pairs' coordinates from the LDS copy (coord is non-decreasing: the smallest lo has the smallest c_lo)
        int wlo = 0, whi = M - 1, min_lo, max_hi;
        bool adj;
        {
            const int lmin = loA < loB ? loA : loB, hmax = hiA > hiB ? hiA : hiB;
            const unsigned gap = (unsigned)((hiA - loA != 1) | (hiB - loB != 1));
            const unsigned z = rows4_max((unsigned)(63 - lmin), (unsigned)hmax, __float_as_uint(fmaxf(best.x, best.y)), gap);
            min_lo = 63 - __builtin_amdgcn_readlane((int)z, 15);
            const float bmx = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)z, 31));
            max_hi = __builtin_amdgcn_readlane((int)z, 47);
            adj = __builtin_amdgcn_readlane((int)z, 63) == 0;
            // (lane = node holds every node's coordinate: a v_readlane is the broadcast -- no scalar load, no LDS round trip in the batch's dependency chain)
            const float amin = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(qn.w), min_lo));
            const float amax = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(qn.w), max_hi));
            const float Rwin = Num<float>::sqrt_fast(bmx + R2win);
            const unsigned long long inw = __ballot(qn.w > amin - Rwin && qn.w < amax + Rwin);
            if (inw) { wlo = (int)__builtin_ctzll(inw); whi = 63 - (int)__builtin_clzll(inw); }
            wlo = __builtin_amdgcn_readfirstlane(wlo); whi = __builtin_amdgcn_readfirstlane(whi);
        }
        E2PHASE(3);
        // ---- unnormalised memberships and their sum over the nodes (:354-383); the first TR nodes of the window go to the tile.
        // Q = sum_n sum_m P_mn |x_n - y_m|^2 is NOT accumulated pair by pair (k_estep does: a squared distance and a multiply-add per pair): with the
        // wave's origin o, |x - y|^2 = |x - o|^2 + 2 (o - y).(x - o) + |o - y|^2, so the batch's share is
        //     sum_n Pt1_n |x_n - o|^2  +  sum_m sum_k d_mk (s_mk + R_mk),     d = o - y_m,  s = sum_n P_mn (x_n - o),  R = s + d P1_m
        // -- the first term per point from what is at hand anyway, the second in the column sums' fixed-point tail (fp64) from the very sums it
        // converts.  The three parts are of the size of Q itself (a batch's points and its window's nodes lie within centimetres of o, like sigma):
        // no cancellation to speak of.
        f2 sum = {0.f, 0.f};
        // t of a node for both points; MODE 0: every point has the node at or below its lower pair node (m <= min lo), 1: at or above its
        // upper one (m >= max hi), 2: per point (adjacent pairs), 3: per point, a pair with the end-node gap in the wave (:332-350)
        auto geo_t = [&](float qw, int m, auto MODE) -> f2 {
            constexpr int mode = decltype(MODE)::value;
            f2 t;
            if constexpr (mode == 0) t = (CLO - sp(qw)) + DLO;
            else if constexpr (mode == 1) t = (sp(qw) - CHI) + DHI;
            else if constexpr (mode == 2) {
                const bool sA = m <= loA, sB = m <= loB;
                f2 c, d;
                c.x = sA ? CLO.x : CHI.x; d.x = sA ? DLO.x : DHI.x;
                c.y = sB ? CLO.y : CHI.y; d.y = sB ? DLO.y : DHI.y;
                const f2 u = sp(qw) - c;
                t.x = __builtin_fabsf(u.x) + d.x; t.y = __builtin_fabsf(u.y) + d.y;
            } else {
                const f2 tl = (CLO - sp(qw)) + DLO, th = (sp(qw) - CHI) + DHI;
                t.x = (m <= loA) ? tl.x : 0.f; t.x = (m >= hiA) ? th.x : t.x;
                t.y = (m <= loB) ? tl.y : 0.f; t.y = (m >= hiB) ? th.y : t.y;
            }
            return t;
        };
        // (adjacent pairs: the nodes up to the wave's smallest lo and from its largest hi on need no per-point decision -- wave-uniform branches)
        auto geo_any = [&](float qw, int m) -> f2 {
            if (!adj) return geo_t(qw, m, std::integral_constant<int, 3>());
            if (m <= min_lo) return geo_t(qw, m, std::integral_constant<int, 0>());
            if (m >= max_hi) return geo_t(qw, m, std::integral_constant<int, 1>());
            return geo_t(qw, m, std::integral_constant<int, 2>());
        };
        auto member = [&](float qw, int m, auto STORE) {
            const f2 t = geo_any(qw, m);
            f2 e = (t * t) * sp(k2);
            if (VIS) e += sp(lvL[m]);
            f2 p; p.x = Num<float>::exp2(e.x); p.y = Num<float>::exp2(e.y);
            sum += p;
            if (decltype(STORE)::value) { float *row = pb + (m - wlo) * kP2Stride + lane; row[0] = p.x; row[64] = p.y; }
        };
        // [from, to]: a node's coordinate is broadcast from its lane (qn.w) -- one v_readlane per node and nothing to wait for
        auto span = [&](int from, int to, auto STORE) {
            for (int m = from; m <= to; ++m)
                member(__uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(qn.w), m)), m, STORE);
        };
        {
            const int wst = (wlo + TR - 1) < whi ? (wlo + TR - 1) : whi;              // last node whose membership is stored
            span(wlo, wst, std::true_type());
            span(wst + 1, whi, std::false_type());
        }
        E2PHASE(4);
        f2 inv;
        {
            const f2 den = sum + sp(cn);
            inv.x = Num<float>::rcp_fast(den.x); inv.y = Num<float>::rcp_fast(den.y);
            if (!full) { inv.x = vA ? inv.x : 0.f; inv.y = vB ? inv.y : 0.f; }
            const f2 qvv = (inv * sum) * r2v;                                   // Pt1_n |x_n - o|^2
            const double qv = (double)(qvv.x + qvv.y);
            acc_ok &= __builtin_fabs(qv) < limQ; accQ += acc_fix(qv, scQ);
            // the normalised points relative to the wave's origin (lane 0's first point), four arrays of 128: column = point
            const f2 wx = inv * ex, wy = inv * ey, wz = inv * ez;
            float *o = pw + lane;
            o[0] = inv.x; o[64] = inv.y; o[128] = wx.x; o[192] = wx.y; o[256] = wy.x; o[320] = wy.y; o[384] = wz.x; o[448] = wz.y;
        }
        {
            // ---- column sums (:386-389): lane = (node of the window, slice of the 128 points), the window in chunks of TR nodes; the memberships of
            //      the chunks behind the first are recomputed (same expression, same bits) instead of being kept in a taller tile
            const int Wtot = whi - wlo + 1;
            for (int c0 = 0; c0 < Wtot; c0 += TR) {
                const int Wn = (Wtot - c0) < TR ? (Wtot - c0) : TR;
                const int wlo_c = wlo + c0;
                if (c0 > 0) {
                    for (int m = wlo_c; m < wlo_c + Wn; ++m) {
                        const float qw = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(qn.w), m));
                        const f2 t = geo_any(qw, m);
                        f2 e = (t * t) * sp(k2);
                        if (VIS) e += sp(lvL[m]);
                        float *row = pb + (m - wlo_c) * kP2Stride + lane;
                        row[0] = Num<float>::exp2(e.x); row[64] = Num<float>::exp2(e.y);
                    }
                }
                wave_lds_sync();
                // up to 8 nodes: 8 slices of 16 points; up to 16: 4 slices of 32.  A slice's points are 4-point groups g4 (column 4 g4) picked so that
                // the 16 lanes a ds_read_b128 serves at a time (2 slices x 8 rows, or 1 slice x 16 rows) fall on 64 different banks.
                const int shift = (TR <= 8 || Wn <= 8) ? 3 : 4;              // wave-uniform
                const int wl = lane & ((1 << shift) - 1), sl = lane >> shift;
                f2 a0 = {0.f, 0.f}, ax = {0.f, 0.f}, ay = {0.f, 0.f}, az = {0.f, 0.f};
                if (wl < Wn) {
                    const float *prow = pb + wl * kP2Stride;
                    auto four = [&](int g4) {
                        const f4 p4 = *(const f4 *)(prow + 4 * g4);
                        const f4 w0 = *(const f4 *)(pw + 4 * g4), wx = *(const f4 *)(pw + 128 + 4 * g4), wy = *(const f4 *)(pw + 256 + 4 * g4), wz = *(const f4 *)(pw + 384 + 4 * g4);
                        const f2 pl = {p4.x, p4.y}, ph = {p4.z, p4.w};
                        a0 = fma2(pl, (f2){w0.x, w0.y}, a0); a0 = fma2(ph, (f2){w0.z, w0.w}, a0);
                        ax = fma2(pl, (f2){wx.x, wx.y}, ax); ax = fma2(ph, (f2){wx.z, wx.w}, ax);
                        ay = fma2(pl, (f2){wy.x, wy.y}, ay); ay = fma2(ph, (f2){wy.z, wy.w}, ay);
                        az = fma2(pl, (f2){wz.x, wz.y}, az); az = fma2(ph, (f2){wz.z, wz.w}, az);
                    };
                    if (shift == 3) {
                        const int gb = ((sl & 1) << 3) | (sl >> 1);
#pragma unroll
                        for (int i = 0; i < 4; ++i) four(gb | ((i & 1) << 4) | ((i >> 1) << 2));
                    } else {
#pragma unroll
                        for (int i = 0; i < 8; ++i) four(sl * 8 + i);
                    }
                }
                float s0 = a0.x + a0.y, sx = ax.x + ax.y, sy = ay.x + ay.y, sz = az.x + az.y;
                // ---- the slices' partial sums folded together and the fixed-point tail with ONE value per lane (k_estep's scheme: row 0 converts Rx,
                //      row 1 Rz, row 2 Ry, row 3 P1).   R_k = s_k + (o_k - y_k) w0;  P1 = w0.
                s0 = fold32(s0, s0);
                float u = fold32(sx, sy), v = fold32(sz, sz);
                s0 = fold16(s0, s0); u = fold16(u, v);
                if (shift == 3) { s0 += row_ror8(s0); u += row_ror8(u); }
                {
                    typedef __attribute__((address_space(3))) long long lds_i64;
                    const V4<float> ym = nodesL[wlo_c + (wl < Wn ? wl : 0)];
                    lds_i64 *acn = (lds_i64 *)(accL + (size_t)(wlo_c + wl) * 4);
                    const double w0 = (double)s0;
                    const int g = lane >> 4;                                              // 0 Rx, 1 Rz, 2 Ry, 3 P1
                    const bool mine = wl < Wn && (shift != 3 || (lane & 8) == 0);         // (8-lane slices: both halves of a row hold the totals, the lower one converts)
                    const bool gx = g == 0, gy = g == 2, gp = g == 3;
                    const float ok = gx ? cx : (gy ? cy : cz), yk = gx ? ym.x : (gy ? ym.y : ym.z);
                    const int k = gx ? 1 : (gy ? 2 : (gp ? 0 : 3));
                    const double d = gp ? 1.0 : (double)ok - (double)yk, a = gp ? 0.0 : (double)u;
                    const double val = ::fma(d, w0, a);
                    acc_ok &= !mine || __builtin_fabs(val) < (gp ? limP : limR);
                    if (mine) __hip_atomic_fetch_add(acn + k, acc_fix(val, gp ? scP : scR), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    {   // the nodes' part of Q: d (s + R) per (node, coordinate) -- the P1 lanes have d = 1, a = 0 and add nothing of it
                        const double dq = (mine && !gp) ? d * (a + val) : 0.0;
                        acc_ok &= __builtin_fabs(dq) < limQn; accQ += acc_fix(dq, scQ);
                    }
                    wave_lds_sync();
                }
            }
        }
        E2PHASE(5);
        X = Xn; Y = Yn; Z = Zn;
        E2PHASE(6);
    }

    // ---- the workgroup's share into the accumulators of this iteration's parity, replica row = workgroup % kAccRows (integer atomics)
    long long *iscr = (long long *)scratch;
    {
        const long long qw = wave_sum_i64(accQ);
        if (lane == 0) iscr[wave] = qw;
    }
    if (__ballot(!acc_ok) != 0ull && lane == 0) {
        // a contribution beyond the fixed point's range, or not a number: the registration ends here with an error (k_estep's rule)
        IterState *sw = f.st;
        sw->status = TDLO_E_NUMERIC; sw->converged = 0; sw->done = 1;
    }
    __syncthreads();
    long long *arow = f.acc + ((size_t)(TDLO_AS_GLOBAL(IterState, f.st)->it & 1) * kAccRows + (chunk & (acc_rows_used(f) - 1))) * acc_stride(M);
    for (int i = tid; i < 4 * M; i += EB) acc_add(arow, (i & 3) * M + (i >> 2), accL[i]);
    if (tid == 0) {
        long long q = 0;
#pragma unroll
        for (int w = 0; w < NWE; ++w) q += iscr[w];
        acc_add(arow, 4 * M, q);
    }
    E2PHASE(7);
#ifdef TDLO_ESTEP_PHASES
    if (tid == 0 && chunk == f.nblkE / 2) { for (int i = 0; i < 10; ++i) f.dbg[48 + i] = ph_acc[i]; }
#endif
}

template <bool VIS, int TR>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5, 5))) void k_estep2(const FrameDev *__restrict__ frames) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const FrameDev &f = frames[blockIdx.y];
    if ((int)blockIdx.x >= f.nblkE) return;
    estep2_chunk<VIS, TR>(f, (int)blockIdx.x, smem);
}

// ------------------------------------------------------------------------------------------------------------------------------------------
// A batch's whole EM loop in ONE launch (round 6 EXPERIMENT, off by default: TDLO_BATCH_PERSIST=1; measured 5.1 ms against 1.27 ms per C3 call -- DESIGN.md 3.2c
// says why: a frame's own chain chunk -> M-step -> chunk is longer than an iteration of all frames' E-steps, the M-step's body as a called function spills, and
// every hand-over is an agent-scope atomic plus a cache invalidate / write-back).  With a launch per E-step and per M-step a batch's stream groups leave the GPU without an E-step for a
// third of the time: a frame's M-step is one wave for 8 us, and whatever the host enqueues, kernels of one stream wait for each other as
// wholes.  Here the dependency is per FRAME: the loop's work is the sequence of tickets (iteration, frame, chunk) -- chunk = one workgroup's share of the
// frame's E-step --, the resident workgroups draw them in order; the workgroup that completes the last chunk of (iteration, frame) runs that frame's M-step
// (the chain smoother's body, tdlo_mstep_chain_body.h) and raises mdone[frame]; a workgroup that draws a chunk of iteration k + 1 first makes sure the
// frame's M-step k is through (it nearly always is: a frame's chunks come round once per iteration, its M-step takes half of one).  So other frames'
// E-steps fill the GPU while a frame's M-step runs, with no host in the loop and no kernel boundary between iterations.
// Deadlock-free: a ticket's holder waits only for the M-step of an EARLIER ticket's frame, whose runner holds an earlier ticket and is running.  Every
// wait is bounded (2 s): a workgroup that gives up marks its frame TDLO_E_EXCHANGE and raises `abort`, everybody leaves at their next ticket, and the host
// repeats the call on the launch-per-step loop (run_frames).  What the kernels compute is what k_estep2 and k_mstep_chain compute, in the same order per
// frame: the same bits (tests/test_estep2_gpu.py).
// ctl: [1] abort; then per frame a 4 KB block with mdone (M-steps completed), then per frame a 4 KB block with edone (chunks reported); zeroed per call.
constexpr int kBatchCtlStride = 1024;      // words between two frames' counters (4 KB)
// (the two bodies as functions of their own: inlined side by side the register allocator held both bodies' values at once -- 128 VGPRs and a hundred spills.
//  Each refers to the launch's dynamic LDS by its own declaration, so that its accesses stay LDS accesses.)
template <int TR>
__device__ __noinline__ void batch_estep(const FrameDev *f, int chunk) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    estep2_chunk<false, TR>(*f, chunk, smem);
}
__device__ __noinline__ void batch_mstep(const FrameDev *f) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    mstep_chain_run<float, false, false, false, false>(*f, 0, smem);
}
template <int TR>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_batch_loop(const FrameDev *__restrict__ frames, unsigned *__restrict__ ctl, int F, int nblk, int iters, int edone_off) {
    __shared__ unsigned s_tk[4];          // [0] this round's ticket, [1] the frame's mdone as prefetched, [2] abort as seen, [3] answers of thread 0
    const int tid = threadIdx.x;
    const unsigned per_it = (unsigned)F * (unsigned)nblk, total = per_it * (unsigned)iters;
    // (a frame's two counters in 4 KB blocks of their own: agent-scope atomics are performed where the XCDs meet, ~100 ns each and one after the other per memory
    //  channel -- with all frames' counters in one cache line the 78 400 chunk reports of a call took 5.5 ms)
    unsigned *mdone = ctl + kBatchCtlStride, *edone = ctl + edone_off;
    // (tickets are dealt round-robin, workgroup w takes w, w + G, w + 2 G, ...: drawn from ONE counter they cost 100 ns each -- agent-scope atomics on a single
    //  address are serialised where the XCDs meet -- 8 ms for a call's 78 400.  Every workgroup still takes its tickets in increasing order, which is all the
    //  deadlock argument needs.)
    unsigned nx_ticket = blockIdx.x, nx_md = 0;
    if (tid == 0 && nx_ticket < total) nx_md = __hip_atomic_load(mdone + (size_t)((nx_ticket % per_it) / (unsigned)nblk) * kBatchCtlStride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (;;) {
        if (tid == 0) { s_tk[0] = nx_ticket; s_tk[1] = nx_md; s_tk[2] = __hip_atomic_load(ctl + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
        __syncthreads();
        const unsigned ticket = s_tk[0];
        const bool stop = s_tk[2] != 0u;
        unsigned md = s_tk[1];
        __syncthreads();
        if (ticket >= total || stop) break;
        const unsigned it = ticket / per_it, r = ticket - it * per_it;
        const int fi = (int)(r / (unsigned)nblk), c = (int)(r - (unsigned)fi * (unsigned)nblk);
        const FrameDev &f = frames[fi];
        // the next ticket and its frame's progress are requested now and looked at when this chunk is done
        nx_ticket = ticket + gridDim.x;
        if (tid == 0) nx_md = nx_ticket < total ? __hip_atomic_load(mdone + (size_t)((nx_ticket % per_it) / (unsigned)nblk) * kBatchCtlStride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
        // the frame's M-step of the iteration before must be through (its nodes, its state, the cleared accumulator rows)
        if (md < it) {
            if (tid == 0) {
                const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
                unsigned v;
                while ((v = __hip_atomic_load(mdone + (size_t)fi * kBatchCtlStride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < it) {
                    if (__hip_atomic_load(ctl + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;
                    __builtin_amdgcn_s_sleep(2);
                    if (__builtin_amdgcn_s_memrealtime() - t0 > 200000000ull) break;       // 2 s of the 100 MHz clock
                }
                s_tk[3] = v;
            }
            __syncthreads();
            md = s_tk[3];
            __syncthreads();
            if (md < it) {          // gave up (or told to): the frame's registration ends with an error, everybody leaves at their next ticket
                if (tid == 0) {
                    IterState *sw = f.st; sw->status = TDLO_E_EXCHANGE; sw->converged = 0; sw->done = 1;
                    __hip_atomic_store(ctl + 1, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                }
                break;
            }
        }
        // (cache maintenance by ONE wave per workgroup: an agent-scope acquire / release is an invalidate / write-back of the XCD's whole L2 -- issued by every wave of
        //  every ticket they took 20 ms per call where the loop's work is 1; the CU's vector cache and the scalar cache are shared by the workgroup's waves, and
        //  the other waves' own stores have reached the L2 when their workgroup-scope release -- a wait for their memory counters -- is through)
        if (tid < 64) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); __builtin_amdgcn_s_dcache_inv(); }      // (the nodes come through scalar loads)
        __syncthreads();
        if (c < f.nblkE) batch_estep<TR>(&f, c);
        // report the chunk; the workgroup that completes the frame's E-step runs its M-step
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");      // this thread's atomics on the accumulators have been performed
        __syncthreads();
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            s_tk[3] = __hip_atomic_fetch_add(edone + (size_t)fi * kBatchCtlStride, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u == (it + 1u) * (unsigned)nblk ? 1u : 0u;
        }
        __syncthreads();
        const bool last = s_tk[3] != 0u;
        __syncthreads();
        if (last) {
            if (tid < 64) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // every workgroup's sums
            __syncthreads();
            __builtin_amdgcn_s_setprio(3);
            batch_mstep(&f);
            __builtin_amdgcn_s_setprio(0);
            __syncthreads();                                    // every thread's stores to the nodes, Y and the state have been performed
            if (tid == 0) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); __hip_atomic_store(mdone + (size_t)fi * kBatchCtlStride, it + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); }
        }
    }
}

size_t estep2_lds_bytes(int M, int tr) {
    return sizeof(V4<float>) * (size_t)M + sizeof(float) * (size_t)((M + 3) & ~3) + sizeof(float) * 4 * 4 * kE2Points + sizeof(float) * (size_t)4 * tr * kP2Stride +
           16 * sizeof(double) + sizeof(long long) * (size_t)M * 4;
}

// (ev_start / ev_stop: measurement aid of tdlo_device.hip -- events bound to the dispatch itself)
hipError_t launch_estep2(const FrameDev *fd, const FrameDev *fh, int F, hipStream_t s, hipEvent_t ev_start, hipEvent_t ev_stop) {
    const int M = fh[0].M, tr = fh[0].estep2;
    const bool vis = fh[0].vis_branch != 0;
    int gx = 0;
    for (int i = 0; i < F; ++i) gx = fh[i].nblkE > gx ? fh[i].nblkE : gx;
    const dim3 grid(gx, F), block(256);
    const size_t lds = estep2_lds_bytes(M, tr);
#define TDLO_E2L(VIS, TR) do { \
        if (lds > 64 * 1024) { const hipError_t e_ = hipFuncSetAttribute((const void *)k_estep2<VIS, TR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); if (e_ != hipSuccess) return e_; } \
        if (ev_start) hipExtLaunchKernelGGL((k_estep2<VIS, TR>), grid, block, lds, s, ev_start, ev_stop, 0, fd); \
        else hipLaunchKernelGGL((k_estep2<VIS, TR>), grid, block, lds, s, fd); } while (0)
    if (tr == 8) { if (vis) TDLO_E2L(true, 8); else TDLO_E2L(false, 8); }
    else { if (vis) TDLO_E2L(true, 16); else TDLO_E2L(false, 16); }
#undef TDLO_E2L
    return hipGetLastError();
}

size_t batch_loop_ctl_words(int F) { return (size_t)kBatchCtlStride * (1 + 2 * (size_t)F); }

// the whole loop of a batch in one launch (k_batch_loop); ctl: batch_loop_ctl_words(F) zeroed words on the device
hipError_t launch_batch_loop(const FrameDev *fd, const FrameDev *fh, int F, int iters, unsigned *ctl, hipStream_t s) {
    const int M = fh[0].M, tr = fh[0].estep2;
    int nblk = 0;
    for (int i = 0; i < F; ++i) nblk = fh[i].nblkE > nblk ? fh[i].nblkE : nblk;
    const size_t lds = std::max(estep2_lds_bytes(M, tr), sizeof(double) * ChainCarve(M).total);
    const int edone_off = kBatchCtlStride * (1 + F);
    // four workgroups of 128 VGPRs per CU (the M-step's body needs them): every workgroup of the launch resident, or the tickets' order would not protect from deadlock
    int cus = 256;
    { int dev = 0; hipDeviceProp_t pr; if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0) cus = pr.multiProcessorCount; }
    const long long want = (long long)F * nblk;
    const int grid = (int)std::min<long long>(want, 4LL * cus);
    hipError_t e;
#define TDLO_BL(TR) do { \
        if (lds > 64 * 1024) { e = hipFuncSetAttribute((const void *)k_batch_loop<TR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); if (e != hipSuccess) return e; } \
        hipLaunchKernelGGL((k_batch_loop<TR>), dim3(grid), dim3(256), lds, s, fd, ctl, F, nblk, iters, edone_off); } while (0)
    if (tr == 8) TDLO_BL(8); else TDLO_BL(16);
#undef TDLO_BL
    return hipGetLastError();
}

}  // namespace tdlo
