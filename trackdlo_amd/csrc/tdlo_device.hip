// tdlo_device.hip -- gfx950 (CDNA4, wave64) kernels for TrackDLO's EM registration loop.
//
// Reference path: trackdlo/src/trackdlo.cpp:161-441 (trackdlo::cpd_lle).  Nothing here is a
// translation of that code: the reference materialises five dense M x N fp64 matrices per
// iteration on one CPU thread; here no M x N matrix ever exists.
//
// Kernels (one EM iteration = [k_dmin] -> k_estep -> M-step, all on one stream, no host sync):
//   k_prune_pass1 / k_setup / k_prune_scatter   once per call   (:177-273)
//   k_dmin    per-node minimum distance to the cloud (visibility weighting only, :278-296, :358-372)
//   k_estep   fused distances + nearest-node + geodesic membership + normalisation + column sums
//             (:278-389): thread = point, nodes via scalar loads, 64x64 lane-transposed LDS tile so
//             that lane = node accumulates P1 / PX without cross-lane reductions
//             -- the sums leave the kernel as 64-bit fixed-point integers added with atomics (tdlo_devcommon.h: acc_*)
//   k_mstep*  the M-step (:392-437): without the LLE term the chain smoother of tdlo_mstep_chain.hip (O(M)); with it -- and as
//             comparators -- the dense eliminations here (k_mstep_fast), in tdlo_mstep_generic.h and tdlo_mstep_big.hip
//
// Numerics: all device geometry lives in a frame centred on the centroid of the incoming nodes
// (every formula of the path is translation invariant).  In TDLO_PREC_F32 the E-step works in fp32
// on fp32-rounded centred coordinates; sums leave each 64-point tile in fp32, are converted to fixed point per
// wave and batch, and are integers from there on (exact, order-independent).  sigma2 uses the algebraically identical residual form
//   sum_mn P_mn |x_n - T_m|^2 = Q - 2 sum_m d_m.R_m + sum_m P1_m |d_m|^2,
//   Q = sum_mn P_mn |x_n - y_m|^2,  R_m = PX_m - P1_m y_m (accumulated directly by the E-step),  d_m = T_m - y_m
// instead of :418-422's difference of three large traces, which would cancel catastrophically in fp32.
#include "tdlo_devcommon.h"
#include "tdlo_estep_wide.h"
#include "tdlo_mstep_generic.h"
#include <hip/hip_ext.h>
#include <type_traits>
#include <cstdio>
#include <cstdlib>

namespace tdlo {

typedef double dbl2 __attribute__((ext_vector_type(2)));

// ------------------------------------------------------------------------------------------------
// prune, trackdlo.cpp:177-195, fused with the sigma2 initialisation sum of :263-273
// ------------------------------------------------------------------------------------------------
// One point against the M nodes staged in LDS (Yl: x | y | z, M each, kPrunePad doubles readable behind them): the smallest |y_m - x|^2 with the
// FIRST index that attains it (strict <, as the reference's loop keeps the first minimum), and the sum of all M values in node order (:263-273).
// Per node the compiler had made 3 subtractions, 3 multiply-adds, a compare, THREE selects (a 64-bit select is two v_cndmask_b32) and the sum's
// add, with the next compare waiting for the selected minimum, and it waited for every group's LDS reads right after requesting them.  Here the
// minimum is one v_min_f64 (a NaN or an infinite value leaves it alone exactly as the failed compare did), compare and index select hang off the
// OLD minimum (loop-carried chain: one instruction), and a group's coordinates are requested one group ahead (no index clamp: the reads behind
// the chain's end land in the pad and are never used).  Same values, same order of the sum: the bits of before.
constexpr int kPrunePad = 4;
__device__ __forceinline__ void nearest_node(const double *Yl, int M, double x, double y, double z, double &best_o, int &a0_o, double &sum_o) {
    double best = 1e300, sum = 0;
    int a0 = 0;
    double nx[4], ny[4], nz[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { nx[k] = Yl[k]; ny[k] = Yl[M + k]; nz[k] = Yl[2 * M + k]; }
    int m0 = 0;
    for (; m0 + 4 <= M; m0 += 4) {
        double qx[4], qy[4], qz[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { qx[k] = nx[k]; qy[k] = ny[k]; qz[k] = nz[k]; }
#pragma unroll
        for (int k = 0; k < 4; ++k) { nx[k] = Yl[m0 + 4 + k]; ny[k] = Yl[M + m0 + 4 + k]; nz[k] = Yl[2 * M + m0 + 4 + k]; }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const double dx = qx[k] - x, dy = qy[k] - y, dz = qz[k] - z;
            const double d2 = dx * dx + dy * dy + dz * dz;
            a0 = d2 < best ? m0 + k : a0;
            best = __builtin_fmin(best, d2);
            sum += d2;
        }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        if (m0 + k < M) {                               // (wave-uniform)
            const double dx = nx[k] - x, dy = ny[k] - y, dz = nz[k] - z;
            const double d2 = dx * dx + dy * dy + dz * dz;
            a0 = d2 < best ? m0 + k : a0;
            best = __builtin_fmin(best, d2);
            sum += d2;
        }
    }
    best_o = best; a0_o = a0; sum_o = sum;
}

__global__ __launch_bounds__(kBlock) void k_prune_pass1(const FrameDev *__restrict__ frames) {
    const FrameDev &f = frames[blockIdx.y];
    if ((int)blockIdx.x >= f.nprune_blocks) return;
    __shared__ double scratch[4];
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int *lh = (int *)smem;                            // M: kept points of this block per nearest node
    const int N0 = f.N0, M = f.M;
    double *Yl = (double *)(lh + ((M + 3) & ~3));     // 3M: the nodes, staged once per block (a scalar load per node and
    for (int m = threadIdx.x; m < M; m += kBlock) lh[m] = 0;                           // coordinate stalled every trip of the loop)
    for (int i = threadIdx.x; i < 3 * M; i += kBlock) Yl[i] = f.Yin[i];
    __syncthreads();
    double ssum = 0;
    for (int tt = 0; tt < f.prune_tiles; ++tt) {          // consecutive 256-point tiles of this workgroup
        const int n = (blockIdx.x * f.prune_tiles + tt) * kBlock + threadIdx.x;
        const bool valid = n < N0;
        double x = 0, y = 0, z = 0;
        if (valid) { x = f.Xraw[n]; y = f.Xraw[(size_t)N0 + n]; z = f.Xraw[2 * (size_t)N0 + n]; }
        double best, sum;
        int a0;
        nearest_node(Yl, M, x, y, z, best, a0, sum);
        const bool keep = valid && (::sqrt(best) < 0.1);
        // the kept points are stored sorted by their nearest node (stable), which makes the points of a
        // wave spatially coherent: the E-step then only touches the few nodes with non-zero membership
        if (valid) f.bucket[n] = keep ? (unsigned short)a0 : (unsigned short)0xffff;
        if (keep) atomicAdd(&lh[a0], 1);
        ssum += block_sum(keep ? sum : 0.0, scratch);     // per tile, tiles in order: one tile per workgroup gives the former sums
    }
    __syncthreads();
    for (int m = threadIdx.x; m < M; m += kBlock) f.hist[(size_t)blockIdx.x * M + m] = lh[m];
    if (threadIdx.x == 0) f.blksum[blockIdx.x] = ssum;
}

// ---- the fused prologue's grid barrier (k_prologue: up to 64 point workgroups + one node workgroup, all resident at once) ------------------
// The point workgroups publish their counts with agent-scope stores, wait for them to be performed (vmcnt) and take a ticket; the last one
// re-arms the counter and RELEASES the launch: the barrier word goes to (epoch << 1).  Everybody who needs the counts spins on the word
// (agent-scope loads) and reads them with agent-scope loads.  (A release fence per workgroup would be an L2 write-back each: k_dmin.)
// The wait is bounded (round 5; the reference's cpd_lle is straight-line CPU code and cannot hang, trackdlo.cpp:161-441): co-residency of the
// launch's workgroups is an assumption -- a GPU with masked or partitioned CUs, or one kept busy by other long-running kernels, may leave some of
// them unscheduled while the resident ones spin.  After kFuseSpinTicks (2 s, like every other in-kernel wait of the library) a waiting workgroup
// ABANDONS the launch: the word goes to (epoch << 1) | 1.  Releasing and abandoning are both compare-and-swaps from a value of ANOTHER epoch, so
// an epoch's outcome is decided exactly once and every workgroup of the launch -- those scheduled late included -- follows the same one; an
// abandoned launch ends its registration(s) with TDLO_E_FUSE (fused_fail), the M-step behind it reports that to the host, and the host repeats
// the call on the three-kernel route (tdlo_api.cpp, fuse_fallback).
constexpr int kFuseMaxBlocks = 64;       // point workgroups of the fused prologue: clouds of up to 16 384 points
constexpr int kFuseMaxNodes = 256;       // thread = node in its offset computation
constexpr unsigned long long kFuseSpinTicks = 200000000ull;      // s_memrealtime ticks (100 MHz): 2 s
constexpr unsigned kFuseWithhold = 0x80000000u;                  // test hook in the epoch argument (TDLO_FUSE_FORCE_TIMEOUT): the last point workgroup takes no ticket
// decides epoch's outcome unless it is decided already; returns the decided word
__device__ __forceinline__ unsigned fuse_decide(unsigned *word, unsigned epoch, unsigned want) {
    unsigned v = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while ((v >> 1) != epoch) {
        if (__hip_atomic_compare_exchange_strong(word, &v, want, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return want;
    }
    return v;
}
__device__ __forceinline__ void fuse_arrive(const FrameDev &f, unsigned nblk, unsigned epoch, bool withhold) {      // all threads of a point workgroup
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0 && !withhold) {
        const unsigned old = __hip_atomic_fetch_add(f.sync + 100, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old == nblk - 1u) {
            __hip_atomic_store(f.sync + 100, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            (void)fuse_decide(f.sync + 101, epoch, epoch << 1);
        }
    }
}
__device__ __forceinline__ bool fuse_wait(const FrameDev &f, unsigned epoch, int *ok_lds) {           // all threads of a workgroup; false: the launch was abandoned
    if (threadIdx.x == 0) {
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        unsigned v, spins = 0;
        while (((v = __hip_atomic_load(f.sync + 101, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >> 1) != epoch) {
            __builtin_amdgcn_s_sleep(1);
            if ((++spins & 31u) == 0u && __builtin_amdgcn_s_memrealtime() - t0 > kFuseSpinTicks) { v = fuse_decide(f.sync + 101, epoch, (epoch << 1) | 1u); break; }
        }
        *ok_lds = (int)((v & 1u) ^ 1u);
    }
    __syncthreads();
    return *ok_lds != 0;
}

// One workgroup per frame: scan of the prune counts, centring, chain coordinate + kernel G
// (:214-233), H*G / H*Y0 (:396-401), iteration-0 constants.
// host_up != nullptr (one frame per call): the host-supplied block [descriptor | Yin | aJ | aYd | H] is read straight from pinned host memory
// and copied to its place in the slot's node block (dev_up) by this workgroup -- no host-to-device copy in front of the kernel.
// FUSED: this is the node workgroup of k_prologue; the counts are scanned by the point workgroups themselves, the kept-point count and the
// sigma2 initialisation sum are formed at the end from what they published.
// MAXN: capacity of the static LDS arrays in nodes (the fused prologue serves chains of up to kFuseMaxNodes: its kernel also holds the point
// workgroups' arrays, and both must fit the 64 KB a kernel may declare statically).
template <typename T, bool FUSED, int MAXN = kMaxNodes>
__device__ __forceinline__ void setup_body(const FrameDev &f, int split_mode, const double *__restrict__ host_up, double *__restrict__ dev_up, int up_doubles,
                                           int yin_off, unsigned fuse_epoch) {
    IterState *st = f.st;
    __shared__ double sd[kBlock];
    __shared__ double sctr[3];
    __shared__ int sN;
    __shared__ double sS;
    const int t = threadIdx.x, M = f.M;
#ifdef TDLO_CHAIN_STAMPS      // phase stamps of the setup kernel (instrumented build only): f.dbg[16 ..]
#define SSTAMP(i) do { if (t == 0) f.dbg[16 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define SSTAMP(i) do { } while (0)
#endif
    SSTAMP(0);
    // counting sort by nearest node: hist[block][node] -> start offset of that (node, block) run.
    // 256 threads = 64 nodes x 4 chunks of blocks; per node an exclusive scan over the blocks in block order.
    const int nb = f.nprune_blocks;
    __shared__ double sY[3 * MAXN];
    __shared__ double sc[MAXN];
    if (host_up) {      // the nodes from pinned host memory, then the whole upload block to its place in device memory: every load of a trip is
        // requested before the first store (a load-store loop pays one PCIe round trip, ~2 us, per trip); 16 bytes per load
        // (the nodes are part of the block: they go to LDS out of the same registers -- ONE round trip over PCIe, ~1.7 us, not two)
        const dbl2 *src = (const dbl2 *)host_up;
        dbl2 *dst = (dbl2 *)dev_up;
        const int n2 = up_doubles >> 1;                          // (the block's parts are padded to 16 bytes)
        for (int i0 = 0; i0 < n2; i0 += 8 * kBlock) {
            dbl2 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int i = i0 + u * kBlock + t; v[u] = src[i < n2 ? i : n2 - 1]; }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + u * kBlock + t;
                if (i < n2) dst[i] = v[u];
                const int e = 2 * i - yin_off;                  // (yin_off is even: the parts start on 16-byte boundaries)
                if (i < n2 && e >= 0 && e < 3 * M) { sY[e] = v[u].x; if (e + 1 < 3 * M) sY[e + 1] = v[u].y; }
            }
        }
    } else {   // the node block, requested first: it arrives while the counts are scanned (the barriers of the scan cover it)
        const auto Yg = TDLO_AS_GLOBAL(double, f.Yin);
        for (int i = t; i < 3 * M; i += kBlock) sY[i] = Yg[i];
    }
    __shared__ int stot[MAXN];               // kept points per node -> first index of the node's run
    __shared__ int csum[MAXN / 64][4][64];   // kept points per (node, quarter of the prune blocks)
    const int ml = t & 63, ch = t >> 6;
    const int cb0 = (int)((long long)nb * ch / 4), cb1 = (int)((long long)nb * (ch + 1) / 4);
    const bool reuse = FUSED || (f.reuse_sorted != 0 && !split_mode);      // the sorted cloud of the previous registration serves (or the point workgroups of the fused prologue scan): no counts to scan here
    if (!reuse) {
    {
        // (global address space and 16 independent loads per trip: a load-add chain over ~50 blocks costs a memory latency each.  Chains beyond
        //  64 nodes: the loads of ALL node groups of a block chunk are requested together -- 782 blocks x 300 nodes at C5 took 5 groups x 13
        //  dependent trips per pass)
        const auto hg = TDLO_AS_GLOBAL_RW(int, f.hist);
        auto pass1 = [&](auto NGc, auto Uc) __attribute__((always_inline)) {
            constexpr int NG = decltype(NGc)::value, U = decltype(Uc)::value;
            int run[NG];
#pragma unroll
            for (int g = 0; g < NG; ++g) run[g] = 0;
            for (int b = cb0; b < cb1; b += U) {
                int v[NG][U];
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    const int m = 64 * g + ml, mc = m < M ? m : M - 1;
#pragma unroll
                    for (int u = 0; u < U; ++u) v[g][u] = hg[(size_t)(b + u < cb1 ? b + u : cb1 - 1) * M + mc];
                }
#pragma unroll
                for (int g = 0; g < NG; ++g)
#pragma unroll
                    for (int u = 0; u < U; ++u) if (b + u < cb1) run[g] += v[g][u];
            }
#pragma unroll
            for (int g = 0; g < NG; ++g) if (64 * g < M) csum[g][ch][ml] = (cb1 <= cb0 || 64 * g + ml >= M) ? 0 : run[g];
        };
        if (M <= 64) pass1(std::integral_constant<int, 1>(), std::integral_constant<int, 16>());
        else if (M <= 256) pass1(std::integral_constant<int, 4>(), std::integral_constant<int, 8>());
        else if constexpr (MAXN > 256) {
            if (M <= 512) pass1(std::integral_constant<int, 8>(), std::integral_constant<int, 8>());
            else pass1(std::integral_constant<int, MAXN / 64>(), std::integral_constant<int, 4>());
        }
    }
    SSTAMP(1);
    {
        const int per = (nb + kBlock - 1) / kBlock;
        const int b0 = min(nb, t * per), b1 = min(nb, b0 + per);
        double s = 0;
        for (int b = b0; b < b1; ++b) s += f.blksum[b];
        sd[t] = s;
    }
    }
    __syncthreads();
    // first index of every node's run: exclusive prefix sum of the per-node totals (integers: any order of the additions gives the same
    // result) -- two nodes per thread, a shuffle scan inside the waves, the waves' totals through LDS; and the sigma2 initialisation sum in a
    // FIXED order: every wave's 64 partial sums by the butterfly of wave_sum, the four waves' sums left to right.  (Round 2 had thread 0 walk the
    // M nodes and thread 64 add 256 numbers one after the other: 6000 clocks = 2.5 us of every call.)
    if (!reuse) {
        constexpr int NPT = MAXN > 2 * kBlock ? MAXN / kBlock : 2;          // consecutive nodes per thread (two; four for the 1024-node capacity)
        auto tot_of = [&](int m) __attribute__((always_inline)) { const int g = m >> 6, l = m & 63; return m < M ? (csum[g][0][l] + csum[g][1][l]) + (csum[g][2][l] + csum[g][3][l]) : 0; };
        int vq[NPT], vsum = 0;
#pragma unroll
        for (int q = 0; q < NPT; ++q) { vq[q] = tot_of(NPT * t + q); vsum += vq[q]; }
        int incl = vsum;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(incl, d); if ((t & 63) >= d) incl += o; }
        __shared__ int wtot[4];
        if ((t & 63) == 63) wtot[t >> 6] = incl;
        const double ssum = wave_sum(sd[t]);
        __shared__ double wsd[4];
        if ((t & 63) == 0) wsd[t >> 6] = ssum;
        __syncthreads();
        int base = 0;
        for (int w = 0; w < (t >> 6); ++w) base += wtot[w];
        int excl = base + incl - vsum;
#pragma unroll
        for (int q = 0; q < NPT; ++q) { if (NPT * t + q < M) stot[NPT * t + q] = excl; excl += vq[q]; }
        if (t == 0) { sN = wtot[0] + wtot[1] + wtot[2] + wtot[3]; sS = ((wsd[0] + wsd[1]) + wsd[2]) + wsd[3]; }
    }
    if (t == 0 && reuse && !FUSED) { sN = (int)f.keep[0]; sS = f.keep[1]; }
    __syncthreads();
    SSTAMP(2);
    if (!reuse) {
        // second pass over the counts: (node, block) count -> where that run starts = node's first index + the earlier quarters + the
        // earlier blocks of this quarter, written to hist_off (NOT in place: behind stores into the array it reads, every trip's loads
        // waited for the previous trip's stores -- 1.7 us per trip against 0.7 us in the first pass)
        const auto hg = TDLO_AS_GLOBAL(int, f.hist);
        const auto ho = TDLO_AS_GLOBAL_RW(int, f.hist_off);
        auto pass2 = [&](auto NGc, auto Uc) __attribute__((always_inline)) {
            constexpr int NG = decltype(NGc)::value, U = decltype(Uc)::value;
            int r2[NG];
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                const int m = 64 * g + ml;
                r2[g] = m < M ? stot[m] : 0;
                for (int c2 = 0; c2 < ch; ++c2) r2[g] += (64 * g < M) ? csum[g][c2][ml] : 0;
            }
            for (int b = cb0; b < cb1; b += U) {
                int v[NG][U];
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    const int m = 64 * g + ml, mc = m < M ? m : M - 1;
#pragma unroll
                    for (int u = 0; u < U; ++u) v[g][u] = hg[(size_t)(b + u < cb1 ? b + u : cb1 - 1) * M + mc];
                }
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    const int m = 64 * g + ml;
#pragma unroll
                    for (int u = 0; u < U; ++u) if (b + u < cb1 && m < M) { ho[(size_t)(b + u) * M + m] = r2[g]; r2[g] += v[g][u]; }
                }
            }
        };
        if (M <= 64) pass2(std::integral_constant<int, 1>(), std::integral_constant<int, 16>());
        else if (M <= 256) pass2(std::integral_constant<int, 4>(), std::integral_constant<int, 8>());
        else if constexpr (MAXN > 256) {
            if (M <= 512) pass2(std::integral_constant<int, 8>(), std::integral_constant<int, 8>());
            else pass2(std::integral_constant<int, MAXN / 64>(), std::integral_constant<int, 4>());
        }
    }
    SSTAMP(3);
    // centring offset, chain coordinate: the node block is staged in LDS first -- the serial sums below (same
    // left-to-right order as the reference) would otherwise pay a global-memory round trip per term
    if (t < 192) {          // centring offset = centroid of the nodes: wave w sums coordinate w (lane l the nodes l, l + 64, ...; then the butterfly of wave_sum: a fixed order)
        const int d = t >> 6, l = t & 63;
        double a = 0;
        for (int m = l; m < M; m += 64) a += sY[d * M + m];
        a = wave_sum(a);
        if (l == 0) { sctr[d] = a / M; f.ctr[d] = a / M; }
    }
    for (int i = t; i < M - 1; i += kBlock) {
        const double dx = sY[i + 1] - sY[i], dy = sY[M + i + 1] - sY[M + i], dz = sY[2 * M + i + 1] - sY[2 * M + i];
        sc[i + 1] = ::sqrt(dx * dx + dy * dy + dz * dz);
    }
    __syncthreads();
    if (t == 0) {   // same left-to-right running sum as :219-223
        double cur = 0; sc[0] = 0;
        for (int i = 0; i < M - 1; ++i) { cur += sc[i + 1]; sc[i + 1] = cur; }
    }
    SSTAMP(4);
    for (int i = t; i < 3 * M; i += kBlock) { const double v = sY[i] - sctr[i / M]; f.Y[i] = v; f.Y0[i] = v; f.Yout[i] = sY[i]; }
    __syncthreads();
    V4<T> *nodes = (V4<T> *)f.nodes;
    for (int m = t; m < M; m += kBlock) {
        V4<T> q; q.x = (T)(sY[m] - sctr[0]); q.y = (T)(sY[M + m] - sctr[1]); q.z = (T)(sY[2 * M + m] - sctr[2]); q.w = (T)sc[m];
        nodes[m] = q;
        f.coord[m] = sc[m];
        f.dminbits[m] = ~0ull;
    }
    for (int i = t; i < 2 * kAccRows * acc_stride(M); i += kBlock) f.acc[i] = 0;      // the E-step's accumulators, both parities
    const double beta = f.beta;
    // state-space form of G for the chain smoother (tdlo_mstep_chain.hip): one link per pair of consecutive nodes
    for (int i = t; i < M; i += kBlock) {
        double o[8];
        if (i == 0) { const double s = ::sqrt(2.0) / beta, sf2 = 1.0 / (2.0 * ::sqrt(2.0) * beta); o[0] = sf2; o[1] = s * s * sf2; o[2] = 1.0 / o[0]; o[3] = 1.0 / o[1]; o[4] = o[5] = o[6] = o[7] = 0.0; }      // Pinf and its inverse
        else chain_link(beta, sc[i] - sc[i - 1], o);
#pragma unroll
        for (int q = 0; q < 8; ++q) f.chain[8 * (size_t)i + q] = o[q];
    }
    SSTAMP(5);
    if (f.need_G) {         // the dense M-steps only (LLE term, comparators): the chain smoother works from the links above
        const auto Gw = TDLO_AS_GLOBAL_RW(double, f.G);
        for (int j = t >> 6; j < M; j += kBlock / 64)
            for (int i = t & 63; i < M; i += 64) {
                const double dd = fabs(sc[i] - sc[j]);
                Gw[(size_t)j * M + i] = 1.0 / (2 * beta * 2 * beta) * ::exp(-::sqrt(2.0) * dd / beta) * (2 * dd + ::sqrt(2.0) * beta);
            }
    }
    __syncthreads();
    if (f.include_lle && f.lle_band) {
        // Banded LLE M-step (tdlo_mstep_band.hip): one column record per step of each direction of the elimination (BandPlan), column j
        // of lambda K + lle_weight H in the direction's own order of the unknowns, rows j - 12 .. j, each at the position of its row's
        // slot (row mod 13).  K: joint precision of the states (f_i, f'_i) of the chain, block tridiagonal: diagonal block of node
        // b = (b == 0 ? Pinf^-1 : Q_b^-1) + Phi_{b+1}^T Q_{b+1}^-1 Phi_{b+1}, block (node b, node b - 1) = -Q_b^-1 Phi_b (link b between
        // nodes b - 1 and b).  H enters through its 7 diagonals (:236-237: the rows of I - L reach +-3 nodes).  Direction 0 walks the
        // unknowns u = 0, 1, 2, ...; direction 1 (twisted plan) walks u~ = nUp - 1 - u from the chain's tail and leaves out the block
        // R x R of the 12 unknowns between the two sides (it belongs to direction 0's records).  Unknowns that do not exist (padding,
        // the other side's, the columns entering behind the last pivot) are identity records.
        const auto Hbg = TDLO_AS_GLOBAL(double, f.Hb);        // H through its 13 diagonals: Hb[13 i + u] = H(i, i - 6 + u) (symmetric: checked by the host)
        const auto lk = TDLO_AS_GLOBAL(double, f.chain);
        const BandPlan bp(M);
        const double lam = f.lambda, gam = f.lle_weight;
        for (int jj = t; jj < bp.nRecT + bp.nRecB; jj += kBlock) {
            const int dir = jj >= bp.nRecT ? 1 : 0, j = dir ? jj - bp.nRecT : jj;
            double *o = f.band + 16 * (size_t)jj;
            o[7] = 0.0; o[11] = 0.0; o[15] = 0.0;                            // the right-hand side's positions (the M-step fills them)
            const int sj = j % kBandSlots;
            const bool real = dir ? (j >= bp.D && j < bp.mB + 12) : (j < bp.limT);
            if (!real) {
                for (int q = 0; q < kBandSlots; ++q) o[band_rec_pos(q)] = q == sj ? 1.0 : 0.0;
                continue;
            }
            const int uj = dir ? bp.nUp - 1 - j : j;                         // the unknown: node b, component tj (0: f, 1: f')
            const int b = uj >> 1, tj = uj & 1;
            double kd0, kd1, kd2, ko[4] = {0.0, 0.0, 0.0, 0.0}, kn[4] = {0.0, 0.0, 0.0, 0.0};
            // everything this record reads from memory is requested here, indices clamped instead of branched: both links and H's seven values
            // arrive after ONE round trip (behind the branches below they had been three in a row)
            const int sgn = dir ? 1 : -1;                                     // rows at or above the diagonal in the direction's order: nodes b, b + sgn, ...
            double La[8], Lb[8], h[7];
#pragma unroll
            for (int q = 0; q < 8; ++q) { La[q] = lk[8 * (size_t)(b > 0 ? b : 1) + q]; Lb[q] = lk[8 * (size_t)(b + 1 < M ? b + 1 : M - 1) + q]; }
#pragma unroll
            for (int d = 0; d < 7; ++d) { const int a = b + sgn * d, ac = a < 0 ? 0 : (a > M - 1 ? M - 1 : a); const double hv = Hbg[(size_t)13 * b + (ac - b + 6)]; h[d] = (!tj && a >= 0 && a < M) ? hv : 0.0; }
            // diagonal block of node b (ff, fp, pp); ko: block (node b rows, node b - 1 columns); kn: block (node b + 1 rows, node b columns)
            if (b == 0) { const double s = ::sqrt(2.0) / beta, sf2 = 1.0 / (2.0 * ::sqrt(2.0) * beta); kd0 = 1.0 / sf2; kd1 = 0.0; kd2 = 1.0 / (s * s * sf2); }
            else {
                const double *L = La;          // link b, formed a phase ago by thread b (behind the barrier above)
                const double rdet = 1.0 / (L[4] * L[6] - L[5] * L[5]);
                const double qa = L[6] * rdet, qb = -L[5] * rdet, qd = L[4] * rdet;                  // Q^-1
                kd0 = qa; kd1 = qb; kd2 = qd;
                ko[0] = -(qa * L[0] + qb * L[2]); ko[1] = -(qa * L[1] + qb * L[3]);                 // -(Q^-1 Phi): row f
                ko[2] = -(qb * L[0] + qd * L[2]); ko[3] = -(qb * L[1] + qd * L[3]);                 //              row f'
            }
            if (b + 1 < M) {
                const double *L = Lb;
                const double rdet = 1.0 / (L[4] * L[6] - L[5] * L[5]);
                const double qa = L[6] * rdet, qb = -L[5] * rdet, qd = L[4] * rdet;
                const double t11 = qa * L[0] + qb * L[2], t12 = qa * L[1] + qb * L[3], t21 = qb * L[0] + qd * L[2], t22 = qb * L[1] + qd * L[3];
                kd0 += L[0] * t11 + L[2] * t21; kd1 += L[0] * t12 + L[2] * t22; kd2 += L[1] * t12 + L[3] * t22;      // Phi^T Q^-1 Phi
                kn[0] = -t11; kn[1] = -t12; kn[2] = -t21; kn[3] = -t22;
            }
            // The column has at most 11 non-zero entries: node b's own two unknowns, the two of the neighbour node on the side of the rows
            // already in the window (K), and the f unknowns of the 6 nodes beyond (H).  All of H's values are requested before any is used
            // (one memory latency instead of one per entry), and the entries are written straight to their positions: no loop over the 13 slots.
#pragma unroll
            for (int q = 0; q < kBandSlots; ++q) o[band_rec_pos(q)] = 0.0;
            auto put = [&](int ui, double v) __attribute__((always_inline)) {    // entry (row unknown ui, this column)
                if (ui < 0 || ui >= 2 * M) return;
                const int i = dir ? bp.nUp - 1 - ui : ui;                       // the row's index in the direction's order
                if (i > j || i < j - (kBandSlots - 1)) return;
                if (dir && i >= bp.mB && j >= bp.mB) return;                    // direction 1 leaves R x R to direction 0
                o[band_rec_pos(i % kBandSlots)] = v;
            };
            // (put() drops the rows behind the diagonal in the direction's order, e.g. f'_b for column f_b in direction 0)
            const int an = b + sgn;                                            // the neighbour node on the window's side
            // its coupling with this column: K[2b + tj][2(b-1) + ti] = ko[2 tj + ti],  K[2(b+1) + ti][2b + tj] = kn[2 ti + tj]
            const double Kf = dir ? (tj ? kn[1] : kn[0]) : (tj ? ko[2] : ko[0]), Kp = dir ? (tj ? kn[3] : kn[2]) : (tj ? ko[3] : ko[1]);
            put(2 * b, tj ? lam * kd1 : fma(gam, h[0], lam * kd0));
            put(2 * b + 1, lam * (tj ? kd2 : kd1));
            put(2 * an, tj ? lam * Kf : fma(gam, h[1], lam * Kf));
            put(2 * an + 1, lam * Kp);
            if (!tj) {
#pragma unroll
                for (int d = 2; d < 7; ++d) put(2 * (b + sgn * d), gam * h[d]);
            }
        }
        // (13 loads requested at once; same ascending order of the terms as the dense product.  Dealt out from the LAST thread down: at production
        //  size the column records above occupy waves 0 and 1, these 3 M sums waves 2 and 3 -- side by side instead of one after the other)
        for (int e = kBlock - 1 - t; e < 3 * M; e += kBlock) {
            const int i = e % M, d = e / M;
            double hv[13];
#pragma unroll
            for (int u = 0; u < 13; ++u) hv[u] = Hbg[(size_t)13 * i + u];      // H(i, k), k = i - 6 + u
            double a = 0;
#pragma unroll
            for (int u = 0; u < 13; ++u) { const int k = i - 6 + u; if (k >= 0 && k < M) a += hv[u] * sY[d * M + k]; }
            f.HY0[e] = a;
        }
    } else if (f.include_lle) {
        // H G and H Y0 (:396-401).  Global address space + unrolled k loop: independent loads in flight instead of one
        // flat load per multiply-add (this block was 50 us of a 65 us kernel).
        const auto Hg = TDLO_AS_GLOBAL(double, f.H);
        const auto Gr = TDLO_AS_GLOBAL(double, f.G);
        // (a banded H -- the library's own always is: +-6 nodes -- leaves exact zeros outside the band: the k loop runs over the band only, the same
        //  sums; at 1024 nodes the full product would be 10^9 multiply-adds on this one workgroup)
        const int hb = f.h_banded;
        for (int e = t; e < M * M; e += kBlock) {
            const int i = e % M, j = e / M;
            const int k0 = hb ? (i > 6 ? i - 6 : 0) : 0, k1 = hb ? (i + 6 < M - 1 ? i + 6 : M - 1) : M - 1;
            double a = 0;
#pragma unroll 8
            for (int k = k0; k <= k1; ++k) a += Hg[(size_t)k * M + i] * Gr[(size_t)j * M + k];
            f.HG[e] = a;
        }
        for (int e = t; e < 3 * M; e += kBlock) {
            const int i = e % M, d = e / M;
            const int k0 = hb ? (i > 6 ? i - 6 : 0) : 0, k1 = hb ? (i + 6 < M - 1 ? i + 6 : M - 1) : M - 1;
            double a = 0;
#pragma unroll 8
            for (int k = k0; k <= k1; ++k) a += Hg[(size_t)k * M + i] * sY[d * M + k];
            f.HY0[e] = a;
        }
    }
    SSTAMP(6);
    // (FUSED: the kept-point count, the sigma2 initialisation sum and with them the iteration-0 state are formed by point workgroup 0 of
    //  k_prologue, which has the counts in its registers behind the grid barrier anyway -- fused_iter0; this workgroup used to wait for the
    //  barrier and fetch the counts here: two more memory round trips, ~4 300 clocks, at the end of the kernel's longest workgroup)
    SSTAMP(7);
    if (FUSED) return;
    if (t == 0) {
        const int N = sN;
        if (!reuse && !split_mode) { f.keep[0] = (double)N; f.keep[1] = sS; }
        st->N = N; st->sum_d2 = sS;
        st->it = 0; st->converged = 1; st->crit = 0; st->Np = 0;
        st->status = 0; st->done = 0; st->retries = 0; st->retry_pending = 0; st->sh_boost = 0;
        if (!split_mode) {
            if (N == 0) { st->status = TDLO_E_EMPTY; st->done = 1; st->sigma2 = f.sigma2_in; }
            else {
                double sigma2 = f.sigma2_in;
                if (sigma2 == 0) sigma2 = sS / (3.0 * (double)M * (double)N);     // :271-273
                set_iter_consts(f, st, sigma2, (double)N);
            }
        }
    }
}

// SINGLE: the frame descriptor arrives by value in the kernarg segment (with host_up the copy in device memory does not exist yet)
template <typename T, bool SINGLE>
__global__ __launch_bounds__(kBlock) void k_setup(const FrameDev *__restrict__ frames, const FrameDev f0, int split_mode, const double *__restrict__ host_up,
                                                  double *__restrict__ dev_up, int up_doubles, int yin_off) {
    setup_body<T, false>(SINGLE ? f0 : frames[blockIdx.x], split_mode, SINGLE ? host_up : nullptr, dev_up, up_doubles, yin_off, 0u);
}

// ------------------------------------------------------------------------------------------------
// The whole prologue of one small frame in ONE launch (production size: a few thousand points, 45 nodes; the registrations of a steady-state
// tracker converge in one or two iterations, so prune + scan + setup + scatter -- three launches, two dependent-dispatch gaps and a copy --
// were more than half of a frame's GPU time).  Workgroups 0 .. nb-1 take 256 points each: prune (:177-195) and nearest node exactly as
// k_prune_pass1, counts published, grid barrier, then every workgroup forms ITS start offsets itself (thread = node: the node's total over
// all workgroups, the exclusive scan over the nodes, the counts of the workgroups in front of it -- the numbers k_setup's scan gives), and
// scatters its points like k_prune_scatter from the registers they are still in.  Workgroup nb is k_setup's node work (setup_body<FUSED>),
// running beside the others.  All nb + 1 <= 65 workgroups are resident at once (256 CUs), which is what lets them wait for each other.
// The bits are those of the three-kernel form (same arithmetic, same orders; `test_fused_prologue_equals_the_three_kernel_form`).
// ------------------------------------------------------------------------------------------------
// PAIR (tracking_step with every node visible, tdlo_api.cpp PairNext): one more node workgroup, blockIdx = nb + 1, does the same node work for the
// NEXT registration on this cloud (descriptor f2, its own upload block and node block) -- that registration starts from the same nodes, so the
// sorted cloud, the counts and the centring offset are the ones formed here, and it needs no prologue of its own.
// The registration's iteration-0 state from the kept-point count N and the sigma2 initialisation sum (:263-273): what the last lines of
// setup_body write, for the fused prologue (one thread of point workgroup 0; split_mode does not exist there)
__device__ __forceinline__ void fused_iter0(const FrameDev &f, int N, double sS) {
    IterState *st = f.st;
    f.keep[0] = (double)N; f.keep[1] = sS;
    st->N = N; st->sum_d2 = sS;
    st->it = 0; st->converged = 1; st->crit = 0; st->Np = 0;
    st->status = 0; st->done = 0; st->retries = 0; st->retry_pending = 0; st->sh_boost = 0;
    if (N == 0) { st->status = TDLO_E_EMPTY; st->done = 1; st->sigma2 = f.sigma2_in; }
    else {
        double sigma2 = f.sigma2_in;
        if (sigma2 == 0) sigma2 = sS / (3.0 * (double)f.M * (double)N);     // :271-273
        set_iter_consts(f, st, sigma2, (double)N);
    }
}

// An abandoned launch (fuse_wait): the registration ends here, before its first iteration -- every kernel behind the prologue is a no-op on a
// registration that is done, and the first M-step reports the status to the host (host_publish)
__device__ __forceinline__ void fused_fail(const FrameDev &f) {
    IterState *st = f.st;
    f.keep[0] = 0.0; f.keep[1] = 0.0;
    st->N = 0; st->sum_d2 = 0.0;
    st->it = 0; st->converged = 0; st->crit = 0; st->Np = 0;
    st->retries = 0; st->retry_pending = 0; st->sh_boost = 0; st->sigma2 = f.sigma2_in;
    st->status = TDLO_E_FUSE; st->done = 1;
}

template <typename T, bool PAIR>
__global__ __launch_bounds__(kBlock) void k_prologue(const FrameDev f, const double *__restrict__ host_up, double *__restrict__ dev_up, int up_doubles,
                                                     int yin_off, unsigned epoch_arg, const FrameDev f2, const double *__restrict__ host_up2,
                                                     double *__restrict__ dev_up2, int up_doubles2) {
    const int nb = f.nprune_blocks;
    const unsigned epoch = epoch_arg & ~kFuseWithhold;
    if ((int)blockIdx.x >= nb) {
        const bool second = PAIR && (int)blockIdx.x > nb;
        setup_body<T, true, kFuseMaxNodes>(second ? f2 : f, 0, second ? host_up2 : host_up, second ? dev_up2 : dev_up, second ? up_doubles2 : up_doubles, yin_off, epoch);
        return;
    }
    __shared__ double scratch[4];
    __shared__ double Yl[3 * kFuseMaxNodes + kPrunePad];
    __shared__ double sctr[3];
    __shared__ int lh[kFuseMaxNodes], base[kFuseMaxNodes], wtot[4], fuse_ok;
    __shared__ unsigned long long lmask[4 * kFuseMaxNodes];      // per (wave, node): which lanes of the wave keep a point nearest to that node
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, b = blockIdx.x;
    const int N0 = f.N0, M = f.M;
#ifdef TDLO_CHAIN_STAMPS      // phase stamps of point workgroup 0 (instrumented build only): f.dbg[24 ..]
#define PSTAMP(i) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); if (t == 0 && b == 0) f.dbg[24 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define PSTAMP(i) do { } while (0)
#endif
    PSTAMP(0);
    for (int m = t; m < M; m += kBlock) lh[m] = 0;
    for (int i = t; i < 4 * M; i += kBlock) lmask[i] = 0ull;
    for (int i = t; i < 3 * M; i += kBlock) Yl[i] = host_up[yin_off + i];
    const int n = b * kBlock + t;
    const bool valid = n < N0;
    double x = 0, y = 0, z = 0;
    {   // (a cloud that is still in pinned host memory -- tracking_step, FrameDev::Xhost -- goes to its place in device memory from here)
        const auto src = TDLO_AS_GLOBAL(double, f.Xhost != nullptr ? f.Xhost : f.Xraw);
        if (valid) { x = src[n]; y = src[(size_t)N0 + n]; z = src[2 * (size_t)N0 + n]; }
        if (f.Xhost != nullptr && valid) { double *xw = (double *)f.Xraw; xw[n] = x; xw[(size_t)N0 + n] = y; xw[2 * (size_t)N0 + n] = z; }
    }
    __syncthreads();
    PSTAMP(1);
    // ---- prune + nearest node (k_prune_pass1)
    double best, sum;
    int a0;
    nearest_node(Yl, M, x, y, z, best, a0, sum);
    const bool keep = valid && (::sqrt(best) < 0.1);
    const int bk = keep ? a0 : 0xffff;
    if (keep) atomicAdd(&lh[a0], 1);
    const double ssum = block_sum(keep ? sum : 0.0, scratch);
    __syncthreads();
    PSTAMP(2);
    for (int m = t; m < M; m += kBlock) __hip_atomic_store(f.hist + (size_t)b * M + m, lh[m], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (t == 0) __hip_atomic_store((unsigned long long *)f.blksum + b, (unsigned long long)__double_as_longlong(ssum), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // (beside the wait) the centring offset, in k_setup's order: wave d sums coordinate d -- lane l the nodes l, l + 64, ... -- then the butterfly
    if (t < 192) {
        const int d = t >> 6;
        double a = 0;
        for (int m = lane; m < M; m += 64) a += Yl[d * M + m];
        a = wave_sum(a);
        if (lane == 0) sctr[d] = a / M;
    }
    fuse_arrive(f, (unsigned)nb, epoch, (epoch_arg & kFuseWithhold) != 0u && b == nb - 1);
    PSTAMP(3);
    if (!fuse_wait(f, epoch, &fuse_ok)) {
        // abandoned (this workgroup or another one waited out the limit): the counts are incomplete, nothing may be scattered from them
        if (b == 0 && t == 0) { fused_fail(f); if (PAIR) fused_fail(f2); }
        return;
    }
    PSTAMP(4);
    // ---- this workgroup's start offsets: thread = node
    int Nk = 0;
    double Sk = 0.0;
    {
        int tot = 0, pre = 0;
        // (workgroup 0 also forms the registration's iteration-0 state, fused_iter0: thread i asks for workgroup i's share of the sigma2
        //  initialisation sum in the same round trip as the counts)
        unsigned long long sh0 = 0ull;
        if (b == 0 && t < nb) sh0 = __hip_atomic_load((unsigned long long *)f.blksum + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (t < M) {
            // (the counts other workgroups published: agent-scope loads, ~1900 clocks each trip -- eight at a time were three trips in a row at
            //  production size (20 point workgroups).  All of a tier's loads are requested before any is added, the index clamped instead of tested)
            auto gather = [&](auto KC) __attribute__((always_inline)) {
                constexpr int K = decltype(KC)::value;
                int v[K];
#pragma unroll
                for (int bb = 0; bb < K; ++bb) v[bb] = __hip_atomic_load(f.hist + (size_t)(bb < nb ? bb : nb - 1) * M + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                for (int bb = 0; bb < K; ++bb) { tot += bb < nb ? v[bb] : 0; pre += bb < b ? v[bb] : 0; }      // (b < nb: a clamped repeat never counts in front of b)
            };
            if (nb <= 8) gather(std::integral_constant<int, 8>());
            else if (nb <= 16) gather(std::integral_constant<int, 16>());
            else if (nb <= 32) gather(std::integral_constant<int, 32>());
            else gather(std::integral_constant<int, kFuseMaxBlocks>());
        }
        int incl = tot;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(incl, d); if (lane >= d) incl += o; }
        if (lane == 63) wtot[w] = incl;
        __shared__ double wsd2[4];
        if (b == 0) {           // the sum in the order of the unfused kernel: thread i holds workgroup i's share, the butterfly of wave_sum, the waves left to right
            const double ssum = wave_sum(__longlong_as_double((long long)sh0));
            if (lane == 0) wsd2[w] = ssum;
        }
        __syncthreads();
        int bs = 0;
        for (int q = 0; q < w; ++q) bs += wtot[q];
        if (t < M) base[t] = bs + incl - tot + pre;
        if (b == 0 && t == 0) { Nk = wtot[0] + wtot[1] + wtot[2] + wtot[3]; Sk = ((wsd2[0] + wsd2[1]) + wsd2[2]) + wsd2[3]; }
    }
    __syncthreads();
    PSTAMP(5);
    // ---- stable scatter (the order k_prune_scatter produces): a point goes behind the points of the earlier waves and of the lower lanes of
    //      its own wave that share its nearest node.  Every lane sets its bit in the (wave, node) mask with one LDS atomic (the masks were
    //      cleared before the grid barrier) instead of the wave walking its distinct nodes one ballot at a time (up to 45 trips of ~60 clocks)
    if (keep) __hip_atomic_fetch_or(lmask + w * M + bk, 1ull << lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __syncthreads();
    if (keep) {
        int dst = base[bk] + __popcll(lmask[w * M + bk] & ((1ull << lane) - 1ull));
        for (int i = 0; i < w; ++i) dst += __popcll(lmask[i * M + bk]);
        T *xs = (T *)f.Xs;
        const size_t ld = f.ldx;
        xs[dst] = (T)(x - sctr[0]);
        xs[ld + dst] = (T)(y - sctr[1]);
        xs[2 * ld + dst] = (T)(z - sctr[2]);
    }
    PSTAMP(6);
#undef PSTAMP
    if (b == 0 && t == 0) {
        // the kept-point count (the nodes' totals: integers, any order) and the sigma2 initialisation sum -> iteration-0 state of the registration
        // (of both registrations of a pair: same cloud, same nodes, their own parameters and sigma2); one thread, behind the scatter: nobody waits for it
        fused_iter0(f, Nk, Sk);
        if (PAIR) fused_iter0(f2, Nk, Sk);
    }
}

template <typename T>
__global__ __launch_bounds__(kBlock) void k_prune_scatter(const FrameDev *__restrict__ frames) {
    const FrameDev &f = frames[blockIdx.y];
    if ((int)blockIdx.x >= f.nprune_blocks) return;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int *wcnt = (int *)smem;                          // 4 x M: kept points per (wave, nearest node) of the current tile
    const int M = f.M;
    int *base = wcnt + 4 * M;                         // M: where this workgroup's next point of a node goes
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int m = threadIdx.x; m < M; m += kBlock) base[m] = f.hist_off[(size_t)blockIdx.x * M + m];
    for (int tt = 0; tt < f.prune_tiles; ++tt) {          // the workgroup's tiles in order: the sort stays stable
        const int n = (blockIdx.x * f.prune_tiles + tt) * kBlock + threadIdx.x;
        for (int i = threadIdx.x; i < 4 * M; i += kBlock) wcnt[i] = 0;
        __syncthreads();
        const int b = (n < f.N0) ? (int)f.bucket[n] : 0xffff;
        const bool keep = b != 0xffff;
        // rank of this point among the earlier points of the wave with the same nearest node (stable order)
        int rank = 0;
        unsigned long long remaining = __ballot(keep);
        while (remaining) {
            const int leader = (int)__builtin_ctzll(remaining);
            const int b0 = __builtin_amdgcn_readlane(b, leader);
            const unsigned long long mask = __ballot(keep && b == b0);
            if (keep && b == b0) rank = __popcll(mask & ((1ull << lane) - 1ull));
            if (lane == leader) wcnt[w * M + b0] = __popcll(mask);
            remaining &= ~mask;
        }
        __syncthreads();
        if (keep) {
            int dst = base[b] + rank;
            for (int i = 0; i < w; ++i) dst += wcnt[i * M + b];
            T *xs = (T *)f.Xs;
            const size_t N0 = f.N0, ld = f.ldx;
            xs[dst] = (T)(f.Xraw[n] - f.ctr[0]);
            xs[ld + dst] = (T)(f.Xraw[N0 + n] - f.ctr[1]);
            xs[2 * ld + dst] = (T)(f.Xraw[2 * N0 + n] - f.ctr[2]);
        }
        if (tt + 1 < f.prune_tiles) {
            __syncthreads();
            for (int m = threadIdx.x; m < M; m += kBlock) base[m] += wcnt[m] + wcnt[M + m] + wcnt[2 * M + m] + wcnt[3 * M + m];
            __syncthreads();
        }
    }
}

// split mode: global N and sum_d2 arrive from the all-reduce
__global__ void k_split_set_global(const FrameDev *__restrict__ frames, double Nglob, double Sglob) {
    const FrameDev &f = frames[0];
    IterState *st = f.st;
    if (threadIdx.x == 0) {
        if (Nglob <= 0) { st->status = TDLO_E_EMPTY; st->done = 1; return; }
        double sigma2 = f.sigma2_in;
        if (sigma2 == 0) sigma2 = Sglob / (3.0 * (double)f.M * Nglob);
        set_iter_consts(f, st, sigma2, Nglob);
    }
}

// the same with the two numbers in device memory (tdlo_split_run with an RCCL communicator: the all-reduce of the kept-point
// count and the sigma2-initialisation sum runs on the stream, no host round trip)
__global__ void k_split_init_pack(const FrameDev *__restrict__ frames, double *__restrict__ init2) {
    const IterState *st = frames[0].st;
    if (threadIdx.x == 0) { init2[0] = (double)st->N; init2[1] = st->sum_d2; }
}
// [all done without an error ? 1 : 0, status (0 or negative)] of this rank, for the MIN all-reduce that makes the decision to stop -- and an
// error of ONE rank (a shard whose sums leave the fixed point's range, say) -- common to all ranks: a rank-local decision would leave the
// others inside a collective
__global__ void k_split_poll_pack(const FrameDev *__restrict__ frames, double *__restrict__ out2) {
    const IterState *st = frames[0].st;
    if (threadIdx.x == 0) { out2[0] = (st->done && st->status == 0) ? 1.0 : 0.0; out2[1] = (double)st->status; }
}
__global__ void k_split_set_global_dev(const FrameDev *__restrict__ frames, const double *__restrict__ init2) {
    const FrameDev &f = frames[0];
    IterState *st = f.st;
    if (threadIdx.x == 0) {
        const double Nglob = init2[0], Sglob = init2[1];
        if (Nglob <= 0) { st->status = TDLO_E_EMPTY; st->done = 1; return; }
        double sigma2 = f.sigma2_in;
        if (sigma2 == 0) sigma2 = Sglob / (3.0 * (double)f.M * Nglob);
        set_iter_consts(f, st, sigma2, Nglob);
    }
}

// One-shot exchange, once per registration: every rank writes its [kept points, sum d2] into every peer's inbox (peer
// stores over xGMI on a multi-GPU node), raises its flag there, waits for all flags in its own inbox and adds the R
// contributions in rank order -- the same bits on every rank.  One wave; lane = peer.
__global__ void k_xch_init(const FrameDev *__restrict__ frames) {
    const FrameDev &f = frames[0];
    IterState *st = f.st;
    const int R = f.xch_nranks, me = f.xch_rank, lane = threadIdx.x;
    const unsigned long long tag = ((unsigned long long)f.xch_epoch << 32) | 0x80000000ull;
    if (lane < R) {
        xch_word *peer = xch_ptr(f.xch_inbox[lane]);
        xch_store_f64(peer + xch_off_init(R) + 2 * me, (double)st->N);
        xch_store_f64(peer + xch_off_init(R) + 2 * me + 1, st->sum_d2);
        xch_release();
        xch_store(peer + xch_off_flag_init(R) + me, tag);
    }
    const xch_word *own = xch_ptr(f.xch_inbox[me]);
    bool ok = true;
    if (lane < R) ok = xch_wait(own + xch_off_flag_init(R) + lane, tag);
    ok = __all(ok);
    xch_acquire();
    if (lane == 0) {
        if (!ok) { st->status = TDLO_E_EXCHANGE; st->done = 1; return; }
        double Nglob = 0, Sglob = 0;
        for (int r = 0; r < R; ++r) { Nglob += xch_load_f64(own + xch_off_init(R) + 2 * r); Sglob += xch_load_f64(own + xch_off_init(R) + 2 * r + 1); }
        if (Nglob <= 0) { st->status = TDLO_E_EMPTY; st->done = 1; return; }
        double sigma2 = f.sigma2_in;
        if (sigma2 == 0) sigma2 = Sglob / (3.0 * (double)f.M * Nglob);
        set_iter_consts(f, st, sigma2, Nglob);
    }
}

// ------------------------------------------------------------------------------------------------
// per-node shortest distance to the cloud, trackdlo.cpp:278-296 (only consumed by :358-372)
// ------------------------------------------------------------------------------------------------
// thread = point, one wave per batch of 64 points (grid stride); nodes arrive through scalar loads, four at a time.  Every lane
// keeps the running minimum of ITS point sequence for each of 64 nodes in registers -- one v_min per (point, node) pair, no
// LDS in the loop (the former version went through a 64 x 64 transposition tile per batch: one LDS write and one LDS read per
// pair, 30 us at 250 000 points against 5 us now) -- and the 64 lanes are reduced once per wave at the end (DPP), then the four
// waves through LDS, then one atomicMin per node and workgroup.  Chains beyond 64 nodes take one pass over the points per 64.
template <typename T, int NCH>
__global__ __launch_bounds__(kBlock) void k_dmin(const FrameDev *__restrict__ frames, int nblk) {
    const FrameDev &f = frames[blockIdx.y];
    IterState *st = f.st;
    if (!f.vis_branch || st->done) return;
    __shared__ T red[4 * 64];
    const int N = st->N, M = f.M;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const auto xs = TDLO_AS_GLOBAL(T, f.Xs);
    const size_t ld = f.ldx;
    const int nbatch = (N + 63) >> 6;
#pragma unroll 1
    for (int c = 0; c < NCH; ++c) {
        const int mb = c * kChunk;
        if (mb >= M) break;                                // block-uniform
        const int ng = ((M - mb < kChunk ? M - mb : kChunk) + 3) >> 2;      // groups of four nodes in this chunk
        T rmin[kChunk];
#pragma unroll
        for (int j = 0; j < kChunk; ++j) rmin[j] = Num<T>::inf();
        for (int batch = blockIdx.x * 4 + wave; batch < nbatch; batch += nblk * 4) {
            const int n = batch * 64 + lane;
            T x = Num<T>::inf(), y = Num<T>::inf(), z = Num<T>::inf();      // a lane without a point: every distance comes out +inf
            if (n < N) { x = xs[n]; y = xs[ld + n]; z = xs[2 * ld + n]; }
            // the node loads do not depend on the batch: left alone, the compiler hoists all 16 of them out of this loop (256 SGPRs),
            // spills them to vector lanes and pays a v_readlane per operand; an opaque copy of the base address per batch keeps
            // them where they are
            unsigned long long nbase = (unsigned long long)(uintptr_t)f.nodes;
            asm volatile("" : "+s"(nbase));
#pragma unroll
            for (int g = 0; g < kChunk / 4; ++g) {
                if (g < ng) {                              // wave-uniform; entries behind the last node stay inside the slot's node block
                    const Node4<T> q4 = load_node4<T>((const void *)(uintptr_t)nbase, mb + 4 * g);
#pragma unroll
                    for (int k = 0; k < 4; ++k) {          // (entries behind the last node of a partial group give a minimum nobody reads)
                        const T dx = x - q4.v[4 * k], dy = y - q4.v[4 * k + 1], dz = z - q4.v[4 * k + 2];
                        rmin[4 * g + k] = tmin(rmin[4 * g + k], dx * dx + dy * dy + dz * dz);
                    }
                }
                if (g & 1) __builtin_amdgcn_sched_barrier(0);      // at most two groups' nodes (32 SGPRs) in flight: hoisting all 16 loads spills
            }
        }
        // the wave's minimum of node mb + j goes to lane j
        T mine = Num<T>::inf();
#pragma unroll
        for (int j = 0; j < kChunk; ++j) {                 // (nodes behind the chain's end: a minimum nobody reads)
            const T r = wave_min_nonneg(rmin[j]);
            mine = lane == j ? r : mine;
        }
        // block-level min over the 4 waves, then one atomicMin per node and workgroup
        red[wave * 64 + lane] = mine;
        __syncthreads();
        if (wave == 0) {
            const T r = tmin(tmin(red[lane], red[64 + lane]), tmin(red[128 + lane], red[192 + lane]));
            const int m = mb + lane;
            if (m < M && r < Num<T>::inf()) atomicMin(&f.dminbits[m], Num<T>::bits(r));
        }
        __syncthreads();
    }
    // N-split with the one-shot exchange: the workgroup that finishes last hands the shard's minima to every peer, waits
    // for theirs and leaves the global minimum in dminbits for the E-step (what the MIN all-reduce does in the RCCL form).
    // One workgroup waits, so that shards sharing a GPU (tests) cannot starve each other of CUs.
    if (f.xch_nranks > 1 || (f.xch_nranks == 1 && f.xch_self)) {      // (a lone rank: its own minima are the global ones, as in the plain call)
        __shared__ int s_last;
        // the atomicMins are device-scope read-modify-writes: once they are acknowledged (vmcnt) they are performed where every
        // CU's device-scope atomics meet, so the ticket needs no cache write-back in front of it (an agent-scope fence per
        // workgroup had cost 16 us per launch at 250 000 points)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned old = __hip_atomic_fetch_add(f.sync + 8, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_last = (old == gridDim.x - 1u);
            if (s_last) __hip_atomic_store(f.sync + 8, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // re-armed for the next launch
        }
        __syncthreads();
        if (!s_last) return;
        const int R = f.xch_nranks, me = f.xch_rank, Mc = f.xch_mcap, par = st->it & 1, t = threadIdx.x;
        const unsigned long long tag = ((unsigned long long)f.xch_epoch << 32) | (unsigned)(st->it + 1);
        for (int m = t; m < M; m += kBlock) {
            const unsigned long long b = atomicMin(&f.dminbits[m], ~0ull);      // a device-scope read of the value all atomicMins produced
            const double v = b == ~0ull ? 1e300 : Num<T>::from_bits(b);      // ~0: no point on this shard
            for (int q = 0; q < R; ++q) xch_store_f64(xch_ptr(f.xch_inbox[q]) + xch_off_dmin(R, Mc) + ((size_t)par * R + me) * Mc + m, v);
        }
        xch_release();
        __syncthreads();
        if (t < R) xch_store(xch_ptr(f.xch_inbox[t]) + xch_off_flag_dmin(R) + par * R + me, tag);
        const xch_word *own = xch_ptr(f.xch_inbox[me]);
        __shared__ int s_ok;
        if (t == 0) s_ok = 1;
        __syncthreads();
        if (t < R && !xch_wait(own + xch_off_flag_dmin(R) + par * R + t, tag)) s_ok = 0;
        __syncthreads();
        xch_acquire();
        if (!s_ok) { if (t == 0) { st->status = TDLO_E_EXCHANGE; st->done = 1; } return; }
        for (int m = t; m < M; m += kBlock) {
            double mn = 1e300;
            for (int r = 0; r < R; ++r) mn = ::fmin(mn, xch_load_f64(own + xch_off_dmin(R, Mc) + ((size_t)par * R + r) * Mc + m));
            f.dminbits[m] = Num<T>::bits((T)mn);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// E-step, trackdlo.cpp:278-389
// ------------------------------------------------------------------------------------------------
// EB = threads per workgroup (256 or 512); SINGLE: the frame descriptor arrives by value in the
// kernarg segment (no dependent loads before the first useful one).
template <typename T, int NCH, bool VIS, int EB, bool SINGLE>
__global__ __launch_bounds__(EB) void k_estep(const FrameDev *__restrict__ frames, const FrameDev f0) {
    constexpr int NWE = EB / 64;
    const FrameDev &f = SINGLE ? f0 : frames[blockIdx.y];
    if (!SINGLE && (int)blockIdx.x >= f.nblkE) return;      // (one frame: the grid IS nblkE -- no scalar round trip in front of the other kernarg loads)
#ifdef TDLO_ESTEP_STAMPS
    if (threadIdx.x == 0) atomicMin(&f.dbg[32], (unsigned long long)__builtin_amdgcn_s_memrealtime());
#endif
#ifdef TDLO_ESTEP_STAMPS
#define ESTAMP(i) do { if (threadIdx.x == 0 && blockIdx.x == 0) f.dbg[40 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define ESTAMP(i) do { } while (0)
#endif
    // -DTDLO_ESTEP_PHASES: shader clocks per phase, summed over the batches of wave 0 of the middle workgroup (scripts/gpu_ephases.py)
#ifdef TDLO_ESTEP_PHASES
    unsigned long long ph_prev = __builtin_amdgcn_s_memtime(), ph_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define EPHASE(i) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); ph_acc[i] += t_ - ph_prev; ph_prev = t_; } while (0)
#else
#define EPHASE(i) do { } while (0)
#endif
    ESTAMP(0);
    const auto stg = TDLO_AS_GLOBAL(IterState, f.st);
#ifdef TDLO_TIMELINE      // wall-clock (100 MHz) begin of the first workgroup / end of the last one, iterations 20..27 (scripts/gpu_timeline.py)
    const int tl_it = stg->it - 20;
    if (threadIdx.x == 0 && blockIdx.x == 0 && tl_it >= 0 && tl_it < 8) f.dbg[4 * tl_it] = __builtin_amdgcn_s_memrealtime();
#endif
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int M = f.M;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // tile rows: one frame (SINGLE) of up to 64 nodes cannot fill the GPU anyway, so it keeps the whole window in the tile; batches use the
    // small tile (two workgroups per CU) and recompute the memberships of the later chunks of a wide window.  Chains beyond 64 nodes use the
    // small tile as well: their windows are far wider than any tile (most chunks are recomputed either way), and the 64-row tile of an
    // fp64 workgroup is 133 KB -- one workgroup per CU, one wave per SIMD, every dependent instruction a stall
    constexpr int TR = tile_rows<T>(NCH);
    constexpr int RT = (NCH == 1 && SINGLE) ? kChunk : TR;
    constexpr int RS = (RT / TR) * TR;                     // stored rows: whole chunks only
    const int rows = M < RT ? M : RT;
    // LDS carve (every offset a multiple of 16 bytes)
    V4<T> *nodesL = (V4<T> *)smem;                                    // M
    V4<T> *pts = nodesL + M;                                          // NWE x kPtsStride: point i of a wave at i + (i >> 4), see the column sums
    T *lvL = (T *)(pts + NWE * kPtsStride);                           // M rounded up to 4
    T *pbase = lvL + ((M + 3) & ~3);
    T *pb = pbase + (size_t)wave * rows * kPStride;
    double *scratch = (double *)(pbase + (((size_t)NWE * rows * kPStride + 7) & ~(size_t)3));   // 16-byte aligned, stays an LDS pointer

    const auto nodes = TDLO_AS_CONST(V4<T>, f.nodes);
    const auto xs = TDLO_AS_GLOBAL(T, f.Xs);
    const size_t ld = f.ldx;
    // first wave of loads: this wave's first points, iteration state, nodes for the LDS copy
    const int batch0 = blockIdx.x * NWE + wave;
    T x = 0, y = 0, z = 0;
    {
        const int n = batch0 * 64 + lane;
        if (n < f.N0) { x = xs[n]; y = xs[ld + n]; z = xs[2 * ld + n]; }     // N <= N0: always in bounds
    }
    // spin-ahead loop (FrameDev::spin_on; one frame, fp32, up to 64 nodes, no visibility term): this launch was dispatched while the M-step in front of
    // it still runs -- the points above are on their way, everything the M-step writes (state, nodes) is read behind the wait
    constexpr bool SPINNABLE = SINGLE && NCH == 1 && sizeof(T) == 4 && !VIS;
    const bool spin = SPINNABLE && f.spin_on != 0;
    bool spin_lost = false;
    if (spin) {
        if (!f.spin_first) {
            int *okl = (int *)scratch;
            if (tid == 0) *okl = spin_wait_word(f.sync + kSpinWordM, f.spin_wait) ? 1 : 0;
            __syncthreads();
            spin_lost = *okl == 0;
            __syncthreads();
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        __builtin_amdgcn_s_dcache_inv();          // the nodes come through scalar loads: lines of the previous iteration's nodes may sit in the scalar cache
    }
    // every workgroup of a spin-ahead launch reports itself when it is through, whatever way it leaves (the M-step behind it counts them)
    auto spin_report = [&]() {
        if (!spin) return;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");        // this thread's atomics on the accumulators have been performed
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(f.sync + kSpinWordE, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    };
    if (spin_lost) {      // the M-step in front never reported (2 s): the registration ends with an error; the count still goes up so that nothing behind waits in turn
        if (tid == 0) { IterState *sw = f.st; sw->status = TDLO_E_EXCHANGE; sw->converged = 0; sw->done = 1; }
        spin_report();
        return;
    }
    // ---- EVERYTHING this kernel reads from memory before its first barrier is requested here, back to back, in ONE round trip with the points above: the
    // iteration state, this thread's node for the LDS copy, this lane's node.  (Until round 6 the state's `done` was waited for on its own -- the branch below
    // needs it -- then N, k2, c_norm were requested, waited for, then the nodes: three round trips in a row, and the boost, the window radius and the parity
    // one more each behind the barrier: 3 500 clocks of a 8 000-clock workgroup at C2 spent waiting for memory, scripts/archive/gpu_estamps.py + the ISA.)
    // (fp32 only: the fp64 instantiations -- two waves per SIMD, registers full, chains of hundreds of nodes -- measured 0.4 % ... 4 % SLOWER with the requests up
    //  front and their values held across the kernel; they keep the order of before)
    constexpr bool EARLY = sizeof(T) == 4;
    const int done = stg->done;
    const int it0 = EARLY ? stg->it : 0;
    int N; T k2, cn;
    if constexpr (EARLY) { N = stg->N; k2 = (T)stg->k2; cn = (T)stg->c_norm; }
    const int shb_e = EARLY ? stg->sh_boost : 0;
    const double rwin_e = EARLY ? stg->rwin32 : 0.0;
    const int par_e = it0 & 1;
    const auto qg = TDLO_AS_GLOBAL(V4<T>, f.nodes);
    V4<T> nd0; nd0.x = 0; nd0.y = 0; nd0.z = 0; nd0.w = 0;
    if (EARLY && tid < M) { nd0.x = qg[tid].x; nd0.y = qg[tid].y; nd0.z = qg[tid].z; nd0.w = qg[tid].w; }
    // One frame that cannot fill the GPU, chains of up to 64 nodes (round 6): the kernel is a chain of latencies there (one wave per SIMD: 8 500 clocks per batch, of
    // which the scalar node loads of the two node loops and the LDS reads of the range tests are round trips nothing overlaps).  Lane l keeps node l in registers for the
    // whole kernel and a node's values reach the wave by v_readlane -- the same values in the same operand positions (an SGPR either way): the same bits, no round trip.
#ifdef TDLO_NO_LANE_NODES          // (scripts/build_variant.sh nolane -DTDLO_NO_LANE_NODES: the comparator of scripts/gpu_ab.sh)
    constexpr bool LANE_NODES = false;
#else
    constexpr bool LANE_NODES = SINGLE && NCH == 1;
#endif
    V4<T> qn; qn.x = 0; qn.y = 0; qn.z = 0; qn.w = 0;
    if (LANE_NODES && lane < M) { qn.x = qg[lane].x; qn.y = qg[lane].y; qn.z = qg[lane].z; qn.w = qg[lane].w; }
    if (SINGLE && f.late_aJ != nullptr && f.late_mstep == 0 && blockIdx.x == 0 && (EARLY ? it0 : stg->it) == 0) {
        // (tracking_step's second registration) the priors the host formed while the set-up kernel ran: to their place in the node block, for the
        // M-step of this and every later iteration
        double *dj = (double *)f.aJ, *dy = (double *)f.aYd;
        for (int i = threadIdx.x; i < f.M; i += EB) dj[i] = f.late_aJ[i];
        for (int i = threadIdx.x; i < 3 * f.M; i += EB) dy[i] = f.late_aYd[i];
    }
    if constexpr (EARLY) {
        if (tid < M) nodesL[tid] = nd0;
        for (int m = tid + EB; m < M; m += EB) { V4<T> o; o.x = qg[m].x; o.y = qg[m].y; o.z = qg[m].z; o.w = qg[m].w; nodesL[m] = o; }
    } else {
        N = stg->N; k2 = (T)stg->k2; cn = (T)stg->c_norm;
        for (int m = tid; m < M; m += EB) { V4<T> o; o.x = qg[m].x; o.y = qg[m].y; o.z = qg[m].z; o.w = qg[m].w; nodesL[m] = o; }
    }
    auto lane_val = [&](T v, int src) -> T {          // v of lane src (wave-uniform), as a wave-uniform operand
        if constexpr (sizeof(T) == 4) return __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(v), src));
        else return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), src), __builtin_amdgcn_readlane(__double2loint(v), src));
    };
    if (done) { spin_report(); return; }
    double lv_span = 0;      // VIS: the largest -log2 v_m + log2 v_m' over the nodes -- how far the visibility weights can lower a nearest node's membership against another's
    if (VIS) {
        // P_vis rows, :362-372: v_m = exp(-k_vis * dmin_m) / sum, folded into the exponent as log2 v_m
        double tot = 0, dmx = 0, dmn = 1e300;
        for (int m = tid; m < M; m += EB) {
            double d = ::sqrt(Num<T>::from_bits(f.dminbits[m]));
            if (d > 10000.0) d = 10000.0;                        // initial value of :282
            if (d <= f.vis_thr) d = 0;                           // :291-293
            tot += ::exp(-f.k_vis * d);
            dmx = d > dmx ? d : dmx; dmn = d < dmn ? d : dmn;
        }
        tot = block_sum_n<NWE>(tot, scratch);
        {   // (max and min of d over the nodes: a wave reduction each, the waves through the same scratch)
            dmx = wave_max_nonneg(dmx); dmn = wave_min_nonneg(dmn);
            __syncthreads();
            if (lane == 0) { scratch[wave] = dmx; scratch[NWE + wave] = dmn; }
            __syncthreads();
#pragma unroll
            for (int w_ = 0; w_ < NWE; ++w_) { dmx = scratch[w_] > dmx ? scratch[w_] : dmx; dmn = scratch[NWE + w_] < dmn ? scratch[NWE + w_] : dmn; }
            lv_span = f.k_vis * (dmx - dmn) * 1.4426950408889634;
            if (!(lv_span > 0)) lv_span = 0;
        }
        for (int m = tid; m < M; m += EB) {
            double d = ::sqrt(Num<T>::from_bits(f.dminbits[m]));
            if (d > 10000.0) d = 10000.0;
            if (d <= f.vis_thr) d = 0;
            lvL[m] = (T)(-f.k_vis * d * 1.4426950408889634 - ::log2(tot));
        }
    }
    __syncthreads();
    ESTAMP(1);
    EPHASE(0);

    // running sums in 64-bit fixed point (acc_fix at the grain of one wave x one batch; integer from there on: tdlo_devcommon.h)
    long long accQ = 0;
    const int shb = EARLY ? shb_e : stg->sh_boost;         // (fp64 mode: extra digits for R, twice as many for Q, while sigma is small: set_iter_consts)
    const double scP = acc_scale(f.acc_sh[0]), scR = acc_scale(f.acc_sh[1] + shb), scQ = acc_scale(f.acc_sh[2] + 2 * shb);
    // every converted value is checked against its limit (FrameDev::acc_lim: exact conversion, no wrap-around of the totals; a NaN fails
    // the comparison too): one compare per conversion into a lane mask, looked at once per wave at the end
    const double limP = f.acc_lim[0], limR = f.acc_lim[1] * acc_scale(-shb), limQ = f.acc_lim[2] * acc_scale(-2 * shb);
    // (a node's share of Q -- the column sums' tail -- is one conversion for up to a batch's points: held to the conversion's own exactness bound, 2^51 units)
    const double limQn = acc_scale(51 - (f.acc_sh[2] + 2 * shb));
    bool acc_ok = true;
    // NCH == 1 (M <= 64): windowed variant.  The cloud is sorted by nearest node, so the 64 points of a
    // wave sit on a short piece of the chain, and every membership whose exponent is below -151 (fp32;
    // -1080 in fp64) is EXACTLY zero: only the nodes inside an arc-length window around the wave's
    // nearest-pair range can contribute, the others are skipped -- same sums, bit for bit.
    // [M][4] 64-bit accumulators in LDS (ds_add_u64 without return: integer sums, so neither the order nor who adds matters).  Up to 64 nodes: one set
    // per wave (no contention, 1.6 KB each at M = 50).  Longer chains: ONE set per workgroup -- per-lane register accumulators (4 x NCH 64-bit values
    // and a cross-lane gather per chunk) had held the fp64 kernel at 227 VGPRs, and a set per wave would cost the second workgroup of a CU its LDS
    long long *accL = (long long *)(scratch + 16) + (NCH == 1 ? (size_t)wave * M * 4 : (size_t)0);
    if (NCH == 1) {
        for (int i = lane; i < M * 4; i += 64) accL[i] = 0;
    } else {
        for (int i = tid; i < M * 4; i += EB) accL[i] = 0;
        __syncthreads();
    }
    // Node window: E / |k2| (a squared arc length, left by the M-step; E bits: FrameDev::win_e32 / win_e64), widened by what the visibility
    // weights can take from a nearest node's membership.  A wave leaves node m out when (coord distance to the wave's nearest pairs)^2 exceeds
    // (largest nearest-node distance of the wave)^2 + this: every point's membership of m is then below 2^-E of its largest one
    const T R2win = (T)((EARLY ? rwin_e : stg->rwin64) * (1.0 + (VIS ? lv_span / (sizeof(T) == 4 ? f.win_e32 : f.win_e64) : 0.0)));

    const int nbatch = (N + 63) >> 6;
    for (int batch = batch0; batch < nbatch; batch += f.nblkE * NWE) {
        const int n = batch * 64 + lane;
        const bool valid = n < N;
        EPHASE(6);
        if (batch != batch0) { x = 0; y = 0; z = 0; if (valid) { x = xs[n]; y = xs[ld + n]; z = xs[2 * ld + n]; } }
        else if (!valid) { x = 0; y = 0; z = 0; }     // the speculative first load may have read past the kept points
        EPHASE(1);
        // ---- nearest node: argmax of the Euclidean membership (:298-310) == argmin of d2, first index
        // The cloud is sorted by nearest node, so the wave's points sit in a small ball (centre = lane 0's point, radius
        // rw).  With D_m = |y_m - centre|: every point is within min_m D_m + rw of some node and at least D_m - rw away
        // from node m, so node m can be the nearest node of a point of this wave only if D_m <= min D + 2 rw.  The search
        // runs over the index range of those candidates (a relative margin of 1e-4 dwarfs the rounding of the fp32
        // distances): the same argmin, first index on ties, from typically 5 instead of M candidates.
        int plo = 0, phi = M - 1;
        {
            const T cx = bcast_first(x), cy = bcast_first(y), cz = bcast_first(z);      // all 64 lanes are active here
            T r2 = valid ? (x - cx) * (x - cx) + (y - cy) * (y - cy) + (z - cz) * (z - cz) : T(0);
            // (the test runs on the SQUARES: D_m^2 <= lim^2 -- one square root per wave after the reduction instead of one per lane and chunk, 5 x ~25
            //  fp64 instructions per batch at 300 nodes; the margin covers the rounding of the square as it covers that of the distances)
            T Dm[NCH];
            T dmin_w = Num<T>::inf();
#pragma unroll
            for (int c = 0; c < NCH; ++c) {                  // lane = node 64 c + lane
                const int m = c * kChunk + lane;
                Dm[c] = Num<T>::inf();
                if (m < M) { const V4<T> qq = LANE_NODES ? qn : nodesL[m]; Dm[c] = (qq.x - cx) * (qq.x - cx) + (qq.y - cy) * (qq.y - cy) + (qq.z - cz) * (qq.z - cz); }
                dmin_w = tmin(dmin_w, Dm[c]);
            }
            wave_max_min_nonneg(r2, dmin_w, r2, dmin_w);          // (both reductions in one folded butterfly)
            T lim = (Num<T>::sqrt_fast(dmin_w) + T(2) * Num<T>::sqrt_fast(r2)) * T(1.0001) + T(1e-30);
            lim = lim * lim;
            int first = M, last = -1;
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const unsigned long long cand = __ballot(c * kChunk + lane < M && Dm[c] <= lim);
                if (cand) {
                    const int lo_c = c * kChunk + (int)__builtin_ctzll(cand), hi_c = c * kChunk + 63 - (int)__builtin_clzll(cand);
                    first = lo_c < first ? lo_c : first; last = hi_c > last ? hi_c : last;
                }
            }
            if (last >= 0) { plo = first; phi = last; }
            plo = __builtin_amdgcn_readfirstlane(plo); phi = __builtin_amdgcn_readfirstlane(phi);
        }
        ESTAMP(8);      // (instrumented builds: the candidate range is known; the rest of the phase is the candidates' loop)
        T best = Num<T>::inf();
        int a = plo;
        // candidates in groups of 4: the group's scalar loads are issued together (index clamped to phi), the evaluations
        // beyond phi are skipped by wave-uniform branches -- one scalar-memory latency per group instead of one per node
        if constexpr (LANE_NODES) {
            for (int m = plo; m <= phi; ++m) {           // (the candidate's coordinates from its lane: nothing to wait for)
                const T dx = x - lane_val(qn.x, m), dy = y - lane_val(qn.y, m), dz = z - lane_val(qn.z, m);
                const T d2 = dx * dx + dy * dy + dz * dz;
                if (d2 < best) { best = d2; a = m; }
            }
        } else {
            int m0 = plo;
            for (; m0 + 3 <= phi; m0 += 4) {             // whole groups: ONE scalar load of four nodes, no per-node index clamp or test
                const Node4<T> q4 = load_node4<T>(f.nodes, m0);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const T dx = x - q4.v[4 * k], dy = y - q4.v[4 * k + 1], dz = z - q4.v[4 * k + 2];
                    const T d2 = dx * dx + dy * dy + dz * dz;
                    if (d2 < best) { best = d2; a = m0 + k; }
                }
            }
            if (m0 <= phi) {                             // the last, partial group: the same load (entries behind the last node stay inside
                const Node4<T> q4 = load_node4<T>(f.nodes, m0);          // the slot's node block), evaluations beyond phi skipped wave-uniformly
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (m0 + k <= phi) {
                        const T dx = x - q4.v[4 * k], dy = y - q4.v[4 * k + 1], dz = z - q4.v[4 * k + 2];
                        const T d2 = dx * dx + dy * dy + dz * dz;
                        if (d2 < best) { best = d2; a = m0 + k; }
                    }
                }
            }
        }
        // The reference takes the nearest node as the argmax of exp(-d2 / (2 sigma2)) / (column sum + c) (:298-310).  When even the nearest
        // node's exponent is below -1075 in base 2 every entry of the column has underflowed to exactly zero in fp64, and the argmax of an
        // all-zero column is its FIRST index: node 0 (from there the end-node rule of :313-321 can give node 1 a membership of exp(0) = 1 for a
        // point a metre away -- the reference's behaviour, reproduced).  Reachable once sigma is below d / 38.6 for a kept point d <= 0.1 m
        // from the chain, i.e. sigma2 < 6.7e-6: rare, so the wave looks at it together.
        {
            const bool under = best * k2 < T(-1075);          // (lanes without a point: whatever they decide is masked out below, as their nearest node is)
            if (__builtin_expect(__ballot(under) != 0ull, 0)) {
                const V4<T> q0 = nodesL[0];
                const T dx0 = x - q0.x, dy0 = y - q0.y, dz0 = z - q0.z;
                const T d0 = dx0 * dx0 + dy0 * dy0 + dz0 * dz0;
                if (under) { a = 0; best = d0; }
            }
        }
        ESTAMP(2);
        EPHASE(2);
        // ---- second node by distance (:313-329)
        const int c1 = (a == 0) ? 2 : a - 1;
        const int c2 = (a == M - 1) ? M - 3 : a + 1;
        const V4<T> q1 = nodesL[c1], q2 = nodesL[c2], qa = nodesL[a];
        T dx = x - q1.x, dy = y - q1.y, dz = z - q1.z;
        const T s1 = dx * dx + dy * dy + dz * dz;
        dx = x - q2.x; dy = y - q2.y; dz = z - q2.z;
        const T s2 = dx * dx + dy * dy + dz * dz;
        bool first; T eb;
        if constexpr (sizeof(T) == 4) { first = s1 < s2; eb = Num<T>::sqrt_fast(first ? s1 : s2); }      // the decision of :324 on the squares: one square root
        else { const T e1 = Num<T>::sqrt_fast(s1), e2 = Num<T>::sqrt_fast(s2); first = e1 < e2; eb = first ? e1 : e2; }   // fp64: the reference's comparison of norms
        const int b = first ? c1 : c2;
        const T cb = first ? q1.w : q2.w;
        const T ea = Num<T>::sqrt_fast(best);
        const bool a_lo = a < b;
        const int lo = a_lo ? a : b, hi = a_lo ? b : a;
        const T d_lo = a_lo ? ea : eb, d_hi = a_lo ? eb : ea;
        const T c_lo = a_lo ? qa.w : cb, c_hi = a_lo ? cb : qa.w;

        // ---- node window of this wave
        int wlo = 0, whi = M - 1;
        // fp64 (round 6): the range of the nearest pairs' INDICES and "some point has the end-node gap" go through one folded butterfly of 32-bit keys
        // (the coordinates of the range's ends are looked up: coord is non-decreasing) instead of two more fp64 wave reductions, and they let the
        // membership loop take the nodes below every point's lo / above every point's hi without a per-point decision (k_estep2's scheme)
        int min_lo = 0, max_hi = M - 1;
        bool gap_any = true;
        {
            T amin, amax, bmx;                                                                                          // coord >= 0
            if constexpr (sizeof(T) == 8) {
                const unsigned klo = valid ? (unsigned)(kMaxNodes - lo) : 0u, khi = valid ? (unsigned)hi : 0u, kgap = (valid && hi - lo != 1) ? 1u : 0u;
                const unsigned zz = rows_max_u32(fold16_max(fold32_max(klo, khi), fold32_max(kgap, kgap)));
                min_lo = kMaxNodes - __builtin_amdgcn_readlane((int)zz, 15); max_hi = __builtin_amdgcn_readlane((int)zz, 47);
                gap_any = __builtin_amdgcn_readlane((int)zz, 31) != 0;
                bmx = wave_max_nonneg(valid ? best : T(0));
                amin = nodesL[min_lo].w; amax = nodesL[max_hi].w;
            } else {
                wave_min_max_max_nonneg(valid ? c_lo : Num<T>::inf(), valid ? c_hi : T(0), valid ? best : T(0), amin, amax, bmx);   // (three reductions, one folded butterfly)
            }
            const T Rwin = Num<T>::sqrt_fast(bmx + R2win);
            int first = M, last = -1;
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const int m = c * kChunk + lane;
                const T cm = (m < M) ? (LANE_NODES ? qn.w : nodesL[m].w) : Num<T>::inf();
                const unsigned long long inw = __ballot(m < M && cm > amin - Rwin && cm < amax + Rwin);
                if (inw) {
                    const int lo_c = c * kChunk + (int)__builtin_ctzll(inw), hi_c = c * kChunk + 63 - (int)__builtin_clzll(inw);
                    first = lo_c < first ? lo_c : first; last = hi_c > last ? hi_c : last;
                }
            }
            if (last >= 0) { wlo = first; whi = last; }
            wlo = __builtin_amdgcn_readfirstlane(wlo); whi = __builtin_amdgcn_readfirstlane(whi);
        }

        ESTAMP(3);
        EPHASE(3);
        // ---- unnormalised membership, column sum, Q (:354-383)
        // adj: no point of this wave has the end-node gap (hi == lo + 2), the cheaper form of the exponent applies
        const bool adj = sizeof(T) == 8 ? !gap_any : __ballot(valid && hi - lo != 1) == 0;
        // fp64, chains beyond 64 nodes, a window too wide for the tile (C5's first iterations: the whole chain): lane = node for this batch (tdlo_estep_wide.h)
        if constexpr (sizeof(T) == 8 && NCH > 1) {
            constexpr int WCH = NCH < 5 ? NCH : 5;
            if (whi - wlo + 1 >= f.estep_wide_min && whi - wlo + 1 <= 64 * WCH) {
                estep_wide_batch<WCH, VIS>(lane, wlo, whi, batch * 64, N, adj, min_lo, max_hi, x, y, z, lo, hi, c_lo, d_lo, c_hi, d_hi, k2, cn, nodesL, lvL, (double *)pb, accL,
                                           scP, scR, scQ, limP, limR, limQ, limQn, accQ, acc_ok);
                EPHASE(5);
                continue;
            }
        }
        T sum = 0, qs = 0;
        // fp64: the argument of the exponent without a per-point decision for the nodes at or below every point's lo (m <= min_lo) and at or above
        // every point's hi (m >= max_hi) -- wave-uniform branches; the per-point form in between (and everywhere when a pair has the end-node gap)
        auto geo64 = [&](int m, T cm) -> T {
            if (adj) {
                if (m <= min_lo) { const T t = (c_lo - cm) + d_lo; return t * t; }
                if (m >= max_hi) { const T t = (cm - c_hi) + d_hi; return t * t; }
                return geo_arg_adj<T>(m, lo, cm, c_lo, d_lo, c_hi, d_hi);
            }
            return geo_arg<T>(m, lo, hi, cm, c_lo, d_lo, c_hi, d_hi);
        };
        auto member = [&](const V4<T> &q, int m, auto ADJ, auto STORE) {
            if constexpr (sizeof(T) == 8) {
                // (fp64, round 6) Q is not accumulated pair by pair: with the wave's origin o, sum_m P_mn |x_n - y_m|^2 = Pt1_n |x_n - o|^2 + the nodes' part
                // sum_mk d_mk (s_mk + R_mk), d = o - y_m, formed in the column sums' fixed-point tail from the very sums it converts (k_estep2 has the algebra;
                // in fp64 the three parts' cancellation costs nothing that matters at the mode's 1e-7)
                T e = geo64(m, q.w) * k2;
                if (VIS) e += lvL[m];
                const T p = Num<T>::exp2(e);
                sum += p;
                if (decltype(STORE)::value) pb[(m - wlo) * kPStride + lane] = p;
            } else {
            T e = (decltype(ADJ)::value ? geo_arg_adj<T>(m, lo, q.w, c_lo, d_lo, c_hi, d_hi) : geo_arg<T>(m, lo, hi, q.w, c_lo, d_lo, c_hi, d_hi)) * k2;
            if (VIS) e += lvL[m];
            const T p = Num<T>::exp2(e);
            const T ddx = x - q.x, ddy = y - q.y, ddz = z - q.z;
            const T d2 = ddx * ddx + ddy * ddy + ddz * ddz;
            sum += p;
            qs += p * d2;
            if (decltype(STORE)::value) pb[(m - wlo) * kPStride + lane] = p;          // the chunks of the window that fit the tile
            }
        };
        // [from, to] in groups of 4 nodes: scalar loads first (clamped index), evaluations beyond `to` skipped wave-uniformly
        auto span = [&](int from, int to, auto ADJ, auto STORE) {
            if constexpr (LANE_NODES) {                  // (a node's four values from its lane: no scalar load in the loop)
                for (int m = from; m <= to; ++m) {
                    V4<T> q; q.x = lane_val(qn.x, m); q.y = lane_val(qn.y, m); q.z = lane_val(qn.z, m); q.w = lane_val(qn.w, m);
                    member(q, m, ADJ, STORE);
                }
                return;
            }
            int m0 = from;
            for (; m0 + 3 <= to; m0 += 4) {              // whole groups: one scalar load of four nodes (the scalar unit is shared by the
                const Node4<T> q4 = load_node4<T>(f.nodes, m0);          // CU's four SIMDs: index clamps, address arithmetic and a
#pragma unroll                                                           // wave-uniform branch per node had cost one scalar instruction
                for (int k = 0; k < 4; ++k) {                            // per two vector ones, SQ_INSTS_SALU 8.15 M : VALU 15.1 M)
                    V4<T> q; q.x = q4.v[4 * k]; q.y = q4.v[4 * k + 1]; q.z = q4.v[4 * k + 2]; q.w = q4.v[4 * k + 3];
                    member(q, m0 + k, ADJ, STORE);
                }
            }
            if (m0 <= to) {                              // the last, partial group: the same load, evaluations beyond `to` skipped
                const Node4<T> q4 = load_node4<T>(f.nodes, m0);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (m0 + k <= to) {
                        V4<T> q; q.x = q4.v[4 * k]; q.y = q4.v[4 * k + 1]; q.z = q4.v[4 * k + 2]; q.w = q4.v[4 * k + 3];
                        member(q, m0 + k, ADJ, STORE);
                    }
                }
            }
        };
        {
            const int wst = (wlo + RS - 1) < whi ? (wlo + RS - 1) : whi;              // last node whose membership is stored
            if (adj) { span(wlo, wst, std::true_type(), std::true_type()); span(wst + 1, whi, std::true_type(), std::false_type()); }
            else { span(wlo, wst, std::false_type(), std::true_type()); span(wst + 1, whi, std::false_type(), std::false_type()); }
        }
        ESTAMP(4);
        EPHASE(4);
        const T inv = valid ? Num<T>::rcp_fast(sum + cn) : T(0);
        // column sums are taken relative to a wave-local origin (lane 0's point; the sorted cloud keeps a
        // wave's points within centimetres) and leave as the residual R_m = sum_n P_mn (x_n - y_m):
        // small numbers, so fp32 tile sums lose nothing that matters (everything after a tile's sums is 64-bit fixed point)
        const T ox = bcast_first(x), oy = bcast_first(y), oz = bcast_first(z);
        if constexpr (sizeof(T) == 8) { const T ux = x - ox, uy = y - oy, uz = z - oz; qs = sum * (ux * ux + uy * uy + uz * uz); }      // Pt1_n |x_n - o|^2 (Pt1 = inv * sum)
        { const double qv = (double)(inv * qs); acc_ok &= __builtin_fabs(qv) < limQ; accQ += acc_fix(qv, scQ); }
        V4<T> pw; pw.x = inv; pw.y = inv * (x - ox); pw.z = inv * (y - oy); pw.w = inv * (z - oz);     // (s0, sx) and (sy, sz) pair up for v_pk_fma
        // one entry of padding after every 16 points: the column sums below read 16-point slices with all lanes of a slice on one
        // address (broadcast), and a ds_read_b128 serves 16 lanes that straddle two slices at a time -- 256 bytes apart they fall
        // on the same banks (2-way conflict on every read, SQ_LDS_BANK_CONFLICT = 69 % of the LDS cycles at N = 2 000 000),
        // 272 bytes apart they do not
        pts[wave * kPtsStride + lane + (lane >> 4)] = pw;

        {
            // ---- column sums (:386-389): lane = (node of the window, slice of the 64 points).  The window is summed in
            // chunks of kTileRows nodes (identical order in both tile variants, so a batch gives the bits of a single
            // frame): once sigma is millimetres the whole window is one chunk; the wide windows of the first iterations
            // take several.  With the small batch tile the memberships of the later chunks are recomputed (same
            // expression, same bits) instead of being kept in a 64-row tile.
            const int Wtot = whi - wlo + 1;
            for (int c0 = 0; c0 < Wtot; c0 += TR) {
            const int Wn = (Wtot - c0) < TR ? (Wtot - c0) : TR;
            const int wlo_c = wlo + c0;
            const bool rec = c0 >= RS;                                   // chunk beyond the stored part of the window
            const int rbase = rec ? 0 : c0;                              // first tile row of this chunk
            if (rec) {
#pragma unroll 4
                for (int m = wlo_c; m < wlo_c + Wn; ++m) {
                    const T cm = nodes[m].w;
                    T e;
                    if constexpr (sizeof(T) == 8) e = geo64(m, cm) * k2;
                    else e = (adj ? geo_arg_adj<T>(m, lo, cm, c_lo, d_lo, c_hi, d_hi) : geo_arg<T>(m, lo, hi, cm, c_lo, d_lo, c_hi, d_hi)) * k2;
                    if (VIS) e += lvL[m];
                    pb[(m - wlo_c) * kPStride + lane] = Num<T>::exp2(e);
                }
            }
            wave_lds_sync();
#ifdef TDLO_ESTEP_MFMA_COLSUMS       // (round 3 experiment, measured SLOWER and therefore off: scripts/build_variant.sh mfmacs -DTDLO_ESTEP_MFMA_COLSUMS, scripts/gpu_estep_ab.py)
            if constexpr (sizeof(T) == 4 && NCH == 1) {
                // Column sums on the matrix pipe: [P1 | S] (nodes x 4) = P (nodes x 64 points) [inv | inv (x - o)] (64 x 4) is a GEMM whose
                // reduction dimension is the points -- v_mfma_f32_16x16x4_f32, 16 nodes per tile, K = 4 points per instruction, 16 instructions
                // per 64-point batch (fp32 products and sums like the vector form; fp32-input MFMA runs at the vector rate, but beside the VALU).
                // The operands come straight from the tiles already in LDS: A[i][k] = p(node i, point 16 k + q) for instruction q -- lane i + 16 k
                // reads row i, column 16 k + q of the membership tile (row stride 65: the 64 lanes fall on 64 different banks); B[k][j] = component j
                // of point 16 k + q (lanes j < 4; zero beyond).  Any assignment of the points to (instruction, k) gives the same sums up to the
                // order of the fp32 additions; this one is fixed, so a batch's share is the same bits wherever it is computed.  The accumulator is
                // cleared per batch and converted to fixed point at the grain of one wave x one batch, as before.
                // RESULT (MI355X, N = 2 000 000, M = 50): E-step 34.6 us against 26.9 us for the vector form below; C2 5.05 against 4.6 us; 32-frame batch
                // 32.0 against 25.7 us.  Why: fp32-input MFMA has the VECTOR rate on gfx950 (256 flop / clk / CU either way), a 16 x 16 x 4 tile does
                // 1024 multiply-adds where 8 window nodes x 4 sums x 4 points = 128 are wanted (12 of 16 columns and half the rows are padding), and a wave
                // issues nothing else while its MFMA runs (scripts/ubench/mfma64.hip) -- 16 x 32 clocks per batch against ~250 for the 32 v_pk_fma of the
                // vector form.  MFMA pays where the tile is full; here K is the only large dimension.
                typedef float mfma_f4 __attribute__((ext_vector_type(4)));
                const int ci = lane & 15, ck = lane >> 4;
                const T *prow0 = pb + (size_t)rbase * kPStride + 16 * ck;
                const T *bsrc = (const T *)(pts + wave * kPtsStride + 17 * ck) + (ci < 4 ? ci : 0);
                const float bmask = ci < 4 ? 1.f : 0.f;
                for (int s0n = 0; s0n < Wn; s0n += 16) {                     // 16 nodes of the chunk at a time (a converged window: one tile)
                    // (tile rows behind the chunk's last node belong to nobody: those lanes re-read the last row; their output rows are not looked at)
                    const T *prow = prow0 + (size_t)((s0n + ci) < Wn ? (s0n + ci) : (Wn - 1)) * kPStride;
                    // all 32 operand values are requested before the first MFMA (one LDS latency, not sixteen), two accumulators halve the
                    // chain of dependent MFMAs
                    float av[16], bv[16];
#pragma unroll
                    for (int q = 0; q < 16; ++q) { av[q] = (float)prow[q]; bv[q] = (float)bsrc[4 * q]; }
                    mfma_f4 acc4 = {0.f, 0.f, 0.f, 0.f}, acc4b = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int q = 0; q < 16; q += 2) {
                        acc4 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[q], bv[q] * bmask, acc4, 0, 0, 0);      // (lanes j >= 4 loaded component 0: the mask zeroes them)
                        acc4b = __builtin_amdgcn_mfma_f32_16x16x4f32(av[q + 1], bv[q + 1] * bmask, acc4b, 0, 0, 0);
                    }
                    acc4 += acc4b;
                    // C[row 4 g + r][column j] in lane j + 16 g, register r: lanes j < 4 hold (w0, sx, sy, sz) of nodes 4 g .. 4 g + 3.
                    // The four nodes' coordinates are requested together, the four values formed, then the four integer adds.
                    {
                        const int i0 = s0n + 4 * ck;                         // first of this lane's four nodes of the chunk
                        T ymj[4]; float w0f[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int nd = wlo_c + ((i0 + r) < Wn ? (i0 + r) : (Wn - 1));
                            ymj[r] = ((const T *)(nodesL + nd))[(ci > 0 && ci < 4) ? ci - 1 : 0];
                            w0f[r] = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(acc4[r]), 0x00, 0xf, 0xf, false));   // quad_perm [0,0,0,0]: column 0 of the row
                        }
                        const T oj = ci == 1 ? ox : (ci == 2 ? oy : oz);
                        const double lim = ci == 0 ? limP : limR, sc = ci == 0 ? scP : scR;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const double w0 = (double)w0f[r];
                            const double v = ci == 0 ? w0 : (double)acc4[r] + ((double)oj - (double)ymj[r]) * w0;
                            const bool on = ci < 4 && (i0 + r) < Wn;
                            acc_ok &= !on || __builtin_fabs(v) < lim;
                            if (on) __hip_atomic_fetch_add(accL + (size_t)(wlo_c + i0 + r) * 4 + ci, acc_fix(v, sc), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                        }
                    }
                }
                wave_lds_sync();
                continue;
            }
#endif
            // (a converged window is 5 to 8 nodes: with 8 lanes per slice a lane sums 8 points instead of 16, and the extra
            // level of the slice reduction is one DPP add per sum)
            const int shift = Wn <= 8 ? 3 : (Wn <= 16 ? 4 : 5);          // wave-uniform
            const int wl = lane & ((1 << shift) - 1), sl = lane >> shift;
            const int nj = 1 << shift;                                   // points per slice (= lanes per slice): 8 / 16 / 32
            T s0 = 0, sx = 0, sy = 0, sz = 0;
            if (wl < Wn) {
                for (int h = 0; h < nj; h += 8) {                        // 8 points at a time (same order as one loop over nj)
                    const int i0 = sl * nj + h;
                    const T *prow = pb + (rbase + wl) * kPStride + i0;
                    const V4<T> *pw_ = pts + wave * kPtsStride + i0 + (i0 >> 4);
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const T p = prow[j];
                        const V4<T> w = pw_[j];
                        s0 += p * w.x; sx += p * w.y; sy += p * w.z; sz += p * w.w;
                    }
                }
            }
            // ---- the slices' partial sums folded together, and the fixed-point tail with ONE value per lane.
            // s0 (the node's P1 share, which every residual needs) goes through plain butterfly steps: every lane of a node ends up with it.  The three
            // residual sums are FOLDED: fold32(sx, sy) puts sx's halves into lanes 0..31 and sy's into lanes 32..63 with one exchange, fold16 then
            // leaves row 0 with Sx, row 2 with Sy, rows 1 and 3 with Sz -- so that each row of 16 lanes converts a different value of its nodes:
            // row 0 Rx, row 1 Rz, row 2 Ry, row 3 P1 (from s0).  Same pairs added in the same order as the xor-32 / xor-16 / ror-8 butterflies of
            // before (same bits), in 12 cross-lane instructions instead of 8 trips through the LDS crossbar + 12 more; and the half-rate fp64
            // instructions of the tail (conversions, the residual's FMA, the range check, acc_fix) are issued once instead of four times.
            // (32-lane slices come in two: lanes 0..31 convert Rx then Rz, lanes 32..63 Ry then P1.)   R_k = s_k + (o_k - y_k) w0;  P1 = 0 + 1 w0.
            s0 = fold32(s0, s0);
            T u = fold32(sx, sy), v = fold32(sz, sz);
            if (shift <= 4) { s0 = fold16(s0, s0); u = fold16(u, v); }
            if (shift <= 3) { s0 += row_ror8(s0); u += row_ror8(u); }
            {
                typedef __attribute__((address_space(3))) long long lds_i64;
                const V4<T> ym = nodesL[wlo_c + (wl < Wn ? wl : 0)];
                lds_i64 *acn = (lds_i64 *)(accL + (size_t)(wlo_c + wl) * 4);
                const double w0 = (double)s0;
                const int grp = shift == 5 ? (lane >> 5) * 2 : (lane >> 4);          // which value(s) this lane converts: 0: Rx (then Rz), 1: Rz, 2: Ry (then P1), 3: P1
                const bool mine = wl < Wn && (shift != 3 || (lane & 8) == 0);         // (8-lane slices: both halves of a row hold the totals, the lower one converts)
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    if (r == 1 && shift != 5) break;                 // (wave-uniform)
                    const int g = grp + r;                           // 0 Rx, 1 Rz, 2 Ry, 3 P1
                    const bool gx = g == 0, gy = g == 2, gp = g == 3;
                    const T sk = r == 0 ? u : v, ok = gx ? ox : (gy ? oy : oz), yk = gx ? ym.x : (gy ? ym.y : ym.z);
                    const int k = gx ? 1 : (gy ? 2 : (gp ? 0 : 3));
                    const double d = gp ? 1.0 : (double)ok - (double)yk, a = gp ? 0.0 : (double)sk;
                    const double val = ::fma(d, w0, a);
                    acc_ok &= !mine || __builtin_fabs(val) < (gp ? limP : limR);
                    // (ds_add_u64 without return: one LDS instruction per value instead of read, 64-bit add, write)
                    if (mine) __hip_atomic_fetch_add(acn + k, acc_fix(val, gp ? scP : scR), __ATOMIC_RELAXED, NCH == 1 ? __HIP_MEMORY_SCOPE_WAVEFRONT : __HIP_MEMORY_SCOPE_WORKGROUP);
                    if constexpr (sizeof(T) == 8) {      // the nodes' part of Q: d (s + R) per (node, coordinate); the P1 lanes add nothing of it
                        const double dq = (mine && !gp) ? d * (a + val) : 0.0;
                        acc_ok &= __builtin_fabs(dq) < limQn; accQ += acc_fix(dq, scQ);
                    }
                }
                wave_lds_sync();
            }
            }
        }
        EPHASE(5);
    }

    ESTAMP(5);
    // ---- the workgroup's share: its waves' integer sums added up, then into the accumulators of this iteration's parity, replica row =
    //      workgroup % kAccRows (integer atomics: neither the order of the waves nor that of the workgroups matters)
    long long *iscr = (long long *)scratch;
    {
        const long long qw = wave_sum_i64(accQ);
        if (lane == 0) iscr[wave] = qw;
    }
    if (__ballot(!acc_ok) != 0ull && lane == 0) {
        // a contribution beyond the fixed point's range, or not a number: the sums of this iteration are void.  The registration ends here with
        // an error (the M-step that follows finds done = 1 and leaves Y as it is) instead of continuing on wrapped-around integers.
        IterState *sw = f.st;
        sw->status = TDLO_E_NUMERIC; sw->converged = 0; sw->done = 1;
    }
    __syncthreads();
    ESTAMP(6);
    long long *arow = f.acc + ((size_t)(EARLY ? par_e : (TDLO_AS_GLOBAL(IterState, f.st)->it & 1)) * kAccRows + (blockIdx.x & (acc_rows_used(f) - 1))) * acc_stride(M);
    {
        const long long *accAll = (const long long *)(scratch + 16);
        for (int i = tid; i < 4 * M; i += EB) {
            const int m = i >> 2, k = i & 3;
            long long v = 0;
            if (NCH == 1) {
#pragma unroll
                for (int w = 0; w < NWE; ++w) v += accAll[(size_t)w * M * 4 + i];
            } else {
                v = accAll[i];
            }
            acc_add(arow, k * M + m, v);
        }
    }
    if (tid == 0) {
        long long q = 0;
#pragma unroll
        for (int w = 0; w < NWE; ++w) q += iscr[w];
        acc_add(arow, 4 * M, q);
    }
    spin_report();
    ESTAMP(7);
    EPHASE(7);
#ifdef TDLO_TIMELINE
    if (tid == 0 && tl_it >= 0 && tl_it < 8) atomicMax(&f.dbg[4 * tl_it + 1], (unsigned long long)__builtin_amdgcn_s_memrealtime());
#endif
#ifdef TDLO_ESTEP_PHASES
    if (tid == 0 && (int)blockIdx.x == f.nblkE / 2) { for (int i = 0; i < 10; ++i) f.dbg[48 + i] = ph_acc[i]; }
#endif
#ifdef TDLO_ESTEP_STAMPS
    if (tid == 0) atomicMax(&f.dbg[33], (unsigned long long)__builtin_amdgcn_s_memrealtime());
#endif
}

// (the generic pivoted M-step k_mstep lives in tdlo_mstep_generic.h: tdlo_mstep_big.hip falls back on its body)

// ------------------------------------------------------------------------------------------------
// M-step, fast path (M <= 128): [A | B] and G live in LDS; MB threads arranged as
// (row = t % RS, column slot = t / RS).  Elimination is Gauss-Jordan on [A | B].
//   * include_lle == 0:  A = c I + D G with D >= 0 diagonal and G symmetric positive definite.
//     Row scaling does not change Gaussian elimination, so eliminating A without pivoting is the
//     elimination of the SPD matrix (G + c D^-1) (rows with D_i = 0 stay c e_i): unconditionally
//     stable, no pivot search.  One barrier per column; the reciprocal of the next pivot is
//     produced by the thread that updates it, so no division sits between barrier and update.
//   * include_lle == 1:  A = c I + (D + s H) G is not of that form -> partial pivoting (row
//     permutation kept implicit), pivot found redundantly by every wave.
// SINGLE: the frame descriptor arrives by value in the kernarg segment (no pointer chasing).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double readlane_f64(double v, int src_lane) {     // src_lane must be wave-uniform
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src_lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src_lane);
    return __hiloint2double(hi, lo);
}



// value of lane group g = lane >> 4 out of four wave-uniform-per-group candidates.  v_cndmask with the
// four constant lane masks; written as inline asm because a ?: chain over array elements is turned into
// a dynamically indexed scratch array by the compiler.
__device__ __forceinline__ double sel_by_group(double a0, double a1, double a2, double a3) {
    int lo = __double2loint(a0), hi = __double2hiint(a0);
    const unsigned long long m1 = 0x00000000ffff0000ull, m2 = 0x0000ffff00000000ull, m3 = 0xffff000000000000ull;
    asm("v_cndmask_b32 %0, %0, %1, %2" : "+v"(lo) : "v"(__double2loint(a1)), "s"(m1));
    asm("v_cndmask_b32 %0, %0, %1, %2" : "+v"(hi) : "v"(__double2hiint(a1)), "s"(m1));
    asm("v_cndmask_b32 %0, %0, %1, %2" : "+v"(lo) : "v"(__double2loint(a2)), "s"(m2));
    asm("v_cndmask_b32 %0, %0, %1, %2" : "+v"(hi) : "v"(__double2hiint(a2)), "s"(m2));
    asm("v_cndmask_b32 %0, %0, %1, %2" : "+v"(lo) : "v"(__double2loint(a3)), "s"(m3));
    asm("v_cndmask_b32 %0, %0, %1, %2" : "+v"(hi) : "v"(__double2hiint(a3)), "s"(m3));
    return __hiloint2double(hi, lo);
}

// XCH: the instantiation that carries the one-shot exchange of the N-split (from_sums == 3); the plain loop's kernel is compiled
// without it (its presence alone moved the register allocation: +0.7 us per launch at C2)
template <typename T, int NW, int MC, bool SINGLE, bool MFMA, bool XCH = false>
__global__ __launch_bounds__(NW * 64) void k_mstep_fast(const FrameDev *__restrict__ frames, const FrameDev f0, int from_sums) {
    static_assert(!MFMA || NW == 4, "the MFMA elimination maps one 16-column block to each of 4 waves");
    constexpr int MB = NW * 64, NSLOT = NW;
    const FrameDev &f = SINGLE ? f0 : frames[blockIdx.x];
    IterState *st = f.st;
    const int M = f.M, t = threadIdx.x, lane = t & 63;
    const int slot = __builtin_amdgcn_readfirstlane(t >> 6);      // wave index, wave-uniform
    const int row = lane;
    const bool rowok = row < M;
    const int nS = 4 * M + 1, nSp = (nS + 1) & ~1;
    const int ncol = M + 3;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double *S = (double *)smem;                       // nSp
    double *W = S + ((nSp + 1) & ~1);                 // 3M (+pad)
    double *Tn = W + ((3 * M + 1) & ~1);              // 3M (+pad)
    double *red = Tn + ((3 * M + 1) & ~1);            // 8
    double *colb = red + 8;                           // 2 x 8 x 64: panel exchange buffers (double-buffered)
    double *tmp = colb + 2 * 8 * 64;                  // 12 x 64: G W slices
    double *aux = tmp + 12 * 64;                      // 7 x 64: node - Y0 (3), alpha (Y_ext - Y0) (3), alpha J (1)
    double *Gs = aux + 7 * 64;                        // M x M (column-major, ld = M)
    double *Ut = Gs + (((size_t)M * M + 1) & ~(size_t)1);    // (M + 3) x 64: the eliminated tableau for the back substitution (pivoted variant only)

#ifdef TDLO_CHAIN_STAMPS      // phase stamps only in an instrumented build (scripts/build_variant.sh): an s_memtime behind a full lgkmcnt wait + a store each
#define TDLO_STAMP(i) do { if (t == 0) f.dbg[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define TDLO_STAMP(i) do { } while (0)
#endif
    TDLO_STAMP(0);
#ifdef TDLO_ESTEP_STAMPS
    if (t == 0) { f.dbg[34] = f.dbg[32]; f.dbg[35] = f.dbg[33]; f.dbg[36] = f.dbg[38]; f.dbg[37] = __builtin_amdgcn_s_memrealtime(); f.dbg[32] = ~0ull; f.dbg[33] = 0ull; }
#endif
    const auto stg = TDLO_AS_GLOBAL(IterState, st);
    const int done = stg->done;
    const double sigma2 = stg->sigma2;
    const auto Gg = TDLO_AS_GLOBAL(double, f.G);
    const auto Y0g = TDLO_AS_GLOBAL(double, f.Y0);
    const int lle = f.include_lle, pri = f.has_priors;
    // per-node quantities the later phases need (lane = node): requested now, used after the elimination
    V4<T> nq; nq.x = nq.y = nq.z = nq.w = (T)0;
    double yq[3] = {0, 0, 0}, y0q[3] = {0, 0, 0}, ayq[3] = {0, 0, 0}, ajq = 0;
    if (rowok) {
        const auto ndg = TDLO_AS_GLOBAL(V4<T>, f.nodes);
        const auto Yg = TDLO_AS_GLOBAL(double, f.Y);
        nq.x = ndg[row].x; nq.y = ndg[row].y; nq.z = ndg[row].z; nq.w = ndg[row].w;
#pragma unroll
        for (int d = 0; d < 3; ++d) { yq[d] = Yg[d * M + row]; y0q[d] = Y0g[d * M + row]; }
        if (pri && slot == 0) {
#pragma unroll
            for (int d = 0; d < 3; ++d) ayq[d] = f.aYd[d * M + row];
            ajq = f.aJ[row];
        }
    }
    const double ctr0 = f.ctr[0], ctr1 = f.ctr[1], ctr2 = f.ctr[2];
    // ---- 1. everything that comes from memory is requested up front: the E-step's sums (kAccRows replica rows of fixed-point
    //         accumulators, both iteration parities so that no load waits for the iteration counter), G
    constexpr int GQ = (64 * 64 + MB - 1) / MB;       // G (M <= 64) goes through registers
    double gq[GQ];
#pragma unroll
    for (int u = 0; u < GQ; ++u) { const int i = t + u * MB; gq[u] = i < M * M ? Gg[i] : 0.0; }
    const int itn = stg->it;
    double sq0 = 0, sq1 = 0;
    if (from_sums != 1) {
        if (t < nS) sq0 = acc_read_both(f, t, itn);
        if (t + MB < nS) sq1 = acc_read_both(f, t + MB, itn);
    }
#ifdef TDLO_CHAIN_STAMPS
    if (slot < 4) f.dbg[8 + slot] = __builtin_amdgcn_s_memtime();
#endif
#pragma unroll
    for (int u = 0; u < GQ; ++u) { const int i = t + u * MB; if (i < M * M) Gs[i] = gq[u]; }
    if (done) { if (XCH && from_sums == 3) xch_post_error(f, st, t); return; }
    if (slot == 0) {
        aux[row] = (double)nq.x - y0q[0]; aux[64 + row] = (double)nq.y - y0q[1]; aux[128 + row] = (double)nq.z - y0q[2];
        aux[192 + row] = ayq[0]; aux[256 + row] = ayq[1]; aux[320 + row] = ayq[2]; aux[384 + row] = ajq;
    }
    TDLO_STAMP(1);
    if (from_sums != 1) {
        if (t < nS) S[t] = sq0;
        if (t + MB < nS) S[t + MB] = sq1;
        acc_clear_other<MB>(f, itn, t);
    } else {
        const auto sums = TDLO_AS_GLOBAL(double, f.sums);
        for (int i = t; i < nS; i += MB) S[i] = sums[i];
    }
    __syncthreads();
    if (from_sums == 2) {       // split mode, export only
        for (int i = t; i < nS; i += MB) f.sums[i] = S[i];
        if (t == 0) f.sums[nS] = (double)stg->N;
        return;
    }
    if (XCH && from_sums == 3 && (f.xch_nranks > 1 || f.xch_self)) {      // (a lone rank: its own sums are the total)
        // N-split with the one-shot exchange: the shard's sums go to every peer's inbox (peer stores), the flag follows; then
        // this workgroup waits for the R flags in its own inbox and adds the R contributions in rank order (the same bits
        // on every rank) -- the SUM all-reduce of the RCCL form, inside the M-step, without a launch in between.
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
        if (t == 0) red[7] = 1.0;
        __syncthreads();
        if (t < R) { const int w_ = xch_wait_sums(own + xch_off_flag_sums(R) + par * R + t, tag); if (w_ != 1) red[7] = w_ == 0 ? 0.0 : -1.0; }      // (-1: that peer's own shard failed, kXchErrMark)
        __syncthreads();
        xch_acquire();
        if (red[7] != 1.0) { if (t == 0) { st->status = red[7] == 0.0 ? TDLO_E_EXCHANGE : TDLO_E_NUMERIC; st->done = 1; st->converged = 0; } return; }
        for (int i = t; i < nS; i += MB) {
            double a = 0;
            for (int r = 0; r < R; ++r) a += xch_load_f64(own + so + ((size_t)par * R + r) * sl + i);
            S[i] = a;
        }
        __syncthreads();
    }

    // ---- 2. assemble [A | B] (:392-413) straight into registers: wave = column slot, lane = row,
    //         a[c] = element (row, slot + 8 c)
    TDLO_STAMP(2);
    const double c2 = f.lambda * sigma2, sg = sigma2 * f.lle_weight;
    int singular = 0;
    if constexpr (MFMA) {
    // ---- 2+3 (MFMA variant, include_lle == 0, M <= 60): blocked Gauss-Jordan on the 64 x 64 tableau
    //   [A (cols 0..59, identity-padded) | - | B (cols 61..63)]  held in v_mfma_f64_16x16x4 accumulators:
    //   wave w = columns 16 w .. 16 w + 15, lane (c = lane & 15, g = lane >> 4), C[rb][r] = element
    //   (row 16 rb + 4 r + g, column 16 w + c).  Per panel of 4 pivot columns k0 .. k0 + 3:
    //     a. the owner wave publishes the 4 columns (64 x 4, 2 KB) in LDS; one barrier;
    //     b. the 4 pivot rows are the accumulator register C[k0 / 16][(k0 % 16) / 4]: row k0 + g sits in
    //        lane group g, i.e. they ARE the MFMA B operand.  They are reduced among themselves by 4
    //        division-free row operations (row_i <- p row_i - a_ik row_k; the 4 x 4 pivot block is
    //        tracked redundantly by every lane, the partner row arrives by one cross-lane read) and
    //        then normalised by ONE reciprocal per row -- no reciprocal inside the 4-step chain;
    //     c. every other row:  row_i -= sum_k a_ik U_k  =  4 MFMAs per wave (A operand = the published
    //        columns with the pivot rows zeroed); the pivot rows are replaced by U.
    //   Applying the 4 row operations sequentially (instead of multiplying by an explicit inverse of
    //   the pivot block) keeps the accuracy of the unblocked elimination (measured: 1e-13 vs 1e-12).
    //   A = c I + D G with D >= 0 diagonal, G SPD: elimination without pivoting is stable (see below).
        const int w = slot, cL = lane & 15, gL = lane >> 4;
        const int col = 16 * w + cL;
        const int np4 = (M + 3) >> 2;                 // panels
        mfma_d4 C[4];
        {
            // B = PX - P1 Y0 = R + P1 (y - Y0): the E-step delivers R = PX - P1 y (y = nodes as it saw them)
            // (branch-free: every LDS read is issued unconditionally on a clamped index, the result is selected)
            const int dcol = col >= 61 ? col - 61 : 0, colc = col < M ? col : M - 1;
            const bool isA = col < M, isB = col >= 61;
#pragma unroll
            for (int rb = 0; rb < 4; ++rb) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int rw = 16 * rb + 4 * r + gL, rwc = rw < M ? rw : M - 1;
                    const double p1 = S[rwc];
                    const double gv = Gs[rwc * M + colc];                              // G symmetric: row-major read, conflict-free
                    const double va = (p1 + aux[384 + rwc]) * gv + (rw == col ? c2 : 0.0);
                    double vb = 0.0;                  // right-hand sides live in wave 3 only (wave-uniform branch: the other
                    if (w == 3) vb = S[M + dcol * M + rwc] + p1 * aux[dcol * 64 + rwc] + aux[192 + dcol * 64 + rwc];   // waves skip half of the LDS reads)
                    const double vin = isA ? va : (isB ? vb : 0.0);
                    C[rb][r] = rw < M ? vin : (rw == col ? 1.0 : 0.0);
                }
            }
        }
        TDLO_STAMP(3);
        double *pan = colb;                           // 2 x (64 rows x 4) doubles
        double *strip = colb + 512 + 64 * w;          // per wave: 16 columns x 4 pivot rows
        strip[cL * 4 + gL] = C[0][0];
        if (w == 0 && cL < 4) {
#pragma unroll
            for (int rb = 0; rb < 4; ++rb)
#pragma unroll
                for (int r = 0; r < 4; ++r) pan[(16 * rb + 4 * r + gL) * 4 + cL] = C[rb][r];
        }
        auto panel = [&](auto PC) __attribute__((always_inline)) {
            constexpr int p = decltype(PC)::value;
            const int k0 = 4 * p, rb0 = k0 >> 4, r0 = (k0 & 15) >> 2, rr = k0 & 15;
            const double *buf = pan + (p & 1) * 256;
            __syncthreads();
            if (w < (k0 >> 4)) return;                // wave-uniform: this wave holds finished columns only; staying out of
                                                      // the LDS pipe shortens the panel for the waves that still work
            double ak[4][4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) ak[i][j] = buf[(k0 + i) * 4 + j];
            double aop[4];
#pragma unroll
            for (int rb = 0; rb < 4; ++rb) {
                const double v = buf[(16 * rb + cL) * 4 + gL];
                aop[rb] = (rb == rb0 && cL >= rr && cL < rr + 4) ? 0.0 : -v;
            }
            // all four pivot-row entries of this lane's column (published to the wave's private LDS strip right
            // after the previous panel's update, i.e. off the critical path)
            double Rall[4];
            {
                const double2 q0 = *(const double2 *)(strip + cL * 4), q1 = *(const double2 *)(strip + cL * 4 + 2);
                Rall[0] = q0.x; Rall[1] = q0.y; Rall[2] = q1.x; Rall[3] = q1.y;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const double pv = ak[k][k];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (i == k) continue;
                    const double m = ak[i][k];
                    Rall[i] = fma(pv, Rall[i], -(m * Rall[k]));
#pragma unroll
                    for (int j = k + 1; j < 4; ++j) ak[i][j] = fma(pv, ak[i][j], -(m * ak[k][j]));
                    if (i < k) ak[i][i] *= pv;
                }
            }
            const double R = sel_by_group(Rall[0], Rall[1], Rall[2], Rall[3]);
            const double dg = sel_by_group(ak[0][0], ak[1][1], ak[2][2], ak[3][3]);
            {
                const int e = (__double2hiint(dg) >> 20) & 0x7ff;
                if (e == 0 || e == 0x7ff) singular = 1;      // zero / denormal / non-finite pivot
            }
            const double U = R * fast_rcp(dg);
#pragma unroll
            for (int rb = 0; rb < 4; ++rb) C[rb] = __builtin_amdgcn_mfma_f64_16x16x4f64(aop[rb], U, C[rb], 0, 0, 0);
            C[rb0][r0] = U;
            // next panel: its pivot rows go to this wave's strip, its columns are published by the wave that owns them
            {
                const int k1 = k0 + 4, c1 = k1 & 15;
                if (k1 < 64) strip[cL * 4 + gL] = C[k1 >> 4][(k1 & 15) >> 2];
                if (w == (k1 >> 4) && cL >= c1 && cL < c1 + 4) {
                    double *nb = pan + ((p + 1) & 1) * 256;
#pragma unroll
                    for (int rb = 0; rb < 4; ++rb)
#pragma unroll
                        for (int r = 0; r < 4; ++r) nb[(16 * rb + 4 * r + gL) * 4 + (cL - c1)] = C[rb][r];
                }
            }
        };
        // straight-line panels with static register indices; nested so that leaving early (np4 is block-uniform)
        // never merges accumulator values back into a common path
#define TDLO_P(P) if (np4 > P) { panel(std::integral_constant<int, P>());
        TDLO_P(0) TDLO_P(1) TDLO_P(2) TDLO_P(3) TDLO_P(4) TDLO_P(5) TDLO_P(6) TDLO_P(7) TDLO_P(8) TDLO_P(9) TDLO_P(10) TDLO_P(11)
        TDLO_P(12) TDLO_P(13) TDLO_P(14) }}}}} }}}}} }}}}}
#undef TDLO_P
        if (w == 3 && cL >= 13) {
#pragma unroll
            for (int rb = 0; rb < 4; ++rb)
#pragma unroll
                for (int r = 0; r < 4; ++r) { const int rw = 16 * rb + 4 * r + gL; if (rw < M) W[(cL - 13) * M + rw] = C[rb][r]; }
        }
        singular = __syncthreads_or(singular);
    } else {
    double a[MC];
    {
        const double p1 = rowok ? S[row] : 0.0;
        const double aj = (pri && rowok) ? f.aJ[row] : 0.0;
#pragma unroll
        for (int c = 0; c < MC; ++c) {
            const int j = slot + c * NSLOT;           // wave-uniform
            double v = 0.0;
            if (rowok && j < M) {
                const double gv = Gs[(size_t)j * M + row];
                v = (p1 + aj) * gv + (row == j ? c2 : 0.0);
                if (lle) v += sg * f.HG[(size_t)j * M + row];
            } else if (rowok && j < ncol) {
                // B = PX - P1 Y0 = R + P1 (y - Y0): the E-step delivers R = PX - P1 y (y = nodes as it saw them)
                const int i = (j - M) * M + row, d = j - M;
                const auto ndg = TDLO_AS_GLOBAL(V4<T>, f.nodes);
                const double yd = d == 0 ? (double)ndg[row].x : (d == 1 ? (double)ndg[row].y : (double)ndg[row].z);
                v = S[M + i] + p1 * (yd - Y0g[i]);
                if (lle) v -= sg * f.HY0[i];
                if (pri) v += f.aYd[i];
            }
            a[c] = v;
        }
    }
    // ---- 3. (register variant: the LLE system) Gauss-Jordan elimination on [A | B] with partial pivoting: straight-line
    // code, ONE barrier per NB columns.  wave = column slot, lane = row; a lane keeps its row's entries of the columns
    // j = slot + NB c in registers.  Columns are eliminated in panels of NB (one column from every wave):
    //   * at the start of a panel the NB columns are exchanged through LDS (one barrier) and EVERY
    //     wave keeps a private copy `pc[]` which it updates redundantly while the panel is eliminated --
    //     the pivot column is therefore always local: no per-column hand-off, no spin, no chain of LDS
    //     round trips (the VALU has room: two waves per SIMD run this at full speed each);
    //   * the pivot ROW entries a wave needs are the ones its own lane `pw` holds -> v_readlane.
    // Row operation, division-free: row_i <- (p s) row_i - (a_ik s) row_k with s = 2^-exponent(p), so
    // p s is in [1, 2): rows grow by < 2x per column, no reciprocal on the dependency chain.  A row's
    // pivot entry times its accumulated scale is tracked exactly (`dgv`): x = b / dgv.
    // A = c I + (D + s H) G has no SPD structure and H is ill-conditioned -> partial pivoting, the row
    // permutation stays implicit (`mine`); every wave takes the same decision on its own copy.
    // (Systems without the LLE term never come here: they take the MFMA tableau above or k_mstep_big.)
    TDLO_STAMP(3);
    constexpr int NB = NSLOT;                         // panel width
    static_assert(NB == 8 || NB == 4, "panel width");
    double *pan = colb;                               // 2 x NB x 64 doubles
    int mine = -1;                                    // unknown this row ends up solving
    unsigned long long usedmask = 0;
    double dgv = 1.0;                                 // pivot entry times later row scalings
    // Panels are unrolled so that inside panel p the update loop is the static register range (p, MC):
    // finished columns are skipped without a single data-dependent branch.  (A rolled panel loop that
    // updates every register was measured slower: the extra column updates cost more than the
    // instruction fetches saved.)
#pragma unroll
    for (int p = 0; p < MC; ++p) {                    // panel p = columns p NB .. p NB + NB - 1; mine is register p
        if (p * NB < M) {                             // wave-uniform
            double *buf = pan + (p & 1) * NB * 64;
            buf[slot * 64 + row] = a[p];
            __syncthreads();
            double pc[NB];
#pragma unroll
            for (int s2 = 0; s2 < NB; ++s2) pc[s2] = buf[s2 * 64 + row];
#pragma unroll
            for (int sI = 0; sI < NB; ++sI) {
                const int k = p * NB + sI;
                if (k < M) {                          // wave-uniform
                    const double aik = pc[sI];
                    int pw;
                    {
                        // largest |a_ik| among the rows not yet used, judged by the upper 32 bits of the double (sign off: non-negative
                        // doubles order like their bit patterns; any entry within 2^-20 of the largest is as good a pivot), first lane
                        // among equals: a wave-wide max on v_max_u32 with DPP operands (7 instructions) instead of six dependent
                        // ds_bpermute round trips on the column's critical path (~600 cycles per column)
                        const unsigned key = (rowok && !((usedmask >> row) & 1ull)) ? ((unsigned)__double2hiint(aik) & 0x7fffffffu) + 1u : 0u;
                        const unsigned mx = wave_minmax_u32<true>(key);
                        const unsigned long long hit = __ballot(key == mx);
                        pw = __builtin_amdgcn_readfirstlane((int)__builtin_ctzll(hit));
                        usedmask |= 1ull << pw;
                    }
                    const double pv = readlane_f64(aik, pw);
                    const int e = (__double2hiint(pv) >> 20) & 0x7ff;
                    if (e == 0 || e == 0x7ff) singular = 1;          // zero / denormal / non-finite pivot
                    const double sc = __hiloint2double((2046 - e) << 20, 0);
                    const bool self = (row == pw);
                    const bool upd = !self && mine < 0;       // a row that has served as pivot row is final: Gaussian elimination,
                    const double ps = upd ? pv * sc : 1.0;    // not Gauss-Jordan (see the back substitution below)
                    const double ls = upd ? aik * sc : 0.0;
                    dgv = self ? pv : dgv;
                    mine = self ? k : mine;
#pragma unroll
                    for (int s2 = sI + 1; s2 < NB; ++s2) pc[s2] = fma(ps, pc[s2], -(ls * readlane_f64(pc[s2], pw)));
#pragma unroll
                    for (int c = p + 1; c < MC; ++c) a[c] = fma(ps, a[c], -(ls * readlane_f64(a[c], pw)));
                }
            }
            // this wave's own column of the panel goes back to its register (matters for right-hand-side
            // columns that share the last panel with matrix columns)
            {
                double v = pc[0];
#pragma unroll
                for (int s2 = 1; s2 < NB; ++s2) v = (s2 == slot) ? pc[s2] : v;
                a[p] = v;
            }
        }
    }
    // ---- 3b. back substitution.  The elimination above is Gaussian elimination with partial pivoting (LU), which is backward
    // stable; Gauss-Jordan (reducing the rows above the pivot as well, as this kernel did) is only forward stable: its error
    // in W is not of the form A^-1 dA W, the product G W of :417 does not damp it, and on the ill-conditioned systems of
    // the pre-processing registration it cost ~1e-8 m per solve in T (the oracle's QR: 1e-11, tests/test_solver_error.py).
    // The finished tableau goes to LDS once (column j = 64 consecutive rows); waves 0..2 take one right-hand side each,
    // lane = row: the row's entries of U live in registers, x_k comes from the lane that pivots column k (v_readlane),
    // every row that pivots an earlier column takes b -= U[row][k] x_k.  One LDS round trip, no barrier inside the loop.
    __syncthreads();                                  // the panel buffers / Sg are dead; Ut has its own area
#pragma unroll
    for (int c = 0; c < MC; ++c) { const int j = slot + c * NSLOT; if (j < ncol) Ut[j * 64 + row] = a[c]; }
    __syncthreads();
    if (slot < 3) {
        constexpr int KMAX = (4 * MC - 3) < 64 ? (4 * MC - 3) : 64;      // M <= 4 MC - 3 for this instantiation
        double u[KMAX];
#pragma unroll
        for (int k = 0; k < KMAX; ++k) u[k] = Ut[(k < M ? k : 0) * 64 + row];
        double b = Ut[(M + slot) * 64 + row];
        const double rinv = mine >= 0 ? 1.0 / dgv : 0.0;
#pragma unroll
        for (int k = KMAX - 1; k >= 0; --k) {
            if (k < M) {                              // wave-uniform
                const unsigned long long hit = __ballot(mine == k);
                const int pl = hit ? (int)__builtin_ctzll(hit) : 0;
                const double w = readlane_f64(b, pl) * readlane_f64(rinv, pl);
                b = (mine >= 0 && mine < k) ? fma(-u[k], w, b) : b;
            }
        }
        if (rowok && mine >= 0) W[slot * M + mine] = b * rinv;
    }
    singular = __syncthreads_or(singular);
    }   // !MFMA

    // ---- 4. T = Y0 + G W (:417): waves 0..5 = (coordinate d, half of the k range), lane = node
    TDLO_STAMP(4);
    if (NW >= 6) {
        if (slot < 6 && rowok) {
            const int d = slot % 3, ks = slot / 3;
            double v = 0;
            for (int k = ks; k < M; k += 2) v += Gs[(size_t)k * M + row] * W[d * M + k];
            tmp[slot * 64 + row] = v;
        }
        __syncthreads();
        if (slot < 3 && rowok) Tn[slot * M + row] = y0q[slot] + (tmp[slot * 64 + row] + tmp[(slot + 3) * 64 + row]);
    } else {
        // wave = quarter of the k range, lane = node, three coordinates at once; fixed-order sum of the quarters
        const int kq = (M + 3) >> 2, kb = slot * kq, ke = (kb + kq) < M ? (kb + kq) : M;
        double v0 = 0, v1 = 0, v2 = 0;
        {
            const int rc = rowok ? row : 0;
#pragma unroll
            for (int u = 0; u < 16; ++u) {            // kq <= 16; straight-line so that the LDS reads are batched
                const int k = kb + u, kc = k < M ? k : M - 1;
                const double gk = (k < ke) ? Gs[kc * M + rc] : 0.0;
                v0 += gk * W[kc]; v1 += gk * W[M + kc]; v2 += gk * W[2 * M + kc];
            }
        }
        tmp[(slot * 3 + 0) * 64 + row] = v0; tmp[(slot * 3 + 1) * 64 + row] = v1; tmp[(slot * 3 + 2) * 64 + row] = v2;
        __syncthreads();
        if (slot < 3 && rowok)
            Tn[slot * M + row] = y0q[slot] + (((tmp[slot * 64 + row] + tmp[(3 + slot) * 64 + row]) + tmp[(6 + slot) * 64 + row]) + tmp[(9 + slot) * 64 + row]);
    }
    __syncthreads();

    // ---- 5. sigma2 (residual form of :418-422) and the convergence criterion (:424): waves 0..3 each
    //         reduce one of the four sums over the nodes
    TDLO_STAMP(5);
    if (slot < 4) {
        double v = 0;
        if (rowok) {
            const int m = row;
            const double yx = (double)nq.x, yy = (double)nq.y, yz = (double)nq.z;   // nodes as the E-step saw them
            const double p1 = S[m];
            const double dx = Tn[m] - yx, dy = Tn[M + m] - yy, dz = Tn[2 * M + m] - yz;
            if (slot == 0) v = p1;
            else if (slot == 1) v = dx * S[M + m] + dy * S[2 * M + m] + dz * S[3 * M + m];
            else if (slot == 2) v = p1 * (dx * dx + dy * dy + dz * dz);
            else { const double ex = yq[0] - Tn[m], ey = yq[1] - Tn[M + m], ez = yq[2] - Tn[2 * M + m]; v = ::sqrt(ex * ex + ey * ey + ez * ez); }
        }
        v = wave_sum(v);
        if (lane == 0) red[slot] = v;
    }
    __syncthreads();
    const double s_np = red[0], s_dr = red[1], s_pd = red[2], s_cr = red[3];
    const double new_sigma2 = (S[4 * M] - 2.0 * s_dr + s_pd) / (s_np * 3.0);
    const double crit = s_cr / (double)M;

    // ---- 6. publish Y, nodes, iteration state
    TDLO_STAMP(6);
    V4<T> *nodes_w = (V4<T> *)f.nodes;
    if (slot == 3 && rowok) {
        V4<T> q; q.x = (T)Tn[row]; q.y = (T)Tn[M + row]; q.z = (T)Tn[2 * M + row]; q.w = nq.w;      // .w = chain coordinate, unchanged
        nodes_w[row] = q;
        f.dminbits[row] = ~0ull;
    }
    if (slot < 3 && rowok) {
        const int i = slot * M + row;
        f.Y[i] = Tn[i];
        f.Yout[i] = Tn[i] + (slot == 0 ? ctr0 : (slot == 1 ? ctr1 : ctr2));
    }
    TDLO_STAMP(7);
#ifdef TDLO_ESTEP_STAMPS
    if (t == 0) f.dbg[38] = __builtin_amdgcn_s_memrealtime();
#endif
    if (t == 0) {
        const int it = stg->it + 1;
        st->it = it; st->crit = crit; st->Np = s_np;
        const double Nc = stg->Nc;
        const bool finite_ok = (new_sigma2 == new_sigma2) && (fabs(new_sigma2) < 1e300) && (new_sigma2 > 0) && !singular;
        if (finite_ok) set_iter_consts(f, st, new_sigma2, Nc);
        else { st->sigma2 = new_sigma2; st->status = TDLO_E_NUMERIC; st->done = 1; st->converged = 0; }
        if (crit < f.tol) st->done = 1;                                   // :424-428
        else if (it >= f.max_iter) { st->converged = 0; st->done = 1; }  // :433-437
    }
}

// ------------------------------------------------------------------------------------------------
// Caller-side visibility pre-pass: per-node shortest distance to the RAW cloud,
// trackdlo/src/trackdlo_node.cpp:257-277.  fp64 on the uploaded fp64 points so that the `<=
// visibility_threshold` decision at :316 is taken on the same numbers as the reference.
// thread = point, grid-stride; per node a wave-level min, then one atomicMin per wave on ordered bits.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_node_min_dist(const double *__restrict__ X, int N, const double *__restrict__ Y, int M,
                                                         unsigned long long *__restrict__ out_bits) {
    const int lane = threadIdx.x & 63;
    for (int base = blockIdx.x * kBlock; base < N; base += gridDim.x * kBlock) {
        const int n = base + threadIdx.x;
        const bool valid = n < N;
        double x = 0, y = 0, z = 0;
        if (valid) { x = X[n]; y = X[(size_t)N + n]; z = X[2 * (size_t)N + n]; }
        for (int m = 0; m < M; ++m) {
            double d2 = valid ? node_point_d2(Y[m], Y[M + m], Y[2 * M + m], x, y, z) : __builtin_huge_val();
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) d2 = ::fmin(d2, __shfl_xor(d2, o));
            if (lane == 0) atomicMin(&out_bits[m], (unsigned long long)__double_as_longlong(d2));
        }
    }
}

// The same in ONE launch and without copies (round 5): the nodes are read from pinned host memory by the kernel, the workgroup that draws the last
// ticket hands the M minima to pinned host memory, re-arms the minima (+inf) and the ticket for the next call, and raises the word the host waits on.
// state: [0 .. M) minima as ordered bits (armed: +inf), [kMaxNodes] tickets.  Same arithmetic in the same order as k_node_min_dist.
__global__ __launch_bounds__(kBlock) void k_node_min_dist_direct(const double *__restrict__ X, int N, const double *__restrict__ Yhost, int M,
                                                                unsigned long long *__restrict__ state, unsigned long long *__restrict__ res_host, unsigned epoch) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double *Yl = (double *)smem;
    __shared__ int slast;
    const int t = threadIdx.x, lane = t & 63;
    for (int i = t; i < 3 * M; i += kBlock) Yl[i] = Yhost[i];
    __syncthreads();
    for (int base = blockIdx.x * kBlock; base < N; base += gridDim.x * kBlock) {
        const int n = base + t;
        const bool valid = n < N;
        double x = 0, y = 0, z = 0;
        if (valid) { x = X[n]; y = X[(size_t)N + n]; z = X[2 * (size_t)N + n]; }
        for (int m = 0; m < M; ++m) {
            double d2 = valid ? node_point_d2(Yl[m], Yl[M + m], Yl[2 * M + m], x, y, z) : __builtin_huge_val();
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) d2 = ::fmin(d2, __shfl_xor(d2, o));
            if (lane == 0) (void)__hip_atomic_fetch_min(state + m, (unsigned long long)__double_as_longlong(d2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t == 0) slast = __hip_atomic_fetch_add(state + kMaxNodes, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned long long)(gridDim.x - 1);
    __syncthreads();
    if (!slast) return;
    for (int m = t; m < M; m += kBlock) {
        res_host[1 + m] = __hip_atomic_load(state + m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(state + m, 0x7ff0000000000000ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (t == 0) __hip_atomic_store(state + kMaxNodes, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t == 0) __hip_atomic_store(res_host, (unsigned long long)epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

hipError_t launch_node_min_dist_direct(const double *X, int N, const double *Yhost, int M, unsigned long long *state, unsigned long long *res_host, unsigned epoch, hipStream_t s) {
    int blocks = (N + kBlock - 1) / kBlock;
    if (blocks > 1024) blocks = 1024;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_node_min_dist_direct, dim3(blocks), dim3(kBlock), sizeof(double) * 3 * M, s, X, N, Yhost, M, state, res_host, epoch);
    return hipGetLastError();
}

hipError_t launch_node_min_dist(const double *X, int N, const double *Y, int M, unsigned long long *out_bits, hipStream_t s) {
    int blocks = (N + kBlock - 1) / kBlock;
    if (blocks > 1024) blocks = 1024;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_node_min_dist, dim3(blocks), dim3(kBlock), 0, s, X, N, Y, M, out_bits);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
static inline int nch_for(int M) { const int c = (M + kChunk - 1) / kChunk; return c <= 1 ? 1 : (c <= 2 ? 2 : (c <= 4 ? 4 : (c <= 8 ? 8 : 16))); }

size_t mstep_lds_bytes(int M) {
    const int nS = 4 * M + 1, ld = M | 1;
    size_t b = sizeof(double) * (size_t)(((nS + 1) & ~1) + 6 * M + 16) + sizeof(int) * 2 * (size_t)((M + 3) & ~3) + 16;
    if (M <= kLdsSolveMaxM) b += sizeof(double) * (size_t)ld * (M + 3);
    return b;
}

template <typename K> static hipError_t set_lds(K kernel, size_t bytes) {
    if (bytes > 64 * 1024) return hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    return hipSuccess;
}

#define TDLO_TRY(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return e_; } while (0)

template <typename T, int EB> static size_t estep_lds_bytes(int M, bool single) {
    const int rt = (M <= kChunk && single) ? kChunk : tile_rows<T>(M <= kChunk ? 1 : 2);
    const int rows = M < rt ? M : rt;
    constexpr int NWE = EB / 64;
    const size_t tile = sizeof(T) * (((size_t)NWE * rows * kPStride + 7) & ~(size_t)3);
    const size_t red = (size_t)NWE * 64 * 4 * sizeof(double);
    size_t b = sizeof(V4<T>) * (size_t)M + sizeof(V4<T>) * NWE * kPtsStride + sizeof(T) * (size_t)((M + 3) & ~3);
    b += (tile > red ? tile : red) + 16 * sizeof(double) + 64;
    b += sizeof(double) * (size_t)(M <= kChunk ? NWE : 1) * M * 4;     // [M][4] 64-bit accumulators: per wave up to 64 nodes, per workgroup beyond
    return b;
}

// measurement aid (tdlo_profile_kernel kind 10): when set, the E-step is launched with start/stop events bound to the
// dispatch itself (hipExtLaunchKernelGGL), i.e. the same begin/end timestamps a kernel trace reports
static thread_local hipEvent_t g_estep_ev[2] = {nullptr, nullptr};      // (per launching thread: another context's launches on another host thread must not pick these up)
// the same for the M-step dispatch (k_mstep_fast here, k_mstep_mcu in tdlo_mstep_big.hip): tdlo_profile_iteration
thread_local hipEvent_t g_mstep_ev[2] = {nullptr, nullptr};

template <typename T, int EB> static hipError_t launch_estep_TE(const FrameDev *fd, const FrameDev *fh, int F, hipStream_t s) {
    const int M = fh[0].M, nch = nch_for(M);
    const bool vis = fh[0].vis_branch != 0;
    int gx = 0;
    for (int i = 0; i < F; ++i) gx = fh[i].nblkE > gx ? fh[i].nblkE : gx;
    const dim3 grid(gx, F), block(EB);
    const bool single = F == 1 && fh[0].wide_tile;
    const size_t lds = estep_lds_bytes<T, EB>(M, single);
#define TDLO_E2(NCH, VIS, SINGLE) do { TDLO_TRY(set_lds(k_estep<T, NCH, VIS, EB, SINGLE>, lds)); \
        if (g_estep_ev[0]) hipExtLaunchKernelGGL((k_estep<T, NCH, VIS, EB, SINGLE>), grid, block, lds, s, g_estep_ev[0], g_estep_ev[1], 0, fd, fh[0]); \
        else hipLaunchKernelGGL((k_estep<T, NCH, VIS, EB, SINGLE>), grid, block, lds, s, fd, fh[0]); } while (0)
#define TDLO_E(NCH, VIS) do { if (single) TDLO_E2(NCH, VIS, true); else TDLO_E2(NCH, VIS, false); } while (0)
    if (vis) { switch (nch) { case 1: TDLO_E(1, true); break; case 2: TDLO_E(2, true); break; case 4: TDLO_E(4, true); break; case 8: TDLO_E(8, true); break; default: TDLO_E(16, true); } }
    else     { switch (nch) { case 1: TDLO_E(1, false); break; case 2: TDLO_E(2, false); break; case 4: TDLO_E(4, false); break; case 8: TDLO_E(8, false); break; default: TDLO_E(16, false); } }
#undef TDLO_E
#undef TDLO_E2
    return hipGetLastError();
}

template <typename T> static hipError_t launch_estep_T(const FrameDev *fd, const FrameDev *fh, int F, hipStream_t s) {
    if (sizeof(T) == 4 && fh[0].estep2) {            // two points per lane (tdlo_estep2.hip): clouds and batches that fill the GPU
        return launch_estep2(fd, fh, F, s, g_estep_ev[0], g_estep_ev[1]);
    }
    if (fh[0].eb == 512) return launch_estep_TE<T, 512>(fd, fh, F, s);
    return launch_estep_TE<T, 256>(fd, fh, F, s);
}

// workgroups of k_dmin (4 waves, a wave = a grid-stride sequence of 64-point batches): a wave pays a fixed epilogue (64 nodes x
// a cross-lane minimum, about two batches' worth of instructions), so large clouds give every wave several batches
static inline int dmin_blocks(const FrameDev *fh, int F) {
    int nb = 1;
    for (int i = 0; i < F; ++i) { const int b = (fh[i].N0 + 63) / 64; nb = b > nb ? b : nb; }
    const int per_wave = nb >= 8192 ? 8 : (nb >= 2048 ? 4 : (nb >= 512 ? 2 : 1));
    const int gx = (nb + 4 * per_wave - 1) / (4 * per_wave);
    return gx < 1 ? 1 : (gx > 1024 ? 1024 : gx);
}

template <typename T> static hipError_t launch_dmin_T(const FrameDev *fd, const FrameDev *fh, int F, hipStream_t s) {
    const int M = fh[0].M, nch = nch_for(M);
    const int gx = dmin_blocks(fh, F);
    const dim3 grid(gx, F), block(kBlock);
#define TDLO_D(NCH) do { hipLaunchKernelGGL((k_dmin<T, NCH>), grid, block, 0, s, fd, gx); } while (0)
    switch (nch) { case 1: TDLO_D(1); break; case 2: TDLO_D(2); break; case 4: TDLO_D(4); break; case 8: TDLO_D(8); break; default: TDLO_D(16); }
#undef TDLO_D
    return hipGetLastError();
}

template <typename T> static size_t mstep_fast_lds_bytes(int M, int NW, bool pivoted) {
    const int nSp = (4 * M + 2) & ~1;
    size_t d = (size_t)nSp + 2 * (size_t)((3 * M + 1) & ~1) + 8 + 2 * 8 * 64 + 12 * 64 + 7 * 64 + (((size_t)M * M + 1) & ~(size_t)1);
    if (pivoted) d += (size_t)(M + 3) * 64;
    (void)NW;
    return d * sizeof(double);
}

template <typename T, int NW, int MC, bool MFMA = false> static hipError_t launch_mstep_fast(const FrameDev *fd, const FrameDev *fh, int F, int from_sums, hipStream_t s) {
    const size_t lds = mstep_fast_lds_bytes<T>(fh[0].M, NW, !MFMA);
    if (from_sums == 3) {          // one frame (a shard of the split cloud), exchange inside the kernel
        if (F != 1) return hipErrorInvalidValue;
        TDLO_TRY(set_lds(k_mstep_fast<T, NW, MC, true, MFMA, true>, lds));
        hipLaunchKernelGGL((k_mstep_fast<T, NW, MC, true, MFMA, true>), dim3(1), dim3(NW * 64), lds, s, fd, fh[0], from_sums);
    } else if (F == 1) {
        TDLO_TRY(set_lds(k_mstep_fast<T, NW, MC, true, MFMA>, lds));
        if (g_mstep_ev[0]) hipExtLaunchKernelGGL((k_mstep_fast<T, NW, MC, true, MFMA>), dim3(1), dim3(NW * 64), lds, s, g_mstep_ev[0], g_mstep_ev[1], 0, fd, fh[0], from_sums);
        else hipLaunchKernelGGL((k_mstep_fast<T, NW, MC, true, MFMA>), dim3(1), dim3(NW * 64), lds, s, fd, fh[0], from_sums);
    } else {
        TDLO_TRY(set_lds(k_mstep_fast<T, NW, MC, false, MFMA>, lds));
        if (g_mstep_ev[0]) hipExtLaunchKernelGGL((k_mstep_fast<T, NW, MC, false, MFMA>), dim3(F), dim3(NW * 64), lds, s, g_mstep_ev[0], g_mstep_ev[1], 0, fd, fh[0], from_sums);
        else hipLaunchKernelGGL((k_mstep_fast<T, NW, MC, false, MFMA>), dim3(F), dim3(NW * 64), lds, s, fd, fh[0], from_sums);
    }
    return hipGetLastError();
}


template <typename T> static hipError_t launch_mstep_T(const FrameDev *fd, const FrameDev *fh, int F, int from_sums, hipStream_t s) {
    const int M = fh[0].M;
    bool any_lle = false;
    for (int i = 0; i < F; ++i) any_lle = any_lle || fh[i].include_lle;
    if (!any_lle && !fh[0].mstep_dense) return launch_mstep_chain(fd, fh, F, from_sums, fh[0].precision == TDLO_PREC_F64, s);
    if (any_lle && fh[0].lle_band) return launch_mstep_band(fd, fh, F, from_sums, fh[0].precision == TDLO_PREC_F64, s);
    if (M <= 60 && !any_lle) return launch_mstep_fast<T, 4, 1, true>(fd, fh, F, from_sums, s);
    // (M = 61..64 without LLE: the 64-column register tableau has no room for the right-hand sides; the tracer-column
    //  variant of the register path is an order of magnitude less accurate at weak regularisation, so these sizes take
    //  the blocked global-memory path as well)
    if (M > 60 && M <= kChainLdsMaxNodes && !any_lle) return launch_mstep_big(fd, fh, F, from_sums, fh[0].precision == TDLO_PREC_F64, s);
    if (M <= 64) {
        const int mc = (M + 3 + 3) / 4;               // columns per wave: M matrix + 3 right-hand sides
        if (mc <= 6) return launch_mstep_fast<T, 4, 6>(fd, fh, F, from_sums, s);
        if (mc <= 10) return launch_mstep_fast<T, 4, 10>(fd, fh, F, from_sums, s);
        if (mc <= 14) return launch_mstep_fast<T, 4, 14>(fd, fh, F, from_sums, s);
        return launch_mstep_fast<T, 4, 17>(fd, fh, F, from_sums, s);
    } else if (M <= kLdsSolveMaxM) {
        const size_t lds = mstep_lds_bytes(M);
        TDLO_TRY(set_lds((k_mstep<T, true, kBlock>), lds));
        hipLaunchKernelGGL((k_mstep<T, true, kBlock>), dim3(F), dim3(kBlock), lds, s, fd, from_sums, 0);
    } else if (from_sums != 2 && mstep_pivot_mcu_enabled() && M <= kChainLdsMaxNodes) {
        TDLO_TRY(launch_mstep_pivot_mcu(fd, fh, F, from_sums, fh[0].precision == TDLO_PREC_F64, s));
        const size_t lds = mstep_lds_bytes(M);        // redoes an iteration whose hand-offs timed out; otherwise leaves at once
        hipLaunchKernelGGL((k_mstep<T, false, 1024>), dim3(F), dim3(1024), lds, s, fd, from_sums, 1);
    } else {
        // the one-workgroup elimination with the tableau in global memory: the comparator of the multi-workgroup kernels, and what serves the dense
        // cases of chains beyond kChainLdsMaxNodes nodes (the LLE term, lambda = 0, TDLO_MSTEP=dense): O(M^3) on one CU -- correct, not fast
        const size_t lds = mstep_lds_bytes(M);
        TDLO_TRY(set_lds((k_mstep<T, false, 1024>), lds));
        hipLaunchKernelGGL((k_mstep<T, false, 1024>), dim3(F), dim3(1024), lds, s, fd, from_sums, 0);
    }
    return hipGetLastError();
}

hipError_t launch_prune_and_setup(const FrameDev *fd, const FrameDev *fh, int F, hipStream_t s) {
    int gx = 0;
    for (int i = 0; i < F; ++i) gx = fh[i].nprune_blocks > gx ? fh[i].nprune_blocks : gx;
    const bool f64 = fh[0].precision == TDLO_PREC_F64;
    bool reuse = true;                      // every frame's sorted cloud serves as it is: neither prune nor scatter
    for (int i = 0; i < F; ++i) reuse = reuse && fh[i].reuse_sorted;
    if (!reuse) hipLaunchKernelGGL(k_prune_pass1, dim3(gx, F), dim3(kBlock), sizeof(int) * ((fh[0].M + 3) & ~3) + sizeof(double) * (3 * fh[0].M + kPrunePad), s, fd);
    if (f64) hipLaunchKernelGGL((k_setup<double, false>), dim3(F), dim3(kBlock), 0, s, fd, fh[0], 0, (const double *)nullptr, (double *)nullptr, 0, 0);
    else hipLaunchKernelGGL((k_setup<float, false>), dim3(F), dim3(kBlock), 0, s, fd, fh[0], 0, (const double *)nullptr, (double *)nullptr, 0, 0);
    if (!reuse) {
        if (f64) hipLaunchKernelGGL((k_prune_scatter<double>), dim3(gx, F), dim3(kBlock), sizeof(int) * 5 * fh[0].M, s, fd);
        else hipLaunchKernelGGL((k_prune_scatter<float>), dim3(gx, F), dim3(kBlock), sizeof(int) * 5 * fh[0].M, s, fd);
    }
    return hipGetLastError();
}

// One frame, host-supplied block read from pinned host memory by the kernel itself (no copy in front of it): the fused prologue for a cloud of up
// to kFuseMaxBlocks * 256 points on a chain of up to kFuseMaxNodes nodes, or -- when the slot's sorted cloud is reused -- the setup workgroup alone.
bool prologue_direct_ok(const FrameDev &f) {
    return f.reuse_sorted || (f.nprune_blocks <= kFuseMaxBlocks && f.prune_tiles == 1 && f.M <= kFuseMaxNodes);
}
bool prologue_pair_ok(const FrameDev &f) {
    return !f.reuse_sorted && f.nprune_blocks <= kFuseMaxBlocks && f.prune_tiles == 1 && f.M <= kFuseMaxNodes;
}
hipError_t launch_prologue_direct(const FrameDev *fh, const double *host_up, double *dev_up, int up_doubles, int yin_off, unsigned epoch, hipStream_t s,
                                  const FrameDev *f2, const double *host_up2, double *dev_up2, int up_doubles2) {
    const bool f64 = fh[0].precision == TDLO_PREC_F64;
    if (fh[0].reuse_sorted) {
        if (f64) hipLaunchKernelGGL((k_setup<double, true>), dim3(1), dim3(kBlock), 0, s, (const FrameDev *)nullptr, fh[0], 0, host_up, dev_up, up_doubles, yin_off);
        else hipLaunchKernelGGL((k_setup<float, true>), dim3(1), dim3(kBlock), 0, s, (const FrameDev *)nullptr, fh[0], 0, host_up, dev_up, up_doubles, yin_off);
    } else {
        if (f2) {
            const dim3 grid(fh[0].nprune_blocks + 2);
            if (f64) hipLaunchKernelGGL((k_prologue<double, true>), grid, dim3(kBlock), 0, s, fh[0], host_up, dev_up, up_doubles, yin_off, epoch, *f2, host_up2, dev_up2, up_doubles2);
            else hipLaunchKernelGGL((k_prologue<float, true>), grid, dim3(kBlock), 0, s, fh[0], host_up, dev_up, up_doubles, yin_off, epoch, *f2, host_up2, dev_up2, up_doubles2);
        } else {
            const dim3 grid(fh[0].nprune_blocks + 1);
            if (f64) hipLaunchKernelGGL((k_prologue<double, false>), grid, dim3(kBlock), 0, s, fh[0], host_up, dev_up, up_doubles, yin_off, epoch, fh[0], (const double *)nullptr, (double *)nullptr, 0);
            else hipLaunchKernelGGL((k_prologue<float, false>), grid, dim3(kBlock), 0, s, fh[0], host_up, dev_up, up_doubles, yin_off, epoch, fh[0], (const double *)nullptr, (double *)nullptr, 0);
        }
    }
    return hipGetLastError();
}

hipError_t launch_iteration(const FrameDev *fd, const FrameDev *fh, int F, hipStream_t s, int iteration) {
    const bool f64 = fh[0].precision == TDLO_PREC_F64;
    if (fh[0].vis_branch) TDLO_TRY(f64 ? launch_dmin_T<double>(fd, fh, F, s) : launch_dmin_T<float>(fd, fh, F, s));
    TDLO_TRY(f64 ? launch_estep_T<double>(fd, fh, F, s) : launch_estep_T<float>(fd, fh, F, s));
    mstep_parity_hint(iteration);          // (the caller's count of the registration's iterations: the chain M-step fetches that parity's sums only)
    const hipError_t e = f64 ? launch_mstep_T<double>(fd, fh, F, 0, s) : launch_mstep_T<float>(fd, fh, F, 0, s);
    mstep_parity_hint(-1);
    return e;
}

hipError_t launch_iteration_spin(const FrameDev *fd, FrameDev *fh, hipStream_t s_e, hipStream_t s_m, bool first, unsigned *ecount, unsigned *mtag) {
    FrameDev &f = fh[0];
    f.spin_on = 1;
    f.spin_first = first ? 1 : 0; f.spin_wait = *mtag; f.spin_signal = 0;             // E-step: behind the M-step that stored the tag in front of it
    TDLO_TRY(launch_estep_T<float>(fd, fh, 1, s_e));
    *ecount += (unsigned)f.nblkE;
    f.spin_first = 0; f.spin_wait = *ecount; f.spin_signal = ++*mtag;                  // M-step: behind every workgroup of that E-step
    TDLO_TRY(launch_mstep_chain(fd, fh, 1, 0, false, s_m));
    f.spin_on = 0;
    return hipSuccess;
}

hipError_t launch_iteration_timed(const FrameDev *fd, const FrameDev *fh, int F, hipStream_t s, hipEvent_t e_start, hipEvent_t e_stop,
                                  hipEvent_t m_start, hipEvent_t m_stop, int iteration) {
    g_estep_ev[0] = e_start; g_estep_ev[1] = e_stop;
    g_mstep_ev[0] = m_start; g_mstep_ev[1] = m_stop;
    const hipError_t e = launch_iteration(fd, fh, F, s, iteration);
    g_estep_ev[0] = g_estep_ev[1] = nullptr;
    g_mstep_ev[0] = g_mstep_ev[1] = nullptr;
    return e;
}

// which kernel the M-step of these frames dispatches (for the bench line): name and whether its dispatch carries the events
const char *mstep_kernel_name(const FrameDev *fh, int F) {
    const int M = fh[0].M;
    bool any_lle = false;
    for (int i = 0; i < F; ++i) any_lle = any_lle || fh[i].include_lle;
    if (!any_lle && !fh[0].mstep_dense) return M > kChainLdsMaxNodes ? "k_mstep_chain_long" : "k_mstep_chain";
    if (any_lle && fh[0].lle_band) return "k_mstep_band";
    if (M <= 60 && !any_lle) return "k_mstep_fast<MFMA>";
    if (M > kChainLdsMaxNodes) return "k_mstep<1wg>";
    if (!any_lle) return "k_mstep_mcu";
    if (M <= 64) return "k_mstep_fast<pivoted>";
    if (M <= kLdsSolveMaxM) return "k_mstep<LDS>";
    return "k_mstep_pivot_mcu";
}

// kind: 0 E-step, 1 dmin, 2 M-step (from the E-step's accumulators), 3 M-step export-only (split), 4 M-step from global sums (split),
// 5 M-step from the accumulators with the one-shot exchange of the split inside (chain smoother / k_mstep_fast: the caller checks M)
hipError_t launch_estep_only(const FrameDev *fd, const FrameDev *fh, int F, int kind, hipStream_t s) {
    const bool f64 = fh[0].precision == TDLO_PREC_F64;
    switch (kind) {
        case 0: return f64 ? launch_estep_T<double>(fd, fh, F, s) : launch_estep_T<float>(fd, fh, F, s);
        case 1: return f64 ? launch_dmin_T<double>(fd, fh, F, s) : launch_dmin_T<float>(fd, fh, F, s);
        case 2: return f64 ? launch_mstep_T<double>(fd, fh, F, 0, s) : launch_mstep_T<float>(fd, fh, F, 0, s);
        case 3: return f64 ? launch_mstep_T<double>(fd, fh, F, 2, s) : launch_mstep_T<float>(fd, fh, F, 2, s);
        case 4: return f64 ? launch_mstep_T<double>(fd, fh, F, 1, s) : launch_mstep_T<float>(fd, fh, F, 1, s);
        case 5: return f64 ? launch_mstep_T<double>(fd, fh, F, 3, s) : launch_mstep_T<float>(fd, fh, F, 3, s);   // M-step with the one-shot exchange inside
        default: return hipErrorInvalidValue;
    }
}

hipError_t launch_split_setup(const FrameDev *fd, const FrameDev *fh, hipStream_t s) {
    const bool f64 = fh[0].precision == TDLO_PREC_F64;
    hipLaunchKernelGGL(k_prune_pass1, dim3(fh[0].nprune_blocks, 1), dim3(kBlock), sizeof(int) * ((fh[0].M + 3) & ~3) + sizeof(double) * (3 * fh[0].M + kPrunePad), s, fd);
    if (f64) hipLaunchKernelGGL((k_setup<double, false>), dim3(1), dim3(kBlock), 0, s, fd, fh[0], 1, (const double *)nullptr, (double *)nullptr, 0, 0);
    else hipLaunchKernelGGL((k_setup<float, false>), dim3(1), dim3(kBlock), 0, s, fd, fh[0], 1, (const double *)nullptr, (double *)nullptr, 0, 0);
    if (f64) hipLaunchKernelGGL((k_prune_scatter<double>), dim3(fh[0].nprune_blocks, 1), dim3(kBlock), sizeof(int) * 5 * fh[0].M, s, fd);
    else hipLaunchKernelGGL((k_prune_scatter<float>), dim3(fh[0].nprune_blocks, 1), dim3(kBlock), sizeof(int) * 5 * fh[0].M, s, fd);
    return hipGetLastError();
}

// N-split with a device-resident exchange: the per-node minimum goes to / comes from caller-owned device memory as plain
// doubles (what an RCCL all-reduce MIN understands); inside the path it stays in ordered-bits form for atomicMin.
template <typename T>
__global__ void k_split_dmin_xch(const FrameDev *__restrict__ frames, double *__restrict__ xch, int import) {
    const FrameDev &f = frames[0];
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= f.M) return;
    if (import) f.dminbits[m] = Num<T>::bits((T)xch[m]);
    else { const unsigned long long b = f.dminbits[m]; xch[m] = b == ~0ull ? 1e300 : Num<T>::from_bits(b); }   // ~0: no point on this shard
}

hipError_t launch_split_dmin_xch(const FrameDev *fd, const FrameDev *fh, double *xch, int import, hipStream_t s) {
    const dim3 grid((fh[0].M + 63) / 64), block(64);
    if (fh[0].precision == TDLO_PREC_F64) hipLaunchKernelGGL((k_split_dmin_xch<double>), grid, block, 0, s, fd, xch, import);
    else hipLaunchKernelGGL((k_split_dmin_xch<float>), grid, block, 0, s, fd, xch, import);
    return hipGetLastError();
}

hipError_t launch_split_init_pack(const FrameDev *fd, double *init2, hipStream_t s) {
    hipLaunchKernelGGL(k_split_init_pack, dim3(1), dim3(64), 0, s, fd, init2);
    return hipGetLastError();
}
hipError_t launch_split_poll_pack(const FrameDev *fd, double *out2, hipStream_t s) {
    hipLaunchKernelGGL(k_split_poll_pack, dim3(1), dim3(64), 0, s, fd, out2);
    return hipGetLastError();
}
hipError_t launch_split_set_global_dev(const FrameDev *fd, const double *init2, hipStream_t s) {
    hipLaunchKernelGGL(k_split_set_global_dev, dim3(1), dim3(64), 0, s, fd, init2);
    return hipGetLastError();
}
hipError_t launch_xch_init(const FrameDev *fd, hipStream_t s) {
    hipLaunchKernelGGL(k_xch_init, dim3(1), dim3(64), 0, s, fd);
    return hipGetLastError();
}

hipError_t launch_split_set_global(const FrameDev *fd, double Nglob, double Sglob, hipStream_t s) {
    hipLaunchKernelGGL(k_split_set_global, dim3(1), dim3(64), 0, s, fd, Nglob, Sglob);
    return hipGetLastError();
}

int check_device_image() {
    hipFuncAttributes attr;
    const hipError_t e = hipFuncGetAttributes(&attr, (const void *)k_prune_pass1);
    if (e != hipSuccess) fprintf(stderr, "trackdlo_hip: no gfx950 kernel image usable on this device: %s\n", hipGetErrorString(e));
    return e == hipSuccess ? 0 : -1;
}

// test aid (tdlo_debug_exp2): the fp64 E-step's 2^x on an array
__global__ void k_debug_exp2(const double *__restrict__ x, double *__restrict__ y, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = Num<double>::exp2(x[i]);
}
hipError_t launch_debug_exp2(const double *x, double *y, int n, hipStream_t s) {
    hipLaunchKernelGGL(k_debug_exp2, dim3((n + 255) / 256), dim3(256), 0, s, x, y, n);
    return hipGetLastError();
}

}  // namespace tdlo
