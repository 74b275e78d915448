// tdlo_cloud.hip -- depth image -> cloud -> voxel-grid down-sample on the device.
//
// Replaces the step right upstream of tracking_step in the ROS node:
//   trackdlo/src/trackdlo_node.cpp:195-232  masked-pixel back-projection (row-major scan, double arithmetic,
//                                            float storage in pcl::PointXYZRGB)
//   trackdlo/src/trackdlo_node.cpp:235-241  pcl::VoxelGrid<PointXYZRGB>::filter with a cubic leaf, result
//                                            widened to the double matrix X
// pcl::VoxelGrid is third-party (PCL 1.10, not under the reference tree); the algorithm implemented is the
// published one of pcl/filters/impl/voxel_grid.hpp (the test suite's CPU restatement performs the float operations in
// the same order, so the two agree bit for bit): bounding box -> integer cell index -> sort by cell ->
// one centroid per occupied cell, ascending cell index, float sums in input order.
//
// Integer/byte work, HBM-bound and tiny (0.3 M pixels): no MFMA, no LDS tiling to speak of.  Design:
//   * the bounding-box pass also counts the masked pixels per 1024-pixel tile; the key pass compacts them in pixel
//     order (32-bit key = cell index, payload = pixel number); a stable LSD radix sort (8-bit digits, only as many
//     passes as the cell count needs) over the masked pixels only groups the cells;
//   * the sort is three small kernels per pass: per-block digit histogram, one-block exclusive scan, stable
//     scatter (wave-level match by ballots, wave/round offsets through LDS) -- element order is the pixel order;
//   * centroids: the thread of a cell's first element walks the cell's run (runs are short: a handful of pixels
//     per 8 mm cell), recomputing the points from the depth image with the arithmetic of the back-projection.
// The result is written as the slot's raw cloud (N x 3 column-major doubles), i.e. exactly what tdlo_set_cloud
// would have uploaded, so cpd_lle / tracking_step run on it without a host round trip of the cloud.
#include "tdlo_internal.h"
#include "tdlo_devcommon.h"
#include <cstdint>
#include <cstdlib>

namespace tdlo {

namespace {

constexpr int kCB = 256;            // threads per block
constexpr int kItems = 4;           // elements per thread (rounds)
constexpr int kTile = kCB * kItems; // elements per block
constexpr unsigned kSent = 0xffffffffu;

struct Cam { double fx, fy, cx, cy; };

__device__ __forceinline__ void back_project(const unsigned short *__restrict__ depth, int p, int cols, const Cam cam, float &x, float &y, float &z) {
    const int i = p / cols, j = p - i * cols;
    const double pc_z = (double)depth[p] / 1000.0;                  // trackdlo_node.cpp:220
    x = (float)(((double)j - cam.cx) * pc_z / cam.fx);              // :222
    y = (float)(((double)i - cam.cy) * pc_z / cam.fy);              // :223
    z = (float)pc_z;                                                // :224
}

__device__ __forceinline__ unsigned ordered_bits(float v) {
    const unsigned u = __float_as_uint(v);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// bbox[0..2] = min (ordered bits), bbox[3..5] = max, bbox[6] = number of masked pixels
__global__ __launch_bounds__(kCB) void k_cloud_bbox(const unsigned short *__restrict__ depth, const unsigned char *__restrict__ mask, int P, int cols,
                                                    const Cam cam, unsigned *__restrict__ bbox, int *__restrict__ blkcnt) {
    __shared__ int wc[4];
    unsigned mn[3] = {~0u, ~0u, ~0u}, mx[3] = {0u, 0u, 0u};
    int cnt = 0;
#pragma unroll
    for (int u = 0; u < kItems; ++u) {
        const int p = blockIdx.x * kTile + u * kCB + threadIdx.x;
        if (p < P && mask[p] != 0) {                                // :212
            float v[3];
            back_project(depth, p, cols, cam, v[0], v[1], v[2]);
#pragma unroll
            for (int d = 0; d < 3; ++d) { const unsigned o = ordered_bits(v[d]); mn[d] = o < mn[d] ? o : mn[d]; mx[d] = o > mx[d] ? o : mx[d]; }
            ++cnt;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const unsigned a = __shfl_xor(mn[d], o), b = __shfl_xor(mx[d], o);
            mn[d] = a < mn[d] ? a : mn[d]; mx[d] = b > mx[d] ? b : mx[d];
        }
        cnt += __shfl_xor(cnt, o);
    }
    // (the four waves' boxes meet in LDS: seven atomics per tile, not per wave -- the same words are the target of every tile of the image, see cloud_phase_a)
    __shared__ unsigned wb[4][6];
    if ((threadIdx.x & 63) == 0) {
        wc[threadIdx.x >> 6] = cnt;
#pragma unroll
        for (int d = 0; d < 3; ++d) { wb[threadIdx.x >> 6][d] = mn[d]; wb[threadIdx.x >> 6][3 + d] = mx[d]; }
    }
    __syncthreads();
    const int tot = wc[0] + wc[1] + wc[2] + wc[3];
    if (threadIdx.x < 6 && tot > 0) {
        const int d = threadIdx.x;
        unsigned v = wb[0][d];
#pragma unroll
        for (int i = 1; i < 4; ++i) { const unsigned u = wb[i][d]; v = d < 3 ? (u < v ? u : v) : (u > v ? u : v); }
        if (d < 3) atomicMin(&bbox[d], v); else atomicMax(&bbox[d], v);
    }
    if (threadIdx.x == 6 && tot > 0) atomicAdd(&bbox[6], (unsigned)tot);
    if (threadIdx.x == 0) blkcnt[blockIdx.x] = tot;     // masked pixels of this 1024-pixel tile
}

struct Grid { int min_b[3]; int mul1, mul2; float inv; int nodown; };

// Compaction + keys: the masked pixels of tile b go to positions blkoff[b] .. in pixel order (the order of the reference's
// row-major scan, trackdlo_node.cpp:197-198); key = cell index, payload = pixel number.
__global__ __launch_bounds__(kCB) void k_cloud_keys(const unsigned short *__restrict__ depth, const unsigned char *__restrict__ mask, int P, int cols,
                                                    const Cam cam, const Grid g, const int *__restrict__ blkoff,
                                                    unsigned *__restrict__ key, unsigned *__restrict__ val) {
    __shared__ int wc[4];
    __shared__ int run;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    if (t == 0) run = blkoff[blockIdx.x];
    __syncthreads();
#pragma unroll 1
    for (int u = 0; u < kItems; ++u) {
        const int p = blockIdx.x * kTile + u * kCB + t;
        const bool on = p < P && mask[p] != 0;
        const unsigned long long bl = __ballot(on);
        const unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
        if (lane == 0) wc[w] = __popcll(bl);
        __syncthreads();
        if (on) {
            int dst = run + __popcll(bl & below);
            for (int ww = 0; ww < w; ++ww) dst += wc[ww];
            unsigned k = 0;                                         // "leaf size too small": output = input, pixel order
            if (!g.nodown) {
                float x, y, z;
                back_project(depth, p, cols, cam, x, y, z);
                const int i0 = (int)(floorf(x * g.inv) - (float)g.min_b[0]);     // voxel_grid.hpp: ijk = floor(p * inv_leaf) - min_b
                const int i1 = (int)(floorf(y * g.inv) - (float)g.min_b[1]);
                const int i2 = (int)(floorf(z * g.inv) - (float)g.min_b[2]);
                k = (unsigned)(i0 + i1 * g.mul1 + i2 * g.mul2);
            }
            key[dst] = k; val[dst] = (unsigned)p;
        }
        __syncthreads();
        if (t == 0) run += wc[0] + wc[1] + wc[2] + wc[3];
        __syncthreads();
    }
}

// hist[d * nblk + b] = number of elements of block b whose digit is d
__global__ __launch_bounds__(kCB) void k_radix_hist(const unsigned *__restrict__ key, int P, int shift, int nblk, int *__restrict__ hist) {
    __shared__ int h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
#pragma unroll
    for (int u = 0; u < kItems; ++u) {
        const int e = blockIdx.x * kTile + u * kCB + threadIdx.x;
        if (e < P) atomicAdd(&h[(key[e] >> shift) & 255u], 1);
    }
    __syncthreads();
    hist[threadIdx.x * nblk + blockIdx.x] = h[threadIdx.x];
}

// in-place exclusive scan of data[0..n) by ONE block of 1024 threads; *total = sum.  The array is walked in tiles of
// 4096 elements, four consecutive ints per thread (coalesced 16-byte accesses), with a running carry.
__global__ __launch_bounds__(1024) void k_scan_single(int *__restrict__ data, int n, int *__restrict__ total) {
    __shared__ int wsum[16];
    __shared__ int carry_s;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    if (t == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < n; base += 4096) {
        const int i0 = base + 4 * t;
        int v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = (i0 + k < n) ? data[i0 + k] : 0;
        const int s = v[0] + v[1] + v[2] + v[3];
        int incl = s;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(incl, o); if (lane >= o) incl += u; }
        if (lane == 63) wsum[w] = incl;
        __syncthreads();
        int woff = 0;
#pragma unroll
        for (int i = 0; i < 16; ++i) woff += (i < w) ? wsum[i] : 0;
        int tilesum = 0;
#pragma unroll
        for (int i = 0; i < 16; ++i) tilesum += wsum[i];
        int run = carry_s + woff + incl - s;
#pragma unroll
        for (int k = 0; k < 4; ++k) { if (i0 + k < n) data[i0 + k] = run; run += v[k]; }
        __syncthreads();
        if (t == 0) carry_s += tilesum;
        __syncthreads();
    }
    if (t == 0 && total) *total = carry_s;
}

// stable scatter of one radix pass; `hist` holds the scanned offsets
__global__ __launch_bounds__(kCB) void k_radix_scatter(const unsigned *__restrict__ key_in, const unsigned *__restrict__ val_in, int P, int shift, int nblk,
                                                       const int *__restrict__ hist, unsigned *__restrict__ key_out, unsigned *__restrict__ val_out) {
    __shared__ int base[256];
    __shared__ int wcnt[4][256];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    base[t] = hist[t * nblk + blockIdx.x];
#pragma unroll
    for (int i = 0; i < 4; ++i) wcnt[i][t] = 0;
    __syncthreads();
#pragma unroll 1
    for (int u = 0; u < kItems; ++u) {
        const int e = blockIdx.x * kTile + u * kCB + t;
        const bool valid = e < P;
        const unsigned k = valid ? key_in[e] : 0u, v = valid ? val_in[e] : 0u;
        const unsigned d = (k >> shift) & 255u;
        unsigned long long m = __ballot(valid);
#pragma unroll
        for (int bit = 0; bit < 8; ++bit) {                          // lanes of this wave holding the same digit
            const unsigned long long bl = __ballot((d >> bit) & 1u);
            m &= ((d >> bit) & 1u) ? bl : ~bl;
        }
        const unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
        const int rank = __popcll(m & below);
        if (valid && rank == 0) wcnt[w][d] = __popcll(m);
        __syncthreads();
        if (valid) {
            int off = base[d] + rank;
            for (int ww = 0; ww < w; ++ww) off += wcnt[ww][d];
            key_out[off] = k; val_out[off] = v;
        }
        __syncthreads();
        base[t] += wcnt[0][t] + wcnt[1][t] + wcnt[2][t] + wcnt[3][t];
#pragma unroll
        for (int i = 0; i < 4; ++i) wcnt[i][t] = 0;
        __syncthreads();
    }
}

__device__ __forceinline__ bool is_head(const unsigned *__restrict__ key, int i, int nodown) {
    const unsigned k = key[i];
    if (k == kSent) return false;
    return nodown || i == 0 || key[i - 1] != k;
}

__global__ __launch_bounds__(kCB) void k_cloud_heads(const unsigned *__restrict__ key, int P, int nodown, int *__restrict__ cnt) {
    __shared__ int ws[4];
    int c = 0;
#pragma unroll
    for (int u = 0; u < kItems; ++u) {
        const int e = blockIdx.x * kTile + u * kCB + threadIdx.x;
        if (e < P && is_head(key, e, nodown)) ++c;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) cnt[blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
}

// one output point per cell; X is the slot's raw cloud: x[0..n) y[0..n) z[0..n) as doubles, n = *total
__global__ __launch_bounds__(kCB) void k_cloud_centroid(const unsigned short *__restrict__ depth, int P, int cols, const Cam cam, int nodown,
                                                        const unsigned *__restrict__ key, const unsigned *__restrict__ val,
                                                        const int *__restrict__ cnt_scan, const int *__restrict__ total, int cap, double *__restrict__ X) {
    __shared__ int ws[4];
    __shared__ int run;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int n = *total;
    if (n > cap) return;                                             // host reports the error
    if (t == 0) run = cnt_scan[blockIdx.x];
    __syncthreads();
#pragma unroll 1
    for (int u = 0; u < kItems; ++u) {
        const int e = blockIdx.x * kTile + u * kCB + t;
        const bool head = e < P && is_head(key, e, nodown);
        const unsigned long long bl = __ballot(head);
        const unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
        if (lane == 0) ws[w] = __popcll(bl);
        __syncthreads();
        int out = run + __popcll(bl & below);
        for (int ww = 0; ww < w; ++ww) out += ws[ww];
        if (head) {
            const unsigned k = key[e];
            float sx = 0.0f, sy = 0.0f, sz = 0.0f;
            int q = e;
            do {                                                     // CentroidPoint: float sums in input order
                float x, y, z;
                back_project(depth, (int)val[q], cols, cam, x, y, z);
                sx += x; sy += y; sz += z;
                ++q;
            } while (!nodown && q < P && key[q] == k);
            const float c = (float)(q - e);
            X[out] = (double)__fdiv_rn(sx, c); X[(size_t)n + out] = (double)__fdiv_rn(sy, c); X[2 * (size_t)n + out] = (double)__fdiv_rn(sz, c);
        }
        __syncthreads();
        if (t == 0) run += ws[0] + ws[1] + ws[2] + ws[3];
        __syncthreads();
    }
}

}  // namespace

// =================================================================================================================================================
// The whole of depth image -> cloud -> voxel grid in ONE launch (round 5; VERDICT r04 item 3: the multi-launch form above is a bounding-box launch,
// a host round trip for the masked-pixel count and the box, and ~15 more dependent launches to sort a few thousand keys: 0.28 ms per 640 x 480 frame,
// 4.7 x the whole tracking_step behind it).
//
//   phase A, every workgroup (1024 threads, a tile of 4096 pixels): mask -> compaction in pixel order inside the tile (thread = 4 consecutive
//     pixels, one 32-bit mask word), back-projection of the masked pixels (the arithmetic of trackdlo_node.cpp:220-224), the tile's points
//     {x, y, z} as floats into the tile's own region of the workspace (agent-scope 16-byte stores), the tile's count, the bounding box by atomic
//     min / max on order-preserving bits -- then a ticket.  Nobody waits for anybody: the workgroup that draws the LAST ticket carries on alone
//     (no co-residency assumption, nothing to time out).
//   phase B, that one workgroup: tile offsets (block scan of the counts), the grid of pcl::VoxelGrid from the box (the float arithmetic of
//     voxel_grid.hpp's applyFilter, as the host does it for the multi-launch form), every point's cell index, and
//     a stable LSD radix sort of (cell index << rb | rank in pixel order) -- ONE 32-bit word per point, all of it in LDS (up to 32 704 points, the
//     words of a thread in registers between the phases of a pass): 4-bit digits, thread t owns R consecutive words, so equal digits are ordered
//     by (thread, word of the thread) and no lane needs another to rank its words (matching the lanes of a wave by digit costs 8 ballots and
//     ~320 clocks a 64-word round; LDS atomics on a digit most of a round's lanes share -- neighbouring pixels fall into one cell -- serialise):
//     byte counters per (digit, thread) by LDS atomics, one block scan in digit-major order, places by LDS atomics with return.
//     Then the coordinates in sorted order into LDS (all three at once up to 10 900 points, else one at a time), one thread per run of
//     consecutive sorted positions: centroids as float sums in input (= pixel) order straight out of LDS, written as the slot's raw cloud, and the
//     counts to a pinned results word the host waits on.
//   Not taken (the last workgroup says so and the host runs the multi-launch form): more than 32 704 masked pixels, cell-index bits + rank bits
//   beyond 32, "leaf size too small" (voxel_grid.hpp's pass-through), more than 4095 tiles.
// The images are read where they are: device memory after a copy, or -- tdlo_image_buffers -- pinned host memory the caller filled (the mask is
// read once, coalesced, straight over PCIe; depth only where the mask is set).
// Bit-exact to the multi-launch form and to the oracle: the same float operations in the same order.
namespace {

constexpr int kFT = 1024;                // threads per workgroup
constexpr int kFW = kFT / 64;            // waves
constexpr int kFPix = 4 * kFT;           // pixels per tile
constexpr int kFNmax = 32704;            // points the in-LDS sort takes (the CU's 160 KB: 4 bytes a point, 32 KB of counters, ~150 bytes of statics)
constexpr int kFR = (kFNmax + kFT - 1) / kFT;      // sort words per thread (register resident between the phases of a pass): 32
constexpr int kFTmax = 4095;             // tiles (their offsets live in the counters' area before the sort)
constexpr size_t kFLdsE = (size_t)kFNmax * 4, kFLdsCnt = 16 * kFT * 2;
static_assert(kFLdsCnt >= (size_t)(kFTmax + 1) * 4 && kFLdsCnt >= 8 * 512, "the tile offsets and the head bits live in the counters' area");
constexpr size_t kFLds = kFLdsE + kFLdsCnt;         // 163 584 B; with the kernel's static LDS (132 B) just inside the CU's 163 840

struct FusedCloud {
    const unsigned short *depth; const unsigned char *mask;     // padded to 8 bytes beyond P pixels
    int P, cols, T;
    Cam cam;
    float inv_leaf;
    float *ex, *ey, *ez;                 // P each: tile b's points at [b * kFPix ...)
    float *cx, *cy, *cz;                 // kFNmax each: the same points compacted (rank = position in pixel order)
    int *tcnt;                           // T
    unsigned *tw0, *tw1;                 // kFNmax each: the team kernel's sort words, ping and pong
    int *thist;                          // kTK x 256 digit counts + kTK head counts of the team kernel
    unsigned *state;                     // [0..2] box min, [3..5] box max (order-preserving bits), [6] tickets; re-armed by the last workgroup
    double *X; int cap;                  // the slot's raw cloud and its capacity in points
    unsigned long long *res;             // pinned host: [1] = n_raw << 32 | n, then [0] = epoch << 32 | status (1 done, 2 not taken, 3 capacity)
    unsigned epoch;
    // the visibility pre-pass of the same frame (trackdlo_node.cpp:257-277) inside the team kernel: visM nodes (<= 64; 0: not asked for) read from pinned host
    // memory, every team member takes the minima over its own centroids (atomic min on visState, armed with +inf), the member that finishes last hands
    // them to pinned host memory (visOut[1 + m], the layout of k_node_min_dist_direct) and re-arms
    const double *visY; int visM; unsigned long long *visState, *visOut;
    int hook;                            // test hook (TDLO_CLOUD_TEAM_FORCE_TIMEOUT): the team's last workgroup leaves before its first barrier, the others wait out the limit
};

__device__ __forceinline__ float ordered_decode(unsigned o) { return __uint_as_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o); }

// inclusive scan over the 64 lanes: four DPP row shifts inside the rows of 16 lanes (lanes without a source read 0), then the totals of the rows in
// front by three v_readlane -- ~100 clocks where six trips through the LDS crossbar (__shfl_up) took 576 (scripts/ubench/wg1024.hip)
__device__ __forceinline__ int wave_incl_scan_i(int v, int lane) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);      // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);      // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);      // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);      // row_shr:8
    const int r0 = __builtin_amdgcn_readlane(v, 15), r1 = __builtin_amdgcn_readlane(v, 31), r2 = __builtin_amdgcn_readlane(v, 47);
    return v + (lane >= 16 ? r0 : 0) + (lane >= 32 ? r1 : 0) + (lane >= 48 ? r2 : 0);
}
// exclusive scan over the 1024 threads (wtot: 16 ints of LDS, free again on return); *total = sum
__device__ __forceinline__ int block_excl_scan_i(int v, int *wtot, int t, int *total) {
    const int lane = t & 63, w = t >> 6;
    const int incl = wave_incl_scan_i(v, lane);
    if (lane == 63) wtot[w] = incl;
    __syncthreads();
    int off = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < kFW; ++i) { const int x = wtot[i]; off += i < w ? x : 0; tot += x; }
    __syncthreads();
    *total = tot;
    return off + incl - v;
}
__device__ __forceinline__ int bits_for(unsigned long long values) { return values <= 2ull ? 1 : 64 - __builtin_clzll(values - 1ull); }   // bits that hold 0 .. values - 1
// smallest position >= from whose head bit is set, n if there is none (H: one bit per sorted position, nothing set at or beyond n)
__device__ __forceinline__ int next_head(const unsigned long long *H, int from, int n) {
    if (from >= n) return n;
    int wi = from >> 6;
    unsigned long long m = H[wi] >> (from & 63);
    if (m) return from + (int)__builtin_ctzll(m);
    const int nw = (n + 63) >> 6;
    for (++wi; wi < nw; ++wi) { m = H[wi]; if (m) return (wi << 6) + (int)__builtin_ctzll(m); }
    return n;
}

struct FusedGrid { int min_b[3]; int mul1, mul2, rb, kb, take; };

// pcl/filters/impl/voxel_grid.hpp applyFilter: leaf-size check, min_b / div_b / divb_mul in float arithmetic, from the bounding box (ordered bits) and the
// point count; g.take = 0 when the in-LDS sort cannot serve the frame (too many points, cell-index bits + rank bits beyond 32, PCL's pass-through)
__device__ __forceinline__ void cloud_grid(const unsigned *sbox, int n, float inv, FusedGrid &g) {
    int min_b[3], div_b[3];
    bool take = n <= kFNmax;
    long long ddp = 1;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const float mn = ordered_decode(sbox[d]), mx = ordered_decode(sbox[3 + d]);
        const float ext = (mx - mn) * inv;
        if (!(ext >= 0.0f && ext < 2147483000.0f)) take = false;             // (the host's arithmetic decides what happens out there)
        const long long dd = (long long)(take ? ext : 0.0f) + 1;
        ddp = (ddp > 2147483647ll || dd > 2147483647ll) ? 4294967296ll : ddp * dd;
        const float lo = floorf(mn * inv), hi = floorf(mx * inv);
        if (!(lo > -2147483000.0f && hi < 2147483000.0f)) take = false;
        min_b[d] = take ? (int)lo : 0;
        div_b[d] = take ? (int)hi - min_b[d] + 1 : 1;
    }
    if (ddp > 2147483647ll) take = false;                                     // "leaf size too small": the pass-through of the multi-launch form
    const long long cells = (long long)div_b[0] * div_b[1] * div_b[2];
    const int rb = bits_for((unsigned long long)n), kb = bits_for((unsigned long long)(cells > 0 ? cells : 1));
    if (cells <= 0 || cells > 2147483647ll || rb + kb > 32) take = false;
    g.min_b[0] = min_b[0]; g.min_b[1] = min_b[1]; g.min_b[2] = min_b[2];
    g.mul1 = div_b[0]; g.mul2 = div_b[0] * div_b[1]; g.rb = rb; g.kb = kb; g.take = take ? 1 : 0;
}

// phase A of the one-launch kernels, every workgroup: its tile of 4096 pixels -- compaction in pixel order, back-projection, the points to the tile's
// region, count, bounding box -- and a ticket (returned to every thread: 0 .. T - 1 in the order the workgroups finished)
__device__ __forceinline__ unsigned cloud_phase_a(const FusedCloud &a, int *wtot, unsigned *sticket, unsigned *sbx /* kFW x 6 words of LDS nobody uses yet */) {
    const int t = threadIdx.x, lane = t & 63, b = blockIdx.x;
#ifdef TDLO_CLOUD_STAMPS      // wall-clock (100 MHz) split of phase A in the middle tile: words 12 .. 18 behind the state words (scripts/archive/gpu_cloud_stamps.py)
#define ASTAMP(i) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); if (t == 0 && b == a.T / 2) ((unsigned long long *)(a.state + 16))[12 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define ASTAMP(i) do { } while (0)
#endif
    ASTAMP(0);
    {
        const int p0 = b * kFPix + 4 * t;
        unsigned m4 = 0;
        if (p0 < a.P) m4 = *(const unsigned *)(a.mask + p0);
        ASTAMP(1);
        if (p0 + 4 > a.P) {                                                   // (the last word of the image: bytes beyond it do not count)
            const int keep = a.P - p0;
            m4 = keep <= 0 ? 0u : (m4 & (0xffffffffu >> (8 * (4 - keep))));
        }
        const int k0 = (m4 & 0xffu) != 0u, k1 = (m4 & 0xff00u) != 0u, k2 = (m4 & 0xff0000u) != 0u, k3 = (m4 & 0xff000000u) != 0u;
        const int c = k0 + k1 + k2 + k3;
        uint2 d4 = make_uint2(0u, 0u);
        if (c) d4 = *(const uint2 *)(a.depth + p0);
        ASTAMP(2);
        int total;
        int rnk = block_excl_scan_i(c, wtot, t, &total);
        ASTAMP(3);
        unsigned mn[3] = {~0u, ~0u, ~0u}, mx[3] = {0u, 0u, 0u};
        if (c) {
            int i = p0 / a.cols, j = p0 - i * a.cols;
            const size_t dst = (size_t)b * kFPix;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const bool on = (k == 0 ? k0 : (k == 1 ? k1 : (k == 2 ? k2 : k3))) != 0;
                if (on) {
                    const unsigned dv = ((k < 2 ? d4.x : d4.y) >> (16 * (k & 1))) & 0xffffu;
                    const double pc_z = (double)dv / 1000.0;                              // trackdlo_node.cpp:220
                    const float x = (float)(((double)j - a.cam.cx) * pc_z / a.cam.fx);    // :222
                    const float y = (float)(((double)i - a.cam.cy) * pc_z / a.cam.fy);    // :223
                    const float z = (float)pc_z;                                          // :224
                    __hip_atomic_store(a.ex + dst + rnk, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(a.ey + dst + rnk, y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(a.ez + dst + rnk, z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ++rnk;
                    const unsigned o0 = ordered_bits(x), o1 = ordered_bits(y), o2 = ordered_bits(z);
                    mn[0] = o0 < mn[0] ? o0 : mn[0]; mx[0] = o0 > mx[0] ? o0 : mx[0];
                    mn[1] = o1 < mn[1] ? o1 : mn[1]; mx[1] = o1 > mx[1] ? o1 : mx[1];
                    mn[2] = o2 < mn[2] ? o2 : mn[2]; mx[2] = o2 > mx[2] ? o2 : mx[2];
                }
                if (++j == a.cols) { j = 0; ++i; }
            }
        }
        ASTAMP(4);
        // the tile's box: every wave's six values by DPP (six trips each through the LDS crossbar before), the sixteen waves' through LDS, then SIX atomics
        // per tile -- one per wave with a masked pixel had been ~6 000 read-modify-writes on six words per 640 x 480 image, serialised where
        // the device's atomics meet: most of this phase's 23 us (scripts/archive/gpu_cloud_stamps.py)
        {
            const bool any = __ballot(c != 0) != 0ull;                        // (wave-uniform)
            unsigned r6[6] = {~0u, ~0u, ~0u, 0u, 0u, 0u};
            if (any) {
#pragma unroll
                for (int d = 0; d < 3; ++d) { r6[d] = wave_minmax_u32<false>(mn[d]); r6[3 + d] = wave_minmax_u32<true>(mx[d]); }
            }
            if (lane < 6) sbx[6 * (t >> 6) + lane] = lane == 0 ? r6[0] : (lane == 1 ? r6[1] : (lane == 2 ? r6[2] : (lane == 3 ? r6[3] : (lane == 4 ? r6[4] : r6[5]))));
        }
        __syncthreads();
        if (t < 6 && total > 0) {
            unsigned v = sbx[t];
#pragma unroll
            for (int i = 1; i < kFW; ++i) { const unsigned u = sbx[6 * i + t]; v = t < 3 ? (u < v ? u : v) : (u > v ? u : v); }
            if (t < 3) (void)__hip_atomic_fetch_min(a.state + t, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else (void)__hip_atomic_fetch_max(a.state + t, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (t == 0) __hip_atomic_store(a.tcnt + b, total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                      // this wave's stores and atomics have been performed
        __syncthreads();
        ASTAMP(5);
        if (t == 0) *sticket = __hip_atomic_fetch_add(a.state + 6, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        ASTAMP(6);
    }
#undef ASTAMP
    return *sticket;
}



__global__ __launch_bounds__(kFT) void k_cloud_fused(const FusedCloud a) {
    extern __shared__ __attribute__((aligned(16))) char fsm[];
    unsigned *E = (unsigned *)fsm;                                           // kFNmax sort words (through sw(): bank swizzle)
    unsigned *cnt32 = (unsigned *)(fsm + kFLdsE);                            // 32 KB: per (digit, thread) counts as bytes, then start offsets as 16-bit words
    int *toff = (int *)cnt32;                                                // (before the sort) T + 1 tile offsets
    __shared__ int wtot[kFW];
    __shared__ unsigned sbox[8];
    __shared__ unsigned sticket;
    __shared__ FusedGrid sgrid;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
#ifdef TDLO_CLOUD_STAMPS      // phase stamps of the last workgroup (instrumented build only, scripts/archive/gpu_cloud_stamps.py): 64-bit words behind the state words
#define FSTAMP(i) do { __syncthreads(); if (t == 0) ((unsigned long long *)(a.state + 16))[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define FSTAMP(i) do { } while (0)
#endif

    // ================= phase A: this tile
    if (cloud_phase_a(a, wtot, &sticket, (unsigned *)fsm) != (unsigned)(a.T - 1)) return;

    // ================= phase B: the workgroup with the last ticket
    FSTAMP(0);
    // ---- the tiles' offsets, the count, the box
    int n;
    {
        int carry = 0;
        for (int b0 = 0; b0 < a.T; b0 += kFT) {
            const int bb = b0 + t;
            const int c = bb < a.T ? __hip_atomic_load(a.tcnt + bb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
            if (b0 == 0 && t < 6) sbox[t] = __hip_atomic_load(a.state + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (the same round trip)
            int tot;
            const int ex = block_excl_scan_i(c, wtot, t, &tot);
            if (bb < a.T) toff[bb] = carry + ex;
            carry += tot;
        }
        n = carry;
        if (t == 0) toff[a.T] = n;
    }
    // ---- pcl's grid from the box (one thread; the others wait)
    const float inv = a.inv_leaf;
    if (t == 0 && n > 0) cloud_grid(sbox, n, inv, sgrid);
    __syncthreads();
    FSTAMP(1);
    auto finish = [&](unsigned status, int n_out) __attribute__((always_inline)) {
        // re-arm the box and the tickets for the next launch; the counts to the host, then the word it waits on
        if (t < 3) __hip_atomic_store(a.state + t, ~0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else if (t < 7) __hip_atomic_store(a.state + t, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                                     // every wave's stores to X have been performed
        if (t == 0) {
            __hip_atomic_store(a.res + 1, ((unsigned long long)(unsigned)n << 32) | (unsigned)n_out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(a.res, ((unsigned long long)a.epoch << 32) | status, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    };
    if (n == 0) { finish(1u, 0); return; }
    if (!sgrid.take) { finish(2u, 0); return; }
    const int mb0 = sgrid.min_b[0], mb1 = sgrid.min_b[1], mb2 = sgrid.min_b[2], mul1 = sgrid.mul1, mul2 = sgrid.mul2, rb = sgrid.rb, kb = sgrid.kb;
    // word i of the sort lives at E[sw(i)]: thread t owns the R consecutive words [t R, t R + R) and the 64 lanes of a wave read them R words apart --
    // with the 32 words of a block rotated by the block's number they fall into different banks whatever R is
    auto sw = [](int i) __attribute__((always_inline)) { return i ^ ((i >> 5) & 31); };
    const int R = (n + kFT - 1) / kFT;                                        // sort words per thread (the last threads' words beyond n do not exist)

    // ---- where every rank's point lies: E[rank] = its index in the tiles' regions
    for (int bb = w; bb < a.T; bb += kFW) {
        const int o = toff[bb], c = toff[bb + 1] - o;
        for (int i = lane; i < c; i += 64) E[o + i] = (unsigned)(bb * kFPix + i);
    }
    __syncthreads();
    FSTAMP(2);
    // ---- gather (rank = r * 1024 + t: coalesced), cell index, sort word; the compacted copy the centroids will read.
    //      (One CU pulls about 10 bytes a clock from memory: 12 bytes a point, requested four points at a time, is what this step costs.)
    {
        unsigned src[4], word[4];
        float px[4], py[4], pz[4];
        for (int r0 = 0; r0 < R; r0 += 4) {
#pragma unroll
            for (int k = 0; k < 4; ++k) { const int g = (r0 + k) * kFT + t; src[k] = g < n ? E[g] : 0u; }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                px[k] = __hip_atomic_load(a.ex + src[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                py[k] = __hip_atomic_load(a.ey + src[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                pz[k] = __hip_atomic_load(a.ez + src[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int g = (r0 + k) * kFT + t;
                word[k] = 0u;
                if (g < n) {
                    const int i0 = (int)(floorf(px[k] * inv) - (float)mb0);      // voxel_grid.hpp: ijk = floor(p * inv_leaf) - min_b
                    const int i1 = (int)(floorf(py[k] * inv) - (float)mb1);
                    const int i2 = (int)(floorf(pz[k] * inv) - (float)mb2);
                    word[k] = ((unsigned)(i0 + i1 * mul1 + i2 * mul2) << rb) | (unsigned)g;
                    a.cx[g] = px[k]; a.cy[g] = py[k]; a.cz[g] = pz[k];
                }
            }
            __syncthreads();                                                 // (the source map's words of this batch have been read by everybody: the sort words take their place, swizzled)
#pragma unroll
            for (int k = 0; k < 4; ++k) { const int g = (r0 + k) * kFT + t; if (g < n) E[sw(g)] = word[k]; }
            __syncthreads();
        }
    }
    FSTAMP(3);
    // ---- stable LSD radix sort of the n words on the cell-index bits [rb, rb + kb), 4 bits a pass.  Thread t owns the words [t R, t R + R): the
    //      order of equal digits is (thread, word of the thread), so a thread needs no other lane to rank its words -- per pass: its words into
    //      registers; per (digit, thread) byte counters by LDS atomics without return (four threads share a 32-bit word); one block scan over the
    //      16 x 1024 counters in digit-major order leaves every (digit, thread)'s start as a 16-bit word; each word's place is what an LDS atomic
    //      WITH return hands back from that start (the atomics of a lane are performed in program order: the thread's words keep their order)
    unsigned reg[kFR];
    for (int sh = 0; sh < kb; sh += 4) {
        const int shift = rb + sh;
#pragma unroll
        for (int r0 = 0; r0 < kFR; r0 += 4)
            if (r0 < R) {                                                    // (wave-uniform)
#pragma unroll
                for (int k = 0; k < 4; ++k) { const int i = t * R + r0 + k; reg[r0 + k] = (r0 + k < R && i < n) ? E[sw(i)] : 0u; }
            }
        {   // counters to zero: 16 KB
            uint4 *z = (uint4 *)cnt32;
            z[t] = make_uint4(0u, 0u, 0u, 0u);
        }
        __syncthreads();
#pragma unroll
        for (int r0 = 0; r0 < kFR; r0 += 4)
            if (r0 < R) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int i = t * R + r0 + k;
                    if (r0 + k < R && i < n) {
                        const unsigned d = (reg[r0 + k] >> shift) & 15u;
                        (void)__hip_atomic_fetch_add(cnt32 + ((d * kFT + t) >> 2), 1u << (8 * (t & 3)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                }
            }
        __syncthreads();
        if (sh == 0) FSTAMP(16);
        {   // exclusive scan in (digit, thread) order: scan thread t holds the counters 16 t .. 16 t + 15
            const uint4 c4 = ((const uint4 *)cnt32)[t];
            const unsigned cw[4] = {c4.x, c4.y, c4.z, c4.w};
            int c[16], sum = 0;
#pragma unroll
            for (int i = 0; i < 16; ++i) { c[i] = (int)((cw[i >> 2] >> (8 * (i & 3))) & 255u); sum += c[i]; }
            int tot;
            int run = block_excl_scan_i(sum, wtot, t, &tot);                 // (its barriers also separate the reads of the byte counters from the stores below)
            unsigned o[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) { const unsigned lo = (unsigned)run; run += c[2 * i]; const unsigned hi = (unsigned)run; run += c[2 * i + 1]; o[i] = lo | (hi << 16); }
            uint4 *ob = (uint4 *)cnt32;
            ob[2 * t] = make_uint4(o[0], o[1], o[2], o[3]);
            ob[2 * t + 1] = make_uint4(o[4], o[5], o[6], o[7]);
        }
        __syncthreads();
        if (sh == 0) FSTAMP(17);
#pragma unroll
        for (int r0 = 0; r0 < kFR; r0 += 4)
            if (r0 < R) {
                unsigned old[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int i = t * R + r0 + k;
                    old[k] = 0u;
                    if (r0 + k < R && i < n) {
                        const unsigned d = (reg[r0 + k] >> shift) & 15u;
                        old[k] = __hip_atomic_fetch_add(cnt32 + ((d * kFT + t) >> 1), 1u << (16 * (t & 1)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int i = t * R + r0 + k;
                    if (r0 + k < R && i < n) E[sw((int)((old[k] >> (16 * (t & 1))) & 0xffffu))] = reg[r0 + k];
                }
            }
        __syncthreads();
        FSTAMP(8 + (sh >> 2));
    }
    FSTAMP(4);
    // ---- one output point per cell, ascending cell index.  The sorted words into registers once more, position p = w S + r 64 + lane this time
    //      (a wave-round = 64 consecutive positions), and the head bits: sorted position p starts a cell's run
    const int S = R * 64;
    unsigned long long *H = (unsigned long long *)cnt32;                     // <= 512 words
#pragma unroll
    for (int r0 = 0; r0 < kFR; r0 += 4)
        if (r0 < R) {
            unsigned prevw[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int p = w * S + (r0 + k) * 64 + lane;
                const bool in = r0 + k < R && p < n;
                reg[r0 + k] = in ? E[sw(p)] : 0u;
                prevw[k] = (in && p > 0) ? E[sw(p - 1)] : 0u;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int p = w * S + (r0 + k) * 64 + lane;
                const unsigned long long hm = __ballot(r0 + k < R && p < n && (p == 0 || (prevw[k] >> rb) != (reg[r0 + k] >> rb)));
                if (lane == 0 && r0 + k < R) H[p >> 6] = hm;
            }
        }
    __syncthreads();
    // thread t takes the runs that start in the sorted positions [t q, t q + q), q <= 32
    const int q = R;
    const int p_lo = t * q < n ? t * q : n, p_hi = (t + 1) * q < n ? (t + 1) * q : n;
    int heads = 0;
    if (p_lo < p_hi) {                                                       // head bits in [p_lo, p_hi): at most two words of H
        const int w0 = p_lo >> 6, w1 = (p_hi - 1) >> 6;
        const unsigned long long m0 = H[w0] >> (p_lo & 63);
        if (w0 == w1) heads = __popcll(m0 << (63 - ((p_hi - 1) & 63) + (p_lo & 63)));
        else heads = __popcll(m0) + __popcll(H[w1] << (63 - ((p_hi - 1) & 63)));
    }
    int ncell;
    const int out0 = block_excl_scan_i(heads, wtot, t, &ncell);
    FSTAMP(5);
    if (ncell > a.cap) { finish(3u, ncell); return; }
    // The coordinates of every sorted position into LDS (the sort words live in registers by now: their LDS is free) -- all three at once when they
    // fit, else one coordinate at a time --, then CentroidPoint's float sums in input order along each run, straight out of LDS, eight values
    // requested at a time
    float *V = (float *)E;
    const unsigned rmask = (1u << rb) - 1u;
    auto stage = [&](const float *cd, float *Vd) __attribute__((always_inline)) {
#pragma unroll
        for (int r0 = 0; r0 < kFR; r0 += 4)
            if (r0 < R) {
                float v[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) { const unsigned g = reg[r0 + k] & rmask; v[k] = cd[g < (unsigned)n ? g : 0u]; }
#pragma unroll
                for (int k = 0; k < 4; ++k) { const int p = w * S + (r0 + k) * 64 + lane; if (r0 + k < R && p < n) Vd[p] = v[k]; }
            }
    };
    if (3 * n <= kFNmax) {
        __syncthreads();
        stage(a.cx, V); stage(a.cy, V + n); stage(a.cz, V + 2 * n);
        __syncthreads();
        if (ncell == 0) { }      // (n > 0: never)
        const float *Vx = V, *Vy = V + n, *Vz = V + 2 * n;
        const size_t ld = (size_t)ncell;
        int out = out0;
        for (int p = next_head(H, p_lo, n); p < p_hi;) {
            const int e = next_head(H, p + 1, n);
            float sx = Vx[p], sy = Vy[p], sz = Vz[p];
            int j = p + 1;
            for (; j + 4 <= e; j += 4) {
                float vx[4], vy[4], vz[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) { vx[k] = Vx[j + k]; vy[k] = Vy[j + k]; vz[k] = Vz[j + k]; }
#pragma unroll
                for (int k = 0; k < 4; ++k) { sx += vx[k]; sy += vy[k]; sz += vz[k]; }
            }
            for (; j < e; ++j) { sx += Vx[j]; sy += Vy[j]; sz += Vz[j]; }
            const float cf = (float)(e - p);
            a.X[out] = (double)__fdiv_rn(sx, cf); a.X[ld + out] = (double)__fdiv_rn(sy, cf); a.X[2 * ld + out] = (double)__fdiv_rn(sz, cf);
            ++out;
            p = e;
        }
        FSTAMP(12);
    } else {
        for (int d = 0; d < 3; ++d) {
            __syncthreads();
            stage(d == 0 ? a.cx : (d == 1 ? a.cy : a.cz), V);
            __syncthreads();
            if (d == 0) FSTAMP(19);
            double *Xd = a.X + (size_t)d * (size_t)ncell;
            int out = out0;
            for (int p = next_head(H, p_lo, n); p < p_hi;) {
                const int e = next_head(H, p + 1, n);
                float sum = V[p];
                int j = p + 1;
                for (; j + 8 <= e; j += 8) {
                    float v[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k) v[k] = V[j + k];
#pragma unroll
                    for (int k = 0; k < 8; ++k) sum += v[k];
                }
                for (; j < e; ++j) sum += V[j];
                Xd[out++] = (double)__fdiv_rn(sum, (float)(e - p));
                p = e;
            }
            FSTAMP(12 + d);
        }
    }
    FSTAMP(6);
    finish(1u, ncell);
#undef FSTAMP
}

}  // namespace


// =================================================================================================================================================
// The same launch with phase B spread over a TEAM of workgroups (round 5, second form).  One compute unit pulls ~10 bytes a clock from memory and pays
// an LDS round trip or a barrier for every step of a 16-wave workgroup: the single finishing workgroup above takes 48 us at 8 000 points and 123 us at
// 32 000.  Here the LAST K = min(8, T) workgroups to draw a ticket form a team: they are running, hence co-resident by construction -- the barriers
// between them cannot deadlock on scheduling, and a workgroup that is not in the team never waits for anything.  The team members wait until all T
// tickets are drawn, then each takes n / K positions:
//   gather (tile of a rank by binary search over the tile offsets), cell index, sort word -> registers, compacted points -> memory;
//   per 8-bit digit pass (LSD, stable): chunks of 64 positions are matched by 8 ballots, (chunk, digit) counts in LDS, the workgroup's digit
//   histogram published | TEAM BARRIER | every workgroup scans the K histograms itself, scatters its words (agent-scope stores) | TEAM BARRIER |
//   takes its slice of the result;
//   heads of its slice, head counts published | TEAM BARRIER | output offsets; the coordinates of its slice (and of the run that crosses its end)
//   into LDS in sorted order, one lane per head: float sums in input order; the last workgroup to finish re-arms the state and reports to the host.
// Cross-workgroup data travels by agent-scope stores and loads (the L2 caches of the eight XCDs are not coherent with each other).  Every wait is
// bounded (2 s): a team that cannot complete abandons the launch, the host re-initialises the state words and runs the multi-launch form.
// Same words, same stable order, same float sums as k_cloud_fused: bit-exact to it, to the multi-launch form and to the oracle.
namespace {

constexpr int kTK = 8;                   // team size
constexpr int kTCh = 64;                 // chunks of 64 positions per team workgroup and pass: 4096 positions
constexpr int kTVcap = 6144;             // sorted positions whose coordinates a team workgroup stages in LDS (its slice + the run that crosses its end)
constexpr size_t kTLdsCnt = (size_t)kTCh * 256 * 2, kTLdsV = (size_t)3 * kTVcap * 4;
constexpr size_t kTLds = kTLdsCnt + 2 * 4 * 256 * 4 + 256 * 4 + kTCh * 8 + kTCh * 4 + kTLdsV;      // 116 KB
static_assert(kTLdsCnt >= (size_t)(kFTmax + 1) * 4, "the tile offsets live in the counters' area during the gather");
static_assert(kTK * kTCh * 64 >= kFNmax, "the team covers every point the one-launch form takes");

// all threads of a team workgroup; idx: which of the launch's barriers.  false: the launch was abandoned (by this workgroup after 2 s, or by another)
__device__ __forceinline__ bool team_barrier(unsigned *state, int idx, int K, int *sflag) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                          // this wave's stores have been performed
    __syncthreads();
    if (threadIdx.x == 0) {
        int ok = 1;
        (void)__hip_atomic_fetch_add(state + 8, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);        // (one counter for all barriers of the launch: barrier idx is passed at K (idx + 1))
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        unsigned spins = 0;
        while (__hip_atomic_load(state + 8, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(K * (idx + 1))) {
            __builtin_amdgcn_s_sleep(1);
            if ((++spins & 15u) == 0u) {
                if (__hip_atomic_load(state + 7, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { ok = 0; break; }
                if (__builtin_amdgcn_s_memrealtime() - t0 > 200000000ull) { __hip_atomic_store(state + 7, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); ok = 0; break; }
            }
        }
        *sflag = ok;
    }
    __syncthreads();
    return *sflag != 0;
}

__global__ __launch_bounds__(kFT) void k_cloud_team(const FusedCloud a) {
    extern __shared__ __attribute__((aligned(16))) char fsm[];
    unsigned short *cnt = (unsigned short *)fsm;                             // [chunk][digit]: lanes of the chunk with the digit, then their start inside the chunk's quarter
    int *toff = (int *)fsm;                                                  // (during the gather) T + 1 tile offsets
    int *qt = (int *)(fsm + kTLdsCnt);                                       // [4][256]: a quarter's (16 chunks') lanes with the digit
    int *qb = qt + 4 * 256;                                                  // [4][256]: the same, exclusive over the quarters
    int *gb = qb + 4 * 256;                                                  // [256]: where this workgroup's words with the digit start in the whole array
    unsigned long long *hm = (unsigned long long *)(gb + 256);               // [chunk]: head bits of the slice
    int *hp = (int *)(hm + kTCh);                                            // [chunk]: heads in the chunks before
    float *V = (float *)(hp + kTCh);                                         // 3 x kTVcap coordinates in sorted order
    __shared__ int wtot[kFW];
    __shared__ unsigned sbox[8];
    __shared__ unsigned sticket;
    __shared__ int sflag, send;
    __shared__ FusedGrid sgrid;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const unsigned long long below = (1ull << lane) - 1ull;
#ifdef TDLO_CLOUD_STAMPS      // phase stamps of team workgroup 0 (instrumented build only, scripts/archive/gpu_cloud_stamps.py)
#define TSTAMP(i) do { __syncthreads(); if (t == 0 && k == 0) ((unsigned long long *)(a.state + 16))[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define TSTAMP(i) do { } while (0)
#endif
#ifdef TDLO_CLOUD_STAMPS
    const unsigned long long rt_begin = __builtin_amdgcn_s_memrealtime();    // (100 MHz) this workgroup's start
#endif
    const unsigned ticket = cloud_phase_a(a, wtot, &sticket, (unsigned *)fsm);
    const int K = a.T < kTK ? a.T : kTK;
    if ((int)ticket < a.T - K) return;                                       // not in the team: done, having waited for nobody
    const int k = (int)ticket - (a.T - K);                                   // rank in the team
#ifdef TDLO_CLOUD_STAMPS
    if (t == 0 && k == 0) { unsigned long long *sw = (unsigned long long *)(a.state + 16); sw[20] = rt_begin; sw[21] = __builtin_amdgcn_s_memrealtime(); }      // start, ticket drawn
#endif

    unsigned status = 1u;
    int n = 0, ncell = 0;
    // (pre-pass) lane = node: its coordinates are requested now, from pinned host memory -- a PCIe round trip that the sort hides
    const int vM = a.visM;
    double vyx = 0.0, vyy = 0.0, vyz = 0.0;
    if (vM > 0 && lane < vM) { vyx = a.visY[lane]; vyy = a.visY[vM + lane]; vyz = a.visY[2 * vM + lane]; }
    // the last team workgroup to come here re-arms the state words for the next launch and reports to the host
    auto finish = [&]() __attribute__((always_inline)) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                                     // every wave's stores to X have been performed
        if (t == 0) send = __hip_atomic_fetch_add(a.state + 15, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(K - 1);
        __syncthreads();
        if (!send) return;
        if (__hip_atomic_load(a.state + 7, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) status = 4u;      // somebody gave the launch up: whatever this workgroup finished, the cloud is not complete
        __syncthreads();
        if (t < 3) __hip_atomic_store(a.state + t, ~0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else if (t < 16) __hip_atomic_store(a.state + t, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (vM > 0 && t >= 64 && t < 64 + vM) {      // (every member's atomic minima were performed before it drew its finish ticket)
            const int m = t - 64;
            const unsigned long long b = __hip_atomic_load(a.visState + m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(a.visOut + 1 + m, b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(a.visState + m, 0x7ff0000000000000ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (t == 0) {
            __hip_atomic_store(a.res + 1, ((unsigned long long)(unsigned)n << 32) | (unsigned)ncell, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(a.res, ((unsigned long long)a.epoch << 32) | status, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    };
    // ---- everybody's ticket
    if (t == 0) {
        int ok = 1;
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        unsigned spins = 0;
        while (__hip_atomic_load(a.state + 6, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)a.T) {
            __builtin_amdgcn_s_sleep(1);
            if ((++spins & 15u) == 0u) {
                if (__hip_atomic_load(a.state + 7, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { ok = 0; break; }
                if (__builtin_amdgcn_s_memrealtime() - t0 > 200000000ull) { __hip_atomic_store(a.state + 7, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); ok = 0; break; }
            }
        }
        sflag = ok;
    }
    __syncthreads();
    if (!sflag) { status = 4u; finish(); return; }
    TSTAMP(0);
#ifdef TDLO_CLOUD_STAMPS
    if (t == 0 && k == 0) ((unsigned long long *)(a.state + 16))[22] = __builtin_amdgcn_s_memrealtime();      // every ticket drawn: the team begins
#endif
    // ---- the tiles' offsets, the count, the box, the grid (every team workgroup for itself: the same numbers)
    {
        int carry = 0;
        for (int b0 = 0; b0 < a.T; b0 += kFT) {
            const int bb = b0 + t;
            const int c = bb < a.T ? __hip_atomic_load(a.tcnt + bb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
            if (b0 == 0 && t < 6) sbox[t] = __hip_atomic_load(a.state + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int tot;
            const int ex = block_excl_scan_i(c, wtot, t, &tot);
            if (bb < a.T) toff[bb] = carry + ex;
            carry += tot;
        }
        n = carry;
        if (t == 0) toff[a.T] = n;
    }
    const float inv = a.inv_leaf;
    if (t == 0 && n > 0) cloud_grid(sbox, n, inv, sgrid);
    __syncthreads();
    if (n == 0) { finish(); return; }
    if (!sgrid.take) { status = 2u; finish(); return; }
    const int mb0 = sgrid.min_b[0], mb1 = sgrid.min_b[1], mb2 = sgrid.min_b[2], mul1 = sgrid.mul1, mul2 = sgrid.mul2, rb = sgrid.rb, kb = sgrid.kb;
    const unsigned rmask = (1u << rb) - 1u;
    const int per = ((n + K - 1) / K + 63) & ~63;                            // positions per team workgroup: whole chunks, at most 4096
    const int s0 = k * per < n ? k * per : n, s1 = s0 + per < n ? s0 + per : n;
    TSTAMP(1);
    // ---- gather: position p = s0 + 64 (w + 16 r) + lane, r < 4 (a wave's chunk: 64 consecutive positions)
    unsigned wreg[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int g = s0 + 64 * (w + 16 * r) + lane;
        wreg[r] = 0xffffffffu;
        if (g < s1) {
            int lo = 0, hi = a.T;                                            // the tile b with toff[b] <= g < toff[b + 1] (empty tiles have equal offsets)
            while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (toff[mid] <= g) lo = mid; else hi = mid; }
            const unsigned src = (unsigned)(lo * kFPix + (g - toff[lo]));
            const float px = __hip_atomic_load(a.ex + src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const float py = __hip_atomic_load(a.ey + src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const float pz = __hip_atomic_load(a.ez + src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int i0 = (int)(floorf(px * inv) - (float)mb0);             // voxel_grid.hpp: ijk = floor(p * inv_leaf) - min_b
            const int i1 = (int)(floorf(py * inv) - (float)mb1);
            const int i2 = (int)(floorf(pz * inv) - (float)mb2);
            wreg[r] = ((unsigned)(i0 + i1 * mul1 + i2 * mul2) << rb) | (unsigned)g;
            __hip_atomic_store(a.cx + g, px, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(a.cy + g, py, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(a.cz + g, pz, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    __syncthreads();                                                         // (the tile offsets are not needed any more: the counters take their place)
    TSTAMP(2);
    // ---- stable LSD radix sort over the team, 8 bits a pass
    unsigned *win = a.tw0, *wout = a.tw1;
    int bar = 0;
    for (int sh = 0; sh < kb; sh += 8) {
        const int shift = rb + sh;
        const unsigned dmask = kb - sh >= 8 ? 255u : ((1u << (kb - sh)) - 1u);
        {
            uint4 *z = (uint4 *)cnt;
            z[t] = make_uint4(0u, 0u, 0u, 0u); z[t + kFT] = make_uint4(0u, 0u, 0u, 0u);      // 32 KB
        }
        __syncthreads();
        int rank[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int c = w + 16 * r, p = s0 + 64 * c + lane;
            const bool valid = p < s1;
            const unsigned long long vm = __ballot(valid);
            rank[r] = 0;
            if (vm != 0ull) {                                                // (wave-uniform)
                const unsigned d = (wreg[r] >> shift) & dmask;
                unsigned long long same = vm;
#pragma unroll
                for (int bit = 0; bit < 8; ++bit) {
                    const unsigned one = (d >> bit) & 1u;
                    const unsigned long long bl = __ballot(one != 0u);
                    same &= bl ^ ((unsigned long long)one - 1ull);           // one ? bl : ~bl
                }
                rank[r] = __popcll(same & below);
                if (valid && rank[r] == 0) cnt[c * 256 + d] = (unsigned short)__popcll(same);
            }
        }
        __syncthreads();
        {   // thread = (digit, quarter of the chunks): the quarter's count, the chunks' starts inside the quarter
            const int d = t & 255, qd = t >> 8;
            int run = 0;
#pragma unroll 4
            for (int c = 16 * qd; c < 16 * qd + 16; ++c) { const int v = cnt[c * 256 + d]; cnt[c * 256 + d] = (unsigned short)run; run += v; }
            qt[qd * 256 + d] = run;
        }
        __syncthreads();
        if (t < 256) {
            const int q0 = qt[t], q1 = qt[256 + t], q2 = qt[512 + t], q3 = qt[768 + t];
            qb[t] = 0; qb[256 + t] = q0; qb[512 + t] = q0 + q1; qb[768 + t] = q0 + q1 + q2;
            __hip_atomic_store(a.thist + k * 256 + t, q0 + q1 + q2 + q3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (a.hook && K > 1 && k == K - 1 && bar == 0) { status = 4u; finish(); return; }      // (test hook: a team member that never arrives)
        if (!team_barrier(a.state, bar++, K, &sflag)) { status = 4u; finish(); return; }
        {   // every workgroup: the team's histograms, the digit's start in the whole array + the words of the workgroups in front
            int tot = 0, before = 0;
            if (t < 256) {
                int h[kTK];
#pragma unroll
                for (int j = 0; j < kTK; ++j) h[j] = j < K ? __hip_atomic_load(a.thist + j * 256 + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
#pragma unroll
                for (int j = 0; j < kTK; ++j) { tot += h[j]; before += j < k ? h[j] : 0; }
            }
            int all;
            const int ex = block_excl_scan_i(tot, wtot, t, &all);
            if (t < 256) gb[t] = ex + before;
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int c = w + 16 * r, p = s0 + 64 * c + lane;
            if (p < s1) {
                const unsigned d = (wreg[r] >> shift) & dmask;
                const int dst = gb[d] + qb[(c >> 4) * 256 + d] + (int)cnt[c * 256 + d] + rank[r];
                __hip_atomic_store(wout + dst, wreg[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        if (!team_barrier(a.state, bar++, K, &sflag)) { status = 4u; finish(); return; }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int p = s0 + 64 * (w + 16 * r) + lane;
            wreg[r] = p < s1 ? __hip_atomic_load(wout + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0xffffffffu;
        }
        { unsigned *tmp = win; win = wout; wout = tmp; }
    }
    TSTAMP(3);
    // ---- heads of the slice (win holds the sorted words now); the word in front of a chunk's first lane comes from memory
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int c = w + 16 * r, p = s0 + 64 * c + lane;
        unsigned prevw = (unsigned)__shfl_up((int)wreg[r], 1);
        if (lane == 0) prevw = (p > 0 && p < s1) ? __hip_atomic_load(win + p - 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
        const unsigned long long m = __ballot(p < s1 && (p == 0 || (prevw >> rb) != (wreg[r] >> rb)));
        if (lane == 0) hm[c] = m;
    }
    __syncthreads();
    int nh;                                                                  // heads in this slice
    {
        const int hc = t < kTCh ? __popcll(hm[t]) : 0;                       // (wave 0: one chunk per lane)
        const int ex = block_excl_scan_i(hc, wtot, t, &nh);
        if (t < kTCh) hp[t] = ex;
        if (t == 0) __hip_atomic_store(a.thist + kTK * 256 + k, nh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // the slice's heads as a list (position inside the slice), so that EVERY lane of the workgroup can take one: a chunk of 64 positions holds about one
    // head, and a wave walking its four chunks one after the other would leave 60 of its lanes idle through four runs
    unsigned short *hl = cnt;                                                // (the counters are done with)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int c = w + 16 * r, p = s0 + 64 * c + lane;
        if (p < s1 && ((hm[c] >> lane) & 1ull)) hl[hp[c] + __popcll(hm[c] & below)] = (unsigned short)(p - s0);
    }
    if (!team_barrier(a.state, bar++, K, &sflag)) { status = 4u; finish(); return; }
    int out0 = 0;
    {
        int h[kTK];
#pragma unroll
        for (int j = 0; j < kTK; ++j) h[j] = j < K ? __hip_atomic_load(a.thist + kTK * 256 + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
#pragma unroll
        for (int j = 0; j < kTK; ++j) { ncell += h[j]; out0 += j < k ? h[j] : 0; }
    }
    if (ncell > a.cap) { status = 3u; finish(); return; }
    TSTAMP(4);
    // ---- the slice's own coordinates are requested now (its words are in registers; they do not depend on where the crossing run ends): the round trip
    //      runs beside the search below instead of behind it
    float gx[4], gy[4], gz[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int q = s0 + 64 * (w + 16 * r) + lane;
        const unsigned wq = q < s1 ? (wreg[r] & rmask) : 0u;
        gx[r] = __hip_atomic_load(a.cx + wq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        gy[r] = __hip_atomic_load(a.cy + wq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        gz[r] = __hip_atomic_load(a.cz + wq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // ---- where the run that crosses the end of the slice ends (the next head at or behind s1)
    int e_k = s1;
    if (s1 > s0 && s1 < n) {
        const unsigned lastw = __hip_atomic_load(win + s1 - 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int q0 = s1;; q0 += kFT) {                                      // (uniform trip count: every thread sees the same sflag)
            const int q = q0 + t;
            const bool stop = q >= n || (__hip_atomic_load(win + (q < n ? q : n - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> rb) != (lastw >> rb);
            const unsigned long long sm = __ballot(stop);
            if (lane == 0) wtot[w] = sm ? (int)__builtin_ctzll(sm) : 64;
            __syncthreads();
            int first = kFT;
#pragma unroll
            for (int i = kFW - 1; i >= 0; --i) if (wtot[i] < 64) first = 64 * i + wtot[i];
            __syncthreads();
            if (first < kFT) { e_k = q0 + first; break; }
        }
    }
    // ---- the coordinates of [s0, min(e_k, s0 + kTVcap)) in sorted order into LDS; beyond that (a run of thousands of points) they are read from memory
    const int vend = e_k < s0 + kTVcap ? e_k : s0 + kTVcap;
#pragma unroll
    for (int r = 0; r < 4; ++r) {                                            // the slice itself
        const int q = s0 + 64 * (w + 16 * r) + lane;
        if (q < s1) { V[q - s0] = gx[r]; V[kTVcap + q - s0] = gy[r]; V[2 * kTVcap + q - s0] = gz[r]; }
    }
    for (int q = s1 + t; q < vend; q += kFT) {                               // the run that crosses its end
        const unsigned wq = __hip_atomic_load(win + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & rmask;
        V[q - s0] = __hip_atomic_load(a.cx + wq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        V[kTVcap + q - s0] = __hip_atomic_load(a.cy + wq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        V[2 * kTVcap + q - s0] = __hip_atomic_load(a.cz + wq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    TSTAMP(5);
    // ---- one lane per head: CentroidPoint's float sums in input order
    float ccx[4] = {0.0f, 0.0f, 0.0f, 0.0f}, ccy[4] = {0.0f, 0.0f, 0.0f, 0.0f}, ccz[4] = {0.0f, 0.0f, 0.0f, 0.0f};      // this thread's centroids (nh <= 4096 = 4 per thread)
    {
        const size_t ld = (size_t)ncell;
        int rr = 0;
        for (int i = t; i < nh; i += kFT, ++rr) {
            const int p = s0 + (int)hl[i];
            const int e = i + 1 < nh ? s0 + (int)hl[i + 1] : e_k;             // the next head inside the slice, or where the run that crosses its end stops
            const int out = out0 + i;
            float sx = 0.0f, sy = 0.0f, sz = 0.0f;
            int j = p;
            const int el = e < vend ? e : vend;
            for (; j + 16 <= el; j += 16) {                                   // (sixteen points' reads in flight: a cell of the 1280 x 720 image holds ~80 points, the sums are sequential)
                float vx[16], vy[16], vz[16];
#pragma unroll
                for (int q = 0; q < 16; ++q) { vx[q] = V[j + q - s0]; vy[q] = V[kTVcap + j + q - s0]; vz[q] = V[2 * kTVcap + j + q - s0]; }
#pragma unroll
                for (int q = 0; q < 16; ++q) { sx += vx[q]; sy += vy[q]; sz += vz[q]; }
            }
            for (; j + 8 <= el; j += 8) {                                     // (eight points' reads in flight: a cell of the 1280 x 720 image holds ~80 points, the sums are sequential)
                float vx[8], vy[8], vz[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) { vx[q] = V[j + q - s0]; vy[q] = V[kTVcap + j + q - s0]; vz[q] = V[2 * kTVcap + j + q - s0]; }
#pragma unroll
                for (int q = 0; q < 8; ++q) { sx += vx[q]; sy += vy[q]; sz += vz[q]; }
            }
            for (; j + 4 <= el; j += 4) {
                float vx[4], vy[4], vz[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) { vx[q] = V[j + q - s0]; vy[q] = V[kTVcap + j + q - s0]; vz[q] = V[2 * kTVcap + j + q - s0]; }
#pragma unroll
                for (int q = 0; q < 4; ++q) { sx += vx[q]; sy += vy[q]; sz += vz[q]; }
            }
            for (; j < el; ++j) { sx += V[j - s0]; sy += V[kTVcap + j - s0]; sz += V[2 * kTVcap + j - s0]; }
            for (; j < e; ++j) {                                             // (beyond the staged range)
                const unsigned g = __hip_atomic_load(win + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & rmask;
                sx += __hip_atomic_load(a.cx + g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                sy += __hip_atomic_load(a.cy + g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                sz += __hip_atomic_load(a.cz + g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            const float cf = (float)(e - p);
            const float qx = __fdiv_rn(sx, cf), qy = __fdiv_rn(sy, cf), qz = __fdiv_rn(sz, cf);
            a.X[out] = (double)qx; a.X[ld + out] = (double)qy; a.X[2 * ld + out] = (double)qz;
#pragma unroll
            for (int q = 0; q < 4; ++q) if (q == rr) { ccx[q] = qx; ccy[q] = qy; ccz[q] = qz; }
        }
    }
    TSTAMP(6);
    if (vM > 0) {
        // ---- the visibility pre-pass over this member's centroids: they go to LDS (the staged coordinates are done with), then wave = a sixteenth of
        //      them, lane = node: the distance of k_node_min_dist_direct (node_point_d2 on the centroid as stored: the float widened to double), the
        //      minimum over the waves through LDS, one atomic minimum per node and member
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; ++q) { const int i = t + q * kFT; if (i < nh) { V[i] = ccx[q]; V[kTVcap + i] = ccy[q]; V[2 * kTVcap + i] = ccz[q]; } }
        __syncthreads();
        double mn = __builtin_huge_val();
        if (lane < vM) {
            for (int i = w; i < nh; i += kFW) mn = ::fmin(mn, node_point_d2(vyx, vyy, vyz, (double)V[i], (double)V[kTVcap + i], (double)V[2 * kTVcap + i]));
        }
        double *red = (double *)qt;                                          // kFW x 64 doubles = 8 KB: the quarter counts qt | qb, done with
        red[w * 64 + lane] = mn;
        __syncthreads();
        if (w == 0 && lane < vM && nh > 0) {
#pragma unroll
            for (int j = 1; j < kFW; ++j) mn = ::fmin(mn, red[j * 64 + lane]);
            (void)__hip_atomic_fetch_min(a.visState + lane, (unsigned long long)__double_as_longlong(mn), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
#ifdef TDLO_CLOUD_STAMPS
    if (t == 0 && k == 0) ((unsigned long long *)(a.state + 16))[23] = __builtin_amdgcn_s_memrealtime();
#endif
    finish();
#undef TSTAMP
}

}  // namespace

size_t cloud_fused_ws_bytes(int P) {       // beside the multi-launch form's workspace: state words, compacted points, tile counts
    const size_t T = ((size_t)P + kFPix - 1) / kFPix;
    return 256 + 3 * (size_t)kFNmax * sizeof(float) + 2 * (size_t)kFNmax * sizeof(unsigned) + (kTK * 256 + 64) * sizeof(int) + ((T * sizeof(int) + 255) & ~(size_t)255);
}
int cloud_fused_max_points() { return kFNmax; }
bool cloud_fused_ok(int P) { return ((size_t)P + kFPix - 1) / kFPix <= (size_t)kFTmax; }

// ws: the multi-launch form's workspace (three of its four P-word sort buffers serve as the tiles' regions), fws: cloud_fused_ws_bytes(P)
// bytes kept by the kernels themselves between the launches (first == true: the state words are initialised by a copy in front of the launch)
hipError_t launch_cloud_fused(const unsigned short *depth, const unsigned char *mask, int P, int cols, const double cam[4], float inv_leaf, void *ws, void *fws,
                              bool first, bool team, double *Xraw, int cap, unsigned long long *res_pinned, unsigned epoch, hipStream_t s,
                              const double *vis_nodes_pinned, int vis_M, unsigned long long *vis_state, unsigned long long *vis_out_pinned) {
    {   // on every launch, for the CURRENT device, like every other launcher of the library: HIP keeps the attribute per device function, so a
        // process-wide "done once" flag would leave a second context on another GPU (the exchange tests, a one-process multi-GPU caller) launching
        // 116 / 160 KB of dynamic LDS without it; the call is a table look-up
        hipError_t e = team ? hipFuncSetAttribute((const void *)k_cloud_team, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kTLds)
                            : hipFuncSetAttribute((const void *)k_cloud_fused, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kFLds);
        if (e != hipSuccess) return e;
    }
    FusedCloud a;
    a.depth = depth; a.mask = mask; a.P = P; a.cols = cols; a.T = (P + kFPix - 1) / kFPix;
    a.cam = Cam{cam[0], cam[1], cam[2], cam[3]};
    a.inv_leaf = inv_leaf;
    a.ex = (float *)ws; a.ey = a.ex + P; a.ez = a.ey + P;
    a.state = (unsigned *)fws;                                   // (at a fixed place: another image size finds the words re-armed by the last launch)
    a.cx = (float *)((char *)fws + 256); a.cy = a.cx + kFNmax; a.cz = a.cy + kFNmax;
    a.tw0 = (unsigned *)(a.cz + kFNmax); a.tw1 = a.tw0 + kFNmax;
    a.thist = (int *)(a.tw1 + kFNmax);
    a.tcnt = a.thist + kTK * 256 + 64;
    a.X = Xraw; a.cap = cap; a.res = res_pinned; a.epoch = epoch;
    a.visY = vis_nodes_pinned; a.visM = (team && vis_nodes_pinned && vis_M > 0 && vis_M <= 64) ? vis_M : 0; a.visState = vis_state; a.visOut = vis_out_pinned;
    {   // test hook: the n-th team launch of the process loses a team member (the others give the launch up after 2 s; the host runs the multi-launch form)
        static int hook_at = getenv("TDLO_CLOUD_TEAM_FORCE_TIMEOUT") ? atoi(getenv("TDLO_CLOUD_TEAM_FORCE_TIMEOUT")) : 0;
        a.hook = (team && hook_at > 0 && --hook_at == 0) ? 1 : 0;
    }
    if (first) {
        static const unsigned init[16] = {~0u, ~0u, ~0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};      // box, tickets, team: abandon word, barrier counter, finish tickets (word 15)
        const hipError_t e = hipMemcpyAsync(a.state, init, sizeof init, hipMemcpyHostToDevice, s);
        if (e != hipSuccess) return e;
    }
    if (team) hipLaunchKernelGGL(k_cloud_team, dim3(a.T), dim3(kFT), kTLds, s, a);
    else hipLaunchKernelGGL(k_cloud_fused, dim3(a.T), dim3(kFT), kFLds, s, a);
    return hipGetLastError();
}

size_t cloud_ws_bytes(int P) {
    const size_t nblk = ((size_t)P + kTile - 1) / kTile;
    // key/val double buffers, histogram (256 x nblk), per-block head counts, bbox + counters
    return sizeof(unsigned) * 4 * (size_t)P + sizeof(int) * (256 * nblk + nblk + 64) + 256;
}

static int *blkcnt_of(void *ws, int P) {          // per-tile masked-pixel counts live where the head counts go later
    const int nblk = (P + kTile - 1) / kTile;
    return (int *)((unsigned *)ws + 4 * (size_t)P) + 256 * (size_t)nblk;
}

hipError_t launch_cloud_bbox(const unsigned short *depth, const unsigned char *mask, int P, int cols, const double cam[4], unsigned *bbox, void *ws, hipStream_t s) {
    const Cam c{cam[0], cam[1], cam[2], cam[3]};
    const int nblk = (P + kTile - 1) / kTile;
    hipLaunchKernelGGL(k_cloud_bbox, dim3(nblk), dim3(kCB), 0, s, depth, mask, P, cols, c, bbox, blkcnt_of(ws, P));
    return hipGetLastError();
}

hipError_t launch_cloud_voxels(const unsigned short *depth, const unsigned char *mask, int P, int cols, const double cam[4],
                               const int min_b[3], int mul1, int mul2, float inv_leaf, int nodown, int passes, int n,
                               void *ws, int *total_dev, int cap, double *Xraw, hipStream_t s) {
    const Cam c{cam[0], cam[1], cam[2], cam[3]};
    Grid g; g.min_b[0] = min_b[0]; g.min_b[1] = min_b[1]; g.min_b[2] = min_b[2]; g.mul1 = mul1; g.mul2 = mul2; g.inv = inv_leaf; g.nodown = nodown;
    // n = number of masked pixels (known to the host from the bounding-box pass): everything after the compaction
    // works on n elements, typically a few per cent of the image
    const int nblkP = (P + kTile - 1) / kTile, nblk = (n + kTile - 1) / kTile;
    unsigned *keyA = (unsigned *)ws, *valA = keyA + P, *keyB = valA + P, *valB = keyB + P;
    int *hist = (int *)(valB + P), *cnt = blkcnt_of(ws, P);
    hipLaunchKernelGGL(k_scan_single, dim3(1), dim3(1024), 0, s, cnt, nblkP, (int *)nullptr);
    hipLaunchKernelGGL(k_cloud_keys, dim3(nblkP), dim3(kCB), 0, s, depth, mask, P, cols, c, g, cnt, keyA, valA);
    for (int pass = 0; pass < passes; ++pass) {
        hipLaunchKernelGGL(k_radix_hist, dim3(nblk), dim3(kCB), 0, s, keyA, n, 8 * pass, nblk, hist);
        hipLaunchKernelGGL(k_scan_single, dim3(1), dim3(1024), 0, s, hist, 256 * nblk, (int *)nullptr);
        hipLaunchKernelGGL(k_radix_scatter, dim3(nblk), dim3(kCB), 0, s, keyA, valA, n, 8 * pass, nblk, hist, keyB, valB);
        unsigned *tk = keyA; keyA = keyB; keyB = tk;
        unsigned *tv = valA; valA = valB; valB = tv;
    }
    hipLaunchKernelGGL(k_cloud_heads, dim3(nblk), dim3(kCB), 0, s, keyA, n, nodown, cnt);
    hipLaunchKernelGGL(k_scan_single, dim3(1), dim3(1024), 0, s, cnt, nblk, total_dev);
    hipLaunchKernelGGL(k_cloud_centroid, dim3(nblk), dim3(kCB), 0, s, depth, n, cols, c, nodown, keyA, valA, cnt, total_dev, cap, Xraw);
    return hipGetLastError();
}

}  // namespace tdlo
