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
#include <cstdint>

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
    if ((threadIdx.x & 63) == 0 && cnt > 0) {
#pragma unroll
        for (int d = 0; d < 3; ++d) { atomicMin(&bbox[d], mn[d]); atomicMax(&bbox[3 + d], mx[d]); }
        atomicAdd(&bbox[6], (unsigned)cnt);
    }
    if ((threadIdx.x & 63) == 0) wc[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) blkcnt[blockIdx.x] = wc[0] + wc[1] + wc[2] + wc[3];     // masked pixels of this 1024-pixel tile
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
