// tdlo_reg.hip -- plain GMM-EM registration `reg`.
//
// trackdlo/src/utils.cpp:21-82: M Gaussian centroids fitted to the cloud with the Euclidean membership only (no
// coherence kernel, no prune, no stopping rule): centroids start on a 0.1 m segment of the y axis (:24-29),
// sigma2 = sum ||y_m - x_n||^2 / (3 M N) (:36-45), then max_iter times
//     P = exp(-d2 / 2 sigma2) / (colsum + c),  c = (2 pi sigma2)^1.5 mu / (1 - mu) M / N          (:55-58)
//     Y = (P X) ./ P1,   sigma2 = sum P d2 / (3 sum P)   with d2 measured from the OLD centroids  (:60-80)
// Not called by the reference's node today; it is the initialiser the E-step naturally provides.  fp64 like the
// reference; not a hot path: thread = point, two passes over the (few) centroids, per-node wave sums, per-block
// partials added up in block order by a one-block M-step kernel that also prepares the next iteration's constants --
// the whole loop is enqueued without a host round trip.  Centroids with P1 = 0 become NaN exactly as in the reference.
#include "tdlo_internal.h"

namespace tdlo {
namespace {

constexpr int kRB = 256;

__device__ __forceinline__ double wsum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// state: [0] sigma2, [1] c, [2] Np (last), then Y (3 M, column-major) at +8
// part: per block 5 M + 1 doubles: P1 (M), PX (3 M), sum P d2 per node (M), [5 M] = sum of d2 (init mode)
__global__ __launch_bounds__(kRB) void k_reg_estep(const double *__restrict__ X, int N, int M, const double *__restrict__ state, int init,
                                                   double *__restrict__ part) {
    extern __shared__ double sm[];              // Y (3 M) | acc (4 waves x (5 M + 1))
    double *Ys = sm, *acc = sm + 3 * M;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, nacc = 5 * M + 1;
    for (int i = t; i < 3 * M; i += kRB) Ys[i] = state[8 + i];
    for (int i = t; i < 4 * nacc; i += kRB) acc[i] = 0.0;
    __syncthreads();
    const double sigma2 = state[0], c = state[1];
    const double k2 = -0.5 / sigma2;
    double *my = acc + w * nacc;
    for (int base = blockIdx.x * kRB; base < N; base += gridDim.x * kRB) {
        const int n = base + t;
        const bool ok = n < N;
        const double x = ok ? X[n] : 0.0, y = ok ? X[(size_t)N + n] : 0.0, z = ok ? X[2 * (size_t)N + n] : 0.0;
        if (init) {
            double s = 0.0;
            for (int m = 0; m < M; ++m) { const double dx = Ys[m] - x, dy = Ys[M + m] - y, dz = Ys[2 * M + m] - z; s += dx * dx + dy * dy + dz * dz; }
            s = wsum(ok ? s : 0.0);
            if (lane == 0) my[5 * M] += s;
            continue;
        }
        double den = 0.0;
        for (int m = 0; m < M; ++m) {
            const double dx = Ys[m] - x, dy = Ys[M + m] - y, dz = Ys[2 * M + m] - z;
            den += exp(k2 * (dx * dx + dy * dy + dz * dz));                        // :55
        }
        const double rden = ok ? 1.0 / (den + c) : 0.0;                            // :58
        for (int m = 0; m < M; ++m) {
            const double dx = Ys[m] - x, dy = Ys[M + m] - y, dz = Ys[2 * M + m] - z;
            const double d2 = dx * dx + dy * dy + dz * dz;
            const double p = exp(k2 * d2) * rden;
            const double s0 = wsum(p), s1 = wsum(p * x), s2 = wsum(p * y), s3 = wsum(p * z), s4 = wsum(p * d2);
            if (lane == 0) { my[m] += s0; my[M + m] += s1; my[2 * M + m] += s2; my[3 * M + m] += s3; my[4 * M + m] += s4; }
        }
    }
    __syncthreads();
    for (int i = t; i < nacc; i += kRB) part[(size_t)blockIdx.x * nacc + i] = ((acc[i] + acc[nacc + i]) + acc[2 * nacc + i]) + acc[3 * nacc + i];
}

__global__ __launch_bounds__(kRB) void k_reg_mstep(int N, int M, int nblk, double mu, int init, const double *__restrict__ part, double *__restrict__ state) {
    extern __shared__ double S[];               // 5 M + 1
    __shared__ double red[2];
    const int t = threadIdx.x, nacc = 5 * M + 1;
    for (int i = t; i < nacc; i += kRB) { double a = 0.0; for (int b = 0; b < nblk; ++b) a += part[(size_t)b * nacc + i]; S[i] = a; }
    __syncthreads();
    if (t == 0) {
        double sigma2;
        if (init) sigma2 = S[5 * M] / (3.0 * (double)M * (double)N);              // :45
        else {
            double num = 0.0, np = 0.0;
            for (int m = 0; m < M; ++m) { num += S[4 * M + m]; np += S[m]; }      // :70-79
            sigma2 = num / (np * 3.0);
            state[2] = np;
        }
        state[0] = sigma2;
        state[1] = pow(2.0 * M_PI * sigma2, 1.5) * mu / (1.0 - mu) * (double)M / (double)N;   // :57
        red[0] = sigma2;
    }
    if (!init) for (int i = t; i < 3 * M; i += kRB) state[8 + i] = S[M + i] / S[i % M];       // :60-68 (0 / 0 -> NaN like the reference)
}

}  // namespace

// k_reg_estep: 8 (23 M + 4) bytes of dynamic LDS, at most 160 KB per workgroup on gfx950
int reg_max_nodes() { return (int)((160 * 1024 / sizeof(double) - 4) / 23); }

size_t reg_ws_doubles(int M, int nblk) { return 8 + 3 * (size_t)M + (size_t)nblk * (5 * (size_t)M + 1) + 8; }

hipError_t launch_reg(const double *X, int N, int M, double mu, int max_iter, int nblk, double *ws, hipStream_t s) {
    double *state = ws, *part = ws + 8 + 3 * (size_t)M + ((3 * M) & 1);
    const size_t lds_e = sizeof(double) * (3 * (size_t)M + 4 * (5 * (size_t)M + 1)), lds_m = sizeof(double) * (5 * (size_t)M + 1);
    if (lds_e > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)k_reg_estep, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_e);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(k_reg_estep, dim3(nblk), dim3(kRB), lds_e, s, X, N, M, state, 1, part);
    hipLaunchKernelGGL(k_reg_mstep, dim3(1), dim3(kRB), lds_m, s, N, M, nblk, mu, 1, part, state);
    for (int it = 0; it < max_iter; ++it) {
        hipLaunchKernelGGL(k_reg_estep, dim3(nblk), dim3(kRB), lds_e, s, X, N, M, state, 0, part);
        hipLaunchKernelGGL(k_reg_mstep, dim3(1), dim3(kRB), lds_m, s, N, M, nblk, mu, 0, part, state);
    }
    return hipGetLastError();
}

}  // namespace tdlo
