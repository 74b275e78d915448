// How much does it cost an E-step-shaped launch (98 workgroups x 512 threads, ~5 us of work each) to finish with 201 int64
// atomic adds per workgroup into K replica rows instead of one coalesced 201-float row?  And what does a one-workgroup reader pay
// for K rows against 98 rows?   build: hipcc --offload-arch=gfx950 -O3 scripts/ubench/atom.hip -o scripts/ubench/atom
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
__global__ void work(float *rows, long long *acc, int mode, int K, int spin, float *sink) {
    // fake work
    float a = threadIdx.x * 1e-3f;
    for (int i = 0; i < spin; ++i) a = a * 1.0001f + 0.5f;
    if (a == 12345.f) sink[0] = a;
    const int e = threadIdx.x;
    if (e < 201) {
        if (mode == 0) rows[(size_t)blockIdx.x * 204 + e] = a;
        else {
            unsigned xcc = 0;
            if (mode == 2) { asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc)); xcc &= 7; } else xcc = blockIdx.x % K;
            const long long v = (long long)(a * 1024.f) + 1;
            __hip_atomic_fetch_add(acc + (size_t)(xcc % K) * 256 + e, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}
__global__ void reader(const float *rows, const long long *acc, int mode, int K, int nb, double *out) {
    const int t = threadIdx.x;
    double s = 0;
    if (mode == 0) { for (int b = 0; b < nb; ++b) if (t < 201) s += rows[(size_t)b * 204 + t]; }
    else { for (int k = 0; k < K; ++k) if (t < 201) s += (double)acc[(size_t)k * 256 + t]; }
    out[t] = s;
}
int main() {
    float *rows, *sink; long long *acc; double *out;
    hipMalloc(&rows, 1024 * 204 * 4); hipMalloc(&acc, 64 * 256 * 8); hipMalloc(&sink, 64); hipMalloc(&out, 512 * 8);
    hipMemset(acc, 0, 64 * 256 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int spin = 2500;
    for (int nb : {98, 512}) for (int mode = 0; mode < 3; ++mode) for (int K : {1, 8, 32}) {
        if (mode == 0 && K != 1) continue;
        std::vector<float> tw, tr;
        for (int rep = 0; rep < 40; ++rep) {
            float ms;
            hipEventRecord(e0); hipLaunchKernelGGL(work, dim3(nb), dim3(512), 0, 0, rows, acc, mode, K, spin, sink); hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1); tw.push_back(ms * 1e3f);
            hipEventRecord(e0); hipLaunchKernelGGL(reader, dim3(1), dim3(256), 0, 0, rows, acc, mode, K, nb, out); hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1); tr.push_back(ms * 1e3f);
        }
        std::sort(tw.begin(), tw.end()); std::sort(tr.begin(), tr.end());
        printf("blocks %3d  mode %d (%s) K %2d: work kernel %.2f us (median), reader %.2f us\n", nb, mode, mode == 0 ? "rows" : (mode == 1 ? "atomics, row = block %% K" : "atomics, row = XCC id"), K, tw[20], tr[20]);
    }
    return 0;
}
