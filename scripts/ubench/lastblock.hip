// Does a "last workgroup runs the follow-up" fusion beat two dependent launches?  E-step-shaped launch (196 workgroups x 256 threads, each
// ending with int64 atomics into 8 replica rows) followed by a one-workgroup consumer of the rows:
//   A. two kernels on one stream (dependent dispatch)             B. one kernel: release fence + ticket; the last workgroup consumes
// build: hipcc --offload-arch=gfx950 -O3 scripts/ubench/lastblock.hip -o scripts/ubench/lastblock
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
__device__ __forceinline__ float spin_work(int spin) { float a = threadIdx.x * 1e-3f; for (int i = 0; i < spin; ++i) a = a * 1.0001f + 0.5f; return a; }
__device__ __forceinline__ void consume(const long long *acc, double *out, int spin2) {
    const int t = threadIdx.x;
    long long s = 0;
    if (t < 201) for (int r = 0; r < 8; ++r) s += __hip_atomic_load(acc + (size_t)r * 256 + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    double a = (double)s;
    for (int i = 0; i < spin2; ++i) a = a * 1.0000001 + 0.5;      // the M-step's dependent chain
    out[t] = a;
}
__global__ void producer(long long *acc, int spin, float *sink) {
    const float a = spin_work(spin);
    if (a == 12345.f) sink[0] = a;
    if (threadIdx.x < 201) __hip_atomic_fetch_add(acc + (size_t)(blockIdx.x % 8) * 256 + threadIdx.x, (long long)(a * 1024.f) + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__global__ void consumer(const long long *acc, double *out, int spin2) { consume(acc, out, spin2); }
__global__ void fused(long long *acc, unsigned *ticket, int spin, int spin2, float *sink, double *out) {
    const float a = spin_work(spin);
    if (a == 12345.f) sink[0] = a;
    if (threadIdx.x < 201) __hip_atomic_fetch_add(acc + (size_t)(blockIdx.x % 8) * 256 + threadIdx.x, (long long)(a * 1024.f) + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __shared__ unsigned last;
    __syncthreads();                                  // (the atomics of this workgroup are issued)
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        last = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1 ? 1u : 0u;
    }
    __syncthreads();
    if (!last) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    if (threadIdx.x == 0) *ticket = 0;
    consume(acc, out, spin2);
}
int main() {
    long long *acc; unsigned *ticket; float *sink; double *out;
    hipMalloc(&acc, 8 * 256 * 8); hipMalloc(&ticket, 64); hipMalloc(&sink, 64); hipMalloc(&out, 256 * 8);
    hipMemset(acc, 0, 8 * 256 * 8); hipMemset(ticket, 0, 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int spin = 300, spin2 = 1000, reps = 50;     // ~5 us of producer work, ~8 us of consumer chain
    for (int mode = 0; mode < 2; ++mode) {
        std::vector<float> ts;
        for (int trial = 0; trial < 20; ++trial) {
            hipEventRecord(e0);
            for (int r = 0; r < reps; ++r) {
                if (mode == 0) { hipLaunchKernelGGL(producer, dim3(196), dim3(256), 0, 0, acc, spin, sink); hipLaunchKernelGGL(consumer, dim3(1), dim3(256), 0, 0, acc, out, spin2); }
                else hipLaunchKernelGGL(fused, dim3(196), dim3(256), 0, 0, acc, ticket, spin, spin2, sink, out);
            }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); ts.push_back(ms * 1e3f / reps);
        }
        std::sort(ts.begin(), ts.end());
        printf("%s: %.2f us per (producer + consumer) pair (median of 20 x %d)\n", mode ? "fused (last workgroup consumes)" : "two dependent launches", ts[10], reps);
    }
    return 0;
}
