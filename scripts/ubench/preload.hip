// Does kernarg preloading (-mllvm -amdgpu-kernarg-preload-count=N: the leading scalar / pointer arguments arrive in SGPRs with the wave instead of
// by an s_load from the kernarg segment) shorten a short dependent kernel?  Two consumers of the same 16 accumulator rows, one taking its pointers in
// a by-value struct (never preloaded), one as leading pointer arguments; chains of dependent launches timed with events.
// build: hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-kernarg-preload-count=14 scripts/ubench/preload.hip -o scripts/ubench/preload
#include <hip/hip_runtime.h>
#include <cstdio>
struct Args { long long *acc; double *small; double *out; int mode; };
__global__ void producer(long long *acc, double *small, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (threadIdx.x < 201) __hip_atomic_fetch_add(acc + (size_t)(blockIdx.x % 8) * 256 + threadIdx.x, (long long)i + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (i < n) small[i] = i * 0.5;
}
__device__ __forceinline__ void body(long long *acc, double *small, double *out) {
    const int t = threadIdx.x;
    const double s0 = small[t & 63];
    long long s = 0;
    if (t < 201) for (int r = 0; r < 16; ++r) s += acc[(size_t)r * 256 + t];
    out[t] = (double)s + s0;
}
__global__ void consumerA(const Args a) { body(a.acc, a.small, a.out); }
__global__ void consumerB(long long *acc, double *small, double *out, int mode) { body(acc, small, out); }
__global__ void emptyA(const Args a) { if (a.mode == 12345) a.out[0] = 1; }
__global__ void emptyB(long long *acc, double *small, double *out, int mode) { if (mode == 12345) out[0] = 1; }
int main() {
    Args a; hipMalloc(&a.acc, 16 * 256 * 8); hipMalloc(&a.small, 4096 * 8); hipMalloc(&a.out, 256 * 8); a.mode = 0;
    hipMemset(a.acc, 0, 16 * 256 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int n = 2000;
    for (int rep = 0; rep < 3; ++rep) for (int v = 0; v < 6; ++v) {
        hipDeviceSynchronize();
        hipEventRecord(e0, 0);
        for (int i = 0; i < n; ++i) {
            if (v >= 2 && v < 4) hipLaunchKernelGGL(producer, dim3(196), dim3(256), 0, 0, a.acc, a.small, 4096);
            if (v == 0 || v == 2) hipLaunchKernelGGL(consumerA, dim3(1), dim3(256), 0, 0, a);
            if (v == 1 || v == 3) hipLaunchKernelGGL(consumerB, dim3(1), dim3(256), 0, 0, a.acc, a.small, a.out, 0);
            if (v == 4) hipLaunchKernelGGL(emptyA, dim3(1), dim3(256), 0, 0, a);
            if (v == 5) hipLaunchKernelGGL(emptyB, dim3(1), dim3(256), 0, 0, a.acc, a.small, a.out, 0);
        }
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const char *nm[6] = {"consumer, struct by value", "consumer, preloaded pointers", "producer + consumer, struct", "producer + consumer, preloaded", "empty, struct", "empty, preloaded"};
        printf("%-34s %7.3f us per round\n", nm[v], ms * 1000 / n);
    }
    return 0;
}
