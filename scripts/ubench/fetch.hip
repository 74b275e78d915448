// What does the first memory round trip of a one-workgroup kernel cost when the data were just written by another kernel (other
// XCDs' atomics / stores)?  Stamps: kernel entry -> kernarg pointer loaded -> first global load returned -> 16 rows summed.
// build: hipcc --offload-arch=gfx950 -O3 scripts/ubench/fetch.hip -o scripts/ubench/fetch
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
struct Args { long long *acc; double *small; unsigned long long *ticks; int mode; };
__global__ void producer(long long *acc, double *small, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (threadIdx.x < 201) __hip_atomic_fetch_add(acc + (size_t)(blockIdx.x % 8) * 256 + threadIdx.x, (long long)i + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (i < n) small[i] = i * 0.5;
}
__global__ void consumer(const Args a, double *out) {
    const int t = threadIdx.x;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    const double s0 = a.small[t & 63];                       // first dependent global load
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    long long s = 0;
    if (t < 201) {
        if (a.mode == 0) { for (int r = 0; r < 16; ++r) s += a.acc[(size_t)r * 256 + t]; }
        else { for (int r = 0; r < 16; ++r) s += __builtin_nontemporal_load(a.acc + (size_t)r * 256 + t); }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t2 = __builtin_amdgcn_s_memtime();
    const double s1 = a.small[64 + (t & 63)];                // a second round trip, same lines' neighbours
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t3 = __builtin_amdgcn_s_memtime();
    out[t] = (double)s + s0 + s1;
    if (t == 0) { a.ticks[0] = t1 - t0; a.ticks[1] = t2 - t1; a.ticks[2] = t3 - t2; }
}
int main() {
    Args a; double *out;
    hipMalloc(&a.acc, 16 * 256 * 8); hipMalloc(&a.small, 4096 * 8); hipMalloc(&a.ticks, 64); hipMalloc(&out, 256 * 8);
    hipMemset(a.acc, 0, 16 * 256 * 8);
    for (int mode = 0; mode < 2; ++mode) for (int prod = 0; prod < 2; ++prod) {
        a.mode = mode;
        std::vector<double> v0, v1, v2;
        for (int rep = 0; rep < 30; ++rep) {
            if (prod) hipLaunchKernelGGL(producer, dim3(196), dim3(256), 0, 0, a.acc, a.small, 4096);
            hipLaunchKernelGGL(consumer, dim3(1), dim3(256), 0, 0, a, out);
            unsigned long long h[3]; hipMemcpy(h, a.ticks, sizeof h, hipMemcpyDeviceToHost);
            v0.push_back(h[0]); v1.push_back(h[1]); v2.push_back(h[2]);
        }
        std::sort(v0.begin(), v0.end()); std::sort(v1.begin(), v1.end()); std::sort(v2.begin(), v2.end());
        printf("%s loads, %s: first load %5.0f ticks, 16 rows %5.0f, second small load %5.0f (medians)\n", mode ? "nontemporal" : "plain", prod ? "after a producer kernel" : "data untouched since the last read", v0[15], v1[15], v2[15]);
    }
    return 0;
}
