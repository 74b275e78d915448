// What would ONE persistent launch for the whole EM loop cost per iteration?  196 producer workgroups (E-step-shaped: read the consumer's
// output, work, int64 atomics into 8 replica rows, ticket) + 1 consumer workgroup (M-step-shaped: waits for the 196 tickets, reads the rows,
// a dependent chain, publishes, raises a flag the producers wait for) -- against the same work as two dependent launches per iteration.
// All hand-offs with agent-scope atomics / L2-bypassing accesses, no release fence (nothing but atomics is written before a ticket).
// build: hipcc --offload-arch=gfx950 -O3 scripts/ubench/persist.hip -o scripts/ubench/persist
#include <hip/hip_runtime.h>
#include <cstdio>
#define AG __HIP_MEMORY_SCOPE_AGENT
__device__ __forceinline__ float spin_work(int spin, float a) { for (int i = 0; i < spin; ++i) a = a * 1.0001f + 0.5f; return a; }
__device__ __forceinline__ double chain(int spin2, double a) { for (int i = 0; i < spin2; ++i) a = a * 1.0000001 + 0.5; return a; }
__global__ void producer(long long *acc, const double *nodes, int it, int spin, float *sink) {
    const float a = spin_work(spin, (float)nodes[threadIdx.x & 63]);
    if (a == 12345.f) sink[0] = a;
    if (threadIdx.x < 201) __hip_atomic_fetch_add(acc + (size_t)((it & 1) * 8 + blockIdx.x % 8) * 256 + threadIdx.x, (long long)(a * 1e-3f) + 1, __ATOMIC_RELAXED, AG);
}
__global__ void consumer(long long *acc, double *nodes, int it, int spin2) {
    const int t = threadIdx.x;
    long long s = 0;
    if (t < 201) for (int r = 0; r < 8; ++r) s += acc[(size_t)((it & 1) * 8 + r) * 256 + t];
    nodes[t] = chain(spin2, (double)s);
}
__global__ void persist(long long *acc, double *nodes, unsigned *ticket, unsigned *flag, int iters, int spin, int spin2, float *sink) {
    const int t = threadIdx.x;
    const unsigned np = gridDim.x - 1;
    if (blockIdx.x == 0) {
        for (int it = 0; it < iters; ++it) {
            if (t == 0) while (__hip_atomic_load(ticket, __ATOMIC_RELAXED, AG) < np * (unsigned)(it + 1)) __builtin_amdgcn_s_sleep(1);
            __syncthreads();
            long long s = 0;
            if (t < 201) for (int r = 0; r < 8; ++r) s += __hip_atomic_load(acc + (size_t)((it & 1) * 8 + r) * 256 + t, __ATOMIC_RELAXED, AG);
            __hip_atomic_store(nodes + t, chain(spin2, (double)s), __ATOMIC_RELAXED, AG);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (t == 0) __hip_atomic_store(flag, (unsigned)(it + 1), __ATOMIC_RELAXED, AG);
        }
    } else {
        for (int it = 0; it < iters; ++it) {
            if (t == 0) while (__hip_atomic_load(flag, __ATOMIC_RELAXED, AG) < (unsigned)it) __builtin_amdgcn_s_sleep(1);
            __syncthreads();
            const float a = spin_work(spin, (float)__hip_atomic_load(nodes + (t & 63), __ATOMIC_RELAXED, AG));
            if (a == 12345.f) sink[0] = a;
            if (t < 201) __hip_atomic_fetch_add(acc + (size_t)((it & 1) * 8 + blockIdx.x % 8) * 256 + t, (long long)(a * 1e-3f) + 1, __ATOMIC_RELAXED, AG);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (t == 0) __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, AG);
        }
    }
}
int main() {
    long long *acc; double *nodes; unsigned *ticket, *flag; float *sink;
    (void)hipMalloc(&acc, 16 * 256 * 8); (void)hipMalloc(&nodes, 256 * 8); hipMalloc(&ticket, 64); hipMalloc(&flag, 64); hipMalloc(&sink, 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 500;
    const int cfg[3][2] = {{0, 0}, {300, 1000}, {120, 600}};       // (producer spin, consumer chain): none / ~5 + ~8 us / ~2 + ~5 us
    for (int c = 0; c < 3; ++c) for (int mode = 0; mode < 2; ++mode) {
        float best = 1e30f;
        for (int trial = 0; trial < 5; ++trial) {
            hipMemset(acc, 0, 16 * 256 * 8); hipMemset(nodes, 0, 256 * 8); hipMemset(ticket, 0, 64); hipMemset(flag, 0, 64);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            if (mode == 0) for (int it = 0; it < iters; ++it) {
                hipLaunchKernelGGL(producer, dim3(196), dim3(256), 0, 0, acc, nodes, it, cfg[c][0], sink);
                hipLaunchKernelGGL(consumer, dim3(1), dim3(256), 0, 0, acc, nodes, it, cfg[c][1]);
            } else hipLaunchKernelGGL(persist, dim3(197), dim3(256), 0, 0, acc, nodes, ticket, flag, iters, cfg[c][0], cfg[c][1], sink);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
        }
        printf("spin %4d / chain %4d  %-28s %7.2f us per iteration\n", cfg[c][0], cfg[c][1], mode ? "one persistent launch" : "two dependent launches", best * 1e3f / iters);
    }
    return 0;
}
