// What one 1024-thread workgroup pays for its basic steps (k_cloud_fused's finishing workgroup): a barrier, an LDS read / write round trip,
// an LDS atomic with and without return, a wave scan.   hipcc --offload-arch=gfx950 -O3 -o wg1024 wg1024.hip && ./wg1024
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(1024) void k(unsigned long long *out, int lds_words) {
    extern __shared__ unsigned sm[];
    const int t = threadIdx.x;
    unsigned long long t0, t1;
    unsigned acc = 0;
    sm[t] = t;
    __syncthreads();
    // 1. barriers alone
    t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < 256; ++i) __syncthreads();
    t1 = __builtin_amdgcn_s_memtime();
    if (t == 0) out[0] = t1 - t0;
    // 2. dependent LDS read chain (pointer chase through sm)
    t0 = __builtin_amdgcn_s_memtime();
    unsigned p = t;
    for (int i = 0; i < 256; ++i) p = sm[p & 1023];
    t1 = __builtin_amdgcn_s_memtime();
    acc += p;
    if (t == 0) out[1] = t1 - t0;
    __syncthreads();
    // 3. independent LDS reads, 8 in flight
    t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < 32; ++i) {
        unsigned v[8];
#pragma unroll
        for (int k2 = 0; k2 < 8; ++k2) v[k2] = sm[(t + 64 * k2 + 517 * i) & 8191];
#pragma unroll
        for (int k2 = 0; k2 < 8; ++k2) acc += v[k2];
    }
    t1 = __builtin_amdgcn_s_memtime();
    if (t == 0) out[2] = t1 - t0;
    __syncthreads();
    // 4. LDS atomic add without return, distinct addresses
    t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < 256; ++i) (void)__hip_atomic_fetch_add(sm + ((t + 33 * i) & 8191), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    t1 = __builtin_amdgcn_s_memtime();
    if (t == 0) out[3] = t1 - t0;
    __syncthreads();
    // 5. LDS atomic add with return, dependent use
    t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < 256; ++i) acc += __hip_atomic_fetch_add(sm + ((t + 33 * i + (acc & 1)) & 8191), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    t1 = __builtin_amdgcn_s_memtime();
    if (t == 0) out[4] = t1 - t0;
    __syncthreads();
    // 6. write + barrier + read (a hand-over through LDS)
    t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < 256; ++i) { sm[(t * 7 + i) & 8191] = acc; __syncthreads(); acc += sm[(t + i) & 8191]; __syncthreads(); }
    t1 = __builtin_amdgcn_s_memtime();
    if (t == 0) out[5] = t1 - t0;
    // 7. wave inclusive scan by __shfl_up
    t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < 256; ++i) {
        int v = (int)acc;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(v, o); if ((t & 63) >= o) v += u; }
        acc = (unsigned)v;
    }
    t1 = __builtin_amdgcn_s_memtime();
    if (t == 0) out[6] = t1 - t0;
    // 8. ballot + 64-bit per-lane mask arithmetic (the match step)
    t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < 256; ++i) {
        unsigned long long same = ~0ull;
        const unsigned d = acc >> (i & 7);
#pragma unroll
        for (int b = 0; b < 8; ++b) { const unsigned one = (d >> b) & 1u; const unsigned long long bl = __ballot(one != 0u); same &= bl ^ ((unsigned long long)one - 1ull); }
        acc += (unsigned)__popcll(same);
    }
    t1 = __builtin_amdgcn_s_memtime();
    if (t == 0) out[7] = t1 - t0;
    if (acc == 0x12345678u) out[8] = acc;
}
int main() {
    unsigned long long *d, h[9];
    hipMalloc(&d, sizeof h);
    for (int lds : {32768, 163000}) {
        hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        for (int rep = 0; rep < 3; ++rep) { hipLaunchKernelGGL(k, dim3(1), dim3(1024), lds, 0, d, lds / 4); hipDeviceSynchronize(); }
        hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
        printf("LDS %d B: barrier %.0f clk | dependent LDS read %.0f | 8 independent LDS reads %.0f per batch | atomic no-return %.0f | atomic return %.0f | "
               "write+barrier+read+barrier %.0f | wave scan (6 shfl) %.0f | 8-ballot match %.0f\n", lds, h[0] / 256.0, h[1] / 256.0, h[2] / 32.0, h[3] / 256.0,
               h[4] / 256.0, h[5] / 256.0, h[6] / 256.0, h[7] / 256.0);
    }
    return 0;
}
