// Instruction-latency probes for one wave64 on gfx950 (fp64 chains, LDS round trips, exec-masked stores).
// build: hipcc --offload-arch=gfx950 -O3 scripts/ubench/lat.hip -o scripts/ubench/lat ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#define N 256
__global__ void probe(double *out, unsigned long long *ticks, double seed) {
    __shared__ double lds[4096];
    const int lane = threadIdx.x;
    for (int i = lane; i < 4096; i += 64) lds[i] = 1.0 + 1e-9 * i;
    __syncthreads();
    double a = seed + lane * 1e-12, b = 1.0000001, c = 1e-9;
    unsigned long long t0, t1; unsigned long long r0, r1;
    int k = 0;
#define BEG() do { __builtin_amdgcn_s_waitcnt(0); t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); __builtin_amdgcn_s_waitcnt(0);} while (0)
#define END() do { __builtin_amdgcn_s_waitcnt(0); t1 = __builtin_amdgcn_s_memtime(); r1 = __builtin_amdgcn_s_memrealtime(); __builtin_amdgcn_s_waitcnt(0); if (lane == 0) { ticks[2*k] = t1 - t0; ticks[2*k+1] = r1 - r0; } ++k; } while (0)
    // 0: dependent fma chain
    BEG();
#pragma unroll
    for (int i = 0; i < N; ++i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
    END();
    // 1: 4 independent fma chains
    double a1 = a + 1, a2 = a + 2, a3 = a + 3;
    BEG();
#pragma unroll
    for (int i = 0; i < N / 4; ++i) {
        asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
        asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a1) : "v"(b), "v"(c));
        asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a2) : "v"(b), "v"(c));
        asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a3) : "v"(b), "v"(c));
    }
    END();
    a += a1 + a2 + a3;
    // 2: dependent rcp chain
    BEG();
#pragma unroll
    for (int i = 0; i < N; ++i) asm volatile("v_rcp_f64 %0, %0" : "+v"(a));
    END();
    // 3: dependent fp32 fma chain
    float fa = (float)a, fb = 1.0001f, fc = 1e-6f;
    BEG();
#pragma unroll
    for (int i = 0; i < N; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(fa) : "v"(fb), "v"(fc));
    END();
    a += fa;
    // 4: LDS pointer chase (ds_read_b64 dependent)
    {
        int idx = lane & 7;
        BEG();
#pragma unroll
        for (int i = 0; i < N; ++i) { const double v = lds[idx]; idx = ((int)(v * 3.0) + i) & 1023; }
        END();
        a += idx;
    }
    // 5: s_nop-free scalar chain: N s_add
    {
        int s = __builtin_amdgcn_readfirstlane(lane);
        BEG();
#pragma unroll
        for (int i = 0; i < N; ++i) asm volatile("s_add_i32 %0, %0, 3" : "+s"(s));
        END();
        a += s;
    }
    // 6: N x (exec-masked ds_write via branch)
    {
        BEG();
#pragma unroll 8
        for (int i = 0; i < N; ++i) { if (lane == 0) lds[2048 + i] = a; asm volatile("" ::: "memory"); }
        END();
    }
    // 7: dependent v_mul_f64
    BEG();
#pragma unroll
    for (int i = 0; i < N; ++i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a) : "v"(b));
    END();
    // 8: N independent ds_read_b128 then wait
    {
        typedef double d2 __attribute__((ext_vector_type(2)));
        d2 acc = {0, 0};
        BEG();
#pragma unroll
        for (int i = 0; i < N; ++i) { acc += *(const d2 *)(lds + 2 * ((i * 7) & 511)); }
        END();
        a += acc.x + acc.y;
    }
    // 9: dependent fma interleaved with 3 independent s_add each (does scalar issue overlap?)
    {
        int s = __builtin_amdgcn_readfirstlane(lane);
        BEG();
#pragma unroll
        for (int i = 0; i < N; ++i) {
            asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
            asm volatile("s_add_i32 %0, %0, 3" : "+s"(s));
            asm volatile("s_add_i32 %0, %0, 5" : "+s"(s));
            asm volatile("s_add_i32 %0, %0, 7" : "+s"(s));
        }
        END();
        a += s;
    }
    // 10: v_readlane x N
    {
        int v = lane, s = 0;
        BEG();
#pragma unroll
        for (int i = 0; i < N; ++i) { s += __builtin_amdgcn_readlane(v, i & 63); }
        END();
        a += s;
    }
    // 11: DPP 64-bit move chain (row_shr:1 on both halves) + add
    {
        BEG();
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(a), 0x111, 0xf, 0xf, false);
            const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(a), 0x111, 0xf, 0xf, false);
            a += __hiloint2double(hi, lo);
        }
        END();
    }
    out[blockIdx.x * 64 + lane] = a;
}
int main() {
    double *out; unsigned long long *ticks;
    hipMalloc(&out, 64 * 8 * 4); hipMalloc(&ticks, 64 * 8);
    hipMemset(ticks, 0, 64 * 8);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, out, ticks, 1.5);
    unsigned long long h[64];
    hipMemcpy(h, ticks, sizeof h, hipMemcpyDeviceToHost);
    const char *names[] = {"dep fma_f64", "4 indep fma_f64 chains", "dep rcp_f64", "dep fma_f32", "LDS pointer chase b64", "dep s_add", "exec-masked ds_write (branch)", "dep mul_f64", "indep ds_read_b128 + add", "fma_f64 + 3 s_add", "v_readlane + s_add", "dpp mov x2 + add_f64"};
    for (int k = 0; k < 12; ++k) printf("%-32s memtime %6.2f ticks/op   realtime %7.3f ns/op\n", names[k], (double)h[2*k] / N, (double)h[2*k+1] * 10.0 / N);
    return 0;
}
