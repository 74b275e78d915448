// Latency probes for the banded LLE M-step (csrc/tdlo_mstep_band.hip) on one wave64 of gfx950: dependent v_mfma_f64_16x16x4 chains,
// MFMA -> VALU -> MFMA round trips, v_readlane -> VALU, DPP broadcasts.
// build: hipcc --offload-arch=gfx950 -O3 scripts/ubench/mfma64.hip -o scripts/ubench/mfma64 ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#define N 128
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void probe(double *out, unsigned long long *ticks, double seed) {
    const int lane = threadIdx.x;
    double a = seed * 1e-3 + lane * 1e-9, b = 1e-3, c = 1e-9, eps = 1e-30;
    d4 C = {seed, seed + 1, seed + 2, seed + 3}, D = {seed, seed - 1, seed - 2, seed - 3};
    unsigned long long t0, t1;
    int k = 0;
#define BEG() do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory"); } while (0)
#define END() do { asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory"); if (lane == 0) ticks[k] = t1 - t0; ++k; } while (0)
#define MF(Cv, av, bv) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(Cv) : "v"(av), "v"(bv))
#define FMA(d, x, y, z) asm volatile("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(x), "v"(y), "v"(z))
    // 0: dependent MFMA chain through C
    BEG();
#pragma unroll
    for (int i = 0; i < N; ++i) MF(C, a, b);
    END();
    // 1: two independent MFMA chains
    BEG();
#pragma unroll
    for (int i = 0; i < N / 2; ++i) { MF(C, a, b); MF(D, a, b); }
    END();
    // 2: MFMA -> one VALU fma on the result -> MFMA (A operand from the result)
    BEG();
#pragma unroll
    for (int i = 0; i < N; ++i) { MF(C, a, b); double c0 = C[0]; FMA(a, c0, eps, a); }
    END();
    // 3: MFMA -> fma -> fma -> MFMA
    BEG();
#pragma unroll
    for (int i = 0; i < N; ++i) { MF(C, a, b); double c1 = C[1]; FMA(b, c1, eps, b); FMA(a, b, eps, a); }
    END();
    // 4: dependent fma chain
    BEG();
#pragma unroll
    for (int i = 0; i < N; ++i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
    END();
    // 5: fma -> readlane pair -> fma with the SGPR operand (dependent)
    BEG();
#pragma unroll
    for (int i = 0; i < N; ++i) {
        int slo, shi, alo = __double2loint(a), ahi = __double2hiint(a);
        asm volatile("v_readlane_b32 %0, %2, 5\n\tv_readlane_b32 %1, %3, 5" : "=s"(slo), "=s"(shi) : "v"(alo), "v"(ahi));
        const double sv = __hiloint2double(shi, slo);
        asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(a) : "s"(sv), "v"(eps));
    }
    END();
    // 6: fma -> DPP row broadcast (two halves) -> fma
    BEG();
#pragma unroll
    for (int i = 0; i < N; ++i) {
        int xlo, xhi, alo = __double2loint(a), ahi = __double2hiint(a);
        asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %1, %3 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "=&v"(xlo), "=&v"(xhi) : "v"(alo), "v"(ahi));
        const double xb = __hiloint2double(xhi, xlo);
        FMA(a, xb, eps, a);
    }
    END();
    // 7: dependent rcp chain
    BEG();
#pragma unroll
    for (int i = 0; i < N; ++i) asm volatile("v_rcp_f64 %0, %0" : "+v"(a));
    END();
    // 8: rcp + third-order correction, dependent on the previous result (the prediction's chain)
    BEG();
#pragma unroll
    for (int i = 0; i < N; ++i) { double r0, e, p2; asm volatile("v_rcp_f64 %0, %1" : "=v"(r0) : "v"(a)); asm volatile("v_fma_f64 %0, -%1, %2, 1.0" : "=v"(e) : "v"(a), "v"(r0)); asm volatile("v_fma_f64 %0, %1, %1, %1" : "=v"(p2) : "v"(e)); asm volatile("v_fma_f64 %0, %1, %2, %1" : "=v"(a) : "v"(r0), "v"(p2)); }
    END();
    // 9: MFMA with distinct destination (D = A B + C, C untouched) dependent through A
    BEG();
#pragma unroll
    for (int i = 0; i < N; ++i) { asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %3" : "=&v"(D) : "v"(a), "v"(b), "v"(C)); double d0 = D[0]; FMA(a, d0, eps, a); }
    END();
    // 10: 4 independent fma (throughput)
    double a1 = a + 1, a2 = a + 2, a3 = a + 3;
    BEG();
#pragma unroll
    for (int i = 0; i < N / 4; ++i) {
        asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c)); asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a1) : "v"(b), "v"(c));
        asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a2) : "v"(b), "v"(c)); asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a3) : "v"(b), "v"(c));
    }
    END();
    // 11: f32 MFMA 16x16x4 dependent chain (for comparison)
    {
        typedef float f4 __attribute__((ext_vector_type(4)));
        f4 F = {(float)seed, 1.f, 2.f, 3.f}; float fa = (float)a, fb = 1e-3f;
        BEG();
#pragma unroll
        for (int i = 0; i < N; ++i) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(F) : "v"(fa), "v"(fb));
        END();
        a += F[0];
    }
    // 12: v_mfma_f64_4x4x4 dependent chain
    {
        double q = a;
        BEG();
#pragma unroll
        for (int i = 0; i < N; ++i) asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0" : "+v"(q) : "v"(a), "v"(b));
        END();
        a += q;
    }
    // 13..16: what overlaps with a dependent MFMA chain?  between two MFMAs: 8 independent f64 fma / 8 f32 fma / 8 integer adds / 8 v_mov_b32
    {
        double z0 = a + 1, z1 = a + 2, z2 = a + 3, z3 = a + 4;
        BEG();
#pragma unroll
        for (int i = 0; i < N; ++i) { MF(C, a, b);
            asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(z0) : "v"(b), "v"(c)); asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(z1) : "v"(b), "v"(c));
            asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(z2) : "v"(b), "v"(c)); asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(z3) : "v"(b), "v"(c));
            asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(z0) : "v"(b), "v"(c)); asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(z1) : "v"(b), "v"(c));
            asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(z2) : "v"(b), "v"(c)); asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(z3) : "v"(b), "v"(c)); }
        END();
        float y0 = (float)a, y1 = y0 + 1, y2 = y0 + 2, y3 = y0 + 3, fb2 = 1.0001f, fc2 = 1e-6f;
        BEG();
#pragma unroll
        for (int i = 0; i < N; ++i) { MF(C, a, b);
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(y0) : "v"(fb2), "v"(fc2)); asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(y1) : "v"(fb2), "v"(fc2));
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(y2) : "v"(fb2), "v"(fc2)); asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(y3) : "v"(fb2), "v"(fc2));
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(y0) : "v"(fb2), "v"(fc2)); asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(y1) : "v"(fb2), "v"(fc2));
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(y2) : "v"(fb2), "v"(fc2)); asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(y3) : "v"(fb2), "v"(fc2)); }
        END();
        int i0 = lane, i1 = lane + 1, i2 = lane + 2, i3 = lane + 3;
        BEG();
#pragma unroll
        for (int i = 0; i < N; ++i) { MF(C, a, b);
            asm volatile("v_add_u32 %0, %0, 3" : "+v"(i0)); asm volatile("v_add_u32 %0, %0, 3" : "+v"(i1)); asm volatile("v_add_u32 %0, %0, 3" : "+v"(i2)); asm volatile("v_add_u32 %0, %0, 3" : "+v"(i3));
            asm volatile("v_add_u32 %0, %0, 3" : "+v"(i0)); asm volatile("v_add_u32 %0, %0, 3" : "+v"(i1)); asm volatile("v_add_u32 %0, %0, 3" : "+v"(i2)); asm volatile("v_add_u32 %0, %0, 3" : "+v"(i3)); }
        END();
        BEG();
#pragma unroll
        for (int i = 0; i < N; ++i) { MF(C, a, b);
            asm volatile("v_rcp_f64 %0, %0" : "+v"(z0)); asm volatile("v_rcp_f64 %0, %0" : "+v"(z1)); }
        END();
        // 17: the same 8 f64 fma WITHOUT the MFMA (their own cost)
        BEG();
#pragma unroll
        for (int i = 0; i < N; ++i) {
            asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(z0) : "v"(b), "v"(c)); asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(z1) : "v"(b), "v"(c));
            asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(z2) : "v"(b), "v"(c)); asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(z3) : "v"(b), "v"(c));
            asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(z0) : "v"(b), "v"(c)); asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(z1) : "v"(b), "v"(c));
            asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(z2) : "v"(b), "v"(c)); asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(z3) : "v"(b), "v"(c)); }
        END();
        a += z0 + z1 + z2 + z3 + y0 + y1 + y2 + y3 + i0 + i1 + i2 + i3;
    }
    out[lane] = a + a1 + a2 + a3 + C[0] + C[1] + C[2] + C[3] + D[0] + D[1] + D[2] + D[3] + b;
}
int main() {
    double *out; unsigned long long *ticks;
    hipMalloc(&out, 64 * 8); hipMalloc(&ticks, 64 * 8);
    hipMemset(ticks, 0, 64 * 8);
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, out, ticks, 1.5); hipDeviceSynchronize(); }
    unsigned long long h[64]; hipMemcpy(h, ticks, sizeof h, hipMemcpyDeviceToHost);
    const char *names[] = {"dependent v_mfma_f64_16x16x4 (C chain)", "two independent MFMA chains", "MFMA -> fma -> MFMA (A operand)", "MFMA -> fma -> fma -> MFMA",
                           "dependent v_fma_f64", "fma -> v_readlane x2 -> fma(SGPR)", "fma -> DPP row_newbcast x2 -> fma", "dependent v_rcp_f64",
                           "rcp + 3 fma + add, dependent", "MFMA (dst != C) -> fma -> MFMA", "4 independent v_fma_f64", "dependent v_mfma_f32_16x16x4", "dependent v_mfma_f64_4x4x4", "MFMA + 8 independent f64 fma", "MFMA + 8 independent f32 fma", "MFMA + 8 independent v_add_u32", "MFMA + 2 independent v_rcp_f64", "8 independent f64 fma alone"};
    const int per[] = {N, N, N, N, N, N, N, N, N, N, N, N, N, N, N, N, N, N};
    for (int i = 0; i < 18; ++i) printf("%-45s %8.1f clocks per iteration\n", names[i], (double)h[i] / per[i]);
    return 0;
}
