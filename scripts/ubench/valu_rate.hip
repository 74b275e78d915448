// Vector-issue rates with the GPU full (2048 workgroups x 4 waves = 8 waves per SIMD): lane-instructions per second of a few
// instruction kinds, eight independent chains per wave.  Settles the denominator of bench.py's valu_issue_frac.
// build: hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int kIters = 2048;

#define BODY8(INS) \
  asm volatile(INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7) \
    : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(b), "v"(c) : "vcc", "s12", "s13", "s14")

#define I_FMA(i)   "v_fma_f32 %" #i ", %" #i ", %8, %9\n"
#define I_MUL(i)   "v_mul_f32 %" #i ", %" #i ", %8\n"
#define I_ADDU(i)  "v_add_u32 %" #i ", %" #i ", %8\n"
#define I_EXP(i)   "v_exp_f32 %" #i ", %" #i "\n"
#define I_RCP(i)   "v_rcp_f32 %" #i ", %" #i "\n"
#define I_SQRT(i)  "v_sqrt_f32 %" #i ", %" #i "\n"
#define I_CND(i)   "v_cndmask_b32 %" #i ", %" #i ", %8, vcc\n"
#define I_MAX(i)   "v_max_f32 %" #i ", %" #i ", %8\n"
#define I_DPP(i)   "v_add_f32_dpp %" #i ", %" #i ", %8 row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define I_CVT(i)   "v_cvt_i32_f32 %" #i ", %" #i "\n"
#define I_CMP(i)   "v_cmp_lt_f32 vcc, %" #i ", %8\n"
#define I_LDEXP(i) "v_ldexp_f32 %" #i ", %" #i ", %8\n"
#define I_CNDS(i)  "v_cndmask_b32 %" #i ", %" #i ", %8, s[12:13]\n"
#define I_CMPCND(i) "v_cmp_lt_f32 vcc, %" #i ", %8\n v_cndmask_b32 %" #i ", %" #i ", %9, vcc\n"
#define I_FMAS(i)  "v_fma_f32 %" #i ", %" #i ", s12, %9\n"
#define I_MOV(i)   "v_mov_b32 %" #i ", %8\n"
#define I_AND(i)   "v_and_b32 %" #i ", %" #i ", %8\n"
#define I_MIN(i)   "v_min_f32 %" #i ", %" #i ", %8\n"
#define I_MAXU(i)  "v_max_u32 %" #i ", %" #i ", %8\n"
#define I_ADDF(i)  "v_add_f32 %" #i ", %" #i ", %8\n"
#define I_FMAC(i)  "v_fmac_f32 %" #i ", %8, %9\n"
#define I_LSHL(i)  "v_lshlrev_b32 %" #i ", 1, %" #i "\n"
#define I_CVTU(i)  "v_cvt_f32_u32 %" #i ", %" #i "\n"
#define I_MAD24(i) "v_mad_u32_u24 %" #i ", %" #i ", %8, %9\n"
#define I_MULLIT(i) "v_mul_f32 %" #i ", 0x3fb8aa3b, %" #i "\n"
#define I_MED3(i)  "v_med3_f32 %" #i ", %" #i ", %8, %9\n"
#define I_MAX3(i)  "v_max3_f32 %" #i ", %" #i ", %8, %9\n"
#define I_BFE(i)   "v_bfe_u32 %" #i ", %" #i ", 3, 5\n"
#define I_PERM(i)  "v_permlane32_swap_b32 %" #i ", %" #i "\n"
#define I_DPPMOV(i) "v_mov_b32_dpp %" #i ", %" #i " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
#define I_RDL(i)   "v_readlane_b32 s14, %" #i ", 3\n"
#define I_LOG(i)   "v_log_f32 %" #i ", %" #i "\n"
#define I_RSQ(i)   "v_rsq_f32 %" #i ", %" #i "\n"
#define I_ADDS(i)  "v_add_f32 %" #i ", s12, %" #i "\n"
#define I_CMPS(i)  "v_cmp_lt_f32 vcc, s12, %" #i "\n"
#define I_MULS(i)  "v_mul_f32 %" #i ", s12, %" #i "\n"
#define I_FMACS(i) "v_fmac_f32 %" #i ", s12, %8\n"

template <int K> __global__ __launch_bounds__(256) void k32(float* out, float b, float c) {
  float a[8];
  for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 1e-3f + i;
  for (int it = 0; it < kIters; ++it) {
    if (K == 0) BODY8(I_FMA);
    if (K == 1) BODY8(I_MUL);
    if (K == 2) BODY8(I_ADDU);
    if (K == 3) BODY8(I_EXP);
    if (K == 4) BODY8(I_RCP);
    if (K == 5) BODY8(I_SQRT);
    if (K == 6) BODY8(I_CND);
    if (K == 7) BODY8(I_MAX);
    if (K == 8) BODY8(I_DPP);
    if (K == 9) BODY8(I_CVT);
    if (K == 10) BODY8(I_CMP);
    if (K == 11) BODY8(I_LDEXP);
    if (K == 12) BODY8(I_CNDS);
    if (K == 13) BODY8(I_CMPCND);
    if (K == 14) BODY8(I_FMAS);
    if (K == 15) BODY8(I_MOV);
    if (K == 16) BODY8(I_AND);
    if (K == 17) BODY8(I_MIN);
    if (K == 18) BODY8(I_MAXU);
    if (K == 19) BODY8(I_ADDF);
    if (K == 20) BODY8(I_FMAC);
    if (K == 21) BODY8(I_LSHL);
    if (K == 22) BODY8(I_CVTU);
    if (K == 23) BODY8(I_MAD24);
    if (K == 24) BODY8(I_MULLIT);
    if (K == 25) BODY8(I_MED3);
    if (K == 26) BODY8(I_MAX3);
    if (K == 27) BODY8(I_BFE);
    if (K == 28) BODY8(I_PERM);
    if (K == 29) BODY8(I_DPPMOV);
    if (K == 30) BODY8(I_RDL);
    if (K == 31) BODY8(I_LOG);
    if (K == 32) BODY8(I_RSQ);
    if (K == 33) BODY8(I_ADDS);
    if (K == 34) BODY8(I_CMPS);
    if (K == 35) BODY8(I_MULS);
    if (K == 36) BODY8(I_FMACS);
  }
  float s = 0; for (int i = 0; i < 8; ++i) s += a[i];
  if (s == 123.456f) out[0] = s;
}

#define P_FMA(i)  "v_pk_fma_f32 %" #i ", %" #i ", %8, %9\n"
#define P_MUL(i)  "v_pk_mul_f32 %" #i ", %" #i ", %8\n"
#define P_ADD(i)  "v_pk_add_f32 %" #i ", %" #i ", %8\n"
#define D_FMA(i)  "v_fma_f64 %" #i ", %" #i ", %8, %9\n"
#define D_MUL(i)  "v_mul_f64 %" #i ", %" #i ", %8\n"
#define D_ADD(i)  "v_add_f64 %" #i ", %" #i ", %8\n"
#define P_ADDS(i) "v_pk_add_f32 %" #i ", %" #i ", s[12:13]\n"
template <int K> __global__ __launch_bounds__(256) void k64(double* out, double b, double c) {
  double a[8];
  for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 1e-3 + i;
  for (int it = 0; it < kIters; ++it) {
    if (K == 0) BODY8(P_FMA);
    if (K == 1) BODY8(P_MUL);
    if (K == 2) BODY8(P_ADD);
    if (K == 3) BODY8(D_FMA);
    if (K == 4) BODY8(D_MUL);
    if (K == 5) BODY8(D_ADD);
    if (K == 6) BODY8(P_ADDS);
  }
  double s = 0; for (int i = 0; i < 8; ++i) s += a[i];
  if (s == 123.456) out[0] = s;
}

// mixed: a VALU op beside scalar ops or LDS reads of the same wave
__global__ __launch_bounds__(256) void kmix_salu(float* out, float b, float c) {
  float a[8];
  for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 1e-3f + i;
  int s0 = 0;
  for (int it = 0; it < kIters; ++it) {
    asm volatile("v_fma_f32 %0, %0, %9, %10\n s_add_u32 %8, %8, 1\n v_fma_f32 %1, %1, %9, %10\n s_add_u32 %8, %8, 1\n"
                 "v_fma_f32 %2, %2, %9, %10\n s_add_u32 %8, %8, 1\n v_fma_f32 %3, %3, %9, %10\n s_add_u32 %8, %8, 1\n"
                 "v_fma_f32 %4, %4, %9, %10\n s_add_u32 %8, %8, 1\n v_fma_f32 %5, %5, %9, %10\n s_add_u32 %8, %8, 1\n"
                 "v_fma_f32 %6, %6, %9, %10\n s_add_u32 %8, %8, 1\n v_fma_f32 %7, %7, %9, %10\n s_add_u32 %8, %8, 1\n"
                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+s"(s0) : "v"(b), "v"(c) : "scc");
  }
  float s = s0; for (int i = 0; i < 8; ++i) s += a[i];
  if (s == 123.456f) out[0] = s;
}
__global__ __launch_bounds__(256) void kmix_lds(float* out, float b, float c) {
  __shared__ float sh[256 * 2];
  sh[threadIdx.x] = b; sh[threadIdx.x + 256] = c;
  __syncthreads();
  float a[8];
  for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 1e-3f + i;
  unsigned addr = threadIdx.x * 4;
  float l0 = 0, l1 = 0;
  for (int it = 0; it < kIters; ++it) {
    asm volatile("ds_read_b32 %8, %10\n v_fma_f32 %0, %0, %11, %12\n v_fma_f32 %1, %1, %11, %12\n v_fma_f32 %2, %2, %11, %12\n v_fma_f32 %3, %3, %11, %12\n"
                 "ds_read_b32 %9, %10 offset:1024\n v_fma_f32 %4, %4, %11, %12\n v_fma_f32 %5, %5, %11, %12\n v_fma_f32 %6, %6, %11, %12\n v_fma_f32 %7, %7, %11, %12\n"
                 "s_waitcnt lgkmcnt(0)\n"
                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "=&v"(l0), "=&v"(l1) : "v"(addr), "v"(b), "v"(c));
    a[0] += l0; a[4] += l1;
  }
  float s = 0; for (int i = 0; i < 8; ++i) s += a[i];
  if (s == 123.456f) out[0] = s;
}

// shader clocks (s_memtime) against the 100 MHz wall clock over a long VALU loop with the GPU full: the frequency the SIMDs really run at
__global__ __launch_bounds__(256) void kclock(unsigned long long* out, float b, float c) {
  float a[8];
  for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 1e-3f + i;
  unsigned long long t0 = __builtin_readcyclecounter(), w0 = wall_clock64();
  for (int it = 0; it < kIters * 4; ++it) BODY8(I_FMA);
  unsigned long long t1 = __builtin_readcyclecounter(), w1 = wall_clock64();
  float s = 0; for (int i = 0; i < 8; ++i) s += a[i];
  if (blockIdx.x == 0 && threadIdx.x == 0) { out[0] = t1 - t0; out[1] = w1 - w0; out[2] = s; }
}

template <class F> static double time_ms(F launch) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  launch(); hipDeviceSynchronize();
  float best = 1e30f;
  for (int r = 0; r < 5; ++r) { hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms; }
  return best;
}

int main(int argc, char** argv) {
  setvbuf(stdout, nullptr, _IOLBF, 0);
  int blocks = argc > 1 ? atoi(argv[1]) : 2048;
  float* d; CK(hipMalloc(&d, 1024));
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  double clk = p.clockRate * 1e3;   // Hz
  int simds = p.multiProcessorCount * 4;
  printf("%s: %d CUs, %.0f MHz, %d blocks of 256\n", p.name, p.multiProcessorCount, clk / 1e6, blocks);
  auto rep = [&](const char* name, double ms, double per_iter) {
    double n = double(blocks) * 4 * kIters * per_iter;            // wave-instructions
    double per_simd_cycle = n / (ms * 1e-3 * clk) / simds;        // wave-instructions per SIMD per cycle
    printf("%-34s %8.3f ms  %7.2f T lane-instr/s  %5.2f cycles per wave-instruction per SIMD\n", name, ms, n * 64 / (ms * 1e-3) / 1e12, 1.0 / per_simd_cycle);
  };
#define R32(K, name) rep(name, time_ms([&] { hipLaunchKernelGGL(k32<K>, dim3(blocks), dim3(256), 0, 0, d, 1.0001f, 1e-9f); }), 8)
#define R64(K, name) rep(name, time_ms([&] { hipLaunchKernelGGL(k64<K>, dim3(blocks), dim3(256), 0, 0, (double*)d, 1.0001, 1e-9); }), 8)
  {
    unsigned long long* dc; CK(hipMalloc(&dc, 64)); unsigned long long h[3];
    for (int nb : {1, blocks}) {
      hipLaunchKernelGGL(kclock, dim3(nb), dim3(256), 0, 0, dc, 1.0001f, 1e-9f); CK(hipMemcpy(h, dc, 24, hipMemcpyDeviceToHost));
      printf("%d block(s): %llu shader clocks in %llu wall ticks of 10 ns -> %.0f MHz; %.2f clocks per v_fma_f32 of this wave\n", nb, h[0], h[1], h[0] / (h[1] * 1e-8) / 1e6, double(h[0]) / (kIters * 4 * 8));
    }
  }
  R32(0, "v_fma_f32"); R32(1, "v_mul_f32"); R32(2, "v_add_u32"); R32(7, "v_max_f32"); R32(6, "v_cndmask_b32"); R32(10, "v_cmp_lt_f32 -> vcc");
  R32(8, "v_add_f32 DPP row_shr:1"); R32(9, "v_cvt_i32_f32"); R32(11, "v_ldexp_f32");
  R32(12, "v_cndmask_b32 (SGPR-pair mask)"); R32(13, "v_cmp + v_cndmask pair (per pair)"); R32(14, "v_fma_f32 with an SGPR operand");
  R32(15, "v_mov_b32"); R32(16, "v_and_b32"); R32(17, "v_min_f32"); R32(18, "v_max_u32"); R32(19, "v_add_f32"); R32(20, "v_fmac_f32 (VOP2)");
  R32(21, "v_lshlrev_b32"); R32(22, "v_cvt_f32_u32"); R32(23, "v_mad_u32_u24"); R32(24, "v_mul_f32 with a literal"); R32(25, "v_med3_f32"); R32(26, "v_max3_f32");
  R32(27, "v_bfe_u32"); R32(28, "v_permlane32_swap"); R32(29, "v_mov_b32 DPP quad_perm"); R32(30, "v_readlane_b32"); R32(31, "v_log_f32"); R32(32, "v_rsq_f32");
  R32(33, "v_add_f32 (VOP2) with an SGPR operand"); R32(35, "v_mul_f32 (VOP2) with an SGPR operand"); R32(36, "v_fmac_f32 (VOP2) with an SGPR operand"); R32(34, "v_cmp_lt_f32 with an SGPR operand");
  R32(3, "v_exp_f32"); R32(4, "v_rcp_f32"); R32(5, "v_sqrt_f32");
  R64(0, "v_pk_fma_f32 (2 fp32 per lane)"); R64(1, "v_pk_mul_f32"); R64(2, "v_pk_add_f32");
  R64(6, "v_pk_add_f32 with an SGPR pair");
  R64(3, "v_fma_f64"); R64(4, "v_mul_f64"); R64(5, "v_add_f64");
  rep("v_fma_f32 + s_add_u32 alternating", time_ms([&] { hipLaunchKernelGGL(kmix_salu, dim3(blocks), dim3(256), 0, 0, d, 1.0001f, 1e-9f); }), 8);
  rep("8 v_fma_f32 + 2 ds_read_b32 (VALU)", time_ms([&] { hipLaunchKernelGGL(kmix_lds, dim3(blocks), dim3(256), 0, 0, d, 1.0001f, 1e-9f); }), 10);
  return 0;
}
