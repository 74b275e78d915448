// Packed fp32 issue rates with the GPU full (2048 workgroups x 4 waves): does v_pk_fma_f32 (two FMAs per lane and instruction) issue like v_fma_f32?
// With an SGPR-pair operand?  build: hipcc --offload-arch=gfx950 -O3 pk_rate.hip -o pk_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
constexpr int kIters = 2048;
#define BODY8(INS) asm volatile(INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7) \
    : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(b), "v"(c), "s"(sb))
#define I_PKFMA(i)  "v_pk_fma_f32 %" #i ", %" #i ", %8, %9\n"
#define I_PKMUL(i)  "v_pk_mul_f32 %" #i ", %" #i ", %8\n"
#define I_PKADD(i)  "v_pk_add_f32 %" #i ", %" #i ", %8\n"
#define I_PKFMAS(i) "v_pk_fma_f32 %" #i ", %" #i ", %10, %9\n"
#define I_PKADDS(i) "v_pk_add_f32 %" #i ", %" #i ", %10\n"
template <int K> __global__ __launch_bounds__(256) void k(f2 *out, f2 b, f2 c, f2 sbv) {
    f2 a[8];
    for (int i = 0; i < 8; ++i) { a[i].x = threadIdx.x * 1e-3f + i; a[i].y = a[i].x + 0.5f; }
    f2 sb; sb.x = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, sbv.x))); sb.y = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, sbv.y)));
    for (int it = 0; it < kIters; ++it) {
        if (K == 0) BODY8(I_PKFMA);
        if (K == 1) BODY8(I_PKMUL);
        if (K == 2) BODY8(I_PKADD);
        if (K == 3) BODY8(I_PKFMAS);
        if (K == 4) BODY8(I_PKADDS);
    }
    f2 s = a[0]; for (int i = 1; i < 8; ++i) s += a[i];
    if (s.x == 123.456f) out[threadIdx.x] = s;
}
template <int K> float run(f2 *d) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    f2 b = {1.0001f, 0.9999f}, c = {1e-6f, -1e-6f};
    hipLaunchKernelGGL(k<K>, dim3(2048), dim3(256), 0, 0, d, b, c, b);
    hipEventRecord(e0); hipLaunchKernelGGL(k<K>, dim3(2048), dim3(256), 0, 0, d, b, c, b); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
    f2 *d; hipMalloc(&d, 4096);
    const char *names[5] = {"v_pk_fma_f32", "v_pk_mul_f32", "v_pk_add_f32", "v_pk_fma_f32 with an SGPR-pair operand", "v_pk_add_f32 with an SGPR-pair operand"};
    float ms[5] = {run<0>(d), run<1>(d), run<2>(d), run<3>(d), run<4>(d)};
    for (int i = 0; i < 5; ++i) {
        const double inst = 2048.0 * 4 * kIters * 8;      // wave-instructions
        printf("%-42s %.3f ms  %.2f T lane-instr/s (x2 flops each)  %.2f cycles per wave-instruction per SIMD at 1.9 GHz\n", names[i], ms[i], inst * 64 / (ms[i] * 1e-3) / 1e12,
               ms[i] * 1e-3 * 1.9e9 / (inst / 1024.0));
    }
    return 0;
}
