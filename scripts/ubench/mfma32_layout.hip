// Operand / result layout of v_mfma_f32_16x16x4_f32 on gfx950, checked with an asymmetric product.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void k(float *out) {
    const int lane = threadIdx.x, i = lane & 15, k = lane >> 4;
    // A[i][k] = 1 + i + 100 k   (lane i + 16 k),  B[k][j] = (j + 1) * (k == 0 ? 1 : (k == 1 ? 10 : (k == 2 ? 0 : 0)))  (lane j + 16 k)
    const float a = 1.f + i + 100.f * k;
    const float b = (i + 1) * (k == 0 ? 1.f : (k == 1 ? 10.f : 0.f));
    f4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[lane * 4 + r] = c[r];
}
int main() {
    float *d; hipMalloc(&d, 256 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d); hipDeviceSynchronize();
    float h[256]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    // expected C[i][j] = A[i][0] B[0][j] + A[i][1] B[1][j] = (1 + i)(j + 1) + (101 + i) 10 (j + 1)
    int bad_a = 0, bad_b = 0;
    for (int lane = 0; lane < 64; ++lane) for (int r = 0; r < 4; ++r) {
        const int j = lane & 15, g = lane >> 4;
        const int ia = 4 * g + r, ib = g + 4 * r;      // hypothesis a: row = 4 g + r; hypothesis b: row = g + 4 r
        const float ea = (1 + ia) * (j + 1) + (101 + ia) * 10.f * (j + 1), eb = (1 + ib) * (j + 1) + (101 + ib) * 10.f * (j + 1);
        if (h[lane * 4 + r] != ea) ++bad_a;
        if (h[lane * 4 + r] != eb) ++bad_b;
    }
    printf("row = 4 g + r: %d mismatches; row = g + 4 r: %d mismatches; C[lane 0] = %g %g %g %g\n", bad_a, bad_b, h[0], h[1], h[2], h[3]);
    return 0;
}
