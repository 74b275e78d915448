// How fast does a kernel pull a 640 x 480 byte mask (307 200 B) out of pinned host memory, by bytes per lane and workgroups in flight?  (k_cloud_team's phase A)
// build: hipcc --offload-arch=gfx950 -O3 -o pcie_read pcie_read.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
template <typename V> __global__ void k_read(const V *__restrict__ src, int n, unsigned *__restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned acc = 0;
    if (i < n) { const V v = src[i]; const unsigned *w = (const unsigned *)&v; for (unsigned k = 0; k < sizeof(V) / 4; ++k) acc |= w[k]; }
    if (acc == 0x12345678u) out[0] = acc;      // (keeps the load alive)
}
template <typename V> static float run(const void *src, int bytes, int threads, unsigned *out, int reps) {
    const int n = bytes / (int)sizeof(V), blocks = (n + threads - 1) / threads;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k_read<V>, dim3(blocks), dim3(threads), 0, 0, (const V *)src, n, out);
    hipEventRecord(e0, 0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k_read<V>, dim3(blocks), dim3(threads), 0, 0, (const V *)src, n, out);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f / reps;
}
int main() {
    const int bytes = 640 * 480;
    void *h; hipHostMalloc(&h, 4 << 20, hipHostMallocDefault); memset(h, 0, 4 << 20);
    void *d; hipMalloc(&d, 4 << 20); hipMemset(d, 0, 4 << 20);
    unsigned *out; hipMalloc(&out, 64);
    for (int sz : {bytes, 1280 * 720}) {
        printf("%d bytes from pinned host memory, us per launch (back to back):\n", sz);
        for (int th : {256, 1024}) {
            printf("  %4d threads: 4 B/lane %.2f   8 B/lane %.2f   16 B/lane %.2f\n", th, run<unsigned>(h, sz, th, out, 300), run<uint2>(h, sz, th, out, 300), run<uint4>(h, sz, th, out, 300));
        }
        printf("  the same from device memory (1024 threads): 4 B/lane %.2f  16 B/lane %.2f\n", run<unsigned>(d, sz, 1024, out, 300), run<uint4>(d, sz, 1024, out, 300));
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int i = 0; i < 10; ++i) hipMemcpyAsync(d, h, sz, hipMemcpyHostToDevice, 0);
        hipEventRecord(e0, 0);
        for (int i = 0; i < 100; ++i) hipMemcpyAsync(d, h, sz, hipMemcpyHostToDevice, 0);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        printf("  hipMemcpyAsync host -> device: %.2f us per copy (back to back)\n", ms * 10.f);
    }
    return 0;
}
