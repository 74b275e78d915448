// bar.hip -- can the host CPU store straight into device memory (PCIe BAR), and what does a host -> kernel hand-over cost that way?
// A kernel launched ahead spins on a flag word, then reads a payload of 4 x 45 doubles and reports through pinned host memory (as tracking_step's
// M-step launched ahead of its priors does: FrameDev::spec_flag, late_aJ / late_aYd, the results mailbox).  Variants of where flag + payload live:
//   host    pinned host memory (hipHostMalloc): the kernel polls and reads over PCIe            -- what the library does
//   dev     hipMalloc'd device memory written by the CPU through the BAR mapping (if the process can dereference it at all)
//   fine    hipExtMallocWithFlags(hipDeviceMallocFinegrained) / uncached, likewise
// usage: bar [rounds]          build: hipcc --offload-arch=gfx950 -O2 -o bar bar.hip
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <csetjmp>
#include <csignal>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <immintrin.h>
#include <thread>
#include <vector>
#include <algorithm>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k_wait(const unsigned long long *flag, const double *payload, int n, unsigned long long want, double *mbox) {
    __shared__ int go;
    if (threadIdx.x == 0) {
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        int g = 0;
        for (;;) {
            if (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == want) { g = 1; break; }
            if (__builtin_amdgcn_s_memrealtime() - t0 > 100000000ull) break;      // 1 s
            __builtin_amdgcn_s_sleep(4);
        }
        go = g;
    }
    __syncthreads();
    double a = 0;
    if (go) for (int i = threadIdx.x; i < n; i += blockDim.x) a += __hip_atomic_load(payload + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    for (int d = 32; d >= 1; d >>= 1) a += __shfl_xor(a, d);
    __shared__ double w[4];
    if ((threadIdx.x & 63) == 0) w[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) {
        mbox[1] = w[0] + w[1] + w[2] + w[3];
        __hip_atomic_store((unsigned long long *)mbox, go ? want : ~0ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

static sigjmp_buf g_jmp;
static void on_segv(int) { siglongjmp(g_jmp, 1); }

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char **argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 2000, n = 180;
    int large_bar = -1;
    hipDeviceGetAttribute(&large_bar, hipDeviceAttributeIsLargeBar, 0);
    printf("hipDeviceAttributeIsLargeBar = %d\n", large_bar);
    double *mbox; CHK(hipHostMalloc((void **)&mbox, 64, hipHostMallocDefault));
    hipStream_t s; CHK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    struct Var { const char *name; void *buf; bool ok; };
    std::vector<Var> vars;
    { void *p = nullptr; CHK(hipHostMalloc(&p, 4096, hipHostMallocDefault)); vars.push_back({"host ", p, true}); }
    { void *p = nullptr; if (hipMalloc(&p, 4096) == hipSuccess) vars.push_back({"dev  ", p, true}); }
    { void *p = nullptr; if (hipExtMallocWithFlags(&p, 4096, hipDeviceMallocFinegrained) == hipSuccess) vars.push_back({"fine ", p, true}); else (void)hipGetLastError(); }
    { void *p = nullptr; if (hipExtMallocWithFlags(&p, 4096, hipDeviceMallocUncached) == hipSuccess) vars.push_back({"uncch", p, true}); else (void)hipGetLastError(); }
    signal(SIGSEGV, on_segv); signal(SIGBUS, on_segv);
    for (auto &v : vars) {
        if (v.buf == vars[0].buf) continue;
        if (sigsetjmp(g_jmp, 1) == 0) { volatile unsigned long long *q = (volatile unsigned long long *)v.buf; q[0] = 0; (void)q[0]; }
        else { v.ok = false; printf("%s: the host cannot dereference this memory (fault)\n", v.name); }
    }
    for (auto &v : vars) {
        if (!v.ok) continue;
        unsigned long long *flag = (unsigned long long *)v.buf;
        double *payload = (double *)v.buf + 8;
        std::vector<double> lat;
        int bad = 0;
        for (int r = 0; r < rounds + 20; ++r) {
            const unsigned long long want = 0x100000000ull + r;
            ((volatile unsigned long long *)mbox)[0] = 0;
            hipLaunchKernelGGL(k_wait, dim3(1), dim3(256), 0, s, flag, payload, n, want, mbox);
            const double t_l = now_us();
            while (now_us() - t_l < 30.0) { }                    // the kernel is certainly polling by now
            double sum = 0;
            const double t0 = now_us();
            for (int i = 0; i < n; ++i) { payload[i] = r + i; sum += r + i; }
            _mm_sfence();
            __atomic_store_n(flag, want, __ATOMIC_RELEASE);
            _mm_sfence();
            while (__atomic_load_n((unsigned long long *)mbox, __ATOMIC_ACQUIRE) == 0) { if (now_us() - t0 > 2e6) break; }
            const double t1 = now_us();
            const unsigned long long got = __atomic_load_n((unsigned long long *)mbox, __ATOMIC_ACQUIRE);
            if (got != want || mbox[1] != sum) ++bad;
            if (r >= 20) lat.push_back(t1 - t0);
            CHK(hipStreamSynchronize(s));
        }
        std::sort(lat.begin(), lat.end());
        printf("%s: host write -> result visible: median %.2f us, p10 %.2f, p90 %.2f (%d rounds, %d wrong or timed out)\n", v.name, lat[lat.size() / 2], lat[lat.size() / 10],
               lat[lat.size() * 9 / 10], rounds, bad);
    }
    return 0;
}
