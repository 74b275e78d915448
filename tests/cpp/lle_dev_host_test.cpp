// The device routine that forms the LLE regulariser (trackdlo_amd/csrc/tdlo_lle_dev.h: thread = node, 6 x 6 systems embedded and predicated) compiled for
// the HOST with one "thread" (MB = 1: the loops cover every node, the barriers are no-ops) against the library's host routine
// (tdlo_calc_lle_regulariser), bit for bit: the CPU-side check of the routine's LOGIC -- neighbourhoods at the chain's ends, the embedding, the
// predication, the 1e-5 regularisation of trackdlo.cpp:139-144.  (That the GPU build performs the same operations -- no fused multiply-adds, IEEE
// division -- is what tests/test_lle_device_gpu.py checks on the device.)
// build: g++ -O2 -std=c++17 tests/cpp/lle_dev_host_test.cpp -o tests/cpp/lle_dev_host_test -Ltrackdlo_amd -ltrackdlo_hip -Wl,-rpath,$PWD/trackdlo_amd
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>
#include "../../include/trackdlo_hip.h"
#define __device__
#define __forceinline__ inline
#define __HIP_MEMORY_SCOPE_AGENT 0
template <typename T> static inline T __hip_atomic_load(const T *p, int, int) { return *p; }
static inline void __syncthreads() {}
using std::fabs;
#include "../../trackdlo_amd/csrc/tdlo_lle_dev.h"

static unsigned long long rs = 0x9E3779B97F4A7C15ull;
static double ur() { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return (rs >> 11) * (1.0 / 9007199254740992.0); }
static double nr() { return std::sqrt(-2 * std::log(ur() + 1e-300)) * std::cos(6.283185307179586 * ur()); }

int main() {
    int bad = 0, cases = 0;
    for (int rep = 0; rep < 600; ++rep) {
        const int kind = rep % 6;
        const int M = rep < 40 ? 1 + rep % 20 : 1 + (int)(ur() * 256);
        std::vector<double> Y(3 * (size_t)M), Hh(13 * (size_t)M), Hd(13 * (size_t)M), Ab(7 * (size_t)M);
        double p[3] = {0.1, -0.2, 0.7};
        const double step = kind == 5 ? 1.0 : 0.01;
        for (int m = 0; m < M; ++m) {
            if (kind == 1) { p[0] += 0.01; }                                          // a straight line along x: rank-1 Gram matrices, exact zero pivots
            else if (kind == 2) { p[0] += 0.003; p[1] += 0.004; p[2] += 0.012; }        // straight, oblique
            else if (kind == 3 && m == M / 2 && m > 0) { }                              // a node on top of the one before
            else if (kind == 4) { }                                                     // every node at the same point
            else { for (int d = 0; d < 3; ++d) p[d] += step * nr(); }
            for (int d = 0; d < 3; ++d) Y[(size_t)d * M + m] = p[d];
        }
        if (tdlo_calc_lle_regulariser(Y.data(), M, nullptr, Hh.data()) != 0) { std::printf("host routine failed at M = %d\n", M); return 2; }
        tdlo::lle_band_device<1>(Y.data(), M, Hd.data(), Ab.data(), 0);
        ++cases;
        for (size_t i = 0; i < Hh.size(); ++i) {
            const bool both_nan = Hh[i] != Hh[i] && Hd[i] != Hd[i];
            if (!both_nan && std::memcmp(&Hh[i], &Hd[i], sizeof(double)) != 0) {
                if (bad < 5) std::printf("case %d (kind %d, M = %d): Hb[%zu] host %.17g device routine %.17g\n", rep, kind, M, i, Hh[i], Hd[i]);
                ++bad; break;
            }
        }
    }
    std::printf("%d chains, %d with a differing bit\n", cases, bad);
    return bad ? 1 : 0;
}
