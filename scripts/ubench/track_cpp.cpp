// tracking_step at production size (N = 5000, M = 45) from plain C++ through the C ABI: ms per frame without a Python caller,
// and a host-side breakdown (set TDLO_TRACK_PROFILE=1 in the library for its own stage timers).
// build: g++ -O2 -std=c++17 scripts/ubench/track_cpp.cpp -o scripts/ubench/track_cpp -Ltrackdlo_amd -ltrackdlo_hip -Wl,-rpath,$PWD/trackdlo_amd
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../include/trackdlo_hip.h"
static unsigned long long rs = 88172645463325252ull;
static double ur() { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return (rs >> 11) * (1.0 / 9007199254740992.0); }
static double nr() { return std::sqrt(-2 * std::log(ur() + 1e-300)) * std::cos(6.283185307179586 * ur()); }
int main() {
    const int M = 45, N = 5000;
    std::vector<double> Y0(3 * M), X(3 * (size_t)N), coord(M);
    for (int m = 0; m < M; ++m) { const double s = m / (double)(M - 1); Y0[m] = 0.88 * (s - 0.5); Y0[M + m] = 0.08 * std::sin(6.283185307179586 * s); Y0[2 * M + m] = 0.6 + 0.03 * std::cos(9.42477796076938 * s); }
    coord[0] = 0; for (int i = 1; i < M; ++i) { double d2 = 0; for (int d = 0; d < 3; ++d) { const double e = Y0[d * M + i] - Y0[d * M + i - 1]; d2 += e * e; } coord[i] = coord[i - 1] + std::sqrt(d2); }
    const bool occl = getenv("OCCL") && atoi(getenv("OCCL")) > 0;
    // points along the rope (not along nodes 17 .. 25 when that stretch is hidden), 2 mm of noise, 5 mm + off away from the nodes in y
    auto gen_cloud = [&](std::vector<double> &C, double off) {
        for (int n = 0; n < N; ++n) {
            int i; do { i = (int)(ur() * (M - 1)); } while (occl && i >= 17 && i <= 24);
            const double t = ur();
            for (int d = 0; d < 3; ++d) C[(size_t)d * N + n] = (double)(float)((1 - t) * Y0[d * M + i] + t * Y0[d * M + i + 1] + 0.002 * nr() + (d == 1 ? 0.005 + off : 0.0));
        }
    };
    gen_cloud(X, 0.0);
    tdlo_config cfg{}; tdlo_default_config(&cfg); cfg.max_points = 1 << 16; cfg.max_nodes = 64;
    int err = 0; tdlo_ctx *ctx = tdlo_create(&cfg, &err);
    if (!ctx) { std::printf("tdlo_create -> %d\n", err); return 1; }
    tdlo_tracker *t = tdlo_tracker_create(ctx, 0, M, 0.008, 0.35, 50000, 3, 50, 0.1, 50, 0.0002, 3.0, 1.0, 10.0);
    tdlo_tracker_initialize_nodes(t, Y0.data()); tdlo_tracker_initialize_geodesic_coord(t, coord.data(), M);
    std::vector<int> vis(M); for (int i = 0; i < M; ++i) vis[i] = i;
    if (occl) {
        // a stretch in the middle of the rope is hidden: no points along nodes 18 .. 24, those nodes not visible (the two registrations of a frame
        // then start from different node sets: trackdlo.cpp:913-921)
        vis.clear(); for (int i = 0; i < M; ++i) if (i < 18 || i > 24) vis.push_back(i);
    }
    const int nv = (int)vis.size();
    tdlo_stats st[2];
    for (int r = 0; r < 20; ++r) if (tdlo_tracker_tracking_step(t, X.data(), N, vis.data(), nv, vis.data(), nv, nullptr, st)) { std::printf("FAIL %s\n", tdlo_last_error(ctx)); return 1; }
    const int R = getenv("FRAMES") ? atoi(getenv("FRAMES")) : 2000;      // (FRAMES=500000: a soak run)
    long fails = 0;
    if (getenv("MOVE") && atoi(getenv("MOVE")) > 0) {
        // a rope that keeps moving: 16 clouds (own noise each) along a sway of MOVE tenths of a millimetre per frame in y -- the registrations take
        // more than one iteration, and every frame's cloud comes from a different buffer
        const double amp = 1e-4 * atoi(getenv("MOVE"));
        const int K = 16;
        std::vector<std::vector<double>> Xs(K, std::vector<double>(3 * (size_t)N));
        for (int k = 0; k < K; ++k) {
            const double off = amp * (k < K / 2 ? k : K - k);
            gen_cloud(Xs[k], off);
        }
        for (int r = 0; r < 64; ++r) tdlo_tracker_tracking_step(t, Xs[r % K].data(), N, vis.data(), nv, vis.data(), nv, nullptr, st);
        long it0 = 0, it1 = 0;
        const auto t0 = std::chrono::steady_clock::now();
        for (int r = 0; r < R; ++r) { if (tdlo_tracker_tracking_step(t, Xs[r % K].data(), N, vis.data(), nv, vis.data(), nv, nullptr, st)) { if (!fails++) std::printf("FAIL at frame %d: %s\n", r, tdlo_last_error(ctx)); } it0 += st[0].iters; it1 += st[1].iters; }
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / R;
        std::printf("C++ caller: tracking_step N=%d M=%d%s, rope moving %.1f mm per frame: %.4f ms/frame (iterations per frame: pre %.2f, main %.2f)\n", N, M, occl ? ", nodes 18-24 hidden" : "", amp * 1e3, ms, it0 / (double)R, it1 / (double)R);
        if (fails) std::printf("%ld frames FAILED\n", fails);
        tdlo_tracker_destroy(t); tdlo_destroy(ctx);
        return fails ? 1 : 0;
    }
    const auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < R; ++r) if (tdlo_tracker_tracking_step(t, X.data(), N, vis.data(), nv, vis.data(), nv, nullptr, st)) { if (!fails++) std::printf("FAIL at frame %d: %s\n", r, tdlo_last_error(ctx)); }
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / R;
    std::printf("C++ caller: tracking_step N=%d M=%d%s: %.4f ms/frame (pre %d it host %.3f ms, main %d it host %.3f ms)\n", N, M, occl ? ", nodes 18-24 hidden" : "", ms, st[0].iters, st[0].host_ms, st[1].iters, st[1].host_ms);
    if (fails) std::printf("%ld frames FAILED\n", fails);
    tdlo_tracker_destroy(t); tdlo_destroy(ctx);
    return fails ? 1 : 0;
}
