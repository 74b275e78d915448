// The N-split of trackdlo::cpd_lle (BASELINE.json configs[3]) driven from plain C++ through the C ABI -- no Python, no torch in the
// process: what a ROS node with one process (or thread) per GPU would do.
//   1. plain call on the whole cloud (tdlo_cpd_lle)                                              -> reference result
//   2. tdlo_split_run with an RCCL communicator made by tdlo_rccl_unique_id / tdlo_rccl_comm_init (one rank: this box has one GPU;
//      the library binds librccl at run time and issues ncclAllReduce itself)                  -> must equal 1 bit for bit
//   3. tdlo_split_run with the one-shot exchange, two ranks = two contexts on two host threads, each with half of the cloud
//      (peer-written inboxes as plain device pointers).  On a box with two or more GPUs the ranks take devices 0 and 1 -- the peer
//      stores then cross xGMI, tdlo_xch_bind enables the peer mapping --, on a one-GPU box they share device 0
//      -> both ranks the same bits, and the plain call's result to the stated fp32-mode tolerance (the halves are pruned / sorted separately).
// build: __graft_entry__.build();  run: tests/test_split_native_gpu.py::test_cpp_driver (GPU box).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "../../include/trackdlo_hip.h"

static unsigned long long rng_state = 88172645463325252ull;
static double urand() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return (rng_state >> 11) * (1.0 / 9007199254740992.0); }
static double nrand() { return std::sqrt(-2 * std::log(urand() + 1e-300)) * std::cos(6.283185307179586 * urand()); }

#define CHECK(call) do { const int rc_ = (call); if (rc_ != TDLO_OK) { std::printf("FAIL %s -> %d\n", #call, rc_); return 1; } } while (0)

int main() {
    setenv("GPU_MAX_HW_QUEUES", "8", 0);       // two ranks on ONE GPU: their streams must not share a hardware queue (a rank waits for its peer)
    setenv("NCCL_SOCKET_IFNAME", "lo", 0);
    const int M = 40, N = 24000, iters = 12;
    std::vector<double> Y0(3 * M), X(3 * (size_t)N);
    for (int m = 0; m < M; ++m) { const double s = m / (double)(M - 1); Y0[m] = 0.78 * (s - 0.5); Y0[M + m] = 0.08 * std::sin(6.283185307179586 * s); Y0[2 * M + m] = 0.6 + 0.03 * std::cos(9.42477796076938 * s); }
    for (int n = 0; n < N; ++n) {
        const int i = (int)(urand() * (M - 1)); const double t = urand();
        for (int d = 0; d < 3; ++d) X[(size_t)d * N + n] = (double)(float)((1 - t) * Y0[d * M + i] + t * Y0[d * M + i + 1] + 0.002 * nrand() + (d == 1 ? 0.005 : 0.0));
    }
    tdlo_params p{}; p.beta = 0.35; p.lambda = 50000; p.lle_weight = 10; p.mu = 0.1; p.max_iter = iters; p.tol = 0; p.include_lle = 0; p.visibility_threshold = 0.008; p.precision = TDLO_PREC_F32;
    tdlo_config cfg{}; tdlo_default_config(&cfg); cfg.max_points = N; cfg.max_nodes = 64;
    int fails = 0;

    // 1. plain call
    std::vector<double> Ya = Y0; double s2a = 0; tdlo_stats sta{};
    int err = 0;
    tdlo_ctx *ctx = tdlo_create(&cfg, &err);
    if (!ctx) { std::printf("FAIL tdlo_create -> %d\n", err); return 1; }
    CHECK(tdlo_cpd_lle(ctx, X.data(), N, Ya.data(), M, &s2a, &p, nullptr, 0, nullptr, 0, nullptr, &sta));
    std::printf("plain        : iters %d kept %d sigma2 %.6e\n", sta.iters, sta.n_kept, s2a);

    // 2. RCCL form, one-rank communicator made by the library
    {
        char id[128];
        void *comm = nullptr;
        const int lr = tdlo_rccl_load(nullptr);
        if (lr != TDLO_OK) { std::printf("FAIL tdlo_rccl_load -> %d (%s)\n", lr, tdlo_last_error(ctx)); return 1; }
        CHECK(tdlo_rccl_unique_id(id));
        CHECK(tdlo_rccl_comm_init(ctx, 1, 0, id, &comm));
        CHECK(tdlo_set_cloud(ctx, 0, X.data(), N));
        std::vector<double> Yb = Y0; double s2b = 0; tdlo_stats stb{};
        CHECK(tdlo_split_run(ctx, comm, Yb.data(), M, &s2b, &p, nullptr, 0, nullptr, 0, nullptr, &stb));
        const bool same = std::memcmp(Ya.data(), Yb.data(), sizeof(double) * 3 * M) == 0 && s2a == s2b && stb.iters == sta.iters && stb.n_kept == sta.n_kept;
        std::printf("RCCL, 1 rank : iters %d kept %d sigma2 %.6e  %s\n", stb.iters, stb.n_kept, s2b, same ? "== plain, bit for bit" : "DIFFERS from the plain call");
        if (!same) ++fails;
    }
    tdlo_destroy(ctx);

    // 3. one-shot exchange, two ranks on two threads, half of the cloud each
    {
        const int R = 2;
        tdlo_ctx *c[R] = {nullptr, nullptr};
        void *inbox[R] = {nullptr, nullptr};
        tdlo_config cr = cfg; cr.max_points = N / R + 64;
        const bool two_gpus = tdlo_device_count() >= 2;
        std::printf("one-shot exchange on %s\n", two_gpus ? "devices 0 and 1 (peer stores over xGMI)" : "device 0 (both ranks: this box has one GPU)");
        for (int r = 0; r < R; ++r) { cr.device = two_gpus ? r : 0; c[r] = tdlo_create(&cr, &err); if (!c[r]) { std::printf("FAIL tdlo_create -> %d\n", err); return 1; } CHECK(tdlo_xch_create(c[r], R, 64, &inbox[r])); }
        std::vector<double> Yr[R] = {Y0, Y0}; double s2r[R] = {0, 0}; tdlo_stats str[R] = {}; int rcs[R] = {0, 0};
        auto work = [&](int r) {
            const int n0 = r * (N / R), n1 = (r + 1) * (N / R), n = n1 - n0;
            std::vector<double> Xs(3 * (size_t)n);
            for (int d = 0; d < 3; ++d) std::memcpy(&Xs[(size_t)d * n], &X[(size_t)d * N + n0], sizeof(double) * n);
            int rc = tdlo_xch_bind(c[r], r, R, inbox);
            if (rc == TDLO_OK) rc = tdlo_set_cloud(c[r], 0, Xs.data(), n);
            if (rc == TDLO_OK) rc = tdlo_split_run(c[r], nullptr, Yr[r].data(), M, &s2r[r], &p, nullptr, 0, nullptr, 0, nullptr, &str[r]);
            rcs[r] = rc;
        };
        std::thread t0(work, 0), t1(work, 1);
        t0.join(); t1.join();
        for (int r = 0; r < R; ++r) if (rcs[r] != TDLO_OK) { std::printf("FAIL rank %d -> %d (%s)\n", r, rcs[r], tdlo_last_error(c[r])); ++fails; }
        const bool same = std::memcmp(Yr[0].data(), Yr[1].data(), sizeof(double) * 3 * M) == 0 && s2r[0] == s2r[1];
        double dy = 0; for (int i = 0; i < 3 * M; ++i) dy = std::fmax(dy, std::fabs(Yr[0][i] - Ya[i]));
        std::printf("one-shot, 2  : iters %d kept %d + %d sigma2 %.6e  ranks %s, max|dY| vs plain %.2e m\n", str[0].iters, str[0].n_kept, str[1].n_kept, s2r[0],
                    same ? "identical" : "DIFFER", dy);
        if (!same || dy > 1e-5 || std::fabs(s2r[0] - s2a) > 1e-3 * s2a || str[0].iters != sta.iters || str[0].n_kept + str[1].n_kept != sta.n_kept) ++fails;
        for (int r = 0; r < R; ++r) tdlo_destroy(c[r]);
    }
    std::printf(fails ? "FAILED (%d)\n" : "OK\n", fails);
    return fails ? 1 : 0;
}
