// Exercises the drop-in C++ class (include/trackdlo_shim.hpp) exactly the way the reference's ROS node
// uses `class trackdlo` (trackdlo/src/trackdlo_node.cpp:131-143, :366-369) and checks it against the
// CPU oracle (oracle/ref_cpu.c).  Eigen is not available in this image, so a minimal column-major
// matrix with Eigen::MatrixXd's accessors stands in for it.
#include <cmath>
#include <cstdio>
#include <vector>

#include "../../include/trackdlo_shim.hpp"
extern "C" {
#include "../../oracle/ref_cpu.h"
}

struct MatrixXd {                       // the subset of Eigen::MatrixXd the shim relies on
    int r = 0, c = 0;
    std::vector<double> v;
    MatrixXd() {}
    MatrixXd(int rows, int cols) : r(rows), c(cols), v((size_t)rows * cols, 0.0) {}
    int rows() const { return r; }
    int cols() const { return c; }
    double *data() { return v.data(); }
    const double *data() const { return v.data(); }
    double &operator()(int i, int j) { return v[(size_t)j * r + i]; }
    double operator()(int i, int j) const { return v[(size_t)j * r + i]; }
};
using trackdlo = tdlo::trackdlo_t<MatrixXd>;

static unsigned long long rng_state = 88172645463325252ull;
static double urand() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return (rng_state >> 11) * (1.0 / 9007199254740992.0); }
static double nrand() { return std::sqrt(-2 * std::log(urand() + 1e-300)) * std::cos(6.283185307179586 * urand()); }

int main() {
    const int M = 30, N = 3000;
    MatrixXd Y0(M, 3), X(N, 3);
    for (int m = 0; m < M; ++m) { const double s = m / (double)(M - 1); Y0(m, 0) = 0.58 * (s - 0.5); Y0(m, 1) = 0.08 * std::sin(6.283185307179586 * s); Y0(m, 2) = 0.6 + 0.03 * std::cos(9.42477796076938 * s); }
    for (int n = 0; n < N; ++n) {
        const int i = (int)(urand() * (M - 1)); const double t = urand();
        for (int d = 0; d < 3; ++d) X(n, d) = (double)(float)((1 - t) * Y0(i, d) + t * Y0(i + 1, d) + 0.002 * nrand() + (d == 1 ? 0.005 : 0.0));
    }
    std::vector<double> coord(M, 0.0);
    for (int m = 1; m < M; ++m) { double s = 0; for (int d = 0; d < 3; ++d) s += (Y0(m, d) - Y0(m - 1, d)) * (Y0(m, d) - Y0(m - 1, d)); coord[m] = coord[m - 1] + std::sqrt(s); }
    int fails = 0;

    // ---- cpd_lle through the reference signature vs the oracle
    {
        trackdlo t(M);
        MatrixXd Y = Y0; double sigma2 = 0;
        const bool conv = t.cpd_lle(X, Y, sigma2, 0.35, 50000, 10.0, 0.1, 20, 0.0, false);
        MatrixXd Yr = Y0; double s2r = 0;
        ref_params p{}; p.beta = 0.35; p.lambda = 50000; p.lle_weight = 10; p.mu = 0.1; p.max_iter = 20; p.tol = 0; p.include_lle = 0; p.visibility_threshold = 0.01;
        ref_stats st{};
        ref_cpd_lle(X.data(), N, Yr.data(), M, &s2r, &p, nullptr, 0, nullptr, 0, nullptr, &st, nullptr);
        double dy = 0; for (int i = 0; i < 3 * M; ++i) dy = std::fmax(dy, std::fabs(Y.data()[i] - Yr.data()[i]));
        std::printf("cpd_lle: converged=%d/%d max|dY|=%.3e dsigma2=%.3e\n", (int)conv, st.converged, dy, std::fabs(sigma2 - s2r) / s2r);
        if (dy > 1e-5 || std::fabs(sigma2 - s2r) > 1e-3 * s2r || (int)conv != st.converged) ++fails;
    }
    // ---- the node's usage pattern: default-construct, assign, initialise, step (trackdlo_node.cpp:54, :131-143, :366-369)
    {
        trackdlo tracker;                                  // file-scope global in the node
        tracker = trackdlo(M, 0.008, 0.35, 50000, 3.0, 50.0, 0.1, 30, 0.0002, 3.0, 1.0, 10.0);
        tracker.initialize_nodes(Y0);
        tracker.initialize_geodesic_coord(coord);
        std::vector<int> vis, vis_ext;
        for (int m = 0; m < M; ++m) { vis.push_back(m); vis_ext.push_back(m); }
        ref_tracker *rt = ref_tracker_create(M, 0.008, 0.35, 50000, 3.0, 50.0, 0.1, 30, 0.0002, 3.0, 1.0, 10.0);
        ref_tracker_initialize_nodes(rt, Y0.data());
        ref_tracker_initialize_geodesic_coord(rt, coord.data(), M);
        MatrixXd proj(3, 4);
        for (int step = 0; step < 2; ++step) {
            tracker.tracking_step(X, vis, vis_ext, proj, 720, 1280);
            ref_stats a{}, b{};
            ref_tracking_step(rt, X.data(), N, vis.data(), M, vis_ext.data(), M, nullptr, &a, &b);
            MatrixXd Y = tracker.get_tracking_result();
            MatrixXd G = tracker.get_guide_nodes();
            std::vector<MatrixXd> pri = tracker.get_correspondence_pairs();
            double dy = 0; for (int i = 0; i < 3 * M; ++i) dy = std::fmax(dy, std::fabs(Y.data()[i] - rt->Y[i]));
            std::printf("tracking_step %d: max|dY|=%.3e sigma2=%.6e/%.6e priors=%d/%d guide_rows=%d\n", step, dy, tracker.get_sigma2(), rt->sigma2, (int)pri.size(), rt->K, G.rows());
            // the stated tolerance of the default precision (fp32 E-step): 1e-5 m, pre-processing registration with its own
            // LLE weights included
            if (dy > 1e-5 || (int)pri.size() != rt->K || G.rows() != M) ++fails;
        }
        // ---- copy semantics (trackdlo.h:104-121 has no user-defined copy: EVERY member is copied -- Y_, guide_nodes_, sigma2_,
        //      geodesic_coord_, correspondence_priors_, the parameters).  A copy taken after two steps must answer every getter
        //      like the original and continue the sequence with the same bits.
        {
            trackdlo copy;
            copy = tracker;
            MatrixXd Ya = tracker.get_tracking_result(), Yb = copy.get_tracking_result();
            MatrixXd Ga = tracker.get_guide_nodes(), Gb = copy.get_guide_nodes();
            std::vector<MatrixXd> Pa = tracker.get_correspondence_pairs(), Pb = copy.get_correspondence_pairs();
            bool same = Ya.v == Yb.v && Ga.r == Gb.r && Ga.v == Gb.v && Pa.size() == Pb.size() && !Pa.empty() && tracker.get_sigma2() == copy.get_sigma2();
            for (size_t i = 0; same && i < Pa.size(); ++i) same = Pa[i].v == Pb[i].v;
            tracker.tracking_step(X, vis, vis_ext, proj, 720, 1280);
            copy.tracking_step(X, vis, vis_ext, proj, 720, 1280);
            same = same && tracker.get_tracking_result().v == copy.get_tracking_result().v && tracker.get_sigma2() == copy.get_sigma2();
            std::printf("copy assignment: state and continuation %s\n", same ? "identical" : "DIFFER");
            if (!same) ++fails;
        }
        ref_tracker_destroy(rt);
    }
    std::printf(fails ? "FAILED (%d)\n" : "OK\n", fails);
    return fails;
}
