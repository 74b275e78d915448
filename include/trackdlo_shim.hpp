// trackdlo_shim.hpp -- drop-in `class trackdlo` on top of the C ABI (include/trackdlo_hip.h).
//
// Same public interface as the reference class (trackdlo/include/trackdlo.h:53-102): constructors,
// get_sigma2, get_tracking_result, get_guide_nodes, get_correspondence_pairs,
// initialize_geodesic_coord, initialize_nodes, set_sigma2, cpd_lle, tracking_step -- same argument
// order, defaults and by-value / by-reference conventions.  The ROS node's single call site
// (trackdlo/src/trackdlo_node.cpp:131-143, :366-369) compiles against it unchanged.
//
// The class is a template over the dense matrix type so that it can be compiled and tested without
// Eigen (this image has none); with Eigen available,
//     #include <Eigen/Dense>
//     #include "trackdlo_shim.hpp"
//     using trackdlo = tdlo::trackdlo_t<Eigen::MatrixXd>;
// gives the reference's type.  Requirements on Matrix: column-major doubles, Matrix(rows, cols),
// rows(), cols(), data(), operator()(i, j) -- Eigen::MatrixXd satisfies them and its storage is passed
// to the C ABI without copies or transposes.
//
// The object stays default-constructible and copy-assignable like the reference's (the node
// default-constructs a global and assigns to it later, trackdlo_node.cpp:54, :131): copies share the
// GPU context and own separate tracker state.
#pragma once
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "trackdlo_hip.h"

namespace tdlo {

struct ctx_deleter { void operator()(tdlo_ctx *c) const { tdlo_destroy(c); } };

template <class Matrix>
class trackdlo_t {
public:
    trackdlo_t() {}                                                        // trackdlo.cpp:8
    trackdlo_t(int num_of_nodes) { init(num_of_nodes, nullptr); }          // trackdlo.cpp:10-28 (not explicit: trackdlo.h:58 is not either)
    trackdlo_t(int num_of_nodes, double visibility_threshold, double beta, double lambda, double alpha, double k_vis,
               double mu, int max_iter, double tol, double beta_pre_proc, double lambda_pre_proc, double lle_weight) {
        const double p[11] = {visibility_threshold, beta, lambda, alpha, k_vis, mu, (double)max_iter, tol, beta_pre_proc,
                              lambda_pre_proc, lle_weight};
        init(num_of_nodes, p);                                             // trackdlo.cpp:30-59
    }
    trackdlo_t(const trackdlo_t &o) { *this = o; }
    trackdlo_t &operator=(const trackdlo_t &o) {
        if (this == &o) return *this;
        release();
        ctx_ = o.ctx_; M_ = o.M_; has_params_ = o.has_params_; precision_ = o.precision_;
        for (int i = 0; i < 11; ++i) params_[i] = o.params_[i];
        if (o.trk_) {
            make_tracker();
            // every member of the reference class (trackdlo.h:104-121): Y_, guide_nodes_, sigma2_, the parameters,
            // geodesic_coord_, correspondence_priors_ -- what the implicit copy assignment of the reference copies
            if (tdlo_tracker_copy_state(trk_, o.trk_) != TDLO_OK) throw std::runtime_error("trackdlo: tdlo_tracker_copy_state failed");
            coord_ = o.coord_;
        }
        return *this;
    }
    ~trackdlo_t() { release(); }

    double get_sigma2() { return trk_ ? tdlo_tracker_get_sigma2(trk_) : 0.0; }
    Matrix get_tracking_result() {
        Matrix Y(M_, 3);
        if (trk_) tdlo_tracker_get_tracking_result(trk_, Y.data());
        return Y;
    }
    Matrix get_guide_nodes() {
        std::vector<double> buf(3 * (size_t)M_);
        const int n = trk_ ? tdlo_tracker_get_guide_nodes(trk_, buf.data(), M_) : 0;
        Matrix G(n > 0 ? n : 0, 3);
        for (int i = 0; i < 3 * n; ++i) G.data()[i] = buf[i];
        return G;
    }
    std::vector<Matrix> get_correspondence_pairs() {
        std::vector<double> buf(4 * (size_t)(2 * M_ + 2));
        const int K = trk_ ? tdlo_tracker_get_correspondence_pairs(trk_, buf.data(), 2 * M_ + 2) : 0;
        std::vector<Matrix> out;
        for (int i = 0; i < K; ++i) { Matrix r(1, 4); for (int c = 0; c < 4; ++c) r(0, c) = buf[4 * i + c]; out.push_back(r); }
        return out;
    }
    void initialize_geodesic_coord(std::vector<double> geodesic_coord) {
        need();
        coord_.insert(coord_.end(), geodesic_coord.begin(), geodesic_coord.end());
        tdlo_tracker_initialize_geodesic_coord(trk_, geodesic_coord.data(), (int)geodesic_coord.size());
    }
    void initialize_nodes(Matrix Y_init) { need(); check(tdlo_tracker_initialize_nodes(trk_, Y_init.data())); }
    void set_sigma2(double sigma2) { need(); tdlo_tracker_set_sigma2(trk_, sigma2); }

    // trackdlo.h:80-94
    bool cpd_lle(Matrix X_orig, Matrix &Y, double &sigma2, double beta, double lambda, double lle_weight, double mu,
                 int max_iter = 30, double tol = 0.0001, bool include_lle = true,
                 std::vector<Matrix> correspondence_priors = {}, double alpha = 0, std::vector<int> visible_nodes = {},
                 double k_vis = 0, double visibility_threshold = 0.01) {
        need_ctx();
        tdlo_params p{};
        p.beta = beta; p.lambda = lambda; p.lle_weight = lle_weight; p.mu = mu; p.max_iter = max_iter; p.tol = tol;
        p.include_lle = include_lle ? 1 : 0; p.alpha = alpha; p.k_vis = k_vis; p.visibility_threshold = visibility_threshold;
        p.precision = precision_;
        std::vector<double> pri;
        for (auto &r : correspondence_priors) for (int c = 0; c < 4; ++c) pri.push_back(r(0, c));
        tdlo_stats st{};
        const int rc = tdlo_cpd_lle(ctx_.get(), X_orig.data(), (int)X_orig.rows(), Y.data(), (int)Y.rows(), &sigma2, &p,
                                    pri.empty() ? nullptr : pri.data(), (int)correspondence_priors.size(),
                                    visible_nodes.empty() ? nullptr : visible_nodes.data(), (int)visible_nodes.size(), nullptr, &st);
        check(rc);
        return st.converged != 0;
    }

    // trackdlo.h:96-101; proj_matrix / img_rows / img_cols are unused by the reference body (trackdlo.cpp:900-999)
    void tracking_step(Matrix X_orig, std::vector<int> visible_nodes, std::vector<int> visible_nodes_extended,
                       Matrix /*proj_matrix*/, int /*img_rows*/, int /*img_cols*/) {
        need();
        check(tdlo_tracker_tracking_step(trk_, X_orig.data(), (int)X_orig.rows(), visible_nodes.data(), (int)visible_nodes.size(),
                                         visible_nodes_extended.data(), (int)visible_nodes_extended.size(), nullptr, nullptr));
    }

    // not in the reference: choose fp32 E-step (default) or fp64 everywhere
    void set_precision(int precision) { precision_ = precision; if (trk_) tdlo_tracker_set_precision(trk_, precision); }

private:
    std::shared_ptr<tdlo_ctx> ctx_;
    tdlo_tracker *trk_ = nullptr;
    int M_ = 0;
    bool has_params_ = false;
    double params_[11] = {0};
    int precision_ = TDLO_PREC_F32;
    std::vector<double> coord_;

    void need_ctx() {
        if (ctx_) return;
        int err = 0;
        tdlo_ctx *c = tdlo_create(nullptr, &err);
        if (!c) throw std::runtime_error("trackdlo: no usable MI355X device (tdlo_create failed, code " + std::to_string(err) + ")");
        ctx_ = std::shared_ptr<tdlo_ctx>(c, ctx_deleter());
    }
    void make_tracker() {
        need_ctx();
        trk_ = has_params_
                   ? tdlo_tracker_create(ctx_.get(), 0, M_, params_[0], params_[1], params_[2], params_[3], params_[4], params_[5],
                                         (int)params_[6], params_[7], params_[8], params_[9], params_[10])
                   : tdlo_tracker_create_default(ctx_.get(), 0, M_);
        if (!trk_) throw std::runtime_error("trackdlo: tdlo_tracker_create failed");
        tdlo_tracker_set_precision(trk_, precision_);
    }
    void init(int M, const double *p) {
        M_ = M; has_params_ = p != nullptr;
        if (p) for (int i = 0; i < 11; ++i) params_[i] = p[i];
        make_tracker();
    }
    void need() { if (!trk_) throw std::runtime_error("trackdlo: object was default-constructed; assign a configured tracker first"); }
    void check(int rc) { if (rc != TDLO_OK) throw std::runtime_error(std::string("trackdlo: ") + tdlo_last_error(ctx_.get())); }
    void release() { if (trk_) { tdlo_tracker_destroy(trk_); trk_ = nullptr; } }
};

}  // namespace tdlo

#if defined(EIGEN_WORLD_VERSION) || defined(TDLO_WITH_EIGEN)
#include <Eigen/Dense>
using trackdlo = tdlo::trackdlo_t<Eigen::MatrixXd>;      // the reference's class name and matrix type
#endif
