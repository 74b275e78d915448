/*
 * trackdlo_hip.h -- C ABI of the MI355X (gfx950) implementation of TrackDLO's per-frame EM
 * registration path.
 *
 * This is the drop-in boundary: plain pointers and sizes, no C++/Eigen/torch types.  Every entry
 * point names the reference interface it replaces (paths relative to the RMDLO/trackdlo tree).
 * All matrices are COLUMN-MAJOR doubles, the in-memory layout of Eigen::MatrixXd, so
 * `X.data()` / `Y.data()` of the reference's arguments can be passed straight through:
 *   X : N x 3, leading dimension N   (x[0..N) y[0..N) z[0..N))
 *   Y : M x 3, leading dimension M
 *
 * Error convention: functions return 0 on success or a negative TDLO_E_* code and never throw;
 * tdlo_last_error() returns a human-readable message for the last failure on that context.
 * (The reference path has no error channel at all: cpd_lle only returns `converged`,
 * trackdlo/src/trackdlo.cpp:433-440, which tracking_step discards, :927/:998.)
 *
 * Threading: a context is bound to one GPU and one HIP stream and is NOT thread-safe, matching the
 * reference's single-threaded use (ros::spin(), trackdlo/src/trackdlo_node.cpp:643).
 *
 * There is no CPU fallback: every entry point fails with TDLO_E_NO_DEVICE when no gfx950 device
 * is usable.
 */
#ifndef TRACKDLO_HIP_H
#define TRACKDLO_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define TDLO_ABI_VERSION 3

enum {
    TDLO_OK = 0,
    TDLO_E_NO_DEVICE = -1,   /* no usable HIP device / kernel image */
    TDLO_E_INVALID = -2,     /* bad argument (M < 4 or M > 1024, N <= 0, bad slot, ...) */
    TDLO_E_HIP = -3,         /* HIP runtime error, see tdlo_last_error */
    TDLO_E_EMPTY = -4,       /* the prune (trackdlo.cpp:177-195) removed every point */
    TDLO_E_NUMERIC = -5,     /* non-finite sigma2 / Y or singular system encountered */
    TDLO_E_TRAVERSE = -6,    /* traverse_euclidean would read out of bounds in the reference */
    TDLO_E_EXCHANGE = -7     /* N-split: a peer's contribution did not arrive in time / RCCL reported an error */
};

/* tdlo_params.precision */
enum {
    TDLO_PREC_F32 = 0,       /* fp32 E-step (distances, membership, sums per 64-point tile), fp64 M-step */
    TDLO_PREC_F64 = 1        /* fp64 everywhere */
};

typedef struct tdlo_ctx tdlo_ctx;

typedef struct {
    int device;              /* HIP device ordinal */
    int max_frames;          /* number of frame slots (>= 1); slots are independent clouds/trackers */
    int max_points;          /* initial per-slot capacity in points (grown on demand) */
    int max_nodes;           /* initial capacity in nodes (grown on demand) */
    int estep_blocks;        /* 0 = auto; otherwise workgroups per frame for the E-step */
} tdlo_config;

/* Arguments of trackdlo::cpd_lle after (X_orig, Y, sigma2): trackdlo/include/trackdlo.h:80-94. */
typedef struct {
    double beta;
    double lambda;
    double lle_weight;
    double mu;
    int max_iter;                 /* reference default 30 */
    double tol;                   /* reference default 1e-4 */
    int include_lle;              /* reference default true */
    double alpha;                 /* reference default 0 */
    double k_vis;                 /* reference default 0 */
    double visibility_threshold;  /* reference default 0.01 */
    int precision;                /* TDLO_PREC_*; not in the reference (which is fp64 on the CPU) */
} tdlo_params;

typedef struct {
    int iters;            /* EM iterations executed (the reference only logs this, trackdlo.cpp:426) */
    int converged;        /* return value of trackdlo::cpd_lle */
    int n_kept;           /* N after the prune, trackdlo.cpp:195 */
    int status;           /* 0, or TDLO_E_NUMERIC / TDLO_E_EMPTY */
    double sigma2;        /* final sigma2 (also written through the in/out pointer) */
    float loop_ms;        /* HIP-event time of the EM loop body on the context's stream (trackdlo.cpp:275-438); 0 unless tdlo_set_timing(ctx, 1) */
    float total_ms;       /* HIP-event time of the whole device-side call (prune + setup + loop + readback); 0 unless tdlo_set_timing(ctx, 1) */
    double host_ms;       /* host wall time of the call including uploads and the final synchronise */
    int mstep_retries;    /* iterations whose dense multi-workgroup elimination (a comparator / fall-back M-step, see tdlo_cpd_lle_resident) ran into
                           * the time limit of an inter-workgroup hand-off and were redone by the one-workgroup elimination (normally 0) */
    int sort_reused;      /* 1: this registration reused the slot's pruned, node-sorted cloud of the previous one (same cloud, same nodes, same
                           * precision: tdlo_set_sort_reuse) and skipped the prune of trackdlo.cpp:177-195 -- identical results; 0: it pruned;
                           * 2 (tdlo_tracker_tracking_step's main registration, every node visible): as 1, and its node-side set-up had been done
                           * by the pre-processing registration's prologue as well -- it started at its first E-step (TDLO_PAIR_SETUP=0: never) */
    int band_retry;       /* 1: the banded LLE M-step met a non-positive pivot (or a non-finite sigma2; in fp64 mode also a sigma2 computed from the
                           * data that is too large for the banded form to hold 1e-9 m, see tdlo_debug_band_retries) and the call was repeated on
                           * the dense pivoted eliminations, whose result this is (the reference's solver is a general one, trackdlo.cpp:415) */
} tdlo_stats;

/* ---- context ------------------------------------------------------------------------------- */
int tdlo_abi_version(void);
int tdlo_device_count(void);
void tdlo_default_config(tdlo_config *cfg);
/* Creates a context on cfg->device.  Replaces nothing in the reference (it has no device state);
 * it owns what `class trackdlo`'s members own (trackdlo/include/trackdlo.h:104-121) plus device buffers. */
tdlo_ctx *tdlo_create(const tdlo_config *cfg, int *err);
void tdlo_destroy(tdlo_ctx *ctx);
const char *tdlo_last_error(const tdlo_ctx *ctx);
/* raw hipStream_t of the context (for callers that time or order work on it) */
void *tdlo_stream(tdlo_ctx *ctx);
int tdlo_synchronize(tdlo_ctx *ctx);

/* ---- EM registration ----------------------------------------------------------------------- */
/* Uploads a point cloud into frame slot `slot` (H2D copy, stays resident in HBM).
 * Replaces the by-value `MatrixXd X_orig` argument of cpd_lle / tracking_step (trackdlo.h:80, :96). */
int tdlo_set_cloud(tdlo_ctx *ctx, int slot, const double *X, int N);

/* trackdlo::cpd_lle (trackdlo/src/trackdlo.cpp:161-441) on the cloud resident in `slot`.
 *   Y        in/out  M x 3 column-major           (MatrixXd& Y)
 *   sigma2   in/out                                (double& sigma2; 0 => initialised as at :271-273)
 *   priors   K x 4 row-major [idx, x, y, z]        (std::vector<MatrixXd> correspondence_priors)
 *   visible_nodes, n_vis                           (std::vector<int> visible_nodes)
 *   H_override optional M x M column-major matrix used in place of the LLE regulariser
 *              H = (I-L)^T (I-L) of :236-237 (whose weights are numerically ill-defined).
 * Returns 0 or an error; stats->converged carries the reference's bool result.
 * Chains of 4 .. 1024 nodes (more: TDLO_E_INVALID).  Which kernel solves the M-step (trackdlo.cpp:392-437), by default, for chains of up to 512 nodes
 * (beyond 512: the one-direction smoother k_mstep_chain_long without the LLE term, the one-workgroup dense elimination with it):
 *   include_lle == 0  the chain smoother (csrc/tdlo_mstep_chain.hip: the same linear system in O(M) through the state-space form of
 *                     the kernel G, one workgroup per frame);  lambda == 0 or TDLO_MSTEP=dense: the dense eliminations k_mstep_fast
 *                     (up to 60 nodes) / k_mstep_mcu (one workgroup per 16 rows) -- comparators of the tests;
 *   include_lle == 1  the banded L D L^T in the chain's state (csrc/tdlo_mstep_band.hip, O(M)) when no two consecutive nodes are closer
 *                     than about a millimetre and H is banded and symmetric (the library's own always is); otherwise, after a
 *                     non-positive pivot (stats->band_retry), or with TDLO_MSTEP_LLE=dense: the dense pivoted eliminations
 *                     k_mstep_fast (up to 64 nodes) / k_mstep (up to 128) / k_mstep_pivot_mcu (beyond; TDLO_MSTEP_LLE=1wg keeps it in
 *                     one workgroup).  stats->mstep_retries counts time-outs of the multi-workgroup forms only.
 * Sorted-cloud reuse: the prune (:177-195) and the sort by nearest node depend on the cloud and on the incoming nodes only.  A call
 * whose Y equals, bit for bit, the Y of the previous registration of the same slot in the same precision, with the slot's cloud
 * untouched in between (no tdlo_set_cloud / tdlo_depth_to_cloud / N-split on it), reuses that result instead of pruning again
 * (stats->sort_reused = 1; identical output).  In the reference this happens for the two registrations of a tracking_step whose nodes
 * are all visible (:913-927 and :998 start from the same Y_).  tdlo_set_sort_reuse(ctx, 0) or TDLO_REUSE_SORT=0 turn it off. */
int tdlo_cpd_lle_resident(tdlo_ctx *ctx, int slot, double *Y, int M, double *sigma2,
                          const tdlo_params *params, const double *priors, int K,
                          const int *visible_nodes, int n_vis, const double *H_override,
                          tdlo_stats *stats);

/* Same, taking the cloud from host memory like the reference signature does:
 * tdlo_set_cloud(slot 0) followed by tdlo_cpd_lle_resident. */
int tdlo_cpd_lle(tdlo_ctx *ctx, const double *X, int N, double *Y, int M, double *sigma2,
                 const tdlo_params *params, const double *priors, int K,
                 const int *visible_nodes, int n_vis, const double *H_override,
                 tdlo_stats *stats);

/* Batched form: F independent registrations (slots 0..F-1, clouds already resident), executed
 * concurrently on the GPU.  Arrays are indexed by frame; Y is F consecutive M x 3 blocks; priors/
 * visible_nodes are shared by all frames (pass K = 0 / n_vis = 0 for none).  No reference
 * counterpart: the reference processes one frame per call (BASELINE.json configs[2]).
 * Batches of 8 or more frames run as groups of frames on streams owned by the context (2 groups from 8 frames, 4 from 16, 3 from
 * 28 -- about what the GPU holds at once per E-step launch), staggered by one E-step, so that one group's M-step (one workgroup per
 * frame) overlaps another group's E-step; every frame's result is the same, bit for bit, as from tdlo_cpd_lle_resident
 * (environment TDLO_BATCH_STREAMS=n forces n groups, 1: one stream).  The sorted cloud is reused only when EVERY frame of the batch
 * can reuse its own.  The call returns after all streams have drained. */
int tdlo_cpd_lle_batch(tdlo_ctx *ctx, int F, double *Y, int M, double *sigma2,
                       const tdlo_params *params, const double *priors, int K,
                       const int *visible_nodes, int n_vis, const double *H_override,
                       tdlo_stats *stats /* F entries */);

/* ---- N-split building blocks (BASELINE.json configs[3]) --------------------------------------- */
/* When one frame's cloud is split over several GPUs, each rank holds a shard in slot 0 and the
 * host interleaves these calls with an all-reduce (SUM) of the packed buffer
 *   sums[0..M) = P1, sums[M..4M) = R = PX - P1 y (column-major M x 3, residual w.r.t. the current nodes,
 *   which are identical on every rank), sums[4M] = Q, sums[4M+1] = N_kept
 * and, when visibility weighting is active, an all-reduce (MIN) of dmin[M].
 * They expose the halves of one iteration of trackdlo.cpp:275-438. */
int tdlo_split_begin(tdlo_ctx *ctx, const double *Y, int M, double sigma2, const tdlo_params *params,
                     const double *priors, int K, const int *visible_nodes, int n_vis,
                     const double *H_override, double *init /* out: [n_kept_local, sum_d2_local] */);
int tdlo_split_set_global(tdlo_ctx *ctx, double n_kept_global, double sum_d2_global);
int tdlo_split_dmin(tdlo_ctx *ctx, double *dmin_sq /* out M, local */);
int tdlo_split_estep(tdlo_ctx *ctx, const double *dmin_sq_global /* in M or NULL */, double *sums /* out 4M+2 */);
int tdlo_split_mstep(tdlo_ctx *ctx, const double *sums_global /* in 4M+2 */, int *done /* out */);
int tdlo_split_end(tdlo_ctx *ctx, double *Y, double *sigma2, tdlo_stats *stats);
/* Leaves a split registration without results (e.g. after the global kept-point count came back 0): drains the stream,
 * unbinds the exchange buffers, so that the context can begin the next registration. */
int tdlo_split_abort(tdlo_ctx *ctx);

/* Device-resident exchange (the form the 8-GPU run uses): the two buffers the ranks all-reduce live in DEVICE memory
 * owned by the caller -- e.g. a tensor handed to RCCL -- and the calls below only enqueue work on the context's stream
 * (tdlo_stream), so one EM iteration is   dmin_enqueue, all-reduce MIN d_dmin, estep_enqueue, all-reduce SUM d_sums,
 * mstep_enqueue   with every collective issued on (or ordered after) that stream and NO host synchronisation; the
 * stopping rule is evaluated on the device and read with tdlo_split_poll every few iterations (kernels of a finished
 * registration are no-ops, all ranks see the same flag because they solve the same system).
 *   d_dmin: M doubles, per-node minimum SQUARED distance (1e300 where a shard holds no point); only touched when
 *           visibility weighting is active.   d_sums: 4M+2 doubles, layout as above.
 * Bind before tdlo_split_begin; bind (NULL, NULL) to return to the host-buffer calls. */
int tdlo_split_bind_exchange(tdlo_ctx *ctx, double *d_dmin /* device, M */, double *d_sums /* device, 4M+2 */);
int tdlo_split_dmin_enqueue(tdlo_ctx *ctx);    /* local dmin -> d_dmin */
int tdlo_split_estep_enqueue(tdlo_ctx *ctx);   /* d_dmin (global) -> E-step on the shard -> local sums -> d_sums */
int tdlo_split_mstep_enqueue(tdlo_ctx *ctx);   /* d_sums (global) -> M-step (:392-437), identical on every rank */
int tdlo_split_poll(tdlo_ctx *ctx, int *done, int *iters);   /* synchronises the stream */

/* ---- the split registration driven from C++ (no torch, no Python in the loop) -------------------------------------- */
/* One whole trackdlo::cpd_lle (trackdlo.cpp:161-441) with the cloud split over the ranks: every rank holds its shard in slot 0
 * (tdlo_set_cloud / tdlo_depth_to_cloud) and calls this with the same Y, sigma2 and parameters; every rank returns the same
 * Y, sigma2, iteration count (the replicated M-step solves the same system from the same bits).  stats->n_kept is the
 * shard's count.  Two forms:
 *   nccl_comm != NULL  an RCCL communicator (ncclComm_t) over the ranks: per iteration the all-reduce MIN of dmin[M]
 *       (visibility weighting only) and the all-reduce SUM of the 4M+2 sums, issued by this library on the context's stream
 *       between its kernels (librccl is bound at run time: an RCCL already mapped into the process -- PyTorch's -- is used,
 *       else $TDLO_RCCL_LIB / tdlo_rccl_load, else the system's); any chain length.
 *   nccl_comm == NULL  the ONE-SHOT EXCHANGE bound with tdlo_xch_bind: no collective at all.  Every rank writes its minima /
 *       sums straight into every peer's inbox (peer stores: xGMI on a multi-GPU node) and raises a flag; the last workgroup of
 *       the min-distance kernel and the one-workgroup M-step wait for the R flags in their own inbox and reduce the R
 *       contributions in rank order.  One EM iteration is the three kernels of the unsplit loop, no launch in between.
 *       Any chain length (the one-workgroup M-steps carry the exchange: the chain smoother, and with the LLE term the banded
 *       L D L^T; a registration whose LLE system takes the dense eliminations -- coincident nodes, an H_override that is not banded --
 *       only up to 64 nodes); up to 8 ranks.  A rank that is more than 2 s behind its peers (or gone) makes the waiting kernels give up:
 *       TDLO_E_EXCHANGE on the ranks that waited.  A rank whose own shard fails (TDLO_E_NUMERIC from the E-step's range check) raises its
 *       flag with an error mark: its peers leave the same iteration with TDLO_E_NUMERIC instead of waiting out the limit.  Arguments are validated before anything is exchanged; a shard that loses every point
 *       to the prune still takes part (it contributes zeros).  A group of ONE rank has nobody to exchange with: the per-iteration
 *       exchange is skipped (the plain call's kernels and bits); TDLO_XCH_SELF=1 (read per call) makes it write to and read from its
 *       own inbox like a rank of a larger group -- what the exchange itself costs on one GPU (bench.py, tests).
 * The stopping rule is evaluated on the device and read after iterations 1, 2, 4, 8, 12, ... (tol > 0).
 * With the LLE term, a banded solve that meets a non-positive pivot is repeated on the dense pivoted kernels by all ranks together
 * (stats->band_retry), as tdlo_cpd_lle_resident does; in the one-shot form only for chains of up to 64 nodes (longer: TDLO_E_NUMERIC). */
int tdlo_split_run(tdlo_ctx *ctx, void *nccl_comm, double *Y, int M, double *sigma2, const tdlo_params *params,
                   const double *priors, int K, const int *visible_nodes, int n_vis, const double *H_override, tdlo_stats *stats);
/* One-shot exchange set-up.  tdlo_xch_create allocates this rank's inbox (device memory, zeroed; tdlo_xch_bytes bytes) for
 * `nranks` ranks and chains up to `max_nodes` nodes.  Ranks in other processes export / open it as a HIP IPC handle (64
 * bytes, carried by whatever the host application uses to bootstrap: MPI, a file, a socket); ranks in one process pass the
 * pointers directly.  tdlo_xch_bind takes the device pointers of ALL ranks' inboxes, valid on this context's device, own
 * inbox at [rank]; nranks == 0 unbinds. */
size_t tdlo_xch_bytes(int nranks, int max_nodes);
int tdlo_xch_create(tdlo_ctx *ctx, int nranks, int max_nodes, void **inbox);
int tdlo_xch_ipc_export(tdlo_ctx *ctx, void *handle64);
int tdlo_xch_ipc_open(tdlo_ctx *ctx, const void *handle64, void **peer_inbox);
int tdlo_xch_bind(tdlo_ctx *ctx, int rank, int nranks, void *const *inboxes);
/* Whether this context's GPU can map memory of `peer_device` (hipDeviceCanAccessPeer; 1 for its own device): what a rank asks before it
 * opens a peer's inbox.  Failures of the exchange's set-up that are properties of the node, not errors of the caller -- no fine-grained
 * device memory for the inbox (tdlo_xch_create with more than one rank), an inbox that cannot be mapped (tdlo_xch_ipc_open,
 * tdlo_xch_bind) -- come back as TDLO_E_EXCHANGE: the ranks then agree (e.g. a MIN all-reduce of "set up") to use the RCCL form. */
int tdlo_xch_can_access(tdlo_ctx *ctx, int peer_device, int *can);
/* RCCL bootstrap for hosts that have no communicator of their own: rank 0 makes the 128-byte ncclUniqueId and hands it to
 * the other ranks; every rank then creates its communicator (owned by the context, destroyed with it).  tdlo_rccl_load
 * names the librccl to bind (NULL: search as described above); returns 0 when RCCL is usable. */
int tdlo_rccl_load(const char *path);
int tdlo_rccl_unique_id(void *id128);
int tdlo_rccl_comm_init(tdlo_ctx *ctx, int nranks, int rank, const void *id128, void **comm_out);
/* ncclCommCount / ncclCommUserRank of a communicator (any ncclComm_t of the RCCL this library is bound to): what a rank reports
 * about the group it really is in. */
int tdlo_rccl_comm_count(void *comm, int *nranks, int *rank);

/* ---- tracker object: class trackdlo (trackdlo/include/trackdlo.h:53-130) ---------------------- */
typedef struct tdlo_tracker tdlo_tracker;

/* trackdlo::trackdlo(int num_of_nodes, double visibility_threshold, double beta, double lambda,
 *   double alpha, double k_vis, double mu, int max_iter, double tol, double beta_pre_proc,
 *   double lambda_pre_proc, double lle_weight)  -- trackdlo.cpp:30-59.  Uses slot `slot` of ctx. */
tdlo_tracker *tdlo_tracker_create(tdlo_ctx *ctx, int slot, int num_of_nodes, double visibility_threshold,
                                  double beta, double lambda, double alpha, double k_vis, double mu,
                                  int max_iter, double tol, double beta_pre_proc,
                                  double lambda_pre_proc, double lle_weight);
/* trackdlo::trackdlo(int num_of_nodes) defaults -- trackdlo.cpp:10-28 */
tdlo_tracker *tdlo_tracker_create_default(tdlo_ctx *ctx, int slot, int num_of_nodes);
void tdlo_tracker_destroy(tdlo_tracker *t);
int tdlo_tracker_set_precision(tdlo_tracker *t, int precision);
/* trackdlo::initialize_nodes (trackdlo.cpp:83-86) */
int tdlo_tracker_initialize_nodes(tdlo_tracker *t, const double *Y_init /* M x 3 */);
/* trackdlo::initialize_geodesic_coord (trackdlo.cpp:77-81; appends, like the reference) */
int tdlo_tracker_initialize_geodesic_coord(tdlo_tracker *t, const double *coord, int n);
/* The implicit copy assignment of class trackdlo (trackdlo/include/trackdlo.h:104-121 has no user-defined one: every member is
 * copied -- Y_, guide_nodes_, sigma2_, the eleven parameters, geodesic_coord_, correspondence_priors_).  The ROS node relies on
 * it once (trackdlo_node.cpp:131 assigns a configured object to the file-scope default-constructed one, :54).  Both trackers
 * must have the same number of nodes; dst keeps its own context and slot. */
int tdlo_tracker_copy_state(tdlo_tracker *dst, const tdlo_tracker *src);
/* trackdlo::get_sigma2 / set_sigma2 (trackdlo.cpp:61-63, :88-90) */
double tdlo_tracker_get_sigma2(const tdlo_tracker *t);
void tdlo_tracker_set_sigma2(tdlo_tracker *t, double sigma2);
/* trackdlo::get_tracking_result (trackdlo.cpp:65-67): copies M x 3 */
int tdlo_tracker_get_tracking_result(const tdlo_tracker *t, double *Y_out);
/* trackdlo::get_guide_nodes (trackdlo.cpp:69-71): returns row count, copies rows x 3 */
int tdlo_tracker_get_guide_nodes(const tdlo_tracker *t, double *out, int max_rows);
/* trackdlo::get_correspondence_pairs (trackdlo.cpp:73-75): returns K, copies K x 4 row-major */
int tdlo_tracker_get_correspondence_pairs(const tdlo_tracker *t, double *out, int max_rows);
/* trackdlo::tracking_step (trackdlo.cpp:900-999).  proj_matrix/img_rows/img_cols of the reference
 * signature are unused by its body and therefore not part of the ABI.  H_pre: optional override of
 * the pre-processing registration's LLE matrix (n_vis_ext x n_vis_ext).  stats may be NULL;
 * otherwise stats[0] = pre-processing registration (:927), stats[1] = main registration (:998). */
/* X == NULL: use the cloud already resident in the tracker's slot (tdlo_set_cloud / tdlo_depth_to_cloud); N ignored.
 *
 * How a frame is run (nothing of this changes a bit of any result; each item has an environment switch, read when the context is made, that
 * turns it off -- the comparators of tests/test_direct_path_gpu.py; tdlo_debug_route_count counts how often each was taken):
 *  - X (up to 16 384 points) is copied into the context's pinned staging buffer and read from THERE by the first kernel of the frame, which
 *    also puts it in the slot's device buffer: no host-to-device copy on the stream (TDLO_DIRECT_CLOUD=0: hipMemcpyAsync as tdlo_set_cloud).
 *    X is the caller's again when the function returns, as before.
 *  - With every node visible (n_vis_ext == num_of_nodes) both registrations start from the same nodes Y_ on the same cloud
 *    (:913-927 and :998).  The main registration then (a) reuses the pre-processing registration's pruned, sorted cloud (tdlo_set_sort_reuse),
 *    (b) has its node-side set-up done by one more workgroup of the pre-processing registration's prologue (TDLO_PAIR_SETUP=0: a kernel of
 *    its own), (c) starts from the sums of the pre-processing registration's first E-step instead of repeating it -- same cloud, nodes,
 *    sigma2, mu, no visibility term, and the sums are integers: the same bits (TDLO_PAIR_SUMS=0) -- and (d) has its first M-step
 *    launched right behind the pre-processing registration's first iteration; it waits on the device for the priors that the host
 *    forms from that registration's result (:929-995) and leaves untouched if that registration needs more iterations or ends on an
 *    error (TDLO_SPEC_MSTEP=0: launched when the priors exist).  stats[1].sort_reused is 2 on such a frame.
 *  - With HIDDEN nodes (n_vis_ext < num_of_nodes) the registrations start from different node sets and share nothing -- but the main
 *    registration's prune, sort, set-up, per-node minimum distances and first E-step (:177-389 of the :998 call) depend on nothing the
 *    pre-processing registration produces.  They are launched on a second stream, into a second set of buffers of the context (the cloud
 *    is read from the same pinned staging buffer), as soon as the pre-processing registration's first iterations are on the first one,
 *    and run beside its iterations (6-7 with a stretch of the rope hidden); the first M-step waits behind them for the priors, as in (d),
 *    and the main registration stays on the second stream (TDLO_AHEAD=0: launched when the pre-processing registration has returned;
 *    tdlo_debug_route_count(ctx, 4)).  Such a frame does not leave the next frame's H behind (next item).  Clouds of up to 16 384 points,
 *    chains of up to 256 nodes.
 *  - The M-step that finishes the main registration also forms H = (I - L)^T (I - L) (:236-237) of the nodes it leaves behind, on the
 *    device, bit for bit what tdlo_calc_lle_regulariser gives (tdlo_debug_lle_band_device); the next frame's pre-processing registration
 *    uses it if it starts from exactly those nodes (every node visible, no H_pre) instead of waiting for the host's 6 x 6 factorisations
 *    (TDLO_LLE_NEXT=0).
 *  - Each registration enqueues as many iterations up front as the same registration took in the previous frame (1 .. 8) before the host
 *    looks at its state (trackdlo.cpp:424-428 is decided on the device; kernels of a finished registration are no-ops): consecutive frames
 *    take about the same number, and the GPU neither idles between iterations nor runs more than a no-op or two (TDLO_ITER_HINT=0: one).
 * A steady-state frame at production size (5 000 points, 45 nodes, both registrations converging in their first iteration) is then FOUR
 * kernels and no copy.  The calls of a context must come from one thread at a time (as for every entry point). */
int tdlo_tracker_tracking_step(tdlo_tracker *t, const double *X, int N,
                               const int *visible_nodes, int n_vis,
                               const int *visible_nodes_extended, int n_vis_ext,
                               const double *H_pre, tdlo_stats *stats);

/* ---- plain GMM-EM initial registration ------------------------------------ */
/* reg(pts, Y, sigma2, M, mu, max_iter), trackdlo/src/utils.cpp:21-82 (declared trackdlo/include/utils.h): M centroids
 * fitted to the cloud with the Euclidean membership only; Y (M x 3 column-major) and sigma2 are pure outputs (the
 * reference overwrites both, :24-29, :45).  Exactly max_iter iterations, no stopping rule, fp64.  M <= 890 (per-wave
 * accumulators in LDS; more is TDLO_E_INVALID).  pts == NULL: use the
 * cloud resident in `slot`.  A centroid that attracts no probability mass comes back NaN, as in the reference. */
int tdlo_reg(tdlo_ctx *ctx, int slot, const double *pts, int N, double *Y, double *sigma2, int M, double mu, int max_iter);

/* ---- depth image -> cloud -> voxel-grid down-sample ------------------------ */
/* The step right upstream of tracking_step in the ROS node (trackdlo/src/trackdlo_node.cpp:195-241): every pixel
 * with mask != 0 is back-projected ((u - cx) z / fx, (v - cy) z / fy, z = depth / 1000; float storage like
 * pcl::PointXYZRGB, :212-232) and the points are down-sampled by pcl::VoxelGrid with a cubic leaf of `leaf_size`
 * metres (:235-239; PCL 1.10 algorithm: one centroid per occupied cell, ascending cell index; when the cell count
 * overflows int32 the cloud is passed through unchanged, as PCL does).  depth: rows x cols uint16 millimetres, mask:
 * rows x cols uint8, both row-major (cv::Mat layout).  The result becomes the cloud resident in `slot` -- what
 * tdlo_set_cloud would have uploaded (:241) -- so tdlo_cpd_lle_resident, tdlo_visibility_prepass and
 * tdlo_tracker_tracking_step (with X == NULL) run on it without the cloud ever visiting the host.
 * X_out (optional): n x 3 column-major, leading dimension n, needs x_capacity >= n rows (else TDLO_E_INVALID after
 * the cloud has been made resident); *n_out = n, *n_raw_out = number of masked pixels.  n == 0 is not an error here
 * (the registration calls report TDLO_E_INVALID for an empty cloud). */
int tdlo_depth_to_cloud(tdlo_ctx *ctx, int slot, const unsigned short *depth, const unsigned char *mask, int rows, int cols,
                        double fx, double fy, double cx, double cy, double leaf_size,
                        double *X_out, int x_capacity, int *n_out, int *n_raw_out);
/* Up to 32 704 masked pixels (a 1280 x 720 frame of the reference's camera, launch/realsense_node.launch:7-12, holds about 30 000 on the rope) the
 * whole step is ONE kernel launch (csrc/tdlo_cloud.hip: compaction, back-projection and bounding box per 4096-pixel tile; then the last eight workgroups
 * to finish sort cell index | pixel rank as a team -- k_cloud_team: they are running, hence co-resident, every wait between them is bounded and a team
 * that cannot complete hands the frame to the multi-launch form -- and form the centroids; TDLO_CLOUD_TEAM=0: the ONE workgroup that finishes last does it
 * alone in LDS, k_cloud_fused) whose last workgroup reports the counts through pinned host memory; more masked pixels, a grid whose cell-index bits + rank bits exceed 32, or PCL's pass-through case take the multi-launch
 * form (bounding box, host round trip, radix sort passes, centroids) -- the same bits either way (TDLO_CLOUD_FUSED=0 forces it; tdlo_debug_route_count
 * 6 / 7 count the calls the one-launch kernel served / passed on).
 *
 * tdlo_image_buffers: pinned host buffers of the context for a rows x cols depth image and mask (valid until the next call with a larger image, or
 * tdlo_destroy).  A caller that lets its driver / segmentation write into them -- e.g. cv::Mat(rows, cols, CV_16UC1, depth) and
 * cv::Mat(rows, cols, CV_8UC1, mask) as the destinations of the conversions at trackdlo_node.cpp:167-190 -- and passes exactly these pointers
 * to tdlo_depth_to_cloud has the kernel read the images where they are, over PCIe (the mask once, coalesced; depth only where the mask is set):
 * no host-to-device copy of 3 bytes per pixel.  Any other pointers are copied to the device first, as before. */
int tdlo_image_buffers(tdlo_ctx *ctx, int rows, int cols, unsigned short **depth, unsigned char **mask);

/* ---- caller-side visibility pre-pass ----------------------------------- */
/* What the ROS node computes right before tracking_step (trackdlo/src/trackdlo_node.cpp:257-277, :345-360):
 * each node's shortest distance to the cloud resident in `slot` (M x N distances on the GPU, fp64),
 *   visible_nodes          = { m : dist_m <= visibility_threshold }, ascending          (:316, :326, :346)
 *   visible_nodes_extended = visible_nodes with occluded runs shorter than d_vis (in geodesic_coord) filled in (:350-360)
 * The OpenCV painter's-algorithm self-occlusion test (:279-343, cv::line rasterisation) is NOT part of it.
 * Output arrays need room for M entries; any output pointer may be NULL. */
int tdlo_visibility_prepass(tdlo_ctx *ctx, int slot, const double *Y, int M, double visibility_threshold, double d_vis,
                            const double *geodesic_coord, double *node_dist, int *visible_nodes, int *n_vis,
                            int *visible_nodes_extended, int *n_vis_ext);

/* One frame of the ROS node up to tracking_step in one call (trackdlo/src/trackdlo_node.cpp:195-277, :345-360): tdlo_depth_to_cloud followed by
 * tdlo_visibility_prepass of the tracker's current nodes Y against the cloud it has just made resident in `slot` -- the same outputs as the two
 * calls, bit for bit.  With up to 64 nodes and the one-launch team kernel the pre-pass rides in the SAME launch (every team member takes the minima
 * over the centroids it has just formed; the member that finishes last hands cloud size and minima to pinned host memory together): one launch and one
 * hand-over instead of two of each.  More nodes, TDLO_CLOUD_TEAM=0 / TDLO_CLOUD_FUSED=0 / TDLO_DIRECT_UPLOAD=0, more masked pixels than the one-launch
 * form takes, a team that gave its launch up: the two steps run one behind the other as before (tdlo_debug_route_count(ctx, 8) counts the frames
 * whose pre-pass rode along).  n == 0 (no masked pixel): *n_vis = *n_vis_ext = 0, no error. */
int tdlo_depth_to_cloud_visibility(tdlo_ctx *ctx, int slot, const unsigned short *depth, const unsigned char *mask, int rows, int cols,
                                   double fx, double fy, double cx, double cy, double leaf_size,
                                   const double *Y, int M, double visibility_threshold, double d_vis, const double *geodesic_coord,
                                   double *node_dist, int *visible_nodes, int *n_vis, int *visible_nodes_extended, int *n_vis_ext,
                                   int *n_out, int *n_raw_out);

/* The ROS node's callback from the images to the nodes in one call (trackdlo/src/trackdlo_node.cpp:195-277 and :345-369): tdlo_depth_to_cloud_visibility
 * with the tracker's own nodes, visibility threshold and geodesic coordinates, then tdlo_tracker_tracking_step on the cloud left in the tracker's slot
 * (X == NULL, no H_pre).
 * NOT in this call: the callback's self-occlusion test, trackdlo_node.cpp:279-343 (edges sorted by their distance from the camera and drawn with
 * cv::line of dlo_pixel_width into an image; a node whose projected pixel lies under an edge drawn before its own is left out of visible_nodes).
 * It needs the projection matrix and OpenCV's rasteriser: BY DEFAULT it is the caller's, and for a rope that crosses itself in the image the visible
 * sets this call forms (distance threshold + gap fill only: every node within visibility_threshold of the cloud) are larger than the reference
 * callback's.  Two ways to the callback's sets: (a) the caller runs its own :279-343 (or tdlo_self_occlusion_visible + tdlo_extend_visible_nodes below) on
 * the node distances of tdlo_depth_to_cloud_visibility / tdlo_visibility_prepass and hands visible_nodes / visible_nodes_extended to
 * tdlo_tracker_tracking_step itself; (b) tdlo_tracker_set_self_occlusion(t, proj, dlo_pixel_width) -- this call then applies the library's geometric
 * restatement of the test (parity against OpenCV's rasteriser unpinned, hence off by default).
 * tests/test_cloud_gpu.py::test_a_rope_that_crosses_itself shows the default sets, both ways, and that they agree with each other.  The visible sets of the frame are returned as well (arrays of M ints, may be NULL); stats: the two tdlo_stats of
 * tracking_step.  The result is read with tdlo_tracker_get_tracking_result.  A frame whose mask selects no pixel, or none of whose nodes lies
 * within the visibility threshold of the cloud, is TDLO_E_EMPTY (the reference's callback indexes visible_nodes[size() - 1] there, :351-361);
 * the tracker's state is untouched then. */
int tdlo_tracker_frame_from_depth(tdlo_tracker *t, const unsigned short *depth, const unsigned char *mask, int rows, int cols,
                                  double fx, double fy, double cx, double cy, double leaf_size, double d_vis,
                                  int *visible_nodes, int *n_vis, int *visible_nodes_extended, int *n_vis_ext,
                                  int *n_out, int *n_raw_out, tdlo_stats *stats);

/* The callback's self-occlusion ("painter") test, trackdlo/src/trackdlo_node.cpp:279-343: the edges between consecutive nodes are taken nearest the camera
 * first (:279-290, by the camera distance of their mid-points) and each is drawn as a line of dlo_pixel_width pixels (:337-341) after its two end nodes
 * were looked up in what had been drawn before (:304-334): a node whose projected pixel (proj: 3 x 4 row-major, the reference's proj_matrix; pixel
 * coordinates truncated as by static_cast<int>) lies under an edge nearer than its own is left out; the others are visible when they are within
 * visibility_threshold of the cloud (node_dist[M]: what tdlo_visibility_prepass / tdlo_depth_to_cloud_visibility return, :257-277).  visible_nodes
 * receives the ascending indices (room for M ints), *n_vis their number.  Host code, O(M^2) integer tests, no device work.
 * PARITY UNPINNED against OpenCV: "under an edge" is the geometric content of cv::line with a thickness -- within dlo_pixel_width / 2 of the segment
 * between the end pixels --, not OpenCV's own fixed-point rasteriser (absent from the build image); boundary pixels may differ.  tests: the oracle's
 * literal restatement of the callback's loop (oracle/ref_cpu.c, ref_self_occlusion) gives the same index sets on random self-crossing ropes. */
int tdlo_self_occlusion_visible(const double *Y, int M, const double proj[12], int dlo_pixel_width, const double *node_dist, double visibility_threshold,
                                int *visible_nodes, int *n_vis);
/* trackdlo_node.cpp:345-360: visible_nodes sorted, occluded runs shorter than d_vis (in geodesic_coord) filled in.  Room for the chain's node count. */
int tdlo_extend_visible_nodes(const int *visible_nodes, int n_vis, const double *geodesic_coord, double d_vis, int *visible_nodes_extended, int *n_vis_ext);
/* tdlo_tracker_frame_from_depth applies the self-occlusion test above between its distance pre-pass and the gap fill (the reference callback's order)
 * once the tracker has been given the projection matrix and the rope's width in pixels; proj == NULL switches it off again.  OFF by default (see the
 * parity note above). */
int tdlo_tracker_set_self_occlusion(tdlo_tracker *t, const double *proj, int dlo_pixel_width);

/* ---- host helpers on the path (exported so the parity tests can address them directly) -------- */
/* trackdlo::calc_LLE_weights (trackdlo.cpp:119-159), k as passed at :236 (6). L: M x M col-major out. */
int tdlo_calc_lle_weights(int k, const double *Y, int M, double *L);
/* H = (I - L)^T (I - L) of trackdlo.cpp:236-237 (L = calc_LLE_weights with k = 6) as the registrations form it: H (M x M col-major, optional) by
 * the dense route that the dense LLE M-steps use, Hb (13 M, optional; Hb[13 i + u] = H(i, i - 6 + u)) by the O(M) route of the banded LLE
 * M-step -- the same values bit for bit (tests/test_abi.py). */
int tdlo_calc_lle_regulariser(const double *Y, int M, double *H, double *Hb);
/* line_sphere_intersection (trackdlo/src/utils.cpp:185-241): returns number of points (0..2) */
int tdlo_line_sphere_intersection(const double A[3], const double B[3], const double C[3],
                                  double radius, double out[6]);
/* trackdlo::traverse_euclidean (trackdlo.cpp:584-898): returns number of pairs or TDLO_E_TRAVERSE */
int tdlo_traverse_euclidean(const double *geodesic_coord, int n_coord, const double *guide_nodes, int Mg,
                            const int *visible_nodes, int n_vis, int alignment, int alignment_node_idx,
                            double *out /* (n_coord + 2) x 4 row-major */);

/* ---- frame-level accuracy metric ------------------------------------------ */
/* evaluator::get_piecewise_error (trackdlo/src/evaluator.cpp:258-283, with calc_min_distance :233-256): mean over the
 * nodes of Y_track of the distance to the polyline through Y_true.  Chains are n x 3 column-major.  < 0 on bad input. */
double tdlo_piecewise_error(const double *Y_track, int n_track, const double *Y_true, int n_true);
/* evaluator::compute_error (evaluator.cpp:333-341): the symmetrised mean (E(track,true) + E(true,track)) / 2. */
double tdlo_compute_error(const double *Y_track, int n_track, const double *Y_true, int n_true);

/* ---- measurement ------------------------------------------------------------------------------ */
/* Launches the E-step kernel `reps` times back to back on the context's stream for the state left
 * by the last cpd_lle call on `slot` and returns the HIP-event average per launch (microseconds).
 * kind: 0 = membership/E-step kernel, 1 = per-node min-distance kernel, 2 = M-step kernel;
 * kind 10 = the E-step IN SITU: `reps` (<= 256) real iterations (E-step and M-step alternating as in the loop), each
 * E-step dispatch carrying its own start/stop events (hipExtLaunchKernelGGL) -- the per-dispatch duration a kernel
 * trace reports, measured live on the context's stream. */
int tdlo_profile_kernel(tdlo_ctx *ctx, int slot, int kind, int reps, float *avg_us);
/* One EM iteration IN SITU on the state left by the last cpd_lle call (all frames of the last call): `reps` (<= 256) real
 * iterations on the context's stream; the E-step dispatch and the M-step dispatch of every iteration carry their own HIP
 * start/stop events (hipExtLaunchKernelGGL) -- the per-dispatch durations a kernel trace reports.  *iter_us = stream time
 * per whole iteration (dispatch gaps and the small kernels of the visibility / large-cloud paths included).
 * mstep_kernel receives the name of the M-step kernel these frames dispatch; *mstep_us = -1 when that kernel's dispatch
 * carries no events (the pivoted paths). */
int tdlo_profile_iteration(tdlo_ctx *ctx, int reps, float *estep_us, float *mstep_us, float *iter_us, char *mstep_kernel, int name_cap);
/* Development aid: copies the first n (<= 64) shader-clock stamps that the M-step kernel of the last
 * launch wrote at its phase boundaries (reduce / assemble / eliminate / update / publish). */
int tdlo_debug_stamps(tdlo_ctx *ctx, int slot, unsigned long long *out, int n);
/* Test aid: y[i] = 2^x[i] as the fp64 E-step computes it (its own 17-instruction form, csrc/tdlo_devcommon.h: Num<double>::exp2, not the
 * library's); host arrays of n doubles. */
int tdlo_debug_exp2(tdlo_ctx *ctx, const double *x, double *y, int n);
/* Test aid: which M-step serves registrations WITHOUT the LLE term from now on, process-wide.  0 (default): the chain smoother
 * (csrc/tdlo_mstep_chain.hip: the system of trackdlo.cpp:405-413 solved in O(M) through the state-space form of the kernel G);
 * 1: the dense eliminations of the same system (k_mstep_fast / k_mstep_mcu), kept as comparators.  Returns the previous
 * setting.  The initial setting is 1 when the environment holds TDLO_MSTEP=dense. */
int tdlo_debug_mstep_dense(int on);
/* Test aid: which M-step serves registrations WITH the LLE term (include_lle: the pre-processing registration of tracking_step,
 * trackdlo.cpp:925-927) from now on, process-wide.  0 (default): the banded L D L^T of the system of :396-415 in the state (f, f') of
 * the chain (csrc/tdlo_mstep_band.hip, O(M)), wherever the chain and H allow it -- no two consecutive nodes closer than about a
 * millimetre, H banded like the reference's own (I - L)^T (I - L); 1: always the dense pivoted eliminations (k_mstep_fast<pivoted> /
 * k_mstep / k_mstep_pivot_mcu), kept as comparators and for everything the banded form does not take.  Returns the previous setting.
 * The initial setting is 1 when the environment holds TDLO_MSTEP_LLE=dense. */
int tdlo_debug_mstep_lle_dense(int on);
/* How many calls of this context were repeated on the dense pivoted kernels because the banded L D L^T (which takes no pivots) met a
 * non-positive pivot or produced a non-finite sigma2: an indefinite H_override, or a chain at the edge of the gap test.  The reference's
 * solver is a general one (trackdlo.cpp:415); the caller sees the dense kernels' result, and tdlo_stats.band_retry = 1 on that call
 * (tdlo_cpd_lle*, tdlo_split_run in both forms: the ranks solve the same system and repeat together).  fp64 mode adds one more reason: a
 * sigma2 above 6.25e7 h^3 / beta^4 m2 (h = the chain's mean link length; 0.8 m2 at beta = 5 and 2 cm links, 6e4 m2 at the reference's
 * beta = 0.35) -- met only when a registration starts from sigma2 = 0 on a chain many metres long -- where the state precision's entries,
 * rounded to fp64, leave the banded result a few 1e-9 m from the dense system's; a sigma2 GIVEN above the bound takes the dense kernels
 * without a first attempt (no retry counted).  -1 for a null context. */
long long tdlo_debug_band_retries(tdlo_ctx *ctx);
/* Test aid: the 13 diagonals of H = (I - L)^T (I - L) (see tdlo_calc_lle_regulariser) formed by the DEVICE routine that tdlo_tracker_tracking_step's
 * main registration ends with (csrc/tdlo_lle_dev.h; 1 .. 256 nodes): Hb[13 i + u] = H(i, i - 6 + u), the host routine's values bit for bit. */
int tdlo_debug_lle_band_device(tdlo_ctx *ctx, const double *Y, int M, double *Hb);
/* Test aid: how often tdlo_tracker_tracking_step took its short cuts on this context (none changes a bit of any result; each has an environment
 * switch that turns it off, read when the context is made).  which = 0: main registrations whose node-side set-up had ridden in the
 * pre-processing registration's prologue (every node visible; TDLO_PAIR_SETUP=0); 1: ... that started from the first E-step's sums handed over
 * by the pre-processing registration instead of repeating that E-step (TDLO_PAIR_SUMS=0); 2: ... whose first M-step had been launched ahead of
 * its priors and was released when they were staged (TDLO_SPEC_MSTEP=0); 3: pre-processing registrations whose LLE regulariser had been formed
 * on the device by the M-step that finished the previous frame (TDLO_LLE_NEXT=0); 4: main registrations of frames with hidden nodes whose first
 * iteration had run on the second stream beside the pre-processing registration (TDLO_AHEAD=0); 5: calls repeated on the three-kernel route because
 * the fused prologue's grid barrier was abandoned; 6 / 7: tdlo_depth_to_cloud calls served by the one-launch kernel / passed on by it to the
 * multi-launch form; 8: frames whose visibility pre-pass rode in the depth -> cloud launch (tdlo_depth_to_cloud_visibility); 9: registrations whose
 * E-step was k_estep2 -- two points per lane, csrc/tdlo_estep2.hip: fp32 mode, chains of 8 .. 64 nodes, a cloud or a batch of at least 2048 x 64
 * points, i.e. one that fills the GPU (TDLO_ESTEP2=0: k_estep everywhere, the comparator; =1: wherever eligible, whatever the size).  The two
 * E-step kernels differ in the grain of their fp32 tile sums (one wave x 64 points / x 128 points): each is repeatable bit for bit and held to
 * the reference at the mode's tolerance, but a frame registered alone (k_estep) and the same frame inside a GPU-filling batch (k_estep2) agree
 * to about 1e-8 m, not to the bit.  10: fp64-mode calls repeated without the sigma-following resolution of the E-step's sums because a share was
 * refused under its finer range limits (the repeat runs under the coarse limits of every other mode; only its verdict is reported).
 * 11: calls served by the spin-ahead loop (TDLO_SPIN_AHEAD=1, a round-6 experiment, off by default); 12 / 13: batches whose whole loop ran as one launch
 * (TDLO_BATCH_PERSIST=1, a round-6 experiment, off by default) / such calls repeated on the launch-per-step loop because a wait inside the launch gave up.
 * -1 for a null context or an unknown counter. */
long long tdlo_debug_route_count(tdlo_ctx *ctx, int which);
/* Phase stamps (s_memtime) of the last depth -> cloud launch's finishing workgroup; only a -DTDLO_CLOUD_STAMPS build writes them. */
int tdlo_debug_cloud_stamps(tdlo_ctx *ctx, unsigned long long *out, int n);
/* Test aid: provokes a HIP runtime error inside the library (an invalid copy) and reports it like any other: returns TDLO_E_HIP with the
 * text in tdlo_last_error.  The calls that follow must be unaffected -- HIP keeps a per-thread "last error" that the launch checks of a later
 * call would otherwise read (tests/test_parity_gpu.py::test_a_hip_error_does_not_leak_into_the_next_call). */
int tdlo_debug_fail_hip(tdlo_ctx *ctx);
/* Whether the registrations of this context record the four stream events behind tdlo_stats.loop_ms / total_ms.  Off by default: the
 * reference has no such figures, and the markers cost about 15 us per call (2 % of a 50-iteration call at N = 50 000).  Returns the
 * previous setting (or TDLO_E_INVALID). */
int tdlo_set_timing(tdlo_ctx *ctx, int on);
/* Whether registrations of this context may reuse a slot's pruned, node-sorted cloud (see tdlo_cpd_lle_resident).  On by default
 * (off when the environment holds TDLO_REUSE_SORT=0); the reference prunes in every call (trackdlo.cpp:177-195), and a caller that
 * wants every call to pay for that -- a benchmark that registers the same frame again and again -- turns it off.  Returns the
 * previous setting (or TDLO_E_INVALID). */
int tdlo_set_sort_reuse(tdlo_ctx *ctx, int on);
/* The one-shot exchange of tdlo_split_run with ONE rank bound (tdlo_xch_bind, nranks == 1): by default a lone rank has nobody to exchange with and the
 * kernels skip the per-iteration exchange; on != 0: it stores to, flags and reads back its own inbox like any rank of a larger group (what the exchange
 * itself costs on one GPU -- bench.py's self_exchange_iters_per_s; tests).  The same results either way, bit for bit.  Initial value: TDLO_XCH_SELF
 * when the context was made.  Returns the previous setting. */
int tdlo_set_xch_self(tdlo_ctx *ctx, int on);
/* The PCI bus id of the context's GPU ("0000:c1:00.0"; out needs >= 16 bytes): which /sys/bus/pci/devices/<id> node -- clocks, power, busy percentage --
 * belongs to it (bench.py's clock sampler: the container's /sys/class/drm lists every card of the host, not only the visible one). */
int tdlo_pci_bus_id(tdlo_ctx *ctx, char *out, int len);
/* Development aid: copies the pruned, centred, node-sorted cloud of the last call (N x 3 column-major, widened to
 * double) and the centring offset; returns N. */
int tdlo_debug_read_cloud(tdlo_ctx *ctx, int slot, double *out, int max_points, double *ctr);

#ifdef __cplusplus
}
#endif
#endif
