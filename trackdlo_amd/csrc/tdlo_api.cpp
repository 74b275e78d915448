// tdlo_api.cpp -- C ABI (include/trackdlo_hip.h): context, frame slots, the cpd_lle driver, the
// batched and N-split forms, and the tracker object mirroring class trackdlo
// (trackdlo/include/trackdlo.h:53-130, trackdlo/src/trackdlo.cpp:8-90, :900-999).
#include "../../include/trackdlo_hip.h"
#include "tdlo_internal.h"
#include "tdlo_host.h"
#include "tdlo_rccl.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <atomic>
#include <vector>

using namespace tdlo;

namespace {

#define TDLO_RET(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return e_; } while (0)

constexpr int kMaxEstepBlocks = 4096;      // (upper bound of TDLO_ESTEP_BLOCKS / tdlo_config.estep_blocks; the defaults are 512 and 1024)
constexpr int kBatchStreams = 4;       // streams a batch of frames is spread over (run_frames); more than 4 lose (measured: 6 or 8 fall below one stream)
constexpr int kIterHintMax = 8;        // tdlo_ctx::iter_hint: at most this many iterations go out before the host first looks
constexpr int kEstepWideMin = 129;          // (measured: scripts/gpu_estep_wide_ab.py)
constexpr int kEstep2MinWaves = 2048;   // 64-point batches of a cloud (or of a batch's frames together) from which the E-step is k_estep2 (two points per lane): two waves per SIMD
constexpr int kChunkIters = 4;          // EM iterations per early-exit polling chunk; the first chunks are shorter (1, 1, 2):
                                        // a tracker in steady state converges in one or two iterations

struct Slot {
    unsigned spin_ecount = 0, spin_mtag = 0;      // the spin-ahead loop's running counts (FrameDev::spin_wait / spin_signal): E-step workgroups reported, M-steps tagged
    // cloud-sized
    int cap_points = 0;
    int N0 = 0;
    double *Xraw = nullptr;
    void *Xs = nullptr;
    unsigned short *bucket = nullptr;
    int *hist = nullptr;
    size_t hist_ints = 0;
    double *blksum = nullptr;
    // what the sorted cloud Xs was last made for (prune + counting sort by nearest node depend on the cloud and on the nodes only): a
    // registration of the SAME nodes on the same cloud in the same precision reuses it (FrameDev::reuse_sorted)
    std::vector<double> sorted_Y;
    int sorted_prec = -1;
    bool sorted_valid = false;
    // node-sized (one allocation, carved)
    int cap_nodes = 0;
    double *nodeblk = nullptr;
    size_t nodeblk_doubles = 0;
    unsigned *sync = nullptr;        // 256 words, zeroed once: cross-workgroup hand-off state of the multi-CU M-step
    unsigned fuse_epoch = 0;         // launches of the fused prologue on this slot (its grid barrier's flag carries the number)
    // a second node block: tracking_step with every node visible has the fused prologue of its first registration set up the second one as
    // well (PairNext) -- the second registration then runs out of this block
    int cap_nodes2 = 0;
    double *nodeblk2 = nullptr;
    // the LLE regulariser (13 diagonals) of the nodes the last tracking_step left behind, formed on the device by the M-step that finished its
    // main registration (FrameDev::lle_next): serves the next pre-processing registration if it starts from exactly those nodes
    double *hb_next = nullptr;
    int hb_next_cap = 0;
    std::vector<double> hb_next_Y;
    bool hb_next_valid = false;
};

struct NodeCarve {
    // offsets in doubles into Slot::nodeblk for a given M; the first `upload` doubles are the
    // host-supplied block [descriptor | Yin | aJ | aYd | Hb | H]: without the LLE term it ends in front of Hb, with the banded LLE M-step in
    // front of H (Hb = the 13 diagonals of H, 13 M doubles instead of M^2), with the dense LLE M-steps it is the whole block; [Yout | IterState]
    // comes back.
    size_t fdev, Yin, aJ, aYd, Hb, H, upload;      // fdev: the frame's descriptor travels at the head of the upload block (one frame per call)
    size_t Yout, st, readback;
    size_t ctr, Y, Y0, nodes, coord, G, chain, HG, HY0, dmin, sums, dbg, Ascr, acc, band, total;
    explicit NodeCarve(int M) {
        const size_t m = (size_t)M, mm = m * m;
        size_t o = 0;
        auto take = [&](size_t n) { size_t r = o; o += (n + 1) & ~(size_t)1; return r; };   // keep 16-byte alignment
        fdev = take((sizeof(FrameDev) + 7) / 8);
        Yin = take(3 * m); aJ = take(m); aYd = take(3 * m); Hb = take(13 * m); H = take(mm); upload = o;
        Yout = take(3 * m); st = take((sizeof(IterState) + 7) / 8); readback = o - Yout;
        ctr = take(4); Y = take(3 * m); Y0 = take(3 * m); nodes = take(4 * m); coord = take(m);
        G = take(mm); chain = take(8 * (m + 1)); HG = take(mm); HY0 = take(3 * m); dmin = take(m); sums = take(4 * m + 2); dbg = take(64);
        Ascr = take(std::max((size_t)(M | 1) * (m + 3), mstep_big_scratch_doubles(M)));
        acc = take((size_t)2 * kAccRows * (4 * m + 2));       // the E-step's fixed-point accumulators, two iteration parities
        band = take(band_record_doubles(M));                  // column records of the banded LLE M-step
        total = o;
    }
};

}  // namespace

// Host threads that enqueue the stream groups of a batch side by side.  A 32-frame call is 400 kernel launches of ~4 us of host time
// each -- as long as the GPU needs to run them -- so one enqueueing thread makes the batch host-bound whenever the CPU is a little
// slower than usual (measured: 0.86 ... 1.03 M it/s run to run); every group's launches go to its own stream anyway.
struct EnqueuePool {
    std::vector<std::thread> th;
    std::mutex m;
    std::condition_variable cv, cv_done;
    std::function<int(int)> job;
    int gen = 0, pending = 0, first_err = 0, device = 0;
    bool stop = false;
    void start(int n, int dev) {
        device = dev;
        for (int i = 0; i < n; ++i) th.emplace_back([this, i] { loop(i + 1); });
    }
    void loop(int idx) {
        (void)hipSetDevice(device);
        int seen = 0;
        for (;;) {
            std::function<int(int)> f;
            {
                std::unique_lock<std::mutex> lk(m);
                cv.wait(lk, [&] { return stop || gen != seen; });
                if (stop) return;
                seen = gen; f = job;
            }
            const int rc = f(idx);
            {
                std::lock_guard<std::mutex> lk(m);
                if (rc && !first_err) first_err = rc;
                if (--pending == 0) cv_done.notify_one();
            }
        }
    }
    // f(1) .. f(th.size()) on the workers, f(0) on the caller; returns the first non-zero result
    int run(const std::function<int(int)> &f) {
        {
            std::lock_guard<std::mutex> lk(m);
            job = f; pending = (int)th.size(); first_err = 0; ++gen;
        }
        cv.notify_all();
        const int rc0 = f(0);
        std::unique_lock<std::mutex> lk(m);
        cv_done.wait(lk, [&] { return pending == 0; });
        return rc0 ? rc0 : first_err;
    }
    ~EnqueuePool() {
        { std::lock_guard<std::mutex> lk(m); stop = true; }
        cv.notify_all();
        for (auto &t : th) t.join();
    }
};

struct tdlo_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    double *xfer = nullptr;          // batches: device mirror of [F upload blocks | F read-back blocks], moved with one copy each way
    size_t xfer_doubles = 0;
    hipStream_t stream2[kBatchStreams - 1] = {};   // further groups of a batch: their E-steps overlap another group's one-workgroup-per-frame M-step
    hipEvent_t evx[kBatchStreams] = {}, evj[kBatchStreams] = {};   // fork / join of the batch groups
    // chained E-steps of a batch's stream groups (run_frames): group g's E-step of iteration k goes out behind group g-1's (an event per group and
    // iteration, a ring of kChainRing): the groups' E-steps then run ONE AFTER THE OTHER, each with the GPU to itself, beside the other groups' M-steps.
    // Left to themselves the groups lock into whatever phase the first iteration gave them -- E-steps on top of each other, then a stretch with
    // nothing but M-steps (scripts/gpu_c3_timeline.sh).  Round 6 experiment, TDLO_BATCH_CHAIN=1 switches it on: the events and the host threads waiting for
    // each other cost more than the phase is worth.
    static constexpr int kChainRing = 8;
    hipEvent_t evc[kBatchStreams][kChainRing] = {};
    // 0 (default): never.  1: every iteration -- measured SLOWER, 0.90 - 0.96 M against 1.21 - 1.24 M it/s at C3: the events and the enqueueing threads waiting
    // for each other cost more than the phase is worth.  2: only at four iterations of the call (two groups of equal work that overlap slow each other down
    // equally, so a phase once set should persist) -- 1.18 - 1.21 M against 1.23 M: no gain either (profiles/r06_measured.log).  Kept as comparators.
    int batch_chain = getenv("TDLO_BATCH_CHAIN") ? atoi(getenv("TDLO_BATCH_CHAIN")) : 0;
    hipEvent_t ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // 0..3 timing, 4..5 early-exit polling
    tdlo_config cfg{};
    std::vector<Slot> slots;
    std::vector<FrameDev> fh;        // host copies of the frame descriptors of the last call
    FrameDev *fd = nullptr;          // device array [max_frames]
    double *pin = nullptr;           // pinned staging
    double *reg_ws = nullptr;        // `reg` workspace
    size_t reg_ws_cap = 0;
    void *cloud_ws = nullptr;        // depth -> cloud workspace (images, sort buffers)
    size_t cloud_ws_cap = 0;
    // depth -> cloud in one launch (k_cloud_fused): its own workspace behind the same images, the pinned words the kernel reports through, and the
    // pinned image buffers a caller may fill directly (tdlo_image_buffers: the kernel then reads the images over PCIe, no copy).  TDLO_CLOUD_FUSED=0:
    // the multi-launch form (comparator)
    void *cloud_fws = nullptr;
    size_t cloud_fws_cap = 0;
    bool cloud_fused_first = true;   // the kernel's state words have to be initialised (first launch, after a failed one, after a reallocation)
    unsigned long long *cloud_res = nullptr;
    unsigned cloud_epoch = 0;
    char *img_pin = nullptr;
    size_t img_pin_cap = 0;
    int img_pin_rows = 0, img_pin_cols = 0;
    bool cloud_fused_on = !(getenv("TDLO_CLOUD_FUSED") && atoi(getenv("TDLO_CLOUD_FUSED")) == 0);
    bool cloud_team_on = !(getenv("TDLO_CLOUD_TEAM") && atoi(getenv("TDLO_CLOUD_TEAM")) == 0);      // 0: ONE finishing workgroup (k_cloud_fused) instead of the last eight as a team (k_cloud_team)
    // visibility pre-pass in one launch (k_node_min_dist_direct): minima + ticket on the device (kept armed by the kernel), [word | M minima] in pinned host memory
    unsigned long long *vis_state = nullptr, *vis_res = nullptr;
    unsigned vis_epoch = 0;
    bool vis_armed = false;
    double *vis_nodes_pin = nullptr;     // 3 x 64 doubles in pinned host memory: the nodes of a pre-pass that rides in the depth -> cloud team kernel
    long long cloud_vis_rides = 0;       // how many frames' pre-passes did (tdlo_debug_route_count 8)
    long long cloud_route[2] = {0, 0};   // tdlo_debug_route_count 6 / 7: depth -> cloud calls served by the one-launch kernel / sent on to the multi-launch form by it
    size_t pin_doubles = 0;
    std::string err;
    int last_F = 0;
    bool lle_batch_dense = false;         // run_frames: a frame of this batch cannot take the banded LLE solve, all of them are staged for the dense kernels
    bool lle_dense_once = false;          // run_frames' retry: the banded LLE solve reported a numeric failure, this call repeats with the dense pivoted kernels
    long long band_retries = 0;           // how often that happened (tdlo_debug_band_retries)
    long long route_count[6] = {0, 0, 0, 0, 0, 0};   // tdlo_debug_route_count: 0 paired set-ups taken up, 1 first iterations started from the handed-over sums,
                                               // 2 M-steps released from their wait for priors, 3 pre-processing registrations served by a device-formed H,
                                               // 4 main registrations whose first iteration ran beside the pre-processing registration (PairNext::ahead),
                                               // 5 calls repeated on the three-kernel route because the fused prologue's grid barrier was abandoned (fuse_fallback)
    // The fused prologue (k_prologue) needs all its workgroups resident at once; its grid barrier gives up after 2 s (tdlo_device.hip, fuse_wait) and
    // the registration then ends with TDLO_E_FUSE: the call is repeated on the copy + three-kernel route and this context stops using the fused
    // prologue (a GPU that could not co-schedule <= 66 workgroups once -- masked or partitioned CUs, long-running kernels of other processes -- is
    // not asked again; every wait costs the caller 2 s).  TDLO_FUSE_FORCE_TIMEOUT=n (test hook): the context's n-th fused launch withholds one arrival.
    bool fuse_on = true;
    int fuse_force_timeout = getenv("TDLO_FUSE_FORCE_TIMEOUT") ? atoi(getenv("TDLO_FUSE_FORCE_TIMEOUT")) : 0;
    bool sort_reuse = !(getenv("TDLO_REUSE_SORT") && atoi(getenv("TDLO_REUSE_SORT")) == 0);   // tdlo_set_sort_reuse: a slot's sorted cloud may serve the next registration of the same nodes
    // results mailbox in pinned host memory (FrameDev::host_out / host_prog): [read-back block | progress word], written by the one-workgroup M-steps
    double *mbox = nullptr;
    size_t mbox_doubles = 0;
    unsigned mbox_epoch = 0;
    bool direct_in = !(getenv("TDLO_DIRECT_UPLOAD") && atoi(getenv("TDLO_DIRECT_UPLOAD")) == 0);   // one frame per call: the set-up kernels read the host-supplied block from pinned host
                                                                                                // memory themselves (small clouds: the fused prologue); 0: copy + three launches (comparator)
    double *late_buf = nullptr;           // pinned: [alpha J | alpha (Y_ext - Y0)] of priors that arrive after the set-up kernel was launched (FrameDev::late_aJ)
    size_t late_doubles = 0;
    bool late_on = !(getenv("TDLO_LATE_PRIORS") && atoi(getenv("TDLO_LATE_PRIORS")) == 0);   // tracking_step: priors formed beside the second registration's set-up kernel; 0: before it (comparator)
    bool mbox_on = !(getenv("TDLO_HOST_MAILBOX") && atoi(getenv("TDLO_HOST_MAILBOX")) == 0);   // TDLO_HOST_MAILBOX=0: the read-back copy + stream wait of rounds 1-3 (comparator)
    // tracking_step, every node visible: both registrations start from the SAME nodes on the same cloud, and the second one's node-side set-up
    // (centring, chain links, iteration constants: k_setup's work) depends on nothing the first one produces.  The first registration's fused
    // prologue then runs it as one more workgroup beside its own (k_prologue, blockIdx = nb + 1) into the slot's second node block, and the
    // second registration starts at its E-step: one launch and one dependent-dispatch gap less per frame.  TDLO_PAIR_SETUP=0: off (comparator).
    struct PairNext {
        int state = 0;                    // 0: nothing; 1: asked for by tracking_step (inputs below); 2: set up on the device (f, up)
        int slot = 0, M = 0, n_vis = 0;
        double sigma2 = 0;
        tdlo_params p{};
        std::vector<double> Y;
        FrameDev f{};
        size_t up = 0;
        bool has_sums = false;            // the pre-processing registration's first M-step leaves the first E-step's sums for it (FrameDev::pair_sums)
        // tracking_step with HIDDEN nodes: the two registrations start from different node sets, so nothing is shared -- but the main registration's
        // prologue (prune, sort, set-up), k_dmin and first E-step depend on nothing the pre-processing registration produces either.  state 3 = asked
        // for by tracking_step: the pre-processing registration's run_frames, once its own first iterations are on the stream, launches them on a
        // SECOND stream into the context's twin slot (own cloud buffers, own node block; the cloud is read from the same pinned staging buffer), with
        // the first M-step behind them waiting for the priors (FrameDev::spec_flag, spec_prev == nullptr; FrameDev::late_mstep).  They run beside the
        // pre-processing registration's 6-7 iterations; the main registration (state 2, ahead) then only releases that M-step and stays on the second
        // stream.  TDLO_AHEAD=0: off (comparator).
        bool ahead = false;
        bool cloud_in_pin = false;        // (request) this frame's cloud is in tdlo_ctx::cloud_pin
        int spec = 0;                     // 1: its first M-step is already on the stream, waiting for the priors (FrameDev::spec_flag)
        unsigned spec_epoch = 0;          // ... under this mailbox epoch
    } pair;
    double *pin2 = nullptr;               // pinned: the paired registration's upload block
    size_t pin2_doubles = 0;
    bool pair_on = !(getenv("TDLO_PAIR_SETUP") && atoi(getenv("TDLO_PAIR_SETUP")) == 0);
    Slot twin;                            // PairNext::ahead: the main registration's cloud buffers and node block while it runs beside the pre-processing one
    bool twin_busy = false;               // something launched ahead into the twin slot may still be running on the second stream (drained before its staging block is reused)
    bool ahead_on = !(getenv("TDLO_AHEAD") && atoi(getenv("TDLO_AHEAD")) == 0);
    // tracking_step: the frame's cloud staged in pinned host memory and read from there by the fused prologue's point workgroups (FrameDev::Xhost)
    // instead of a host-to-device copy in front of it (a 9 us copy on the stream, 8 us of host time for the call); cloud_pending: staged for this
    // slot and not yet on the device.  TDLO_DIRECT_CLOUD=0: the copy (comparator)
    double *cloud_pin = nullptr;
    size_t cloud_pin_doubles = 0;
    int cloud_pending = -1;
    bool cloud_direct_on = !(getenv("TDLO_DIRECT_CLOUD") && atoi(getenv("TDLO_DIRECT_CLOUD")) == 0);
    // early exit, one frame per call: how many iterations go out before the host first looks at the registration's state.  1 unless the caller
    // knows better -- tracking_step passes what the same registration took in the previous frame (consecutive frames of a tracker take about
    // the same number; capped: a wrong guess costs a no-op iteration, ~4 us, per iteration too many).  Consumed by the next run_frames.
    int iter_hint = 0, iter_hint_next = 0;      // (_next: of the registration whose first M-step this one launches ahead, PairNext::spec)
    bool iter_hint_on = !(getenv("TDLO_ITER_HINT") && atoi(getenv("TDLO_ITER_HINT")) == 0);
    bool lle_next_on = !(getenv("TDLO_LLE_NEXT") && atoi(getenv("TDLO_LLE_NEXT")) == 0);        // 0: the host forms every LLE regulariser (comparator)
    bool spec_force_timeout = getenv("TDLO_SPEC_FORCE_TIMEOUT") && atoi(getenv("TDLO_SPEC_FORCE_TIMEOUT")) != 0;   // test hook, see run_frames
    bool spec_on = !(getenv("TDLO_SPEC_MSTEP") && atoi(getenv("TDLO_SPEC_MSTEP")) == 0);        // 0: the paired registration's first M-step is launched when its priors exist (comparator)
    bool pair_sums_on = !(getenv("TDLO_PAIR_SUMS") && atoi(getenv("TDLO_PAIR_SUMS")) == 0);     // 0: the paired registration still runs its own first E-step (comparator)
    // the E-step with two points per lane (k_estep2, tdlo_estep2.hip) for clouds and batches that fill the GPU.  TDLO_ESTEP2=0: never (comparator: k_estep
    // everywhere), 1: wherever it is eligible, whatever the size (tests); unset: by size.  TDLO_ESTEP2_ROWS=8|16: rows of its membership tile.
    int estep2_mode = getenv("TDLO_ESTEP2") ? atoi(getenv("TDLO_ESTEP2")) : -1;
    int estep2_rows = (getenv("TDLO_ESTEP2_ROWS") && atoi(getenv("TDLO_ESTEP2_ROWS")) == 16) ? 16 : 8;
    int estep2_blocks = getenv("TDLO_ESTEP2_BLOCKS") ? atoi(getenv("TDLO_ESTEP2_BLOCKS")) : 0;
    bool xch_self = getenv("TDLO_XCH_SELF") && atoi(getenv("TDLO_XCH_SELF")) != 0;      // a lone rank of the one-shot exchange exchanges with its own inbox (tdlo_set_xch_self)
    bool boost_off_once = false;          // run_frames' retry: an fp64-mode E-step refused a share under the sigma-following (finer) limits -- this call repeats with the coarse ones
    long long boost_retries = 0;          // how often that happened (tdlo_debug_route_count 10)
    bool test_boost_fail = getenv("TDLO_TEST_BOOST_FAIL") && atoi(getenv("TDLO_TEST_BOOST_FAIL")) != 0;   // test hook: every fp64-mode call's first attempt is treated as such a refusal
    // Round 6 experiment, OFF by default: one frame's fixed-length loop as a spin-ahead loop -- E-steps on the second stream, M-steps on
    // the first, the kernels parked on device words instead of the streams' dependent dispatches (FrameDev::spin_on).  TDLO_SPIN_AHEAD=1 switches it on.
    bool spin_ahead_on = getenv("TDLO_SPIN_AHEAD") && atoi(getenv("TDLO_SPIN_AHEAD")) != 0;
    // fp64 E-step of chains beyond 64 nodes: batches whose node window is wide go lane = node (tdlo_estep_wide.h); TDLO_ESTEP_WIDE=0: thread = point throughout (comparator)
    int estep_wide_min = getenv("TDLO_ESTEP_WIDE") ? (atoi(getenv("TDLO_ESTEP_WIDE")) > 0 ? atoi(getenv("TDLO_ESTEP_WIDE")) : (1 << 30)) : kEstepWideMin;
    long long spin_calls = 0;             // registrations run that way (tdlo_debug_route_count 11)
    // A batch's whole fixed-length loop in ONE launch (k_batch_loop, tdlo_estep2.hip): tickets (iteration, frame, chunk) drawn by resident workgroups, the workgroup
    // that completes a frame's E-step runs its M-step.  Round 6 EXPERIMENT, OFF by default (TDLO_BATCH_PERSIST=1 switches it on): the same bits, but 5.1 ms
    // against 1.27 ms per C3 call.  A call whose loop kernel gives a wait up is repeated on the launch-per-step loop.
    bool batch_persist_on = getenv("TDLO_BATCH_PERSIST") && atoi(getenv("TDLO_BATCH_PERSIST")) != 0;
    unsigned *batch_ctl = nullptr;        // device: ticket, abort, per-frame progress counters (batch_loop_ctl_words)
    size_t batch_ctl_words = 0;
    long long batch_loop_calls = 0, batch_loop_fallbacks = 0;      // tdlo_debug_route_count 12 / 13
    bool batch_persist_off_once = false;  // run_frames' repeat of a call whose loop kernel gave a wait up
    long long estep2_frames = 0;          // registrations whose E-step was k_estep2 (tdlo_debug_route_count 9)
    bool timing = false;                  // tdlo_set_timing: record the four events behind tdlo_stats.loop_ms / total_ms (~15 us per call)
    EnqueuePool *pool = nullptr;          // made by the first batch that runs on several streams
    // split-mode scratch
    int split_active = 0;
    double *xch_dmin = nullptr;      // caller-owned device memory of the device-resident N-split exchange (tdlo_split_bind_exchange)
    double *xch_sums = nullptr;
    // tdlo_split_run: the whole split registration driven from C++ (RCCL called directly, or the one-shot exchange)
    double *split_buf = nullptr;     // RCCL form: [init 2 | dmin M | sums 4M+2], all-reduced in place on the context's stream
    size_t split_buf_doubles = 0;
    void *own_comm = nullptr;        // communicator made by tdlo_rccl_comm_init (destroyed with the context)
    unsigned long long *xch_own = nullptr;     // one-shot exchange: this rank's inbox (tdlo_xch_create)
    size_t xch_own_words = 0;
    int xch_rank = 0, xch_nranks = 0, xch_mcap = 0, xch_cap_ranks = 0;
    unsigned long long *xch_peer[kMaxXchRanks] = {};
    std::vector<void *> xch_opened;  // peer inboxes opened from IPC handles (closed with the context)
    unsigned xch_calls = 0;          // epoch of the flags: one per registration, in lockstep on all ranks
};

namespace {

// Development aid (TDLO_TRACK_PROFILE=1): host wall time between marks, accumulated per mark and printed when a tracker is destroyed -- where
// a tracking_step's host time goes (scripts/ubench/track_cpp.cpp).  Off: one predictable branch per mark.
struct HostProf {
    static constexpr int kN = 16;
    const char *name[kN] = {"set_cloud call", "guide nodes", "prepare_frame #1", "enqueue prologue #1", "enqueue iteration #1", "wait #1", "results #1",
                            "traverse + priors", "prepare_frame #2", "enqueue prologue #2", "enqueue iteration #2", "wait #2", "results #2", "between registrations", "", ""};
    double us[kN] = {};
    long n[kN] = {};
    bool on = getenv("TDLO_TRACK_PROFILE") && atoi(getenv("TDLO_TRACK_PROFILE")) != 0;
    int base = 0;                        // 0 inside the first registration of a tracking_step, 6 inside the second
    std::chrono::steady_clock::time_point t0;
    void start() { if (on) t0 = std::chrono::steady_clock::now(); }
    void mark(int id) {
        if (!on) return;
        const auto t1 = std::chrono::steady_clock::now();
        us[id] += std::chrono::duration<double, std::micro>(t1 - t0).count(); ++n[id];
        t0 = t1;
    }
    void report() {
        if (!on) return;
        double tot = 0;
        for (int i = 0; i < kN; ++i) if (n[i]) tot += us[i] / (double)n[i];
        fprintf(stderr, "trackdlo_hip host profile (us per call of the phase; sum %.1f):\n", tot);
        for (int i = 0; i < kN; ++i) if (n[i]) fprintf(stderr, "  %-24s %8.2f  x %ld\n", name[i], us[i] / (double)n[i], n[i]);
    }
};
HostProf g_prof;

int fail(tdlo_ctx *c, int code, const std::string &msg) {
    if (c) c->err = msg;
    // a refused launch or attribute leaves HIP's per-thread "last error" set; the launchers of the NEXT call read it after their own
    // (successful) launches and would report this call's failure a second time
    if (code == TDLO_E_HIP) (void)hipGetLastError();
    return code;
}

#define HIPCHK(c, call)                                                                        \
    do {                                                                                       \
        hipError_t e_ = (call);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return fail((c), TDLO_E_HIP, std::string(#call) + ": " + hipGetErrorString(e_));   \
    } while (0)

// Waits for the stream.  TDLO_SPIN_US=n polls hipStreamQuery for the first n microseconds before it blocks.  Measured on the production-size
// tracking_step (N = 5 000, M = 45; scripts/ubench/track_cpp.cpp, scripts/gpu_track.py): with the stream markers of tdlo_set_timing on, the
// blocking wait wakes up late and polling helps (0.189 -> 0.178 ms per frame); WITHOUT them -- the C API's default -- hipStreamSynchronize is the
// faster one (C++ caller 0.129 against 0.143 ms per frame: the runtime spins by itself, and hipStreamQuery is the more expensive call).  Default: off.
hipError_t wait_stream(hipStream_t s) {
    static const int spin_us = getenv("TDLO_SPIN_US") ? atoi(getenv("TDLO_SPIN_US")) : 0;
    if (spin_us > 0) {
        const auto t0 = std::chrono::steady_clock::now();
        for (;;) {
            const hipError_t e = hipStreamQuery(s);
            if (e != hipErrorNotReady) return e;
            if (std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() > spin_us) break;
        }
    }
    return hipStreamSynchronize(s);
}

// Before a buffer kernels may still be reading or polling is replaced: the first stream, and the second one while something launched ahead
// into the twin slot may still run there (an abandoned launch keeps polling the spec word / reading the staged cloud for up to 2 s: ADVICE r04)
hipError_t drain_for_realloc(tdlo_ctx *c) {
    const hipError_t e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) return e;
    if (c->stream2[0]) return hipStreamSynchronize(c->stream2[0]);      // (cheap when idle; only the growth path comes here)
    return hipSuccess;
}

int ensure_pin(tdlo_ctx *c, size_t doubles) {
    if (doubles <= c->pin_doubles) return 0;
    HIPCHK(c, drain_for_realloc(c));
    if (c->pin) hipHostFree(c->pin);
    c->pin = nullptr; c->pin_doubles = 0;
    HIPCHK(c, hipHostMalloc((void **)&c->pin, doubles * sizeof(double), hipHostMallocDefault));
    c->pin_doubles = doubles;
    return 0;
}

int ensure_late(tdlo_ctx *c, size_t doubles) {
    if (doubles <= c->late_doubles) return 0;
    HIPCHK(c, drain_for_realloc(c));
    if (c->late_buf) hipHostFree(c->late_buf);
    c->late_buf = nullptr; c->late_doubles = 0;
    HIPCHK(c, hipHostMalloc((void **)&c->late_buf, doubles * sizeof(double), hipHostMallocDefault));
    std::memset(c->late_buf, 0, doubles * sizeof(double));        // (its last word is the spec flag a kernel launched ahead compares with its epoch)
    c->late_doubles = doubles;
    return 0;
}

int ensure_mbox(tdlo_ctx *c, size_t doubles) {
    if (doubles <= c->mbox_doubles) return 0;
    HIPCHK(c, drain_for_realloc(c));         // (kernels of an earlier call may still report their progress into the old one)
    if (c->mbox) hipHostFree(c->mbox);
    c->mbox = nullptr; c->mbox_doubles = 0;
    HIPCHK(c, hipHostMalloc((void **)&c->mbox, doubles * sizeof(double), hipHostMallocDefault));
    std::memset(c->mbox, 0, doubles * sizeof(double));
    c->mbox_doubles = doubles;
    return 0;
}

// Waits until the M-steps of registration `epoch` have reported `min_it` completed iterations, or that the registration is done (need_done:
// only that).  *word = the progress word seen.  Returns 0; 1 when the stream has drained without that report (the caller falls back on the
// read-back copy); a negative code for a HIP error on the stream.  The wait spins on pinned host memory -- the M-step's own store is the
// earliest moment the host can know -- and looks at the stream only every 0.5 ms, so that a faulting kernel cannot hang the caller.
int mbox_wait(tdlo_ctx *c, hipStream_t stream, unsigned epoch, int min_it, bool need_done, unsigned long long *word) {
    const unsigned long long *w = (const unsigned long long *)(c->mbox + c->mbox_doubles - 2);
    auto t_chk = std::chrono::steady_clock::now();
    for (unsigned spins = 1;; ++spins) {
        const unsigned long long v = __atomic_load_n(w, __ATOMIC_ACQUIRE);
        if ((unsigned)(v >> 32) == epoch) {
            const bool done = (v >> 31) & 1u;
            if (done || (!need_done && (int)(v & 0x7fffffffu) >= min_it)) { *word = v; return 0; }
        }
        if ((spins & 255u) == 0) {
            const auto now = std::chrono::steady_clock::now();
            if (std::chrono::duration<double, std::micro>(now - t_chk).count() > 500.0) {
                t_chk = now;
                const hipError_t e = hipStreamQuery(stream);
                if (e == hipSuccess) {          // drained: whatever was going to be reported has been (the kernel's stores precede its completion)
                    const unsigned long long v2 = __atomic_load_n(w, __ATOMIC_ACQUIRE);
                    if ((unsigned)(v2 >> 32) == epoch && (((v2 >> 31) & 1u) || (!need_done && (int)(v2 & 0x7fffffffu) >= min_it))) { *word = v2; return 0; }
                    return 1;
                }
                if (e != hipErrorNotReady) return fail(c, TDLO_E_HIP, std::string("stream: ") + hipGetErrorString(e));
            }
        }
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
    }
}

int ensure_xfer(tdlo_ctx *c, size_t doubles) {
    if (doubles <= c->xfer_doubles) return 0;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->xfer) hipFree(c->xfer);
    c->xfer = nullptr; c->xfer_doubles = 0;
    HIPCHK(c, hipMalloc((void **)&c->xfer, doubles * sizeof(double)));
    HIPCHK(c, hipMemsetAsync(c->xfer, 0, doubles * sizeof(double), c->stream));
    c->xfer_doubles = doubles;
    return 0;
}

int ensure_points(tdlo_ctx *c, Slot &s, int n) {
    if (n <= s.cap_points) return 0;
    HIPCHK(c, drain_for_realloc(c));
    if (s.Xraw) hipFree(s.Xraw);
    if (s.Xs) hipFree(s.Xs);
    if (s.bucket) hipFree(s.bucket);
    if (s.blksum) hipFree(s.blksum);
    s.Xraw = nullptr; s.Xs = nullptr; s.bucket = nullptr; s.blksum = nullptr;     // a failing hipMalloc below must not leave
    s.cap_points = 0; s.N0 = 0;                                                      // dangling pointers for tdlo_destroy
    const size_t cap = ((size_t)n + 1023) & ~(size_t)1023;
    const size_t nb = cap / kBlock + 1;
    HIPCHK(c, hipMalloc((void **)&s.Xraw, 3 * cap * sizeof(double)));
    HIPCHK(c, hipMalloc((void **)&s.Xs, 3 * cap * sizeof(double)));
    HIPCHK(c, hipMalloc((void **)&s.bucket, cap * sizeof(unsigned short)));
    HIPCHK(c, hipMalloc((void **)&s.blksum, (nb + 2) * sizeof(double)));      // (+ 2: FrameDev::keep)
    s.cap_points = (int)cap;
    s.sorted_valid = false;
    return 0;
}

int ensure_nodes(tdlo_ctx *c, Slot &s, int M) {
    if (M <= s.cap_nodes) return 0;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (s.nodeblk) hipFree(s.nodeblk);
    s.nodeblk = nullptr; s.cap_nodes = 0;
    const int cap = std::max(M, 16);
    NodeCarve nc(cap);
    HIPCHK(c, hipMalloc((void **)&s.nodeblk, nc.total * sizeof(double)));
    HIPCHK(c, hipMemsetAsync(s.nodeblk, 0, nc.total * sizeof(double), c->stream));
    if (!s.sync) {
        HIPCHK(c, hipMalloc((void **)&s.sync, 256 * sizeof(unsigned)));
        HIPCHK(c, hipMemsetAsync(s.sync, 0, 256 * sizeof(unsigned), c->stream));
    }
    s.nodeblk_doubles = nc.total;
    s.cap_nodes = cap;
    return 0;
}

int ensure_nodes2(tdlo_ctx *c, Slot &s, int M) {
    if (M <= s.cap_nodes2) return 0;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (s.nodeblk2) hipFree(s.nodeblk2);
    s.nodeblk2 = nullptr; s.cap_nodes2 = 0;
    const int cap = std::max(M, 16);
    NodeCarve nc(cap);
    HIPCHK(c, hipMalloc((void **)&s.nodeblk2, nc.total * sizeof(double)));
    HIPCHK(c, hipMemsetAsync(s.nodeblk2, 0, nc.total * sizeof(double), c->stream));
    s.cap_nodes2 = cap;
    return 0;
}

int ensure_hb_next(tdlo_ctx *c, Slot &s, int M) {
    if (M <= s.hb_next_cap) return 0;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (s.hb_next) hipFree(s.hb_next);
    s.hb_next = nullptr; s.hb_next_cap = 0; s.hb_next_valid = false;
    const int cap = std::max(M, 64);
    HIPCHK(c, hipMalloc((void **)&s.hb_next, sizeof(double) * 13 * (size_t)cap));
    s.hb_next_cap = cap;
    return 0;
}

int ensure_cloud_pin(tdlo_ctx *c, size_t doubles) {
    if (doubles <= c->cloud_pin_doubles) return 0;
    HIPCHK(c, drain_for_realloc(c));
    if (c->cloud_pin) hipHostFree(c->cloud_pin);
    c->cloud_pin = nullptr; c->cloud_pin_doubles = 0;
    const size_t cap = (doubles + 4095) & ~(size_t)4095;
    HIPCHK(c, hipHostMalloc((void **)&c->cloud_pin, cap * sizeof(double), hipHostMallocDefault));
    c->cloud_pin_doubles = cap;
    return 0;
}

// a cloud staged in pinned memory that no prologue has taken to the device goes there by a copy
int flush_pending_cloud(tdlo_ctx *c) {
    if (c->cloud_pending < 0) return 0;
    Slot &s = c->slots[c->cloud_pending];
    c->cloud_pending = -1;
    HIPCHK(c, hipMemcpyAsync(s.Xraw, c->cloud_pin, 3 * (size_t)s.N0 * sizeof(double), hipMemcpyHostToDevice, c->stream));
    return 0;
}

int ensure_pin2(tdlo_ctx *c, size_t doubles) {
    if (doubles <= c->pin2_doubles) return 0;
    HIPCHK(c, drain_for_realloc(c));
    if (c->pin2) hipHostFree(c->pin2);
    c->pin2 = nullptr; c->pin2_doubles = 0;
    HIPCHK(c, hipHostMalloc((void **)&c->pin2, doubles * sizeof(double), hipHostMallocDefault));
    c->pin2_doubles = doubles;
    return 0;
}

// the word a speculatively launched M-step waits on (FrameDev::spec_flag): the last double of the late-priors buffer
unsigned long long *spec_flag_word(tdlo_ctx *c) { return (unsigned long long *)(c->late_buf + c->late_doubles - 1); }
void spec_release(tdlo_ctx *c, unsigned epoch, bool go) {
    __atomic_store_n(spec_flag_word(c), ((unsigned long long)epoch << 32) | (go ? 1ull : 2ull), __ATOMIC_RELEASE);
}
// whatever happens to the call that launched it, a waiting M-step is told to leave
void spec_abort(tdlo_ctx *c) {
    if (c->pair.spec) { spec_release(c, c->pair.spec_epoch, false); c->pair.spec = 0; }
}

// the number the next fused prologue on this slot carries in its barrier word ((epoch << 1) | abandoned: 31 bits, never 0); bit 31 of the kernel
// argument is the test hook that withholds one arrival
unsigned next_fuse_epoch(tdlo_ctx *c, Slot &sl) {
    sl.fuse_epoch = sl.fuse_epoch >= 0x7fffffffu ? 1u : sl.fuse_epoch + 1u;
    if (c->fuse_force_timeout > 0 && --c->fuse_force_timeout == 0) return sl.fuse_epoch | 0x80000000u;
    return sl.fuse_epoch;
}

// A registration came back with TDLO_E_FUSE: its fused prologue's grid barrier was abandoned (not all workgroups of the launch became resident
// within 2 s).  Nothing of the caller's has been touched.  Everything in flight is told to leave and drained, what the abandoned launch left
// half-made is invalidated (the slots' sorted clouds, the barrier's arrival counter), and the context takes the copy + three-kernel route from now on.
void fuse_fallback(tdlo_ctx *c) {
    spec_abort(c);
    c->pair.state = 0; c->pair.ahead = false; c->pair.has_sums = false;
    (void)hipStreamSynchronize(c->stream);
    if (c->stream2[0]) (void)hipStreamSynchronize(c->stream2[0]);
    c->twin_busy = false;
    auto reset = [&](Slot &sl) {
        sl.sorted_valid = false;
        if (sl.sync) (void)hipMemsetAsync(sl.sync + 100, 0, 2 * sizeof(unsigned), c->stream);      // arrivals, barrier word (the epochs go on counting)
    };
    for (Slot &sl : c->slots) reset(sl);
    reset(c->twin);
    (void)hipStreamSynchronize(c->stream);
    (void)hipGetLastError();
    if (c->fuse_on)
        std::fprintf(stderr, "trackdlo_hip: the fused prologue's workgroups were not co-resident within 2 s (device %d); this context now uses the "
                             "copy + three-kernel route (see INTEGRATION.md, co-residency)\n", c->device);
    c->fuse_on = false;
    ++c->route_count[5];
}

bool same_params(const tdlo_params &a, const tdlo_params &b) {
    return a.beta == b.beta && a.lambda == b.lambda && a.lle_weight == b.lle_weight && a.mu == b.mu && a.max_iter == b.max_iter && a.tol == b.tol &&
           a.include_lle == b.include_lle && a.alpha == b.alpha && a.k_vis == b.k_vis && a.visibility_threshold == b.visibility_threshold &&
           a.precision == b.precision;
}

int check_params(tdlo_ctx *c, int M, const tdlo_params *p) {
    if (!p) return fail(c, TDLO_E_INVALID, "params is null");
    if (M < 4) return fail(c, TDLO_E_INVALID, "M < 4: the reference's neighbour clamps (trackdlo.cpp:313-321) need at least 4 nodes");
    if (M > kMaxNodes) return fail(c, TDLO_E_INVALID, "M > 1024 is not supported by the E-step tiling");
    if (p->max_iter < 0) return fail(c, TDLO_E_INVALID, "max_iter < 0");
    if (p->precision != TDLO_PREC_F32 && p->precision != TDLO_PREC_F64) return fail(c, TDLO_E_INVALID, "bad precision");
    // the kernel G of trackdlo.cpp:233 divides by beta; a negative lambda makes the M-step's system indefinite (the reference's launch files use
    // 0.35 / 3.0 and 50 000 / 1).  lambda == 0 is legal there (A = D G, least-squares solve): it takes the dense eliminations (prepare_frame).
    if (!(p->beta > 0.0)) return fail(c, TDLO_E_INVALID, "beta must be positive (kernel G of trackdlo.cpp:233)");
    if (!(p->lambda >= 0.0)) return fail(c, TDLO_E_INVALID, "lambda must not be negative");
    return 0;
}

// J / Y_extended (trackdlo.cpp:240-260) into a frame's staging block: aJ = alpha * diag(J), aYd = alpha * (Y_extended - Y0)
int stage_priors(tdlo_ctx *c, double *aJ, double *aYd, const double *Y, int M, const double *priors, int K, double alpha) {
    std::fill(aJ, aJ + M, 0.0);
    std::fill(aYd, aYd + 3 * M, 0.0);
    for (int i = 0; i < K; ++i) {
        const int idx = (int)priors[4 * i];
        if (idx < 0 || idx >= M) return fail(c, TDLO_E_INVALID, "correspondence prior index out of range");
        aJ[idx] = alpha;
        for (int d = 0; d < 3; ++d) aYd[d * M + idx] = alpha * (priors[4 * i + 1 + d] - Y[d * M + idx]);
    }
    return 0;
}

// Fills the host-side upload block [Yin | aJ | aYd | H] for one frame and its FrameDev.
// k_estep2 (two points per lane) can serve this frame: fp32 mode, chains of 8 .. 64 nodes
// How many replica rows of the accumulators a frame's E-step workgroups spread their atomics over (FrameDev::acc_rows).  More rows: fewer atomics per address;
// fewer rows: less for the one-workgroup M-step to fetch in the round trip every iteration waits for.  Measured on one box (scripts/gpu_ab.sh; the rows first as
// a build constant 2 / 4 / 8, then -- with the M-step told its iteration's parity, which halves the fetch on its own -- through TDLO_ACC_ROWS):
//   C5 (782 workgroups of an fp64 E-step of 14 us, uneven work; 1 202 sums): M-step 17.59 / 17.73 / 18.14 us, E-step the same -> 2;
//   C2 (196 workgroups, fp32): E-step 5.26 / 4.60 / 4.52 us as a constant; with the parity known, iteration 13.13 (4 rows) against 12.99 us (8) -> 8;
//   C4 (977 workgroups of equal work, finishing together): E-step 17.7 (4) against 16.95 us (8) -> 8.
// The totals are integers: every choice gives the same bits.  Only the plain one-frame chain M-step is instantiated for fewer rows (tdlo_mstep_chain.hip);
// every other M-step adds up all eight (the unused ones are zero).  TDLO_ACC_ROWS=2|4|8 overrides.
static void choose_acc_rows(const tdlo_ctx *c, FrameDev &f, bool merged) {
    static const int env = getenv("TDLO_ACC_ROWS") ? atoi(getenv("TDLO_ACC_ROWS")) : 0;
    (void)c;
    int r = kAccRows;
    if (!merged && !f.include_lle && !f.mstep_dense && f.M > kChunk && f.M <= kChainLdsMaxNodes && f.precision == TDLO_PREC_F64) r = 2;
    if (env == 2 || env == 4 || env == 8) r = env;
    f.acc_rows = r;
}

static bool estep2_eligible(const tdlo_ctx *c, const FrameDev &f) {
    return c->estep2_mode != 0 && f.precision == TDLO_PREC_F32 && f.M >= 8 && f.M <= kChunk;
}
// ... and does: the launch geometry of a frame whose E-step is k_estep2.  `share`: the frame is one of a batch (the batch fills the GPU, every
// wave takes two 128-point batches so that a workgroup's prologue and epilogue are paid half as often -- as k_estep's batches do).
static void estep2_geometry(tdlo_ctx *c, FrameDev &f, bool share) {
    if (!f.estep2) {
        ++c->estep2_frames;
        // the fixed point's grain is one wave x one 128-point batch here: a node's P1 share can reach 128 (prepare_frame sized the exponents for
        // 64), and Q is converted per PAIR of points -- one binary digit less keeps every conversion exact (|v 2^sh| < 2^51); the totals' bound
        // (2^62) only gains from it
        for (int k = 0; k < 3; ++k) { f.acc_sh[k] -= 1; f.acc_lim[k] *= 2.0; }
    }
    f.estep2 = c->estep2_rows;
    f.wide_tile = 0; f.eb = 256;
    const int nb128 = (f.N0 + 127) / 128;
    int nblk = (nb128 + 3) / 4;
    if (share && nblk >= 32) nblk = (nblk + 1) / 2;
    // One frame: every workgroup resident at once (five of 28 KB and 86 VGPRs per CU: 1280 on 256 CUs) and every wave the SAME number of batches -- first
    // the batches per wave that fits the cloud into the resident waves, then the workgroups that many batches need.  (N = 2 000 000: 15 625 batches, 4 per
    // wave, 977 workgroups.  1280 workgroups gave waves of 3 and of 4 batches -- 19.0 us per E-step against 16.9 with 1024 or 1536, scripts/gpu_estep2_sweep.sh.)
    const int cap = c->estep2_blocks > 0 ? c->estep2_blocks : (c->cfg.estep_blocks > 0 ? c->cfg.estep_blocks : 1280);
    if (nblk > cap) { const int per_wave = (nb128 + 4 * cap - 1) / (4 * cap); nblk = (nb128 + 4 * per_wave - 1) / (4 * per_wave); }
    f.nblkE = std::max(1, std::min(nblk, std::min(cap, kMaxEstepBlocks)));
}

int prepare_frame(tdlo_ctx *c, int slot, const double *Y, int M, double sigma2, const tdlo_params *p,
                  const double *priors, int K, const int *vis, int n_vis, const double *H_override,
                  double *stage, FrameDev &f, bool second_block = false, bool hb_may_stay = true) {      // hb_may_stay: f.Hb may point at Slot::hb_next (a batch moves every frame's Hb into its transfer buffer instead)
    Slot &s = slot < 0 ? c->twin : c->slots[slot];      // (slot < 0: the context's twin slot, PairNext::ahead)
    if (s.N0 <= 0) return fail(c, TDLO_E_INVALID, "no cloud resident in slot (call tdlo_set_cloud)");
    int rc = second_block ? ensure_nodes2(c, s, M) : ensure_nodes(c, s, M);
    if (rc) return rc;
    NodeCarve nc(M);
    double *blk = second_block ? s.nodeblk2 : s.nodeblk;
    std::memcpy(stage + nc.Yin, Y, sizeof(double) * 3 * M);
    {
        const int prc = stage_priors(c, stage + nc.aJ, stage + nc.aYd, Y, M, priors, K, p->alpha);
        if (prc) return prc;
    }
    bool lle_band = false, h_banded = false, hb_resident = false;
    double band_s2_max = 1e300;
    if (p->include_lle) {
        double *H = stage + nc.H;
        // The banded L D L^T in the chain's state (tdlo_mstep_band.hip) serves the registration when (i) H is banded like the
        // reference's own (I - L)^T (I - L) -- +-6 nodes; an H_override may be anything --, (ii) it is symmetric there (the band is read
        // from one triangle), (iii) no two consecutive nodes are closer than h_min: the state precision K contains Q^-1 ~ 3 beta^4 / h^3, and
        // the solve's error grows like eps lambda sigma2 K / P1: at h = 1 mm with the pre-processing parameters (beta 3, lambda 1)
        // 1e-13 m, at 0.1 mm 1e-11 m, at 0.01 mm 1e-7 m (scripts/band_gap_study.py); the bound scales with cbrt(lambda beta^4).
        // Everything else (coincident nodes in particular: K is infinite there) keeps the dense pivoted eliminations.
        if (mstep_band_enabled() && !c->lle_dense_once && !c->lle_batch_dense && p->lambda > 0 && p->beta > 0 && M <= kChainLdsMaxNodes) {      // (the banded solve keeps a record per unknown in LDS: up to 512 nodes)
            lle_band = true;
            const double hmin = 1e-3 * std::cbrt(p->lambda * std::pow(p->beta / 3.0, 4));
            double hsum = 0;
            for (int i = 0; i + 1 < M && lle_band; ++i) {
                double d2 = 0;
                for (int d = 0; d < 3; ++d) { const double e = Y[d * M + i + 1] - Y[d * M + i]; d2 += e * e; }
                if (!(d2 >= hmin * hmin)) lle_band = false;
                hsum += std::sqrt(d2);
            }
            // (iv) fp64 mode only: K's entries ~ 3 beta^4 / h^3 are rounded to fp64 when the records are formed, and a smooth displacement field
            // cancels in K x to many digits -- the rounding comes back divided by the data term, ~ eps sigma2 K |x| / P1.  At the reference's scale
            // (sigma2 <= 1e-2 m2) that is 1e-13 m; a registration started from sigma2 = 0 on a chain of 8 .. 10 m begins at sigma2 = 3 .. 6 m2 and
            // with beta = 5 the banded form -- ANY fp64 solve of the stored band: scripts/gpu_band_cond_study.py, DESIGN.md 4 -- is 1.4 .. 5.5e-9 m
            // from the reference's dense system, outside the mode's 1e-9 m.  Measured onset sigma2 beta^4 ~ 1 500 at h = 2 cm; the bound sits a
            // factor 3 below and scales with K.  A sigma2 given by the caller is checked here, one computed on the device (sigma2 == 0,
            // trackdlo.cpp:271-273) by the M-step itself (FrameDev::band_s2_max), which ends in the repeat on the dense kernels (run_frames).
            band_s2_max = 1e300;
            if (lle_band && p->precision == 1 && M > 1) {
                const double hm = hsum / (M - 1);
                band_s2_max = 6.25e7 * hm * hm * hm / std::pow(p->beta, 4);
                if (sigma2 > band_s2_max) lle_band = false;
            }
            if (H_override) {
                for (int j = 0; j < M && lle_band; ++j)
                    for (int i = 0; i < M; ++i) {
                        const double v = H_override[(size_t)j * M + i];
                        if ((std::abs(i - j) > 6 && v != 0.0) || v != H_override[(size_t)i * M + j] || !(v == v)) { lle_band = false; break; }
                    }
            }
        }
        if (lle_band) {
            // the banded solve reads H through its 13 diagonals only (Hb[13 i + u] = H(i, i - 6 + u)): the M x M matrix is neither formed nor uploaded
            double *Hb = stage + nc.Hb;
            if (H_override) {
                for (int i = 0; i < M; ++i)
                    for (int u = 0; u < 13; ++u) { const int j = i - 6 + u; Hb[(size_t)13 * i + u] = (j >= 0 && j < M) ? H_override[(size_t)j * M + i] : 0.0; }
            } else if (c->lle_next_on && hb_may_stay && !second_block && s.hb_next_valid && s.hb_next_Y.size() == 3 * (size_t)M &&
                       std::memcmp(s.hb_next_Y.data(), Y, sizeof(double) * 3 * M) == 0) {
                hb_resident = true;                          // formed on the device at the end of the previous tracking_step (FrameDev::lle_next): the same values
            } else {
                lle_regulariser_band(Y, M, Hb);              // trackdlo.cpp:236-237, O(M)
            }
        } else if (H_override) {
            std::memcpy(H, H_override, sizeof(double) * (size_t)M * M);
            h_banded = true;
            for (int j = 0; j < M && h_banded; ++j)
                for (int i = 0; i < M; ++i) if (std::abs(i - j) > 6 && H_override[(size_t)j * M + i] != 0.0) { h_banded = false; break; }
        } else {
            h_banded = true;
            std::vector<double> L((size_t)M * M);
            lle_weights(6, Y, M, L.data());                  // trackdlo.cpp:236
            lle_regulariser(L.data(), M, H);                 // :237
        }
    }

    std::memset(&f, 0, sizeof f);
    f.N0 = s.N0; f.M = M; f.ldx = s.cap_points;
    const int nbatch = (s.N0 + 63) / 64;
    // 4 waves per workgroup (fp64 tiles are twice as large anyway; fp32: a cloud that cannot fill the GPU gets twice the workgroups of the
    // 8-wave form, one wave per SIMD -- C2: 5.6 -> 5.2 us per E-step -- and at N = 2 000 000 1024 workgroups of 4 waves beat 512 of 8 by
    // 4 %).  The results do not depend on this choice: the sums are integers from the grain of one wave x one batch on.
    f.eb = 256;
    { static const int eb_env = getenv("TDLO_ESTEP_EB") ? atoi(getenv("TDLO_ESTEP_EB")) : 0; if (eb_env == 256 || (eb_env == 512 && M <= 64 && p->precision == TDLO_PREC_F32)) f.eb = eb_env; }
    const int wpb = f.eb / 64;
    int nblk = (nbatch + wpb - 1) / wpb;
    // One frame of moderate size (up to 262 144 points) keeps its whole node window in a 64-row tile (67 KB of LDS per 4-wave
    // workgroup).  A larger cloud is VALU-bound instead: it takes the 24-row tile of the batch path (more workgroups per CU hide the
    // LDS / scalar-load latencies: 47 -> 36 us at N = 2 000 000).  Batches (run_frames, F > 1) always use the small tile.
    f.wide_tile = (M > kChunk || nbatch < 4096) ? 1 : 0;
    // (fp32, chains up to 64 nodes, a cloud that fills the GPU: 2048 workgroups of the 16-row tile -- 26.5 against 27.9 us per E-step at N = 2 000 000)
    int cap = c->cfg.estep_blocks > 0 ? c->cfg.estep_blocks : (f.wide_tile ? 512 : ((M <= kChunk && p->precision == TDLO_PREC_F32) ? 2048 : 1024));
    // (more than 64 nodes: the 24-row tile lets two fp64 workgroups share a CU -- C5: 512 workgroups 21.4 us per converged E-step and 69.7 us per
    //  iteration over a whole call, against 28.4 / 86.3 with 256 workgroups of the former 64-row tile, which filled a CU's LDS alone)
    { static const int cap_env = getenv("TDLO_ESTEP_BLOCKS") ? atoi(getenv("TDLO_ESTEP_BLOCKS")) : 0; if (cap_env > 0) cap = cap_env; }
    cap = std::min(cap, kMaxEstepBlocks);
    f.nblkE = std::max(1, std::min(nblk, cap));
    f.max_iter = p->max_iter; f.include_lle = p->include_lle ? 1 : 0; f.has_priors = K > 0 ? 1 : 0;
    // trackdlo.cpp:358: visible_nodes.size() != Y.rows() && !visible_nodes.empty() && k_vis != 0
    f.vis_branch = (n_vis != M && n_vis != 0 && p->k_vis != 0) ? 1 : 0;
    (void)vis;
    f.precision = p->precision;
    // which M-step: decided here, once per frame (the launchers dispatch on the descriptor, not on the process-wide toggles)
    f.mstep_dense = (mstep_chain_enabled() && p->lambda > 0.0) ? 0 : 1;      // (the chain smoother works in units of 1 / (lambda sigma2))
    f.lle_band = lle_band ? 1 : 0;
    f.band_s2_max = band_s2_max;
    f.h_banded = h_banded ? 1 : 0;
    f.need_G = ((p->include_lle && !lle_band) || (!p->include_lle && f.mstep_dense)) ? 1 : 0;
    {   // test hook: iteration k of the multi-CU M-steps behaves as if a hand-off had timed out (tests/test_parity_gpu.py)
        static const int force_it = getenv("TDLO_MCU_FORCE_TIMEOUT") ? atoi(getenv("TDLO_MCU_FORCE_TIMEOUT")) : -1;
        f.force_timeout_it = force_it;
    }
    // prune / counting-sort workgroups: 256 points each up to 1024 workgroups; larger clouds give every workgroup 2, 4, 8 ...
    // consecutive tiles, so that the one-workgroup scan of the (workgroup, node) histogram in k_setup stays ~1000 rows
    f.prune_tiles = 1;
    while ((s.N0 + kBlock * f.prune_tiles - 1) / (kBlock * f.prune_tiles) > 1024) f.prune_tiles *= 2;
    f.nprune_blocks = (s.N0 + kBlock * f.prune_tiles - 1) / (kBlock * f.prune_tiles);
    {   // the E-step's node window in bits (FrameDev::win_e32): TDLO_WINDOW=exact keeps every membership that is not exactly zero in the arithmetic
        static const bool exact = getenv("TDLO_WINDOW") && getenv("TDLO_WINDOW")[0] == 'e';
        f.win_e32 = exact ? 154.0 : 36.0; f.win_e64 = exact ? 1100.0 : 66.0;
    }
    f.tol = p->tol; f.beta = p->beta; f.lambda = p->lambda; f.lle_weight = p->lle_weight; f.mu = p->mu;
    f.alpha = p->alpha; f.k_vis = p->k_vis; f.vis_thr = p->visibility_threshold; f.sigma2_in = sigma2;
    {   // per (prune block, node) histogram of the counting sort
        const size_t need = (size_t)f.nprune_blocks * M;
        if (need > s.hist_ints) {
            HIPCHK(c, hipStreamSynchronize(c->stream));
            if (s.hist) hipFree(s.hist);
            s.hist = nullptr; s.hist_ints = 0;
            HIPCHK(c, hipMalloc((void **)&s.hist, 2 * need * sizeof(int)));       // counts | start offsets
            s.hist_ints = need;
        }
    }
    f.Xraw = s.Xraw; f.Xs = s.Xs; f.bucket = s.bucket; f.hist = s.hist; f.hist_off = s.hist + (size_t)f.nprune_blocks * M; f.blksum = s.blksum;
    f.keep = s.blksum + (size_t)s.cap_points / kBlock + 1;
    f.reuse_sorted = (c->sort_reuse && s.sorted_valid && s.sorted_prec == p->precision && s.sorted_Y.size() == 3 * (size_t)M &&
                      std::memcmp(s.sorted_Y.data(), Y, sizeof(double) * 3 * M) == 0) ? 1 : 0;
    f.Yin = blk + nc.Yin; f.ctr = blk + nc.ctr; f.Y = blk + nc.Y; f.Y0 = blk + nc.Y0; f.nodes = blk + nc.nodes;
    f.coord = blk + nc.coord; f.G = blk + nc.G; f.chain = blk + nc.chain; f.H = blk + nc.H; f.Hb = blk + nc.Hb; f.HG = blk + nc.HG; f.HY0 = blk + nc.HY0;
    f.aJ = blk + nc.aJ; f.aYd = blk + nc.aYd; f.dminbits = (unsigned long long *)(blk + nc.dmin);
    f.acc = (long long *)(blk + nc.acc);
    f.band = blk + nc.band;
    {   // fixed-point exponents of the accumulators: totals stay below 2^62.  P1_m <= N0.  |R_m| <= sum_n P_mn |x_n - y_m| <= N0 D and
        // Q <= N0 D^2 with D a bound on point-node distances: every kept point is within 0.1 m of a node (:177-195), nodes are within
        // the chain's length of each other; doubled for the motion of the nodes during the registration, rounded up to a power of two
        int ln = 0; while (((long long)1 << ln) < (long long)s.N0 + 1) ++ln;
        double len = 0;
        for (int i = 0; i + 1 < M; ++i) { double d2 = 0; for (int d = 0; d < 3; ++d) { const double e = Y[d * M + i + 1] - Y[d * M + i]; d2 += e * e; } len += std::sqrt(d2); }
        int ld = 0; while (std::ldexp(1.0, ld) < 2.0 * (len + 0.2) && ld < 20) ++ld;
        // ... and one wave's share of one 64-point batch below 2^51 (acc_fix): P1 <= 64, |R| <= 64 D, a point's share of Q <= D^2
        static const int sh_coarser = getenv("TDLO_ACC_COARSER") ? atoi(getenv("TDLO_ACC_COARSER")) : 0;      // experiment: how much of a deviation from the oracle is the sums' resolution (scripts/gpu_acc.py)
        const int shP = std::min(60 - ln, 44) - sh_coarser;
        f.acc_sh[0] = shP; f.acc_sh[1] = shP - ld; f.acc_sh[2] = shP - 2 * ld;
        {   // (experiment, with TDLO_ACC_COARSER: only one class of sums coarser)
            static const int only = getenv("TDLO_ACC_COARSER_ONLY") ? atoi(getenv("TDLO_ACC_COARSER_ONLY")) : -1;
            if (only >= 0) for (int k = 0; k < 3; ++k) if (k != only) f.acc_sh[k] += sh_coarser;
        }
        // D is a heuristic (nodes can be pulled anywhere by a prior, a diverging registration leaves it): the E-step CHECKS every value it
        // converts against these limits -- the conversion stays exact below 2^51, the sum of all batches' shares (P1, R: one per 64-point
        // batch and node; Q: one per point) below 2^62 -- and ends the registration with TDLO_E_NUMERIC beyond them, as it does for NaN
        const int lshare = std::min(51, 67 - std::max(ln, 6)), lpoint = std::min(51, 61 - ln);
        f.acc_lim[0] = std::ldexp(1.0, lshare - f.acc_sh[0]); f.acc_lim[1] = std::ldexp(1.0, lshare - f.acc_sh[1]); f.acc_lim[2] = std::ldexp(1.0, lpoint - f.acc_sh[2]);
    }
    f.acc_boost_off = c->boost_off_once ? 1 : 0;
    f.estep_wide_min = c->estep_wide_min;
    // (acc_rows: choose_acc_rows, once the E-step's geometry is final)
    // one frame whose cloud fills the GPU alone (2048 waves of 64 points: from there on two points per lane win, scripts/gpu_estep2_check.py /
    // profiles/r06_measured.log -- 131 072 points 4.8 against 5.3 us per E-step, 250 000 points (a shard of C4 on eight ranks) 5.5 against 7.3 us)
    if (estep2_eligible(c, f) && (c->estep2_mode == 1 || nbatch >= kEstep2MinWaves)) estep2_geometry(c, f, false);
    choose_acc_rows(c, f, false);
    f.sums = blk + nc.sums; f.Ascr = blk + nc.Ascr; f.Yout = blk + nc.Yout; f.dbg = (unsigned long long *)(blk + nc.dbg);
    f.sync = s.sync;
    f.st = (IterState *)(blk + nc.st);
    if (hb_resident) f.Hb = s.hb_next;
    return 0;
}

// the E-step's fixed-point accumulators of these frames, both parities (the measurement entry points restart the iteration counter)
hipError_t zero_accumulators(const std::vector<FrameDev> &fh, int F, hipStream_t s) {
    for (int i = 0; i < F; ++i) {
        const hipError_t e = hipMemsetAsync(fh[i].acc, 0, sizeof(long long) * 2 * kAccRows * (4 * (size_t)fh[i].M + 2), s);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

size_t upload_doubles(const NodeCarve &nc, const tdlo_params *p, bool band) { return p->include_lle ? (band ? nc.H : nc.upload) : nc.Hb; }

void fill_stats(tdlo_stats *st, const IterState &is) {
    st->iters = is.it; st->converged = is.converged; st->n_kept = is.N; st->status = is.status; st->sigma2 = is.sigma2;
    st->mstep_retries = is.retries;
}

// Priors formed by the host while the set-up kernel of the registration runs (tracking_step's second registration: they come out of the first
// registration's result through traverse_euclidean, 6.5 us of host time that used to sit between the two registrations).  Returns 0 and the
// K x 4 rows, or an error code.
typedef std::function<int(const double *&, int &)> LatePriors;

// PairNext::ahead (tracking_step with hidden nodes; called by the pre-processing registration's run_frames once its own first iterations are on the
// first stream): the main registration's fused prologue, k_dmin and first E-step on the SECOND stream, in the context's twin slot -- they read
// the frame's cloud from the pinned staging buffer like the pre-processing registration's own prologue -- and its first M-step behind them, waiting
// for the priors.  Returns 0 whether or not it launched anything (PairNext::state == 2 && ahead says it did); a negative code for a HIP error.
int launch_ahead(tdlo_ctx *c) {
    tdlo_ctx::PairNext &pn = c->pair;
    const int M = pn.M;
    const int N0 = c->slots[pn.slot].N0;
    const NodeCarve nc(M);
    if (!pn.cloud_in_pin || N0 <= 0 || pn.p.include_lle || pn.p.max_iter <= 0 || M > kChainLdsMaxNodes) return 0;
    if (c->late_doubles < 4 * (size_t)M + 2 || c->mbox_doubles < nc.readback + 4 || c->cloud_pin_doubles < 3 * (size_t)N0) return 0;      // (sized by tracking_step before anything was launched)
    if (!c->stream2[0]) HIPCHK(c, hipStreamCreateWithFlags(&c->stream2[0], hipStreamNonBlocking));
    hipStream_t sb = c->stream2[0];
    Slot &tw = c->twin;
    const size_t need_hist = (size_t)((N0 + kBlock - 1) / kBlock) * M;
    const bool grow = N0 > tw.cap_points || M > tw.cap_nodes || need_hist > tw.hist_ints || nc.upload + 2 > c->pin2_doubles;
    if (grow || c->twin_busy) HIPCHK(c, hipStreamSynchronize(sb));      // (buffers about to be replaced, or a staging block an abandoned launch may still be reading)
    c->twin_busy = false;
    int rc;
    if ((rc = ensure_points(c, tw, N0))) return rc;
    tw.N0 = N0; tw.sorted_valid = false;
    if ((rc = ensure_pin2(c, nc.upload + 2))) return rc;
    FrameDev &f = pn.f;
    if ((rc = prepare_frame(c, -1, pn.Y.data(), M, pn.sigma2, &pn.p, nullptr, 0, nullptr, pn.n_vis, nullptr, c->pin2, f, false))) return rc;
    if (grow) HIPCHK(c, hipStreamSynchronize(c->stream));               // (new buffers are cleared on the first stream)
    if (!f.wide_tile || f.mstep_dense || f.reuse_sorted || !c->fuse_on || !prologue_pair_ok(f)) return 0;      // not this frame: the main registration takes the ordinary route
    pn.up = upload_doubles(nc, &pn.p, false);
    f.Xhost = c->cloud_pin;
    std::memcpy(c->pin2 + nc.fdev, &f, sizeof(FrameDev));
    const unsigned fep = next_fuse_epoch(c, tw);
    c->twin_busy = true;
    HIPCHK(c, launch_prologue_direct(&f, c->pin2, tw.nodeblk, (int)pn.up, (int)nc.Yin, fep, sb));
    f.Xhost = nullptr;                      // (the cloud is in the twin's Xraw for everything that follows)
    f.late_mstep = 1;                       // the priors do not exist yet: this E-step leaves them alone, the M-step reads them itself
    const FrameDev *fdb = (const FrameDev *)(tw.nodeblk + nc.fdev);
    if (f.vis_branch) HIPCHK(c, launch_estep_only(fdb, &f, 1, 1, sb));
    HIPCHK(c, launch_estep_only(fdb, &f, 1, 0, sb));
    unsigned e2 = ++c->mbox_epoch;
    if (e2 == 0) e2 = ++c->mbox_epoch;
    FrameDev fs = f;
    fs.reuse_sorted = 1; fs.has_priors = 1;
    fs.late_aJ = c->late_buf; fs.late_aYd = c->late_buf + M;
    fs.host_out = c->mbox; fs.host_prog = (unsigned long long *)(c->mbox + c->mbox_doubles - 2); fs.host_epoch = e2;
    fs.host_report_it = (pn.p.tol <= 0.0 || pn.p.max_iter <= 2 * kChunkIters) ? 0 : std::max(1, std::min(std::min(c->iter_hint_next, kIterHintMax), pn.p.max_iter));
    fs.spec_flag = spec_flag_word(c); fs.spec_prev = nullptr; fs.spec_epoch = e2;
    HIPCHK(c, launch_mstep_chain(fdb, &fs, 1, 0, fs.precision == TDLO_PREC_F64, sb));
    pn.has_sums = false; pn.spec = 1; pn.spec_epoch = e2; pn.ahead = true; pn.state = 2;
    return 0;
}

// Shared driver of tdlo_cpd_lle_resident / _batch.
int run_frames(tdlo_ctx *c, int F, const int *slots, double *Y, int M, double *sigma2, const tdlo_params *p,
               const double *priors, int K, const int *vis, int n_vis, const double *H_override, tdlo_stats *stats,
               const LatePriors *late = nullptr) {
    const auto t_host0 = std::chrono::steady_clock::now();
    int rc = check_params(c, M, p);
    if (rc) return rc;
    const int iter_hint = c->iter_hint_on ? c->iter_hint : 0;
    if (!c->iter_hint_on) c->iter_hint_next = 0;
    c->iter_hint = 0;
    if (late && F != 1) return fail(c, TDLO_E_INVALID, "late priors: one frame per call");
    if (F < 1 || F > c->cfg.max_frames) return fail(c, TDLO_E_INVALID, "bad frame count");
    NodeCarve nc(M);
    // Batches move their host-supplied blocks and their results with ONE copy each way: every small copy is a 4-5 us blit
    // kernel on the stream, and 32 frames x (upload + read-back [+ flags per polling chunk]) had grown to 0.3 ms of a 2.3 ms
    // call.  The frames' [Yin | aJ | aYd | H] and [Yout | IterState] then live in one device buffer instead of the slots'
    // node blocks (the kernels only see pointers).
    const bool merged = F > 1;
    // how much of a frame's block travels: without the LLE term up to Hb; with it the 13 diagonals Hb when the banded solve serves the call, else
    // the dense H as well.  A batch is staged at the stride of the banded form first; a frame the banded solve cannot take sends the whole
    // batch to the dense kernels (one M-step kernel per launch), and the frames are staged again at the full stride.
    bool band_batch = p->include_lle && mstep_band_enabled() && !c->lle_dense_once;
    size_t up = upload_doubles(nc, p, band_batch);
    size_t ustride = merged ? up : nc.upload;
    // the frame descriptors ride in the same host-to-device copy: one frame -> at the head of its upload block, a batch -> as an array
    // behind the F upload blocks (every separate small copy is a blit kernel plus a dependent-dispatch gap, ~4 us)
    const size_t fdd = ((size_t)F * sizeof(FrameDev) + 15) / 16 * 2;      // doubles of the descriptor array of a batch
    rc = ensure_pin(c, (size_t)F * std::max(nc.upload, nc.readback + 2) + fdd + 2 * (size_t)F * std::max(nc.readback, (sizeof(IterState) + 7) / 8) + 4);
    if (rc) return rc;
    if (merged) { rc = ensure_xfer(c, (size_t)F * (upload_doubles(nc, p, false) + nc.readback) + fdd); if (rc) return rc; }
    // tracking_step's second registration whose node-side set-up the first one's prologue has already done (PairNext): same slot, nodes, sigma2 and
    // parameters as were set up, on the cloud that was sorted for these nodes -- then nothing is staged and no prologue is launched
    bool paired = false;
    bool ahead = false;           // ... or whose whole first iteration up to the M-step has run beside the previous registration, in the twin slot (PairNext::ahead)
    if (late && c->pair.state == 2) {
        const tdlo_ctx::PairNext &pn = c->pair;
        const Slot &sl = c->slots[slots[0]];
        const bool same = F == 1 && !priors && !H_override && pn.slot == slots[0] && pn.M == M && pn.n_vis == n_vis && pn.sigma2 == sigma2[0] && same_params(pn.p, *p) &&
                          std::memcmp(pn.Y.data(), Y, sizeof(double) * 3 * M) == 0 && pn.f.wide_tile != 0;
        if (pn.ahead) paired = ahead = same && c->stream2[0] != nullptr;
        else paired = same && sl.sorted_valid && sl.sorted_prec == p->precision &&
                      sl.sorted_Y.size() == 3 * (size_t)M && std::memcmp(sl.sorted_Y.data(), Y, sizeof(double) * 3 * M) == 0;
        c->pair.state = 0; c->pair.ahead = false;
    }
    // its first M-step may already be waiting on the stream (launched by the previous call, below): released when the priors are staged, told to
    // leave on every other way out of this function
    struct SpecGuard {
        tdlo_ctx *c; unsigned epoch = 0; bool live = false;
        void release(bool go) { if (live) { spec_release(c, epoch, go); live = false; } }
        ~SpecGuard() { release(false); }
    } sg{c};
    if (late && c->pair.spec) { sg.epoch = c->pair.spec_epoch; sg.live = true; c->pair.spec = 0; }
    if (!paired) sg.release(false);
    c->fh.assign(F, FrameDev{});
    if (paired) {
        c->fh[0] = c->pair.f; c->fh[0].reuse_sorted = 1; up = c->pair.up;
        if (ahead) ++c->route_count[4];
        else { ++c->route_count[0]; if (c->pair.has_sums) ++c->route_count[1]; }
    }
    for (int pass = 0; pass < 2 && !paired; ++pass) {
    for (int i = 0; i < F; ++i) {
        rc = prepare_frame(c, slots[i], Y + (size_t)i * 3 * M, M, sigma2[i], p, priors, K, vis, n_vis, H_override,
                           c->pin + (size_t)i * ustride, c->fh[i], false, !merged);
        if (rc) return rc;
        if (merged) {
            FrameDev &f = c->fh[i];
            f.wide_tile = 0;
            // batches fill the GPU: half the workgroups per frame, every wave takes two 64-point batches -- the per-workgroup prologue (nodes
            // to LDS) and epilogue (wave sums, atomics) are paid half as often (32 frames: 879 k -> 991 k it/s in the loop).  The sums do
            // not depend on how batches are dealt out (integer accumulation), so the results are those of the single call, bit for bit.
            if (c->cfg.estep_blocks <= 0 && f.nblkE >= 64) f.nblkE = (f.nblkE + 1) / 2;
            double *bu = c->xfer + (size_t)i * up, *br = c->xfer + (size_t)F * up + fdd + (size_t)i * nc.readback;
            f.Yin = bu + nc.Yin; f.aJ = bu + nc.aJ; f.aYd = bu + nc.aYd;
            if (p->include_lle) { f.Hb = bu + nc.Hb; f.H = bu + nc.H; }
            f.Yout = br; f.st = (IterState *)(br + (nc.st - nc.Yout));
        }
    }
    if (merged) {
        // a batch whose frames together fill the GPU takes the E-step with two points per lane (one choice for the whole launch)
        long long waves = 0;
        bool elig = true;
        for (int i = 0; i < F; ++i) { waves += (c->fh[i].N0 + 63) / 64; elig = elig && estep2_eligible(c, c->fh[i]); }
        const bool two = elig && (c->estep2_mode == 1 || waves >= kEstep2MinWaves);
        for (int i = 0; i < F; ++i) if (two) estep2_geometry(c, c->fh[i], true);      // (a frame prepare_frame had given to k_estep2 on its own size keeps it only if the whole batch does:
        for (int i = 0; i < F; ++i) choose_acc_rows(c, c->fh[i], true);
        // a batch of short fp32 chains without the LLE term whose frames have at most 64 E-step workgroups each (C3: 49): two replica rows -- no more atomics
        // per address than one 50 000-point frame puts on eight -- and a quarter of the sums for every frame's M-step to fetch (one choice for the whole batch)
        {
            static const int env = getenv("TDLO_ACC_ROWS") ? atoi(getenv("TDLO_ACC_ROWS")) : 0;
            bool small = env == 0 && !p->include_lle;
            for (int i = 0; i < F; ++i) small = small && c->fh[i].nblkE <= 64 && !c->fh[i].mstep_dense && c->fh[i].M <= kChunk;
            if (small) for (int i = 0; i < F; ++i) c->fh[i].acc_rows = 2;
        }
        if (!two) for (int i = 0; i < F; ++i) if (c->fh[i].estep2) return fail(c, TDLO_E_INVALID, "internal: a batch's frames disagree about the E-step kernel");   //  same M, precision and mode -- they cannot)
    }
    if (!p->include_lle) break;
    bool all_band = true;
    for (int i = 0; i < F; ++i) all_band = all_band && c->fh[i].lle_band;
    if (all_band) break;
    if (!merged) { up = upload_doubles(nc, p, false); break; }      // one frame: its block was staged at the full stride anyway
    if (pass == 0) { c->lle_batch_dense = true; band_batch = false; up = upload_doubles(nc, p, false); ustride = up; }
    }
    c->lle_batch_dense = false;
    if (!merged && !paired && p->include_lle && c->fh[0].lle_band && c->slots[slots[0]].hb_next != nullptr && c->fh[0].Hb == c->slots[slots[0]].hb_next)
        { up = nc.Hb; ++c->route_count[3]; }      // H's 13 diagonals are on the device already (Slot::hb_next): they do not travel
    // tracking_step's main registration: the M-step that finishes it leaves the next frame's LLE regulariser behind (FrameDev::lle_next; the
    // buffer was sized by tracking_step before anything was launched)
    const bool lle_next = late != nullptr && !merged && !ahead && c->lle_next_on && !p->include_lle && !c->fh[0].mstep_dense && M <= 256 && p->max_iter > 0 &&
                          c->slots[slots[0]].hb_next_cap >= M;      // (ahead: its first M-step is on the second stream already, without this job -- and the next frame's
                                                                    //  pre-processing registration, on the first stream, must not race it)
    if (lle_next) { c->fh[0].lle_next = c->slots[slots[0]].hb_next; c->slots[slots[0]].hb_next_valid = false; }
    // Late priors ride beside the set-up kernel only where the E-step can hand them to the M-step (the one-frame kernel, which takes the frame
    // descriptor by value): otherwise they are formed here, before anything is launched, and staged like ordinary ones.
    bool late_async = late != nullptr && c->fh[0].wide_tile != 0 && (c->late_on || paired);
    if (late && !late_async) {
        const double *lp = nullptr; int lk = 0;
        if ((rc = (*late)(lp, lk))) return rc;
        if ((rc = stage_priors(c, c->pin + nc.aJ, c->pin + nc.aYd, Y, M, lp, lk, p->alpha))) return rc;
        c->fh[0].has_priors = lk > 0 ? 1 : 0;
    }
    g_prof.mark(g_prof.base + 2);
    {   // prune, scan and scatter are skipped per LAUNCH: the sorted clouds are reused only when every frame of the call can reuse its own
        // (a frame that skipped its scan while the batch's scatter ran would have its cloud re-scattered from stale start offsets)
        bool all_reuse = true;
        for (int i = 0; i < F; ++i) all_reuse = all_reuse && c->fh[i].reuse_sorted;
        if (!all_reuse) for (int i = 0; i < F; ++i) c->fh[i].reuse_sorted = 0;
    }
    // One frame per call on a one-workgroup M-step (the chain smoother, the banded LLE solve): the results come back through the pinned
    // mailbox -- the M-step that finishes the registration writes them there itself -- instead of a copy and a stream synchronisation
    const bool use_mbox = !merged && c->mbox_on && p->max_iter > 0 &&
                          ((!p->include_lle && !c->fh[0].mstep_dense) || (p->include_lle && c->fh[0].lle_band));
    unsigned epoch = 0;
    if (!use_mbox) sg.release(false);
    if (use_mbox) {
        rc = ensure_mbox(c, nc.readback + 4);
        if (rc) return rc;
        if (sg.live) epoch = sg.epoch;         // (the waiting M-step reports under the epoch it was launched with)
        else { epoch = ++c->mbox_epoch; if (epoch == 0) epoch = ++c->mbox_epoch; }
        c->fh[0].host_out = c->mbox; c->fh[0].host_prog = (unsigned long long *)(c->mbox + c->mbox_doubles - 2); c->fh[0].host_epoch = epoch;
    }
    hipStream_t s = ahead ? c->stream2[0] : c->stream;      // (a registration that began on the second stream stays there: its kernels are ordered by the stream)
    const bool timing = c->timing;
    if (timing) HIPCHK(c, hipEventRecord(c->ev[0], s));
    const FrameDev *fdp;          // this call's descriptors on the device
    if (merged) {
        std::memcpy(c->pin + (size_t)F * up, c->fh.data(), sizeof(FrameDev) * F);
        HIPCHK(c, hipMemcpyAsync(c->xfer, c->pin, ((size_t)F * up + fdd) * sizeof(double), hipMemcpyHostToDevice, s));
        fdp = (const FrameDev *)(c->xfer + (size_t)F * up);
    } else {
        if (!paired) std::memcpy(c->pin + nc.fdev, c->fh.data(), sizeof(FrameDev));
        fdp = (const FrameDev *)((ahead ? c->twin.nodeblk : (paired ? c->slots[slots[0]].nodeblk2 : c->slots[slots[0]].nodeblk)) + nc.fdev);
    }
    double *const nodeblk_used = merged ? nullptr : (ahead ? c->twin.nodeblk : (paired ? c->slots[slots[0]].nodeblk2 : c->slots[slots[0]].nodeblk));
    if (c->cloud_pending >= 0) {
        // tracking_step staged this frame's cloud in pinned host memory: the fused prologue reads it from there; any other route gets a copy first
        const bool fused = !merged && !paired && c->direct_in && c->fuse_on && c->cloud_pending == slots[0] && !c->fh[0].reuse_sorted && prologue_pair_ok(c->fh[0]);
        if (fused) { c->fh[0].Xhost = c->cloud_pin; c->cloud_pending = -1; }
        else if ((rc = flush_pending_cloud(c))) return rc;
    }
    if (paired) {
        // (set up by the previous call's prologue)
    } else if (!merged && c->direct_in && (c->fuse_on || c->fh[0].reuse_sorted) && prologue_direct_ok(c->fh[0])) {
        // one small frame (or a reused sort): ONE launch reads the block from pinned host memory, puts it in its place and does the whole prologue
        Slot &sl = c->slots[slots[0]];
        const unsigned fep = c->fh[0].reuse_sorted ? 0u : next_fuse_epoch(c, sl);      // (a reused sort: the set-up workgroup alone, no barrier)
        // tracking_step asked for the set-up of its second registration to ride along (PairNext): staged like a frame of its own into the second
        // pinned block / the slot's second node block, one more workgroup of the fused prologue
        tdlo_ctx::PairNext &pn = c->pair;
        const FrameDev *f2 = nullptr;
        if (pn.state == 1) {
            pn.state = 0;
            if (!late && !c->fh[0].reuse_sorted && prologue_pair_ok(c->fh[0]) && pn.slot == slots[0] && pn.M == M && !pn.p.include_lle) {
                const NodeCarve nc2(pn.M);
                if ((rc = ensure_pin2(c, nc2.upload + 2))) return rc;
                if ((rc = prepare_frame(c, pn.slot, pn.Y.data(), pn.M, pn.sigma2, &pn.p, nullptr, 0, nullptr, pn.n_vis, nullptr, c->pin2, pn.f, true))) return rc;
                if (pn.f.wide_tile && !pn.f.mstep_dense && !pn.f.vis_branch) {
                    pn.up = upload_doubles(nc2, &pn.p, false);
                    std::memcpy(c->pin2 + nc2.fdev, &pn.f, sizeof(FrameDev));
                    f2 = &pn.f;
                    // the two registrations' first E-steps are the same computation (see FrameDev::pair_sums) when they share nodes, sigma2, mu and
                    // precision and neither weighs by visibility; the banded M-step of THIS registration then hands the sums over
                    pn.has_sums = p->include_lle && c->fh[0].lle_band && !c->fh[0].vis_branch && p->mu == pn.p.mu && p->precision == pn.p.precision &&
                                  sigma2[0] == pn.sigma2 && std::memcmp(Y, pn.Y.data(), sizeof(double) * 3 * M) == 0 && p->max_iter > 0 && c->pair_sums_on;
                    if (pn.has_sums) c->fh[0].pair_sums = sl.nodeblk2 + nc2.sums;
                }
            }
        }
        HIPCHK(c, launch_prologue_direct(c->fh.data(), c->pin, sl.nodeblk, (int)up, (int)nc.Yin, fep, s, f2, c->pin2, sl.nodeblk2, f2 ? (int)pn.up : 0));
        c->fh[0].Xhost = nullptr;          // (the cloud is in Xraw for everything that follows)
        if (f2) pn.state = 2;
    } else {
        if (!merged) HIPCHK(c, hipMemcpyAsync(c->slots[slots[0]].nodeblk, c->pin, up * sizeof(double), hipMemcpyHostToDevice, s));
        HIPCHK(c, launch_prune_and_setup(fdp, c->fh.data(), F, s));
    }
    g_prof.mark(g_prof.base + 3);
    for (int i = 0; i < F; ++i) {           // the slots' sorted clouds now belong to these nodes (recorded as soon as the prologue is enqueued: whatever fails later, the device's sort is this one)
        Slot &sl = c->slots[slots[i]];
        if (!c->fh[i].reuse_sorted) { sl.sorted_Y.assign(Y + (size_t)i * 3 * M, Y + (size_t)(i + 1) * 3 * M); sl.sorted_prec = p->precision; sl.sorted_valid = true; }
    }
    if (late_async) {
        // the set-up kernel is on its way (it copies the staging block as it is: the priors' part is overwritten by the E-step's copy of what
        // follows); now the host forms the priors and puts them where the first E-step fetches them
        const double *lp = nullptr; int lk = 0;
        rc = (*late)(lp, lk);
        if (!rc) rc = ensure_late(c, 4 * (size_t)M);
        if (!rc) rc = stage_priors(c, c->late_buf, c->late_buf + M, Y, M, lp, lk, p->alpha);
        if (rc) { sg.release(false); (void)hipStreamSynchronize(s); return rc; }        // (the set-up kernel reads the pinned staging block: drained before anybody reuses it)
        if (lk <= 0) sg.release(false);                    // (it was launched expecting priors)
        c->fh[0].has_priors = lk > 0 ? 1 : 0;
        c->fh[0].late_aJ = c->late_buf; c->fh[0].late_aYd = c->late_buf + M;
        g_prof.mark(7);
    }
    if (timing) HIPCHK(c, hipEventRecord(c->ev[1], s));
    // A batch runs as up to kBatchStreams groups of frames on as many streams, each group one E-step behind the previous
    // one: a batch's M-step is one workgroup per frame (F of the 256 CUs busy for 17 us), and meanwhile the other groups'
    // E-steps have the rest of the GPU.  The groups are independent registrations: the results do not depend on the split
    // (TDLO_BATCH_STREAMS=1 disables it).
    static const int ns_env = getenv("TDLO_BATCH_STREAMS") ? atoi(getenv("TDLO_BATCH_STREAMS")) : 0;
    // (32 frames: three groups of 11 / 11 / 10 frames -- 1078 workgroups per E-step launch, about what the GPU holds at once -- give 1.155 M it/s against
    // 1.129 M for four groups of 8, three runs each, on one box and 1.124 M against 1.118 M on another; 24 frames: four groups 0.93 M against 0.91 M; 48 frames: three groups 1.26 M against 1.18 M)
    int NS = ns_env > 0 ? ns_env : (F >= 28 ? 3 : (F >= 16 ? 4 : (F >= 8 ? 2 : 1)));
    NS = std::max(1, std::min(std::min(NS, kBatchStreams), F));
    int goff[kBatchStreams + 1];
    for (int g = 0; g <= NS; ++g) goff[g] = (int)(((long long)F * g) / NS);
    hipStream_t gs[kBatchStreams];
    gs[0] = s;
    for (int g = 1; g < NS; ++g) {            // the further streams are made when a batch first needs them: a context that only ever
        if (!c->stream2[g - 1]) HIPCHK(c, hipStreamCreateWithFlags(&c->stream2[g - 1], hipStreamNonBlocking));   // registers single frames
        gs[g] = c->stream2[g - 1];            // (or a shard) occupies one hardware queue, not four
    }
    bool forked = false;
    bool sums_first = paired && !ahead && c->pair.has_sums;      // the first iteration is its M-step alone, from the sums the previous registration's first M-step left
    bool ahead_first = ahead;                          // the first iteration is on the second stream already: its M-step waits for the priors staged above
    bool spec_released = false;                        // ... and that M-step had been launched ahead and was released by this call
    int enqueued = 0;                                  // iterations this call has put on the stream (or released)
    // test hook: the M-step launched ahead is told to leave instead of being released, as if it had given up waiting (2 s without the host)
    const bool spec_force_timeout = c->spec_force_timeout;
    // the spin-ahead loop (experiment): ONE frame, fixed iteration count, fp32 mode, the chain smoother on up to 63 nodes, no visibility term, results through the mailbox
    const bool spin_mode = c->spin_ahead_on && !merged && F == 1 && !paired && !ahead && late == nullptr && use_mbox && !timing && p->precision == TDLO_PREC_F32 &&
                           !p->include_lle && !c->fh[0].mstep_dense && !c->fh[0].vis_branch && c->fh[0].wide_tile != 0 && c->fh[0].estep2 == 0 && 4 * M + 1 <= 256 &&
                           (p->tol <= 0.0 || p->max_iter <= 2 * kChunkIters) && p->max_iter > 0;
    if (spin_mode) {
        if (!c->stream2[0]) HIPCHK(c, hipStreamCreateWithFlags(&c->stream2[0], hipStreamNonBlocking));
        HIPCHK(c, hipEventRecord(c->evx[0], s));                       // the first E-step behind the prologue; every later one behind its M-step's tag
        HIPCHK(c, hipStreamWaitEvent(c->stream2[0], c->evx[0], 0));
        ++c->spin_calls;
    }
    auto iterate = [&](int n) -> hipError_t {
        for (int it = 0; it < n; ++it) {
            ++enqueued;
            if (spin_mode) {
                Slot &sl = c->slots[slots[0]];
                TDLO_RET(launch_iteration_spin(fdp, c->fh.data(), c->stream2[0], s, enqueued == 1, &sl.spin_ecount, &sl.spin_mtag));
                continue;
            }
            if (sums_first || ahead_first) {
                const bool from_given = sums_first;
                sums_first = false; ahead_first = false;
                if (sg.live) { sg.release(!spec_force_timeout); spec_released = true; if (!ahead) ++c->route_count[2]; }      // it is on the stream already: the priors are staged, off it goes
                else if (from_given) TDLO_RET(launch_mstep_chain(fdp, c->fh.data(), 1, 1, c->fh[0].precision == TDLO_PREC_F64, s));
                else {          // (ahead, and the waiting M-step was sent away -- no priors after all --: it has cleared its E-step's sums, an ordinary iteration follows it)
                    c->fh[0].late_mstep = 0;
                    TDLO_RET(launch_iteration(fdp, c->fh.data(), 1, s));
                }
                c->fh[0].late_mstep = 0;               // (later M-steps find the priors in the node block, where the first one put them)
                continue;
            }
            for (int g = 0; g < NS; ++g) {
                const FrameDev *fdg = fdp + goff[g], *fhg = c->fh.data() + goff[g];
                const int Fg = goff[g + 1] - goff[g];
                if (NS > 1 && !forked && g + 1 < NS) {
                    // first iteration, kernel by kernel (same kernels, same order as launch_iteration), so that the next group
                    // can be released when this group's first E-step has drained
                    if (fhg[0].vis_branch) TDLO_RET(launch_estep_only(fdg, fhg, Fg, 1, gs[g]));
                    TDLO_RET(launch_estep_only(fdg, fhg, Fg, 0, gs[g]));
                    TDLO_RET(hipEventRecord(c->evx[g], gs[g]));
                    TDLO_RET(hipStreamWaitEvent(gs[g + 1], c->evx[g], 0));
                    TDLO_RET(launch_estep_only(fdg, fhg, Fg, 2, gs[g]));
                } else {
                    TDLO_RET(launch_iteration(fdg, fhg, Fg, gs[g], enqueued - 1));      // (this call's count of the registration's iterations: mstep_parity_hint)
                }
            }
            forked = true;
        }
        return hipSuccess;
    };
    auto join = [&]() -> hipError_t {          // everything enqueued on the other streams so far precedes what follows on the first
        if (NS < 2 || !forked) return hipSuccess;
        for (int g = 1; g < NS; ++g) {
            TDLO_RET(hipEventRecord(c->evj[g - 1], gs[g]));
            TDLO_RET(hipStreamWaitEvent(s, c->evj[g - 1], 0));
        }
        return hipSuccess;
    };
    static const bool pool_on = !(getenv("TDLO_BATCH_THREADS") && atoi(getenv("TDLO_BATCH_THREADS")) == 0);
    bool have_readback = false;            // the results are already in pinned memory (early exit after the first iteration; the mailbox)
    auto mbox_done = [&](int min_it, bool need_done, bool *done) -> int {     // 0, or an error code
        unsigned long long w = 0;
        int wr = mbox_wait(c, s, epoch, min_it, need_done, &w);
        if (wr == 1 && spec_released) {
            // The M-step launched ahead of its priors gave up waiting for them (this thread was held up for more than the kernel's 2 s) and left
            // without touching anything; what was enqueued behind it ran as a registration that does its own first E-step.  The state on the
            // device says where that stands: the iterations this call believes to be on the stream are made up, the ordinary way.
            spec_released = false;
            IterState is;
            HIPCHK(c, hipMemcpy(&is, c->fh[0].st, sizeof is, hipMemcpyDeviceToHost));
            int have = is.it;
            // (ahead: the M-step that left has cleared its E-step's sums -- ordinary iterations from the start)
            if (!ahead && !is.done && have == 0 && enqueued > 0) { HIPCHK(c, launch_mstep_chain(fdp, c->fh.data(), 1, 1, c->fh[0].precision == TDLO_PREC_F64, s)); have = 1; }
            for (; !is.done && have < enqueued; ++have) HIPCHK(c, launch_iteration(fdp, c->fh.data(), 1, s));
            wr = mbox_wait(c, s, epoch, min_it, need_done, &w);
        }
        if (wr < 0) return wr;
        if (wr == 1) return fail(c, TDLO_E_HIP, "the stream drained, but the M-step did not report the state of the registration");
        *done = (w >> 31) & 1u;
        return 0;
    };
    auto take_mbox = [&](bool events_recorded) -> int {
        std::memcpy(c->pin, c->mbox, nc.readback * sizeof(double));
        have_readback = true;
        if (timing) {
            if (!events_recorded) { HIPCHK(c, hipEventRecord(c->ev[2], s)); HIPCHK(c, hipEventRecord(c->ev[3], s)); }
            HIPCHK(c, hipEventSynchronize(c->ev[3]));
        }
        return 0;
    };
    // a batch whose frames all take k_estep2 and the chain smoother, fixed iteration count: the whole loop as one launch (k_batch_loop)
    bool persist = merged && c->batch_persist_on && !c->batch_persist_off_once && (p->tol <= 0.0 || p->max_iter <= 2 * kChunkIters) && p->max_iter > 0 &&
                   p->precision == TDLO_PREC_F32 && !p->include_lle && M <= 63;
    for (int i = 0; i < F && persist; ++i) persist = c->fh[i].estep2 != 0 && !c->fh[i].vis_branch && !c->fh[i].mstep_dense;
    if (persist) {
        const size_t words = batch_loop_ctl_words(F);
        if (words > c->batch_ctl_words) {
            HIPCHK(c, hipStreamSynchronize(s));
            if (c->batch_ctl) (void)hipFree(c->batch_ctl);
            c->batch_ctl = nullptr; c->batch_ctl_words = 0;
            HIPCHK(c, hipMalloc((void **)&c->batch_ctl, words * sizeof(unsigned)));
            c->batch_ctl_words = words;
        }
        HIPCHK(c, hipMemsetAsync(c->batch_ctl, 0, words * sizeof(unsigned), s));
        HIPCHK(c, launch_batch_loop(fdp, c->fh.data(), F, p->max_iter, c->batch_ctl, s));
        ++c->batch_loop_calls;
    } else
    if (p->tol <= 0.0 || p->max_iter <= 2 * kChunkIters) {
        // fixed iteration count: enqueue everything, no host involvement.  Several stream groups and enough iterations: the first
        // iteration (which releases the groups one after the other) from this thread, the rest of every group from a thread of its own.
        if (NS > 1 && pool_on && p->max_iter >= 8) {
            HIPCHK(c, iterate(1));
            if (!c->pool) { c->pool = new EnqueuePool; c->pool->start(kBatchStreams - 1, c->device); }
            const int rest = p->max_iter - 1;
            const bool chain = c->batch_chain != 0 && !c->fh[0].vis_branch;
            const bool chain_all = c->batch_chain == 1;
            auto chain_at = [&](int it) { return chain_all || it == 4 || it == 10 || it == 22 || it == 34; };      // (iterations of the enqueueing loop: the call's iteration it + 1)
            if (chain)
                for (int g = 0; g < NS; ++g) for (int r = 0; r < tdlo_ctx::kChainRing; ++r)
                    if (!c->evc[g][r]) HIPCHK(c, hipEventCreateWithFlags(&c->evc[g][r], hipEventDisableTiming));
            // (chain) how far every group's enqueueing thread has come: recorded[g] = iterations whose E-step event group g has recorded, waited[g] =
            // iterations whose event of group g-1 group g has put a wait on -- a thread records event it % ring only when the group behind it has
            // put its wait on the record of iteration it - ring, and waits on an event only when it has been recorded for this iteration
            std::atomic<int> recorded[kBatchStreams], waited[kBatchStreams];
            std::atomic<bool> broken{false};
            for (int g = 0; g < kBatchStreams; ++g) { recorded[g].store(0); waited[g].store(0); }
            const int prc = c->pool->run([&](int g) -> int {
                if (g >= NS) return 0;
                const FrameDev *fdg = fdp + goff[g], *fhg = c->fh.data() + goff[g];
                const int Fg = goff[g + 1] - goff[g];
                if (!chain) {
                    for (int it = 0; it < rest; ++it) { const hipError_t e = launch_iteration(fdg, fhg, Fg, gs[g], 1 + it); if (e != hipSuccess) return (int)e; }
                    return 0;
                }
                auto bail = [&](hipError_t e) { broken.store(true); return (int)e; };
                for (int it = 0; it < rest; ++it) {
                    hipError_t e;
                    if (!chain_at(it)) { if ((e = launch_iteration(fdg, fhg, Fg, gs[g], 1 + it)) != hipSuccess) return bail(e); continue; }
                    // (the event of a chained iteration: every iteration -> a ring with the hazard check below; the four phase-setting iterations -> one event each)
                    const int slot = chain_all ? it % tdlo_ctx::kChainRing : (it == 4 ? 0 : (it == 10 ? 1 : (it == 22 ? 2 : 3)));
                    if (g > 0) {            // behind the E-step of the group in front, same iteration
                        while (recorded[g - 1].load(std::memory_order_acquire) <= it) { if (broken.load()) return 0; std::this_thread::yield(); }
                        if ((e = hipStreamWaitEvent(gs[g], c->evc[g - 1][slot], 0)) != hipSuccess) return bail(e);
                        waited[g].store(it + 1, std::memory_order_release);
                    }
                    if ((e = launch_estep_only(fdg, fhg, Fg, 0, gs[g])) != hipSuccess) return bail(e);
                    if (g + 1 < NS) {
                        while (chain_all && waited[g + 1].load(std::memory_order_acquire) < it + 1 - tdlo_ctx::kChainRing) { if (broken.load()) return 0; std::this_thread::yield(); }
                        if ((e = hipEventRecord(c->evc[g][slot], gs[g])) != hipSuccess) return bail(e);
                        recorded[g].store(it + 1, std::memory_order_release);
                    }
                    if ((e = launch_estep_only(fdg, fhg, Fg, 2, gs[g])) != hipSuccess) return bail(e);
                }
                return 0;
            });
            if (prc) return fail(c, TDLO_E_HIP, std::string("batch enqueue: ") + hipGetErrorString((hipError_t)prc));
        } else {
            HIPCHK(c, iterate(p->max_iter));
        }
        if (use_mbox) {
            if (timing) { HIPCHK(c, hipEventRecord(c->ev[2], s)); HIPCHK(c, hipEventRecord(c->ev[3], s)); }      // behind the last iteration, as the read-back path does
            bool done = false;
            if ((rc = mbox_done(0, true, &done))) return rc;
            if ((rc = take_mbox(true))) return rc;
        }
    } else if (use_mbox) {
        // early exit (trackdlo.cpp:424-428), decided on the device and read from the mailbox: the first iteration is checked eagerly (a tracker
        // in steady state converges in it) -- or the first `iter_hint` iterations, when the caller knows how many the registration took last time
        // (tdlo_ctx::iter_hint) --; after that iterations go out in chunks of 1, 1, 2, 4, 4, ... and the host looks at the progress
        // word of the chunk BEFORE the one it has just enqueued, so that the GPU never idles (kernels of a finished registration are no-ops)
        const int first = std::max(1, std::min(std::min(iter_hint, kIterHintMax), p->max_iter));
        c->fh[0].host_report_it = first;       // (the frame descriptor travels by value with every launch of the one-frame kernels)
        HIPCHK(c, iterate(first));
        if (!late && c->pair.state == 2 && c->pair.has_sums && c->spec_on && c->mbox_on && !timing && c->pair.p.max_iter > 0) {
            // tracking_step's main registration starts with its M-step (PairNext::has_sums): launched NOW, behind this registration's first
            // iteration -- a steady-state tracker converges in it -- and ahead of the priors it needs (FrameDev::spec_flag)
            tdlo_ctx::PairNext &pn = c->pair;
            Slot &sl = c->slots[slots[0]];
            const NodeCarve nc2(pn.M);
            if ((rc = ensure_late(c, 4 * (size_t)pn.M + 2))) return rc;         // (may drain the stream: before anything waits on it)
            unsigned e2 = ++c->mbox_epoch;
            if (e2 == 0) e2 = ++c->mbox_epoch;
            FrameDev fs = pn.f;
            fs.reuse_sorted = 1; fs.has_priors = 1;
            fs.late_aJ = c->late_buf; fs.late_aYd = c->late_buf + pn.M;
            fs.host_out = c->mbox; fs.host_prog = (unsigned long long *)(c->mbox + c->mbox_doubles - 2); fs.host_epoch = e2;
            fs.host_report_it = (pn.p.tol <= 0.0 || pn.p.max_iter <= 2 * kChunkIters) ? 0 : std::max(1, std::min(std::min(c->iter_hint_next, kIterHintMax), pn.p.max_iter));
            fs.spec_flag = spec_flag_word(c); fs.spec_prev = c->fh[0].st; fs.spec_epoch = e2;
            if (c->lle_next_on && pn.M <= 256 && sl.hb_next_cap >= pn.M) fs.lle_next = sl.hb_next;      // (as the registration itself will set it, above)
            HIPCHK(c, launch_mstep_chain((const FrameDev *)(sl.nodeblk2 + nc2.fdev), &fs, 1, 1, fs.precision == TDLO_PREC_F64, s));
            pn.spec = 1; pn.spec_epoch = e2;
        } else if (!late && c->pair.state == 3) {
            // tracking_step with hidden nodes: the main registration's first iteration goes out NOW, on the second stream, beside this registration
            c->pair.state = 0; c->pair.ahead = false;
            if (c->spec_on && c->mbox_on && !timing && c->pair.slot == slots[0] && (rc = launch_ahead(c))) { spec_abort(c); return rc; }
        }
        g_prof.mark(g_prof.base + 4);
        bool stop = false;
        if ((rc = mbox_done(first, false, &stop))) { spec_abort(c); return rc; }
        if (!stop && !c->pair.ahead) c->pair.spec = 0;           // (this registration goes on: the waiting M-step has seen that and left; one launched AHEAD keeps waiting)
        g_prof.mark(g_prof.base + 5);
        int launched = first, chunk = 0;
        while (launched < p->max_iter && !stop) {
            const int n = std::min(chunk < 2 ? 1 : (chunk == 2 ? 2 : kChunkIters), p->max_iter - launched);
            c->fh[0].host_report_it = launched + n;      // the chunk's last M-step reports (a store to host memory costs an M-step ~1 us: not every one)
            HIPCHK(c, iterate(n));
            launched += n;
            if ((rc = mbox_done(launched - n, false, &stop))) return rc;
            ++chunk;
        }
        if (!stop && (rc = mbox_done(0, true, &stop))) return rc;      // the last chunk reaches max_iter, which ends the registration
        if ((rc = take_mbox(false))) return rc;
    } else {
        // early exit (trackdlo.cpp:424-428) is decided on the device; kernels of finished frames are
        // no-ops.  To avoid enqueueing up to max_iter of them, iterations go out in chunks and the
        // `done` flags of chunk i are inspected while chunk i+1 is already running (the GPU never idles).
        IterState *flags = (IterState *)(c->pin + (size_t)F * nc.upload + fdd);     // pinned, 2 x F entries (one frame) ...
        double *fl_rb = c->pin + (size_t)F * nc.upload + fdd;                        // ... or 2 x F read-back blocks (batches)
        // A tracker in steady state converges in its first iteration (launch tolerance 2e-4): the first iteration is followed by the
        // read-back block itself (results + state) and a host sync; if every frame is done the call is over -- no second chunk of
        // no-op kernels, no separate flag copies (two no-op kernels and two 4 us copies per registration: 20 us of a 90 us call).
        HIPCHK(c, iterate(1));
        HIPCHK(c, join());
        if (timing) HIPCHK(c, hipEventRecord(c->ev[2], s));
        if (merged) HIPCHK(c, hipMemcpyAsync(c->pin, c->xfer + (size_t)F * up + fdd, (size_t)F * nc.readback * sizeof(double), hipMemcpyDeviceToHost, s));
        else HIPCHK(c, hipMemcpyAsync(c->pin, nodeblk_used + nc.Yout, nc.readback * sizeof(double), hipMemcpyDeviceToHost, s));
        if (timing) HIPCHK(c, hipEventRecord(c->ev[3], s));
        HIPCHK(c, wait_stream(s));
        {
            bool all = true;
            for (int i = 0; i < F; ++i) {
                IterState is;
                std::memcpy(&is, c->pin + (size_t)i * (merged ? nc.readback : nc.upload) + (nc.st - nc.Yout), sizeof is);
                all = all && is.done != 0;
            }
            have_readback = all;
        }
        int launched = 1, chunk = 0;
        bool stop = have_readback;
        while (launched < p->max_iter && !stop) {
            const int n = std::min(chunk < 2 ? 1 : (chunk == 2 ? 2 : kChunkIters), p->max_iter - launched);      // 1, 1, 2, 4, 4, ...
            HIPCHK(c, iterate(n));
            launched += n;
            const int slotp = chunk & 1;
            if (merged) {           // the whole read-back area (Yout + IterState per frame) in one copy, after the groups have joined
                HIPCHK(c, join());
                HIPCHK(c, hipMemcpyAsync(fl_rb + (size_t)slotp * F * nc.readback, c->xfer + (size_t)F * up + fdd, (size_t)F * nc.readback * sizeof(double),
                                         hipMemcpyDeviceToHost, s));
            } else {
                HIPCHK(c, hipMemcpyAsync(&flags[slotp * F], c->fh[0].st, sizeof(IterState), hipMemcpyDeviceToHost, s));
            }
            HIPCHK(c, hipEventRecord(c->ev[4 + slotp], s));
            if (chunk > 0) {
                const int prev = (chunk - 1) & 1;
                HIPCHK(c, hipEventSynchronize(c->ev[4 + prev]));
                bool all = true;
                for (int i = 0; i < F; ++i) {
                    IterState is;
                    if (merged) std::memcpy(&is, fl_rb + ((size_t)prev * F + i) * nc.readback + (nc.st - nc.Yout), sizeof is);
                    else is = flags[prev * F + i];
                    all = all && is.done != 0;
                }
                stop = all;
            }
            ++chunk;
        }
    }
    const size_t rstride = merged ? nc.readback : nc.upload;
    if (!have_readback) {
        HIPCHK(c, join());
        if (timing) HIPCHK(c, hipEventRecord(c->ev[2], s));
        // the upload block in pinned memory is consumed by now in stream order; reuse it for the readback
        if (merged) HIPCHK(c, hipMemcpyAsync(c->pin, c->xfer + (size_t)F * up + fdd, (size_t)F * nc.readback * sizeof(double), hipMemcpyDeviceToHost, s));
        else HIPCHK(c, hipMemcpyAsync(c->pin, nodeblk_used + nc.Yout, nc.readback * sizeof(double), hipMemcpyDeviceToHost, s));
        if (timing) HIPCHK(c, hipEventRecord(c->ev[3], s));
        HIPCHK(c, wait_stream(s));
    }
    c->last_F = F;
    // The banded L D L^T takes no pivots: a system that is not positive definite in floating point (an indefinite H_override; a chain the gap
    // test let through and rounding did not) ends its registration with TDLO_E_NUMERIC.  The reference's solver is a general one
    // (trackdlo.cpp:415), so the call is repeated once on the dense pivoted kernels -- inputs, Y and sigma2 are untouched so far -- and only
    // their verdict is reported.
    if (p->include_lle && c->fh[0].lle_band && !c->lle_dense_once) {
        bool numeric = false;
        for (int i = 0; i < F; ++i) {
            IterState is;
            std::memcpy(&is, c->pin + (size_t)i * rstride + (nc.st - nc.Yout), sizeof is);
            numeric = numeric || is.status == TDLO_E_NUMERIC;
        }
        if (numeric) {
            if (c->pair.ahead) { spec_abort(c); c->pair.state = 0; c->pair.ahead = false; }      // (the repeat is another call: the main registration takes the ordinary route)
            else c->pair.spec = 0;             // (a paired M-step launched ahead has seen the error status and left)
            c->lle_dense_once = true;
            ++c->band_retries;
            const int rr = run_frames(c, F, slots, Y, M, sigma2, p, priors, K, vis, n_vis, H_override, stats);
            c->lle_dense_once = false;
            return rr;
        }
    }
    // the loop kernel gave a wait up (2 s; a frame ended with TDLO_E_EXCHANGE): the call is repeated once on the launch-per-step loop -- inputs, Y and sigma2 are untouched so far
    if (persist) {
        bool gave_up = false;
        for (int i = 0; i < F && !gave_up; ++i) {
            IterState is;
            std::memcpy(&is, c->pin + (size_t)i * rstride + (nc.st - nc.Yout), sizeof is);
            gave_up = is.status == TDLO_E_EXCHANGE;
        }
        if (gave_up) {
            ++c->batch_loop_fallbacks;
            std::fprintf(stderr, "trackdlo_hip: a wait inside the batch loop kernel gave up after 2 s (device %d); the call is repeated with a launch per step\n", c->device);
            for (int i = 0; i < F; ++i) c->slots[slots[i]].sorted_valid = false;
            c->batch_persist_off_once = true;
            const int rr = run_frames(c, F, slots, Y, M, sigma2, p, priors, K, vis, n_vis, H_override, stats, late);
            c->batch_persist_off_once = false;
            return rr;
        }
    }
    // fp64 mode: the E-step's range check runs against limits that follow sigma (IterState::sh_boost: finer sums while sigma is small).  The extent
    // behind them is a heuristic (D_eff = 2 (0.4 m + 2 sigma)): a registration whose shares exceed it -- nodes dragged far by priors while sigma is
    // small -- would have passed under the coarse limits.  It is not failed for the boost: the call is repeated ONCE without it (inputs, Y, sigma2
    // untouched so far), and only that verdict is reported.
    if (p->precision == TDLO_PREC_F64 && !c->boost_off_once && p->max_iter > 0) {
        bool numeric = c->test_boost_fail;
        for (int i = 0; i < F && !numeric; ++i) {
            IterState is;
            std::memcpy(&is, c->pin + (size_t)i * rstride + (nc.st - nc.Yout), sizeof is);
            numeric = is.status == TDLO_E_NUMERIC;
        }
        if (numeric) {
            if (c->pair.ahead) { spec_abort(c); c->pair.state = 0; c->pair.ahead = false; }
            else c->pair.spec = 0;
            c->boost_off_once = true;
            ++c->boost_retries;
            for (int i = 0; i < F; ++i) c->slots[slots[i]].sorted_valid = false;
            const int rr = run_frames(c, F, slots, Y, M, sigma2, p, priors, K, vis, n_vis, H_override, stats, late);      // (late priors: formed again from the same guide nodes)
            c->boost_off_once = false;
            return rr;
        }
    }
    float loop_ms = 0, total_ms = 0;
    if (timing) { hipEventElapsedTime(&loop_ms, c->ev[1], c->ev[2]); hipEventElapsedTime(&total_ms, c->ev[0], c->ev[3]); }
    const double host_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_host0).count();
    int worst = 0;
    for (int i = 0; i < F; ++i) {
        const double *rb = c->pin + (size_t)i * rstride;
        IterState is;
        std::memcpy(&is, rb + (nc.st - nc.Yout), sizeof is);
        if (is.status == 0 || is.status == TDLO_E_NUMERIC) {
            if (p->max_iter > 0 && is.it > 0) std::memcpy(Y + (size_t)i * 3 * M, rb, sizeof(double) * 3 * M);
            sigma2[i] = is.sigma2;
        }
        if (lle_next && i == 0 && is.status == 0 && is.done != 0 && is.it > 0) {      // (the finishing M-step has formed H of exactly these nodes)
            Slot &sl = c->slots[slots[0]];
            sl.hb_next_Y.assign(Y, Y + 3 * (size_t)M);
            sl.hb_next_valid = true;
        }
        if (p->max_iter == 0) { is.converged = 1; }
        if (stats) {
            fill_stats(&stats[i], is);
            stats[i].loop_ms = loop_ms; stats[i].total_ms = total_ms; stats[i].host_ms = host_ms;
            stats[i].sort_reused = ahead ? 0 : (paired ? 2 : c->fh[i].reuse_sorted); stats[i].band_retry = c->lle_dense_once ? 1 : 0;      // (ahead: its own prune and sort, in the twin slot)
        }
        if (is.status != 0 && worst == 0) worst = is.status;
    }
    g_prof.mark(g_prof.base + 6);
    {   // the measurement and debug entry points relaunch from these descriptors: nothing of this call's hand-overs may ride along (a stale mailbox
        // epoch, late priors copied again at it == 0, another registration's sums overwritten through pair_sums: ADVICE r04)
        FrameDev &f0 = c->fh[0];
        f0.lle_next = nullptr; f0.spec_flag = nullptr; f0.spec_prev = nullptr;
        f0.host_prog = nullptr; f0.host_out = nullptr; f0.host_epoch = 0; f0.host_report_it = 0;
        f0.late_aJ = nullptr; f0.late_aYd = nullptr; f0.late_mstep = 0; f0.pair_sums = nullptr; f0.Xhost = nullptr;
    }
    if (worst != 0 && !late && !c->pair.ahead) c->pair.spec = 0;      // (ahead: tracking_step tells the waiting M-step to leave)
    if (ahead) c->twin_busy = false;       // (what may still be on the second stream are no-op iterations on the twin slot's device buffers)
    if (worst == TDLO_E_FUSE) return TDLO_E_FUSE;      // the fused prologue's barrier was abandoned: the entry point repeats the call on the three-kernel route (fuse_fallback)
    if (worst == TDLO_E_EMPTY) return fail(c, worst, "every point was pruned (no point within 0.1 m of a node, trackdlo.cpp:190)");
    if (worst == TDLO_E_NUMERIC) return fail(c, worst, "non-finite or non-positive sigma2, or singular M-step system");
    return TDLO_OK;
}

}  // namespace

// =================================================================================================
extern "C" {

int tdlo_abi_version(void) { return TDLO_ABI_VERSION; }

int tdlo_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

void tdlo_default_config(tdlo_config *cfg) {
    if (!cfg) return;
    cfg->device = 0; cfg->max_frames = 1; cfg->max_points = 65536; cfg->max_nodes = 64; cfg->estep_blocks = 0;
}

tdlo_ctx *tdlo_create(const tdlo_config *cfg_in, int *err) {
    tdlo_config cfg;
    if (cfg_in) cfg = *cfg_in; else tdlo_default_config(&cfg);
    auto bail = [&](int code) -> tdlo_ctx * { if (err) *err = code; return nullptr; };
    int n = 0;
    hipError_t he = hipGetDeviceCount(&n);
    if (he != hipSuccess || n <= 0 || cfg.device < 0 || cfg.device >= n) {
        fprintf(stderr, "trackdlo_hip: hipGetDeviceCount -> %s, %d device(s), requested %d\n", hipGetErrorString(he), n, cfg.device);
        return bail(TDLO_E_NO_DEVICE);
    }
    he = hipSetDevice(cfg.device);
    if (he != hipSuccess) { fprintf(stderr, "trackdlo_hip: hipSetDevice -> %s\n", hipGetErrorString(he)); return bail(TDLO_E_NO_DEVICE); }
    if (check_device_image() != 0) return bail(TDLO_E_NO_DEVICE);     // no gfx950 code object for this device
    if (cfg.max_frames < 1) cfg.max_frames = 1;
    tdlo_ctx *c = new tdlo_ctx();
    c->device = cfg.device; c->cfg = cfg;
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return bail(TDLO_E_HIP); }
    for (auto &e : c->ev) if (hipEventCreate(&e) != hipSuccess) { delete c; return bail(TDLO_E_HIP); }
    for (auto &e : c->evx) if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { delete c; return bail(TDLO_E_HIP); }
    for (auto &e : c->evj) if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { delete c; return bail(TDLO_E_HIP); }
    c->slots.resize(cfg.max_frames);
    if (hipMalloc((void **)&c->fd, sizeof(FrameDev) * cfg.max_frames) != hipSuccess) { delete c; return bail(TDLO_E_HIP); }
    for (auto &s : c->slots) {
        if (ensure_points(c, s, std::max(cfg.max_points, 1024)) || ensure_nodes(c, s, std::max(cfg.max_nodes, 16))) {
            if (err) *err = TDLO_E_HIP;
            tdlo_destroy(c);
            return nullptr;
        }
    }
    if (err) *err = TDLO_OK;
    return c;
}

void tdlo_destroy(tdlo_ctx *c) {
    if (!c) return;
    hipSetDevice(c->device);
    if (c->stream) hipStreamSynchronize(c->stream);
    for (auto &q : c->stream2) if (q) hipStreamSynchronize(q);
    c->slots.push_back(c->twin);          // (freed with the others)
    for (auto &s : c->slots) {
        if (s.Xraw) hipFree(s.Xraw);
        if (s.Xs) hipFree(s.Xs);
        if (s.bucket) hipFree(s.bucket);
        if (s.blksum) hipFree(s.blksum);
        if (s.hist) hipFree(s.hist);
        if (s.nodeblk) hipFree(s.nodeblk);
        if (s.nodeblk2) hipFree(s.nodeblk2);
        if (s.hb_next) hipFree(s.hb_next);
        if (s.sync) hipFree(s.sync);
    }
    delete c->pool; c->pool = nullptr;
    if (c->fd) hipFree(c->fd);
    if (c->cloud_ws) hipFree(c->cloud_ws);
    if (c->cloud_fws) hipFree(c->cloud_fws);
    if (c->vis_state) hipFree(c->vis_state);
    if (c->vis_res) hipHostFree(c->vis_res);
    if (c->vis_nodes_pin) hipHostFree(c->vis_nodes_pin);
    if (c->cloud_res) hipHostFree(c->cloud_res);
    if (c->img_pin) hipHostFree(c->img_pin);
    if (c->reg_ws) hipFree(c->reg_ws);
    if (c->pin) hipHostFree(c->pin);
    if (c->pin2) hipHostFree(c->pin2);
    if (c->cloud_pin) hipHostFree(c->cloud_pin);
    if (c->mbox) hipHostFree(c->mbox);
    if (c->late_buf) hipHostFree(c->late_buf);
    for (auto &e : c->ev) if (e) hipEventDestroy(e);
    if (c->xfer) hipFree(c->xfer);
    if (c->split_buf) hipFree(c->split_buf);
    for (void *p : c->xch_opened) hipIpcCloseMemHandle(p);
    if (c->xch_own) hipFree(c->xch_own);
    if (c->own_comm) { const RcclApi *r = rccl_api(nullptr, nullptr); if (r) r->CommDestroy(c->own_comm); }
    for (auto &e : c->evx) if (e) hipEventDestroy(e);
    for (auto &e : c->evj) if (e) hipEventDestroy(e);
    for (auto &row : c->evc) for (auto &e : row) if (e) hipEventDestroy(e);
    if (c->batch_ctl) (void)hipFree(c->batch_ctl);
    for (auto &q : c->stream2) if (q) hipStreamDestroy(q);
    if (c->stream) hipStreamDestroy(c->stream);
    delete c;
}

const char *tdlo_last_error(const tdlo_ctx *c) { return c ? c->err.c_str() : "no context (no usable HIP device?)"; }
void *tdlo_stream(tdlo_ctx *c) { return c ? (void *)c->stream : nullptr; }

int tdlo_synchronize(tdlo_ctx *c) {
    if (!c) return TDLO_E_INVALID;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return TDLO_OK;
}

// sync == false: the copy is only enqueued -- for callers that keep X alive until their next synchronisation with the stream
// (tracking_step: the first registration's read-back waits for everything before it)
static int set_cloud_impl(tdlo_ctx *c, int slot, const double *X, int N, bool sync) {
    if (!c) return TDLO_E_INVALID;
    if (slot < 0 || slot >= (int)c->slots.size()) return fail(c, TDLO_E_INVALID, "bad slot");
    if (!X || N <= 0) return fail(c, TDLO_E_INVALID, "empty cloud");
    HIPCHK(c, hipSetDevice(c->device));
    Slot &s = c->slots[slot];
    int rc = ensure_points(c, s, N);
    if (rc) return rc;
    HIPCHK(c, hipMemcpyAsync(s.Xraw, X, 3 * (size_t)N * sizeof(double), hipMemcpyHostToDevice, c->stream));
    if (sync) HIPCHK(c, wait_stream(c->stream));      // caller may free X on return (by-value semantics)
    s.N0 = N; s.sorted_valid = false;
    return TDLO_OK;
}

int tdlo_set_cloud(tdlo_ctx *c, int slot, const double *X, int N) { return set_cloud_impl(c, slot, X, N, true); }

// run_frames for the entry points that register caller-supplied nodes (tdlo_cpd_lle_resident, tdlo_cpd_lle_batch): a registration whose fused
// prologue was abandoned at its grid barrier (TDLO_E_FUSE: internal, no caller ever sees the code) has touched neither Y, sigma2 nor the slots'
// clouds -- the context is taken off the fused prologue (fuse_fallback: arrivals and barrier word reset, fuse_on = false) and the call repeated
// ONCE on the copy + three-kernel route.  (tracking_step's main registration does the same around its own call: it forms its priors again.)
static int run_frames_checked(tdlo_ctx *c, int F, const int *slots, double *Y, int M, double *sigma2, const tdlo_params *p,
                              const double *priors, int K, const int *vis, int n_vis, const double *H_override, tdlo_stats *stats) {
    int rc = run_frames(c, F, slots, Y, M, sigma2, p, priors, K, vis, n_vis, H_override, stats);
    if (rc == TDLO_E_FUSE) {
        fuse_fallback(c);
        rc = run_frames(c, F, slots, Y, M, sigma2, p, priors, K, vis, n_vis, H_override, stats);
        if (rc == TDLO_E_FUSE) rc = fail(c, TDLO_E_HIP, "the fused prologue reported a time-out on the route that does not use it");
    }
    return rc;
}

int tdlo_cpd_lle_resident(tdlo_ctx *c, int slot, double *Y, int M, double *sigma2, const tdlo_params *p,
                          const double *priors, int K, const int *vis, int n_vis, const double *H_override,
                          tdlo_stats *stats) {
    if (!c) return TDLO_E_INVALID;
    if (slot < 0 || slot >= (int)c->slots.size()) return fail(c, TDLO_E_INVALID, "bad slot");
    if (!Y || !sigma2) return fail(c, TDLO_E_INVALID, "null Y / sigma2");
    HIPCHK(c, hipSetDevice(c->device));
    return run_frames_checked(c, 1, &slot, Y, M, sigma2, p, priors, K, vis, n_vis, H_override, stats);
}

int tdlo_cpd_lle(tdlo_ctx *c, const double *X, int N, double *Y, int M, double *sigma2, const tdlo_params *p,
                 const double *priors, int K, const int *vis, int n_vis, const double *H_override, tdlo_stats *stats) {
    int rc = tdlo_set_cloud(c, 0, X, N);
    if (rc) return rc;
    return tdlo_cpd_lle_resident(c, 0, Y, M, sigma2, p, priors, K, vis, n_vis, H_override, stats);
}

int tdlo_cpd_lle_batch(tdlo_ctx *c, int F, double *Y, int M, double *sigma2, const tdlo_params *p,
                       const double *priors, int K, const int *vis, int n_vis, const double *H_override,
                       tdlo_stats *stats) {
    if (!c) return TDLO_E_INVALID;
    if (F < 1 || F > (int)c->slots.size()) return fail(c, TDLO_E_INVALID, "bad frame count");
    if (!Y || !sigma2) return fail(c, TDLO_E_INVALID, "null Y / sigma2");
    HIPCHK(c, hipSetDevice(c->device));
    std::vector<int> slots(F);
    for (int i = 0; i < F; ++i) slots[i] = i;
    return run_frames_checked(c, F, slots.data(), Y, M, sigma2, p, priors, K, vis, n_vis, H_override, stats);      // (F == 1 takes the fused prologue like a single call)
}

// ---- N-split -------------------------------------------------------------------------------------
int tdlo_split_begin(tdlo_ctx *c, const double *Y, int M, double sigma2, const tdlo_params *p,
                     const double *priors, int K, const int *vis, int n_vis, const double *H_override, double *init) {
    if (!c) return TDLO_E_INVALID;
    int rc = check_params(c, M, p);
    if (rc) return rc;
    HIPCHK(c, hipSetDevice(c->device));
    NodeCarve nc(M);
    rc = ensure_pin(c, std::max(nc.upload, nc.readback + 2) + 8 * (size_t)M + 16);
    if (rc) return rc;
    c->fh.assign(1, FrameDev{});
    rc = prepare_frame(c, 0, Y, M, sigma2, p, priors, K, vis, n_vis, H_override, c->pin, c->fh[0]);
    if (rc) return rc;
    c->fh[0].reuse_sorted = 0; c->slots[0].sorted_valid = false;      // a shard is pruned and sorted by the split's own setup
    if (c->xch_sums) c->fh[0].sums = c->xch_sums;       // the reduced sums are exported to / consumed from the caller's buffer
    hipStream_t s = c->stream;
    HIPCHK(c, hipMemcpyAsync(c->slots[0].nodeblk, c->pin, upload_doubles(nc, p, c->fh[0].lle_band != 0) * sizeof(double), hipMemcpyHostToDevice, s));
    HIPCHK(c, hipMemcpyAsync(c->fd, c->fh.data(), sizeof(FrameDev), hipMemcpyHostToDevice, s));
    HIPCHK(c, launch_split_setup(c->fd, c->fh.data(), s));
    HIPCHK(c, hipMemcpyAsync(c->pin, c->slots[0].nodeblk + nc.st, sizeof(IterState), hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    IterState is;
    std::memcpy(&is, c->pin, sizeof is);
    if (init) { init[0] = (double)is.N; init[1] = is.sum_d2; }
    c->split_active = 1;
    c->last_F = 1;
    return TDLO_OK;
}

int tdlo_split_set_global(tdlo_ctx *c, double n_kept_global, double sum_d2_global) {
    if (!c || !c->split_active) return TDLO_E_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, launch_split_set_global(c->fd, n_kept_global, sum_d2_global, c->stream));
    return TDLO_OK;
}

int tdlo_split_dmin(tdlo_ctx *c, double *dmin_sq) {
    if (!c || !c->split_active) return TDLO_E_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    const FrameDev &f = c->fh[0];
    const int M = f.M;
    if (!f.vis_branch) { for (int m = 0; m < M; ++m) dmin_sq[m] = 0; return TDLO_OK; }
    hipStream_t s = c->stream;
    HIPCHK(c, launch_estep_only(c->fd, c->fh.data(), 1, 1, s));
    HIPCHK(c, hipMemcpyAsync(c->pin, f.dminbits, sizeof(double) * M, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    const unsigned long long *b = (const unsigned long long *)c->pin;
    for (int m = 0; m < M; ++m) {
        if (b[m] == ~0ull) { dmin_sq[m] = 1e300; continue; }
        if (f.precision == TDLO_PREC_F64) { double v; std::memcpy(&v, &b[m], 8); dmin_sq[m] = v; }
        else { unsigned u = (unsigned)b[m]; float v; std::memcpy(&v, &u, 4); dmin_sq[m] = (double)v; }
    }
    return TDLO_OK;
}

int tdlo_split_estep(tdlo_ctx *c, const double *dmin_sq_global, double *sums) {
    if (!c || !c->split_active) return TDLO_E_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    const FrameDev &f = c->fh[0];
    const int M = f.M;
    hipStream_t s = c->stream;
    if (f.vis_branch && dmin_sq_global) {
        unsigned long long *b = (unsigned long long *)c->pin;
        for (int m = 0; m < M; ++m) {
            if (f.precision == TDLO_PREC_F64) { double v = dmin_sq_global[m]; std::memcpy(&b[m], &v, 8); }
            else { float v = (float)dmin_sq_global[m]; unsigned u; std::memcpy(&u, &v, 4); b[m] = u; }
        }
        HIPCHK(c, hipMemcpyAsync(f.dminbits, c->pin, sizeof(double) * M, hipMemcpyHostToDevice, s));
    }
    HIPCHK(c, launch_estep_only(c->fd, c->fh.data(), 1, 0, s));
    HIPCHK(c, launch_estep_only(c->fd, c->fh.data(), 1, 3, s));        // the shard's accumulators -> sums
    HIPCHK(c, hipMemcpyAsync(c->pin + M, f.sums, sizeof(double) * (4 * M + 2), hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    std::memcpy(sums, c->pin + M, sizeof(double) * (4 * M + 2));
    return TDLO_OK;
}

int tdlo_split_mstep(tdlo_ctx *c, const double *sums_global, int *done) {
    if (!c || !c->split_active) return TDLO_E_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    const FrameDev &f = c->fh[0];
    const int M = f.M;
    hipStream_t s = c->stream;
    std::memcpy(c->pin, sums_global, sizeof(double) * (4 * M + 2));
    HIPCHK(c, hipMemcpyAsync(f.sums, c->pin, sizeof(double) * (4 * M + 2), hipMemcpyHostToDevice, s));
    HIPCHK(c, launch_estep_only(c->fd, c->fh.data(), 1, 4, s));
    HIPCHK(c, hipMemcpyAsync(c->pin + 4 * M + 4, f.st, sizeof(IterState), hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    IterState is;
    std::memcpy(&is, c->pin + 4 * M + 4, sizeof is);
    if (done) *done = is.done;
    return TDLO_OK;
}

// Device-resident exchange: nothing below synchronises with the host except tdlo_split_poll.
int tdlo_split_bind_exchange(tdlo_ctx *c, double *d_dmin, double *d_sums) {
    if (!c) return TDLO_E_INVALID;
    if (c->split_active) return fail(c, TDLO_E_INVALID, "tdlo_split_bind_exchange inside a split registration");
    if ((d_dmin == nullptr) != (d_sums == nullptr)) return fail(c, TDLO_E_INVALID, "bind both exchange buffers or neither");
    c->xch_dmin = d_dmin; c->xch_sums = d_sums;
    return TDLO_OK;
}

int tdlo_split_dmin_enqueue(tdlo_ctx *c) {
    if (!c || !c->split_active) return TDLO_E_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    if (!c->xch_dmin) return fail(c, TDLO_E_INVALID, "no exchange buffers bound");
    if (!c->fh[0].vis_branch) return TDLO_OK;
    HIPCHK(c, launch_estep_only(c->fd, c->fh.data(), 1, 1, c->stream));
    HIPCHK(c, launch_split_dmin_xch(c->fd, c->fh.data(), c->xch_dmin, 0, c->stream));
    return TDLO_OK;
}

int tdlo_split_estep_enqueue(tdlo_ctx *c) {
    if (!c || !c->split_active) return TDLO_E_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    if (!c->xch_sums) return fail(c, TDLO_E_INVALID, "no exchange buffers bound");
    hipStream_t s = c->stream;
    if (c->fh[0].vis_branch) HIPCHK(c, launch_split_dmin_xch(c->fd, c->fh.data(), c->xch_dmin, 1, s));
    HIPCHK(c, launch_estep_only(c->fd, c->fh.data(), 1, 0, s));
    HIPCHK(c, launch_estep_only(c->fd, c->fh.data(), 1, 3, s));        // the shard's accumulators -> the bound sums buffer
    return TDLO_OK;
}

int tdlo_split_mstep_enqueue(tdlo_ctx *c) {
    if (!c || !c->split_active) return TDLO_E_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    if (!c->xch_sums) return fail(c, TDLO_E_INVALID, "no exchange buffers bound");
    HIPCHK(c, launch_estep_only(c->fd, c->fh.data(), 1, 4, c->stream));
    return TDLO_OK;
}

int tdlo_split_poll(tdlo_ctx *c, int *done, int *iters) {
    if (!c || !c->split_active) return TDLO_E_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t s = c->stream;
    HIPCHK(c, hipMemcpyAsync(c->pin, c->fh[0].st, sizeof(IterState), hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    IterState is;
    std::memcpy(&is, c->pin, sizeof is);
    if (done) *done = is.done;
    if (iters) *iters = is.it;
    return TDLO_OK;
}

int tdlo_split_end(tdlo_ctx *c, double *Y, double *sigma2, tdlo_stats *stats) {
    if (!c || !c->split_active) return TDLO_E_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    const FrameDev &f = c->fh[0];
    const int M = f.M;
    NodeCarve nc(M);
    hipStream_t s = c->stream;
    HIPCHK(c, hipMemcpyAsync(c->pin, c->slots[0].nodeblk + nc.Yout, nc.readback * sizeof(double), hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    IterState is;
    std::memcpy(&is, c->pin + (nc.st - nc.Yout), sizeof is);
    if (Y && is.it > 0) std::memcpy(Y, c->pin, sizeof(double) * 3 * M);
    if (sigma2) *sigma2 = is.sigma2;
    if (stats) { std::memset(stats, 0, sizeof *stats); fill_stats(stats, is); }
    c->split_active = 0;
    return is.status;
}

int tdlo_split_abort(tdlo_ctx *c) {
    if (!c) return TDLO_E_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->split_active = 0;
    c->xch_dmin = nullptr; c->xch_sums = nullptr;
    return TDLO_OK;
}

// ---- tdlo_split_run: the split registration driven from C++ ---------------------------------------------------------
int tdlo_rccl_load(const char *path) {
    std::string why;
    return rccl_api(path, &why) ? TDLO_OK : TDLO_E_EXCHANGE;
}

int tdlo_rccl_unique_id(void *id128) {
    std::string why;
    const RcclApi *r = rccl_api(nullptr, &why);
    if (!r || !id128) return TDLO_E_EXCHANGE;
    RcclApi::UniqueId id;
    if (r->GetUniqueId(&id) != 0) return TDLO_E_EXCHANGE;
    std::memcpy(id128, &id, sizeof id);
    return TDLO_OK;
}

int tdlo_rccl_comm_init(tdlo_ctx *c, int nranks, int rank, const void *id128, void **comm_out) {
    if (!c || !id128 || nranks < 1 || rank < 0 || rank >= nranks) return TDLO_E_INVALID;
    std::string why;
    const RcclApi *r = rccl_api(nullptr, &why);
    if (!r) return fail(c, TDLO_E_EXCHANGE, "RCCL cannot be loaded: " + why);
    HIPCHK(c, hipSetDevice(c->device));
    if (c->own_comm) { r->CommDestroy(c->own_comm); c->own_comm = nullptr; }
    RcclApi::UniqueId id;
    std::memcpy(&id, id128, sizeof id);
    void *comm = nullptr;
    const int rc = r->CommInitRank(&comm, nranks, id, rank);
    if (rc != 0) return fail(c, TDLO_E_EXCHANGE, std::string("ncclCommInitRank: ") + r->GetErrorString(rc));
    c->own_comm = comm;
    if (comm_out) *comm_out = comm;
    return TDLO_OK;
}

int tdlo_rccl_comm_count(void *comm, int *nranks, int *rank) {
    std::string why;
    const RcclApi *r = rccl_api(nullptr, &why);
    if (!r || !comm || !r->CommCount || !r->CommUserRank) return TDLO_E_EXCHANGE;
    int n = 0, me = 0;
    if (r->CommCount(comm, &n) != 0 || r->CommUserRank(comm, &me) != 0) return TDLO_E_EXCHANGE;
    if (nranks) *nranks = n;
    if (rank) *rank = me;
    return TDLO_OK;
}

size_t tdlo_xch_bytes(int nranks, int max_nodes) {
    if (nranks < 1 || nranks > kMaxXchRanks || max_nodes < 4) return 0;
    return xch_words(nranks, max_nodes) * sizeof(unsigned long long);
}

int tdlo_xch_create(tdlo_ctx *c, int nranks, int max_nodes, void **inbox) {
    if (!c) return TDLO_E_INVALID;
    if (nranks < 1 || nranks > kMaxXchRanks || max_nodes < 4 || max_nodes > kChainLdsMaxNodes) return fail(c, TDLO_E_INVALID, "tdlo_xch_create: 1..8 ranks, 4..512 nodes (longer chains: tdlo_split_run with an RCCL communicator)");
    if (c->split_active) return fail(c, TDLO_E_INVALID, "tdlo_xch_create inside a split registration");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->xch_own) { hipFree(c->xch_own); c->xch_own = nullptr; }
    c->xch_nranks = 0;
    const size_t words = xch_words(nranks, max_nodes);
    // peers write this buffer over xGMI while this GPU polls it: it must be uncached (fine-grained) device memory -- a peer's stores to
    // coarse-grained memory are not guaranteed to become visible to a kernel that polls through its local L2, and every wait would then
    // run into its time limit.  Where the runtime does not offer it the exchange is refused (TDLO_E_EXCHANGE) so that callers take the
    // RCCL form instead; a single rank has no peers and may use any memory.
    void *p = nullptr;
    if (hipExtMallocWithFlags(&p, words * sizeof(unsigned long long), hipDeviceMallocUncached) != hipSuccess) {
        (void)hipGetLastError();
        if (nranks > 1) return fail(c, TDLO_E_EXCHANGE, "tdlo_xch_create: no fine-grained (uncached) device memory for the inbox: use tdlo_split_run with an RCCL communicator");
        HIPCHK(c, hipMalloc(&p, words * sizeof(unsigned long long)));
    }
    HIPCHK(c, hipMemset(p, 0, words * sizeof(unsigned long long)));
    HIPCHK(c, hipDeviceSynchronize());
    c->xch_own = (unsigned long long *)p; c->xch_own_words = words; c->xch_mcap = max_nodes; c->xch_cap_ranks = nranks;
    if (inbox) *inbox = p;
    return TDLO_OK;
}

int tdlo_xch_ipc_export(tdlo_ctx *c, void *handle64) {
    if (!c || !handle64) return TDLO_E_INVALID;
    if (!c->xch_own) return fail(c, TDLO_E_INVALID, "tdlo_xch_ipc_export before tdlo_xch_create");
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "hipIpcMemHandle_t");
    HIPCHK(c, hipSetDevice(c->device));
    hipIpcMemHandle_t h;
    HIPCHK(c, hipIpcGetMemHandle(&h, c->xch_own));
    std::memcpy(handle64, &h, sizeof h);
    return TDLO_OK;
}

int tdlo_xch_ipc_open(tdlo_ctx *c, const void *handle64, void **peer_inbox) {
    if (!c || !handle64 || !peer_inbox) return TDLO_E_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    hipIpcMemHandle_t h;
    std::memcpy(&h, handle64, sizeof h);
    void *p = nullptr;
    {   // a GPU that cannot map the peer's memory is not a HIP failure of this library but a property of the node: TDLO_E_EXCHANGE, so that
        // every rank can fall back on the RCCL form together (bench.py: MIN over the ranks of "the exchange is set up")
        const hipError_t e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) { (void)hipGetLastError(); return fail(c, TDLO_E_EXCHANGE, std::string("tdlo_xch_ipc_open: the peer's inbox cannot be mapped on this GPU (") + hipGetErrorString(e) + ")"); }
    }
    c->xch_opened.push_back(p);
    *peer_inbox = p;
    return TDLO_OK;
}

int tdlo_xch_can_access(tdlo_ctx *c, int peer_device, int *can) {
    if (!c || !can) return TDLO_E_INVALID;
    *can = 0;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || peer_device < 0 || peer_device >= n) return fail(c, TDLO_E_INVALID, "tdlo_xch_can_access: no such device");
    if (peer_device == c->device) { *can = 1; return TDLO_OK; }
    int ok = 0;
    HIPCHK(c, hipDeviceCanAccessPeer(&ok, c->device, peer_device));
    *can = ok;
    return TDLO_OK;
}

int tdlo_xch_bind(tdlo_ctx *c, int rank, int nranks, void *const *inboxes) {
    if (!c) return TDLO_E_INVALID;
    if (c->split_active) return fail(c, TDLO_E_INVALID, "tdlo_xch_bind inside a split registration");
    if (nranks == 0) { c->xch_nranks = 0; return TDLO_OK; }
    if (!c->xch_own || !inboxes || nranks != c->xch_cap_ranks || rank < 0 || rank >= nranks) return fail(c, TDLO_E_INVALID, "tdlo_xch_bind: create the inbox for this many ranks first");
    for (int r = 0; r < nranks; ++r) if (!inboxes[r]) return fail(c, TDLO_E_INVALID, "tdlo_xch_bind: null peer inbox");
    if (inboxes[rank] != (void *)c->xch_own) return fail(c, TDLO_E_INVALID, "tdlo_xch_bind: inboxes[rank] is not this context's own inbox");
    // inboxes of contexts of THIS process on other devices (one thread per GPU): plain device pointers -- this GPU must be able to map them
    HIPCHK(c, hipSetDevice(c->device));
    for (int r = 0; r < nranks; ++r) {
        if (r == rank) continue;
        hipPointerAttribute_t at{};
        if (hipPointerGetAttributes(&at, inboxes[r]) != hipSuccess) { (void)hipGetLastError(); continue; }      // (an IPC mapping: opened for this device already)
        if (at.device == c->device) continue;
        int ok = 0;
        HIPCHK(c, hipDeviceCanAccessPeer(&ok, c->device, at.device));
        if (!ok) return fail(c, TDLO_E_EXCHANGE, "tdlo_xch_bind: this GPU cannot access a peer's inbox (no peer mapping between the devices): use the RCCL form");
        const hipError_t e = hipDeviceEnablePeerAccess(at.device, 0);
        if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) return fail(c, TDLO_E_EXCHANGE, std::string("hipDeviceEnablePeerAccess: ") + hipGetErrorString(e));
        (void)hipGetLastError();
    }
    for (int r = 0; r < kMaxXchRanks; ++r) c->xch_peer[r] = r < nranks ? (unsigned long long *)inboxes[r] : nullptr;
    c->xch_rank = rank; c->xch_nranks = nranks;
    return TDLO_OK;
}

int tdlo_split_run(tdlo_ctx *c, void *nccl_comm, double *Y, int M, double *sigma2, const tdlo_params *p, const double *priors, int K,
                   const int *vis, int n_vis, const double *H_override, tdlo_stats *stats) {
    if (!c) return TDLO_E_INVALID;
    const auto t_host0 = std::chrono::steady_clock::now();
    if (!Y || !sigma2) return fail(c, TDLO_E_INVALID, "null Y / sigma2");
    if (c->split_active) return fail(c, TDLO_E_INVALID, "tdlo_split_run inside a split registration");
    int rc = check_params(c, M, p);
    if (rc) return rc;
    const bool oneshot = nccl_comm == nullptr;
    const RcclApi *R = nullptr;
    if (oneshot) {
        if (c->xch_nranks < 1) return fail(c, TDLO_E_INVALID, "tdlo_split_run without a communicator needs the one-shot exchange (tdlo_xch_create / tdlo_xch_bind)");
        if (M > c->xch_mcap) return fail(c, TDLO_E_INVALID, "more nodes than the inbox was created for");
    } else {
        std::string why;
        R = rccl_api(nullptr, &why);
        if (!R) return fail(c, TDLO_E_EXCHANGE, "RCCL cannot be loaded: " + why);
    }
    HIPCHK(c, hipSetDevice(c->device));
    NodeCarve nc(M);
    rc = ensure_pin(c, std::max(nc.upload, nc.readback + 2) + 8 * (size_t)M + 16);
    if (rc) return rc;
    c->fh.assign(1, FrameDev{});
    rc = prepare_frame(c, 0, Y, M, *sigma2, p, priors, K, vis, n_vis, H_override, c->pin, c->fh[0]);
    if (rc) return rc;
    c->fh[0].reuse_sorted = 0; c->slots[0].sorted_valid = false;      // a shard is pruned and sorted by the split's own setup
    FrameDev &f = c->fh[0];
    // the exchange lives in the one-workgroup M-steps: the chain smoother (no LLE term), the banded L D L^T (LLE term) -- any chain
    // length -- and the dense k_mstep_fast (up to 60 / 64 nodes)
    if (oneshot && M > kChainLdsMaxNodes)
        return fail(c, TDLO_E_INVALID, "the one-shot exchange serves chains of up to 512 nodes (the long-chain M-step does not carry it): pass an RCCL communicator");
    if (oneshot && !((!p->include_lle && (!f.mstep_dense || M <= 60)) || (p->include_lle && (f.lle_band || M <= 64))))
        return fail(c, TDLO_E_INVALID, "the one-shot exchange lives in the one-workgroup M-steps: this registration takes a dense multi-workgroup elimination (LLE term on a chain the banded solve cannot take, beyond 64 nodes): pass an RCCL communicator");
    hipStream_t s = c->stream;
    double *b_init = nullptr, *b_dmin = nullptr, *b_sums = nullptr;
    if (oneshot) {
        for (int r = 0; r < kMaxXchRanks; ++r) f.xch_inbox[r] = c->xch_peer[r];
        f.xch_rank = c->xch_rank; f.xch_nranks = c->xch_nranks; f.xch_mcap = c->xch_mcap; f.xch_epoch = ++c->xch_calls;
        f.xch_self = c->xch_self ? 1 : 0;          // a context setting (tdlo_set_xch_self; TDLO_XCH_SELF gives its initial value when the context is made)
        {   // test hook (tests/test_split_native_gpu.py): rank r's E-step refuses every sum as out of range -- an error of ONE shard, which its
            // peers must learn about inside the exchange (kXchErrMark) instead of waiting out the time limit
            static const int fail_rank = getenv("TDLO_TEST_RANGE_FAIL_RANK") ? atoi(getenv("TDLO_TEST_RANGE_FAIL_RANK")) : -1;
            if (fail_rank >= 0 && fail_rank == c->xch_rank) f.acc_lim[0] = f.acc_lim[1] = f.acc_lim[2] = 0.0;
        }
    } else {
        const size_t need = 2 + (size_t)M + 4 * (size_t)M + 2 + 2;
        if (need > c->split_buf_doubles) {
            HIPCHK(c, hipStreamSynchronize(s));
            if (c->split_buf) hipFree(c->split_buf);
            c->split_buf = nullptr; c->split_buf_doubles = 0;
            HIPCHK(c, hipMalloc((void **)&c->split_buf, need * sizeof(double)));
            c->split_buf_doubles = need;
        }
        b_init = c->split_buf; b_dmin = b_init + 2; b_sums = b_dmin + ((M + 1) & ~1);
        f.sums = b_sums;                                 // the export-only M-step writes, and the M-step from sums reads, this buffer
    }
    const bool timing = c->timing;
    if (timing) HIPCHK(c, hipEventRecord(c->ev[0], s));
    HIPCHK(c, hipMemcpyAsync(c->slots[0].nodeblk, c->pin, upload_doubles(nc, p, c->fh[0].lle_band != 0) * sizeof(double), hipMemcpyHostToDevice, s));
    HIPCHK(c, hipMemcpyAsync(c->fd, c->fh.data(), sizeof(FrameDev), hipMemcpyHostToDevice, s));
    HIPCHK(c, launch_split_setup(c->fd, c->fh.data(), s));
    auto nccl = [&](int e, const char *what) -> int {
        if (e == 0) return 0;
        return fail(c, TDLO_E_EXCHANGE, std::string(what) + ": " + R->GetErrorString(e));
    };
    // once per call: kept points and the sigma2-initialisation sum over all shards (trackdlo.cpp:195, :263-273)
    if (oneshot) HIPCHK(c, launch_xch_init(c->fd, s));
    else {
        HIPCHK(c, launch_split_init_pack(c->fd, b_init, s));
        if ((rc = nccl(R->AllReduce(b_init, b_init, 2, kNcclFloat64, kNcclSum, nccl_comm, s), "ncclAllReduce(init)"))) return rc;
        HIPCHK(c, launch_split_set_global_dev(c->fd, b_init, s));
    }
    if (timing) HIPCHK(c, hipEventRecord(c->ev[1], s));
    const bool visb = f.vis_branch != 0;
    int split_it = 0;
    auto iteration = [&]() -> int {
        if (oneshot) {
            if (visb) HIPCHK(c, launch_estep_only(c->fd, c->fh.data(), 1, 1, s));      // k_dmin: its last workgroup exchanges the minima
            HIPCHK(c, launch_estep_only(c->fd, c->fh.data(), 1, 0, s));
            mstep_parity_hint(split_it++);                                             // (this call's count of the registration's iterations: tdlo_mstep_chain.hip)
            const hipError_t me = launch_estep_only(c->fd, c->fh.data(), 1, 5, s);     // M-step with the exchange of the sums inside
            mstep_parity_hint(-1);
            HIPCHK(c, me);
        } else {
            int e;
            if (visb) {
                HIPCHK(c, launch_estep_only(c->fd, c->fh.data(), 1, 1, s));
                HIPCHK(c, launch_split_dmin_xch(c->fd, c->fh.data(), b_dmin, 0, s));
                if ((e = nccl(R->AllReduce(b_dmin, b_dmin, (size_t)M, kNcclFloat64, kNcclMin, nccl_comm, s), "ncclAllReduce(dmin)"))) return e;
                HIPCHK(c, launch_split_dmin_xch(c->fd, c->fh.data(), b_dmin, 1, s));
            }
            HIPCHK(c, launch_estep_only(c->fd, c->fh.data(), 1, 0, s));
            HIPCHK(c, launch_estep_only(c->fd, c->fh.data(), 1, 3, s));                // the shard's accumulators -> b_sums
            if ((e = nccl(R->AllReduce(b_sums, b_sums, 4 * (size_t)M + 2, kNcclFloat64, kNcclSum, nccl_comm, s), "ncclAllReduce(sums)"))) return e;
            HIPCHK(c, launch_estep_only(c->fd, c->fh.data(), 1, 4, s));
        }
        return 0;
    };
    IterState is{};
    int common_status = 0;              // RCCL form: the MIN over the ranks of their status
    for (int it = 1; it <= p->max_iter; ++it) {
        if ((rc = iteration())) return rc;
        // the stopping rule lives on the device (the same flag on every rank: they solve the same system); it is read after
        // iterations 1, 2, 4, 8, 12, ... -- a tracker in steady state converges within a couple of iterations.  With a communicator the
        // decision is taken TOGETHER (MIN all-reduce of [done without error, status]): a rank that stopped on an error of its own -- a
        // shard whose sums leave the fixed point's range -- must not leave its peers inside the next all-reduce
        const bool poll = p->tol > 0.0 && it < p->max_iter && (it == 1 || it == 2 || it == 4 || (it >= 8 && it % 4 == 0));
        if (poll && oneshot) {
            HIPCHK(c, hipMemcpyAsync(c->pin, f.st, sizeof(IterState), hipMemcpyDeviceToHost, s));
            HIPCHK(c, wait_stream(s));
            std::memcpy(&is, c->pin, sizeof is);
            if (is.done) break;
        } else if (poll) {
            HIPCHK(c, launch_split_poll_pack(c->fd, b_init, s));
            if ((rc = nccl(R->AllReduce(b_init, b_init, 2, kNcclFloat64, kNcclMin, nccl_comm, s), "ncclAllReduce(poll)"))) return rc;
            HIPCHK(c, hipMemcpyAsync(c->pin, b_init, 2 * sizeof(double), hipMemcpyDeviceToHost, s));
            HIPCHK(c, wait_stream(s));
            if (c->pin[0] != 0.0 || c->pin[1] < 0.0) { common_status = (int)c->pin[1]; break; }
        }
    }
    if (!oneshot) {                     // every rank leaves with the same verdict
        HIPCHK(c, launch_split_poll_pack(c->fd, b_init, s));
        if ((rc = nccl(R->AllReduce(b_init, b_init, 2, kNcclFloat64, kNcclMin, nccl_comm, s), "ncclAllReduce(status)"))) return rc;
    }
    if (timing) HIPCHK(c, hipEventRecord(c->ev[2], s));
    HIPCHK(c, hipMemcpyAsync(c->pin, c->slots[0].nodeblk + nc.Yout, nc.readback * sizeof(double), hipMemcpyDeviceToHost, s));
    if (timing) HIPCHK(c, hipEventRecord(c->ev[3], s));
    if (!oneshot) HIPCHK(c, hipMemcpyAsync(c->pin + nc.readback, b_init, 2 * sizeof(double), hipMemcpyDeviceToHost, s));
    HIPCHK(c, wait_stream(s));
    std::memcpy(&is, c->pin + (nc.st - nc.Yout), sizeof is);
    if (!oneshot) { common_status = (int)c->pin[nc.readback + 1]; if (common_status < 0 && is.status == 0) is.status = common_status; }
    c->last_F = 1;
    // The banded L D L^T gave up (non-positive pivot, non-finite sigma2): every rank solves the same system from the same sums, so every rank
    // is here with TDLO_E_NUMERIC (RCCL form: the common status; one-shot form: the same pivots, and a rank whose own shard failed has told
    // its peers, kXchErrMark) and repeats the call on the dense pivoted kernels, like run_frames -- Y and sigma2 are untouched so far.  The
    // one-shot exchange lives in one-workgroup kernels only: the dense k_mstep_fast serves up to 64 nodes, longer chains keep the error.
    if (is.status == TDLO_E_NUMERIC && p->include_lle && f.lle_band && !c->lle_dense_once && (!oneshot || M <= 64)) {
        c->lle_dense_once = true;
        ++c->band_retries;
        const int rr = tdlo_split_run(c, nccl_comm, Y, M, sigma2, p, priors, K, vis, n_vis, H_override, stats);
        c->lle_dense_once = false;
        if (stats) stats->band_retry = 1;
        return rr;
    }
    if ((is.status == 0 || is.status == TDLO_E_NUMERIC) && is.it > 0) { std::memcpy(Y, c->pin, sizeof(double) * 3 * M); }
    if (is.status == 0 || is.status == TDLO_E_NUMERIC) *sigma2 = is.sigma2;
    if (stats) {
        std::memset(stats, 0, sizeof *stats);
        fill_stats(stats, is);
        if (p->max_iter == 0) stats->converged = 1;
        if (timing) { hipEventElapsedTime(&stats->loop_ms, c->ev[1], c->ev[2]); hipEventElapsedTime(&stats->total_ms, c->ev[0], c->ev[3]); }
        stats->host_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_host0).count();
    }
    if (is.status == TDLO_E_EMPTY) return fail(c, is.status, "every point of every shard was pruned (no point within 0.1 m of a node, trackdlo.cpp:190)");
    if (is.status == TDLO_E_NUMERIC) return fail(c, is.status, "non-finite or non-positive sigma2, or singular M-step system");
    if (is.status == TDLO_E_EXCHANGE) return fail(c, is.status, "a peer's contribution to the one-shot exchange did not arrive within its time limit");
    return is.status;
}

// ---- plain GMM-EM `reg` ---------------------------------------------------------------------------
int tdlo_reg(tdlo_ctx *c, int slot, const double *pts, int N, double *Y, double *sigma2, int M, double mu, int max_iter) {
    if (!c) return TDLO_E_INVALID;
    if (slot < 0 || slot >= (int)c->slots.size()) return fail(c, TDLO_E_INVALID, "bad slot");
    if (!Y || !sigma2 || M < 1 || max_iter < 0 || !(mu >= 0 && mu < 1)) return fail(c, TDLO_E_INVALID, "bad reg arguments");
    if (M > reg_max_nodes()) return fail(c, TDLO_E_INVALID, "reg: more than " + std::to_string(reg_max_nodes()) + " centroids do not fit the E-step's per-wave accumulators in 160 KB of LDS");
    HIPCHK(c, hipSetDevice(c->device));
    int rc = TDLO_OK;
    if (pts) rc = tdlo_set_cloud(c, slot, pts, N);
    else if (c->slots[slot].N0 <= 0) rc = fail(c, TDLO_E_INVALID, "pts is NULL and no cloud is resident in the slot");
    if (rc) return rc;
    Slot &s = c->slots[slot];
    const int n = s.N0;
    const int nblk = std::max(1, std::min((n + 255) / 256, 256));
    const size_t need = reg_ws_doubles(M, nblk);
    if (need > c->reg_ws_cap) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (c->reg_ws) hipFree(c->reg_ws);
        c->reg_ws = nullptr; c->reg_ws_cap = 0;
        HIPCHK(c, hipMalloc((void **)&c->reg_ws, need * sizeof(double)));
        c->reg_ws_cap = need;
    }
    rc = ensure_pin(c, 8 + 3 * (size_t)M);
    if (rc) return rc;
    double *h = c->pin;
    for (int i = 0; i < 8; ++i) h[i] = 0.0;
    for (int i = 0; i < M; ++i) {                                         // utils.cpp:24-29
        h[8 + i] = 0.0;
        h[8 + M + i] = 0.1 / static_cast<double>(M) * static_cast<double>(i);
        h[8 + 2 * M + i] = 0.0;
    }
    HIPCHK(c, hipMemcpyAsync(c->reg_ws, h, sizeof(double) * (8 + 3 * (size_t)M), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, launch_reg(s.Xraw, n, M, mu, max_iter, nblk, c->reg_ws, c->stream));
    HIPCHK(c, hipMemcpyAsync(h, c->reg_ws, sizeof(double) * (8 + 3 * (size_t)M), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    *sigma2 = h[0];
    std::memcpy(Y, h + 8, sizeof(double) * 3 * (size_t)M);
    return TDLO_OK;
}

// Waits for the one-launch kernel's word in pinned host memory (epoch << 32 | status); like mbox_wait, the stream is looked at every 0.5 ms so that
// a faulting kernel cannot hang the caller.  Returns the status (1, 2, 3), 0 when the stream drained without the word, or a negative code.
static int cloud_wait(tdlo_ctx *c, hipStream_t st, unsigned epoch) {
    auto t_chk = std::chrono::steady_clock::now();
    for (unsigned spins = 1;; ++spins) {
        const unsigned long long v = __atomic_load_n(c->cloud_res, __ATOMIC_ACQUIRE);
        if ((unsigned)(v >> 32) == epoch) return (int)(unsigned)v;
        if ((spins & 255u) == 0) {
            const auto now = std::chrono::steady_clock::now();
            if (std::chrono::duration<double, std::micro>(now - t_chk).count() > 500.0) {
                t_chk = now;
                const hipError_t e = hipStreamQuery(st);
                if (e == hipSuccess) {
                    const unsigned long long v2 = __atomic_load_n(c->cloud_res, __ATOMIC_ACQUIRE);
                    return (unsigned)(v2 >> 32) == epoch ? (int)(unsigned)v2 : 0;
                }
                if (e != hipErrorNotReady) return fail(c, TDLO_E_HIP, std::string("stream: ") + hipGetErrorString(e));
            }
        }
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
    }
}

static size_t img_depth_bytes(size_t P) { return (P * 2 + 255) & ~(size_t)255; }
static size_t img_mask_bytes(size_t P) { return (P + 255) & ~(size_t)255; }

int tdlo_image_buffers(tdlo_ctx *c, int rows, int cols, unsigned short **depth, unsigned char **mask) {
    if (!c) return TDLO_E_INVALID;
    if (rows <= 0 || cols <= 0 || (long long)rows * cols > (1ll << 26) || !depth || !mask) return fail(c, TDLO_E_INVALID, "bad image");
    HIPCHK(c, hipSetDevice(c->device));
    const size_t P = (size_t)rows * cols, need = img_depth_bytes(P) + img_mask_bytes(P);
    if (need > c->img_pin_cap) {
        HIPCHK(c, drain_for_realloc(c));
        if (c->img_pin) hipHostFree(c->img_pin);
        c->img_pin = nullptr; c->img_pin_cap = 0;
        HIPCHK(c, hipHostMalloc((void **)&c->img_pin, need, hipHostMallocDefault));
        std::memset(c->img_pin, 0, need);
        c->img_pin_cap = need;
    }
    c->img_pin_rows = rows; c->img_pin_cols = cols;
    *depth = (unsigned short *)c->img_pin;
    *mask = (unsigned char *)(c->img_pin + img_depth_bytes(P));
    return TDLO_OK;
}

// the visibility pre-pass's device / pinned words (shared by tdlo_visibility_prepass's one-launch route and the depth -> cloud team kernel's own pre-pass)
static int ensure_vis_state(tdlo_ctx *c) {
    if (!c->vis_state) {
        HIPCHK(c, hipMalloc((void **)&c->vis_state, sizeof(unsigned long long) * (kMaxNodes + 8)));
        HIPCHK(c, hipHostMalloc((void **)&c->vis_res, sizeof(unsigned long long) * (kMaxNodes + 8), hipHostMallocDefault));
        std::memset(c->vis_res, 0, sizeof(unsigned long long) * (kMaxNodes + 8));
        c->vis_armed = false;
    }
    if (!c->vis_armed) {
        std::vector<unsigned long long> init(kMaxNodes + 8, 0x7ff0000000000000ull);
        for (int i = kMaxNodes; i < kMaxNodes + 8; ++i) init[i] = 0ull;
        HIPCHK(c, hipMemcpy(c->vis_state, init.data(), sizeof(unsigned long long) * init.size(), hipMemcpyHostToDevice));
        c->vis_armed = true;
    }
    return TDLO_OK;
}

// vis_Y != nullptr: the caller wants the frame's visibility pre-pass as well; *vis_done = true when the one-launch team kernel has left the M squared
// minima in c->vis_res[1 ..] (otherwise the caller runs tdlo_visibility_prepass on the resident cloud)
static int depth_to_cloud_impl(tdlo_ctx *c, int slot, const unsigned short *depth, const unsigned char *mask, int rows, int cols,
                               double fx, double fy, double cx, double cy, double leaf_size,
                               double *X_out, int x_capacity, int *n_out, int *n_raw_out, const double *vis_Y, int vis_M, bool *vis_done) {
    if (!c) return TDLO_E_INVALID;
    if (slot < 0 || slot >= (int)c->slots.size()) return fail(c, TDLO_E_INVALID, "bad slot");
    if (!depth || !mask || rows <= 0 || cols <= 0 || (long long)rows * cols > (1ll << 26)) return fail(c, TDLO_E_INVALID, "bad image");
    if (!(leaf_size > 0) || fx == 0 || fy == 0) return fail(c, TDLO_E_INVALID, "bad leaf size / intrinsics");
    HIPCHK(c, hipSetDevice(c->device));
    Slot &s = c->slots[slot];
    hipStream_t st = c->stream;
    const int P = rows * cols;
    const size_t img = img_depth_bytes(P) + img_mask_bytes(P);
    const size_t need = img + cloud_ws_bytes(P);
    if (need > c->cloud_ws_cap) {
        HIPCHK(c, hipStreamSynchronize(st));
        if (c->cloud_ws) hipFree(c->cloud_ws);
        c->cloud_ws = nullptr; c->cloud_ws_cap = 0;
        HIPCHK(c, hipMalloc(&c->cloud_ws, need));
        c->cloud_ws_cap = need;
    }
    int rc = ensure_pin(c, 16);
    if (rc) return rc;
    if (c->cloud_pending >= 0 && (rc = flush_pending_cloud(c))) return rc;      // (a cloud tracking_step staged for this slot is superseded; its copy is harmless and ordered)
    char *base = (char *)c->cloud_ws;
    const unsigned short *d_depth = (unsigned short *)base;
    const unsigned char *d_mask = (unsigned char *)(base + img_depth_bytes(P));
    char *ws = base + img;
    const double cam[4] = {fx, fy, cx, cy};
    // images the caller wrote into the context's pinned buffers (tdlo_image_buffers) are read where they are; anything else is copied first
    const bool in_place = c->img_pin != nullptr && (const char *)depth == c->img_pin && (const char *)mask == c->img_pin + img_depth_bytes(P) &&
                          c->img_pin_rows == rows && c->img_pin_cols == cols;
    if (in_place) { d_depth = depth; d_mask = mask; }
    else {
        HIPCHK(c, hipMemcpyAsync((void *)d_depth, depth, (size_t)P * 2, hipMemcpyHostToDevice, st));
        HIPCHK(c, hipMemcpyAsync((void *)d_mask, mask, (size_t)P, hipMemcpyHostToDevice, st));
    }
    if (n_out) *n_out = 0;
    if (n_raw_out) *n_raw_out = 0;
    s.N0 = 0; s.sorted_valid = false;
    const float leaf = (float)leaf_size, inv = 1.0f / leaf;
    int n = -1, nraw = 0;
    if (c->cloud_fused_on && cloud_fused_ok(P)) {
        // ---- one launch (k_cloud_fused); its last workgroup reports through pinned host memory
        const size_t fneed = cloud_fused_ws_bytes(P);
        if (fneed > c->cloud_fws_cap) {
            HIPCHK(c, hipStreamSynchronize(st));
            if (c->cloud_fws) hipFree(c->cloud_fws);
            c->cloud_fws = nullptr; c->cloud_fws_cap = 0;
            HIPCHK(c, hipMalloc(&c->cloud_fws, fneed));
            c->cloud_fws_cap = fneed; c->cloud_fused_first = true;
        }
        if (!c->cloud_res) {
            HIPCHK(c, hipHostMalloc((void **)&c->cloud_res, 32 * sizeof(unsigned long long), hipHostMallocDefault));      // [0..1] the report, [4..] phase stamps of an instrumented build
            std::memset(c->cloud_res, 0, 32 * sizeof(unsigned long long));
        }
        if ((rc = ensure_points(c, s, cloud_fused_max_points()))) return rc;      // (an output point per masked pixel at most; sized once)
        if (++c->cloud_epoch == 0) ++c->cloud_epoch;
        // the frame's visibility pre-pass rides along when the team kernel runs (up to 64 nodes): the nodes staged in pinned memory, read by the kernel
        const bool vis_ride = vis_Y != nullptr && vis_M >= 1 && vis_M <= 64 && c->cloud_team_on && c->direct_in;
        if (vis_ride) {
            if ((rc = ensure_vis_state(c))) return rc;
            if (!c->vis_nodes_pin) HIPCHK(c, hipHostMalloc((void **)&c->vis_nodes_pin, sizeof(double) * 3 * 64, hipHostMallocDefault));
            std::memcpy(c->vis_nodes_pin, vis_Y, sizeof(double) * 3 * (size_t)vis_M);
        }
        HIPCHK(c, launch_cloud_fused(d_depth, d_mask, P, cols, cam, inv, ws, c->cloud_fws, c->cloud_fused_first, c->cloud_team_on, s.Xraw, s.cap_points, c->cloud_res, c->cloud_epoch, st,
                                     vis_ride ? c->vis_nodes_pin : nullptr, vis_ride ? vis_M : 0, c->vis_state, c->vis_res));
        c->cloud_fused_first = false;
        const int status = cloud_wait(c, st, c->cloud_epoch);
        if (status != 1 && vis_ride) c->vis_armed = false;      // (whatever the team left in the minima: armed again before their next use)
        if (status < 0) { c->cloud_fused_first = true; return status; }
        if (status == 0) { c->cloud_fused_first = true; return fail(c, TDLO_E_HIP, "the stream drained, but the depth -> cloud kernel did not report"); }
        const unsigned long long w1 = __atomic_load_n(c->cloud_res + 1, __ATOMIC_RELAXED);
        nraw = (int)(unsigned)(w1 >> 32);
        if (status == 1) { n = (int)(unsigned)w1; ++c->cloud_route[0]; if (vis_ride && vis_done) { *vis_done = true; ++c->cloud_vis_rides; } }
        else if (status == 3) return fail(c, TDLO_E_HIP, "voxel grid produced more points than the slot holds");
        else if (status == 4) { c->cloud_fused_first = true; ++c->cloud_route[1]; }      // the team gave the launch up (a wait of 2 s): state words initialised again, the multi-launch form below
        else ++c->cloud_route[1];                 // not taken (too many masked pixels / cells, pass-through): the multi-launch form below
    }
    if (n < 0) {
        // ---- the multi-launch form: bounding box + count, a host round trip, compaction, radix sort passes, centroids
        // the last 64 ints of the workspace: bounding box (6 ordered floats), masked-pixel count, output count
        unsigned *d_bbox = (unsigned *)(ws + cloud_ws_bytes(P) - 256);
        int *d_total = (int *)(d_bbox + 8);
        unsigned *hb = (unsigned *)c->pin;
        hb[0] = hb[1] = hb[2] = ~0u; hb[3] = hb[4] = hb[5] = 0u; hb[6] = 0u; hb[7] = 0u; hb[8] = 0u;
        HIPCHK(c, hipMemcpyAsync(d_bbox, hb, 9 * sizeof(unsigned), hipMemcpyHostToDevice, st));
        HIPCHK(c, launch_cloud_bbox(d_depth, d_mask, P, cols, cam, d_bbox, ws, st));
        HIPCHK(c, hipMemcpyAsync(hb, d_bbox, 8 * sizeof(unsigned), hipMemcpyDeviceToHost, st));
        HIPCHK(c, hipStreamSynchronize(st));
        nraw = (int)hb[6];
        if (n_raw_out) *n_raw_out = nraw;
        if (nraw == 0) return TDLO_OK;
        auto decode = [](unsigned o) { const unsigned u = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o; float f; std::memcpy(&f, &u, 4); return f; };
        float mn[3], mx[3];
        for (int d = 0; d < 3; ++d) { mn[d] = decode(hb[d]); mx[d] = decode(hb[3 + d]); }
        // pcl/filters/impl/voxel_grid.hpp applyFilter: leaf-size check, min_b / div_b / divb_mul (float arithmetic)
        long long dd[3]; int min_b[3], div_b[3];
        for (int d = 0; d < 3; ++d) {
            dd[d] = (long long)((mx[d] - mn[d]) * inv) + 1;
            min_b[d] = (int)std::floor(mn[d] * inv);
            div_b[d] = (int)std::floor(mx[d] * inv) - min_b[d] + 1;
        }
        const int nodown = (dd[0] * dd[1] * dd[2] > 2147483647ll) ? 1 : 0;
        const long long cells = nodown ? 1 : (long long)div_b[0] * div_b[1] * div_b[2];
        if (cells >= 0xffffffffll) return fail(c, TDLO_E_INVALID, "voxel grid has too many cells");
        int passes = 1;
        while (passes < 4 && (1ll << (8 * passes)) <= cells) ++passes;      // every valid key must stay below the all-ones sentinel
        rc = ensure_points(c, s, nraw);
        if (rc) return rc;
        HIPCHK(c, launch_cloud_voxels(d_depth, d_mask, P, cols, cam, min_b, div_b[0], div_b[0] * div_b[1], inv, nodown, passes, nraw,
                                      ws, d_total, s.cap_points, s.Xraw, st));
        HIPCHK(c, hipMemcpyAsync(hb, d_total, sizeof(int), hipMemcpyDeviceToHost, st));
        HIPCHK(c, hipStreamSynchronize(st));
        n = (int)hb[0];
        if (n < 0 || n > s.cap_points) return fail(c, TDLO_E_HIP, "voxel grid produced an impossible point count");
    }
    if (n_raw_out) *n_raw_out = nraw;
    s.N0 = n; s.sorted_valid = false;
    if (n_out) *n_out = n;
    if (X_out && n > 0) {
        if (n > x_capacity) return fail(c, TDLO_E_INVALID, "X_out too small for the down-sampled cloud");
        HIPCHK(c, hipMemcpyAsync(X_out, s.Xraw, 3 * (size_t)n * sizeof(double), hipMemcpyDeviceToHost, st));
        HIPCHK(c, hipStreamSynchronize(st));
    }
    return TDLO_OK;
}

int tdlo_depth_to_cloud(tdlo_ctx *c, int slot, const unsigned short *depth, const unsigned char *mask, int rows, int cols,
                        double fx, double fy, double cx, double cy, double leaf_size,
                        double *X_out, int x_capacity, int *n_out, int *n_raw_out) {
    return depth_to_cloud_impl(c, slot, depth, mask, rows, cols, fx, fy, cx, cy, leaf_size, X_out, x_capacity, n_out, n_raw_out, nullptr, 0, nullptr);
}

// ---- caller-side visibility pre-pass ------------------------------------------------------------
static void vis_threshold_and_fill(const double *min_d2, int M, double visibility_threshold, double d_vis, const double *geodesic_coord, double *node_dist,
                                   int *visible_nodes, int *n_vis, int *visible_nodes_extended, int *n_vis_ext);

int tdlo_visibility_prepass(tdlo_ctx *c, int slot, const double *Y, int M, double visibility_threshold, double d_vis,
                            const double *geodesic_coord, double *node_dist, int *visible_nodes, int *n_vis,
                            int *visible_nodes_extended, int *n_vis_ext) {
    if (!c) return TDLO_E_INVALID;
    if (slot < 0 || slot >= (int)c->slots.size()) return fail(c, TDLO_E_INVALID, "bad slot");
    if (!Y || M < 1 || !geodesic_coord) return fail(c, TDLO_E_INVALID, "null Y / geodesic_coord");
    Slot &s = c->slots[slot];
    if (s.N0 <= 0) return fail(c, TDLO_E_INVALID, "no cloud resident in slot (call tdlo_set_cloud)");
    HIPCHK(c, hipSetDevice(c->device));
    int rc = ensure_nodes(c, s, M);
    if (rc) return rc;
    rc = ensure_pin(c, 4 * (size_t)M + 8);
    if (rc) return rc;
    NodeCarve nc(M);
    hipStream_t st = c->stream;
    if (c->cloud_pending >= 0 && (rc = flush_pending_cloud(c))) return rc;
    std::memcpy(c->pin, Y, sizeof(double) * 3 * M);
    if (c->direct_in && M <= kMaxNodes) {
        // one launch, no copies: the kernel reads the nodes from the pinned staging block and its last workgroup writes the minima to pinned host
        // memory (TDLO_DIRECT_UPLOAD=0: the copies + stream synchronisation below, the comparator)
        if ((rc = ensure_vis_state(c))) return rc;
        if (++c->vis_epoch == 0) ++c->vis_epoch;
        HIPCHK(c, launch_node_min_dist_direct(s.Xraw, s.N0, c->pin, M, c->vis_state, c->vis_res, c->vis_epoch, st));
        auto t_chk = std::chrono::steady_clock::now();
        for (unsigned spins = 1;; ++spins) {
            if ((unsigned)__atomic_load_n(c->vis_res, __ATOMIC_ACQUIRE) == c->vis_epoch) break;
            if ((spins & 255u) == 0) {
                const auto now = std::chrono::steady_clock::now();
                if (std::chrono::duration<double, std::micro>(now - t_chk).count() > 500.0) {
                    t_chk = now;
                    const hipError_t e = hipStreamQuery(st);
                    if (e == hipSuccess) {
                        if ((unsigned)__atomic_load_n(c->vis_res, __ATOMIC_ACQUIRE) == c->vis_epoch) break;
                        c->vis_armed = false;
                        return fail(c, TDLO_E_HIP, "the stream drained, but the visibility pre-pass did not report");
                    }
                    if (e != hipErrorNotReady) { c->vis_armed = false; return fail(c, TDLO_E_HIP, std::string("stream: ") + hipGetErrorString(e)); }
                }
            }
#if defined(__x86_64__)
            __builtin_ia32_pause();
#endif
        }
        std::memcpy(c->pin, c->vis_res + 1, sizeof(double) * M);
    } else {
        double *dY = s.nodeblk + nc.Yin;                                  // reuse the node upload area
        unsigned long long *dbits = (unsigned long long *)(s.nodeblk + nc.dmin);
        for (int m = 0; m < M; ++m) { const unsigned long long inf = 0x7ff0000000000000ull; std::memcpy(c->pin + 3 * M + m, &inf, 8); }
        HIPCHK(c, hipMemcpyAsync(dY, c->pin, sizeof(double) * 3 * M, hipMemcpyHostToDevice, st));
        HIPCHK(c, hipMemcpyAsync(dbits, c->pin + 3 * M, sizeof(double) * M, hipMemcpyHostToDevice, st));
        HIPCHK(c, launch_node_min_dist(s.Xraw, s.N0, dY, M, dbits, st));
        HIPCHK(c, hipMemcpyAsync(c->pin, dbits, sizeof(double) * M, hipMemcpyDeviceToHost, st));
        HIPCHK(c, hipStreamSynchronize(st));
    }
    vis_threshold_and_fill(c->pin, M, visibility_threshold, d_vis, geodesic_coord, node_dist, visible_nodes, n_vis, visible_nodes_extended, n_vis_ext);
    return TDLO_OK;      // an empty visible set is reported as n_vis = 0 (the reference underflows at :351)
}

// trackdlo_node.cpp:348-360: occluded runs shorter than d_vis (in the geodesic coordinate) between two visible nodes are filled in
static void fill_visible_gaps(const std::vector<int> &vis, const double *geodesic_coord, double d_vis, std::vector<int> &ext) {
    ext.clear();
    if (vis.empty()) return;
    for (size_t i = 0; i + 1 < vis.size(); ++i) {
        ext.push_back(vis[i]);
        if (std::fabs(geodesic_coord[vis[i + 1]] - geodesic_coord[vis[i]]) <= d_vis)
            for (int j = 1; j < vis[i + 1] - vis[i]; ++j) ext.push_back(vis[i] + j);
    }
    ext.push_back(vis.back());
}

// thresholding and gap fill on the host (O(M)) from the squared minima: trackdlo_node.cpp:316/:326 (distance test only; the painter test of
// :279-343 is tdlo_self_occlusion_visible, applied by the caller or by a tracker it was switched on for), :345-360
static void vis_threshold_and_fill(const double *min_d2, int M, double visibility_threshold, double d_vis, const double *geodesic_coord, double *node_dist,
                                   int *visible_nodes, int *n_vis, int *visible_nodes_extended, int *n_vis_ext) {
    std::vector<int> vis;
    for (int m = 0; m < M; ++m) {
        const double d = std::sqrt(min_d2[m]);
        if (node_dist) node_dist[m] = d;
        if (d <= visibility_threshold) vis.push_back(m);
    }
    if (n_vis) *n_vis = (int)vis.size();
    if (visible_nodes) std::copy(vis.begin(), vis.end(), visible_nodes);
    std::vector<int> ext;
    fill_visible_gaps(vis, geodesic_coord, d_vis, ext);
    if (n_vis_ext) *n_vis_ext = (int)ext.size();
    if (visible_nodes_extended) std::copy(ext.begin(), ext.end(), visible_nodes_extended);
}

// One frame of the ROS node up to tracking_step (trackdlo_node.cpp:195-277, :345-360) in one call: depth image -> cloud -> voxel grid, and the visibility
// pre-pass of the tracker's current nodes against that cloud.  With up to 64 nodes and the one-launch team kernel the pre-pass rides in the same launch
// (every team member takes the minima over the centroids it has just formed): one launch and one hand-over through pinned memory for both steps.
// Otherwise -- more nodes, TDLO_CLOUD_TEAM=0, too many masked pixels, a team that gave up -- tdlo_visibility_prepass runs behind it: the same numbers.
int tdlo_depth_to_cloud_visibility(tdlo_ctx *c, int slot, const unsigned short *depth, const unsigned char *mask, int rows, int cols,
                                   double fx, double fy, double cx, double cy, double leaf_size,
                                   const double *Y, int M, double visibility_threshold, double d_vis, const double *geodesic_coord,
                                   double *node_dist, int *visible_nodes, int *n_vis, int *visible_nodes_extended, int *n_vis_ext,
                                   int *n_out, int *n_raw_out) {
    if (!c) return TDLO_E_INVALID;
    if (!Y || M < 1 || !geodesic_coord) return fail(c, TDLO_E_INVALID, "null Y / geodesic_coord");
    bool vis_done = false;
    int n = 0;
    int rc = depth_to_cloud_impl(c, slot, depth, mask, rows, cols, fx, fy, cx, cy, leaf_size, nullptr, 0, &n, n_raw_out, Y, M, &vis_done);
    if (n_out) *n_out = n;
    if (rc) return rc;
    if (n_vis) *n_vis = 0;
    if (n_vis_ext) *n_vis_ext = 0;
    if (n == 0) return TDLO_OK;                  // no cloud, nothing visible (tdlo_visibility_prepass would refuse the empty slot)
    if (!vis_done) return tdlo_visibility_prepass(c, slot, Y, M, visibility_threshold, d_vis, geodesic_coord, node_dist, visible_nodes, n_vis, visible_nodes_extended, n_vis_ext);
    if ((rc = ensure_pin(c, (size_t)M + 8))) return rc;
    std::memcpy(c->pin, c->vis_res + 1, sizeof(double) * M);
    vis_threshold_and_fill(c->pin, M, visibility_threshold, d_vis, geodesic_coord, node_dist, visible_nodes, n_vis, visible_nodes_extended, n_vis_ext);
    return TDLO_OK;
}

// ---- measurement ---------------------------------------------------------------------------------
int tdlo_profile_kernel(tdlo_ctx *c, int slot, int kind, int reps, float *avg_us) {
    if (!c || c->last_F < 1 || c->fh.empty()) return TDLO_E_INVALID;
    if (!((kind >= 0 && kind <= 4) || kind == 10) || reps < 1 || (kind == 10 && reps > 256)) return fail(c, TDLO_E_INVALID, "bad kind / reps");
    (void)slot;
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t s = c->stream;
    const int F = c->last_F;
    // the loop left done = 1; clear it so the kernels do their work (state is scratch from here on)
    std::vector<IterState> saved(F);
    for (int i = 0; i < F; ++i) {
        HIPCHK(c, hipMemcpyAsync(&saved[i], c->fh[i].st, sizeof(IterState), hipMemcpyDeviceToHost, s));
    }
    HIPCHK(c, hipStreamSynchronize(s));
    for (int i = 0; i < F; ++i) {
        IterState is = saved[i];
        is.done = 0; is.it = 0;
        std::memcpy(c->pin + i * 16, &is, sizeof is);
        HIPCHK(c, hipMemcpyAsync(c->fh[i].st, c->pin + i * 16, sizeof(IterState), hipMemcpyHostToDevice, s));
    }
    std::vector<FrameDev> fh = c->fh;
    for (auto &f : fh) f.max_iter = 1 << 30;
    for (auto &f : fh) f.tol = -1.0;
    HIPCHK(c, hipMemcpyAsync(c->fd, fh.data(), sizeof(FrameDev) * F, hipMemcpyHostToDevice, s));
    HIPCHK(c, zero_accumulators(fh, F, s));
    if (kind == 10) {
        // E-step in situ: real iterations (E-step, M-step alternating, as in the loop), every E-step dispatch carries its
        // own start/stop events -- the duration a kernel trace reports, not a back-to-back average with hot caches
        std::vector<hipEvent_t> evs(2 * (size_t)reps, nullptr);
        for (auto &e : evs) HIPCHK(c, hipEventCreate(&e));
        for (int w = 0; w < 3; ++w) HIPCHK(c, launch_iteration(c->fd, fh.data(), F, s, w));      // (the counter was set to 0 above)
        for (int r = 0; r < reps; ++r) HIPCHK(c, launch_iteration_timed(c->fd, fh.data(), F, s, evs[2 * r], evs[2 * r + 1], nullptr, nullptr, 3 + r));
        HIPCHK(c, hipStreamSynchronize(s));
        double tot = 0;
        for (int r = 0; r < reps; ++r) { float ms = 0; hipEventElapsedTime(&ms, evs[2 * r], evs[2 * r + 1]); tot += ms; }
        for (auto &e : evs) hipEventDestroy(e);
        if (avg_us) *avg_us = (float)(tot * 1000.0 / reps);
    } else {
    if (kind >= 2) {
        // M-step kinds: every repetition needs this iteration's sums, which an M-step consumes (it clears the other parity's rows and
        // advances the iteration counter) -- so the E-step runs in between, outside the timed spans (one event pair per repetition)
        std::vector<hipEvent_t> evs(2 * (size_t)reps, nullptr);
        for (auto &e : evs) HIPCHK(c, hipEventCreate(&e));
        double tot = 0;
        for (int r = -3; r < reps; ++r) {
            if (fh[0].vis_branch) HIPCHK(c, launch_estep_only(c->fd, fh.data(), F, 1, s));
            HIPCHK(c, launch_estep_only(c->fd, fh.data(), F, 0, s));
            if (kind == 4) HIPCHK(c, launch_estep_only(c->fd, fh.data(), F, 3, s));      // the sums this kind starts from
            if (r >= 0) HIPCHK(c, hipEventRecord(evs[2 * r], s));
            HIPCHK(c, launch_estep_only(c->fd, fh.data(), F, kind, s));
            if (r >= 0) HIPCHK(c, hipEventRecord(evs[2 * r + 1], s));
            if (kind == 3) HIPCHK(c, launch_estep_only(c->fd, fh.data(), F, 4, s));      // export only: finish the iteration
        }
        HIPCHK(c, hipStreamSynchronize(s));
        for (int r = 0; r < reps; ++r) { float ms = 0; hipEventElapsedTime(&ms, evs[2 * r], evs[2 * r + 1]); tot += ms; }
        for (auto &e : evs) hipEventDestroy(e);
        if (avg_us) *avg_us = (float)(tot * 1000.0 / reps);
    } else {
    for (int w = 0; w < 3; ++w) HIPCHK(c, launch_estep_only(c->fd, fh.data(), F, kind, s));
    HIPCHK(c, hipEventRecord(c->ev[0], s));
    for (int r = 0; r < reps; ++r) HIPCHK(c, launch_estep_only(c->fd, fh.data(), F, kind, s));
    HIPCHK(c, hipEventRecord(c->ev[1], s));
    HIPCHK(c, hipStreamSynchronize(s));
    float ms = 0;
    hipEventElapsedTime(&ms, c->ev[0], c->ev[1]);
    if (avg_us) *avg_us = ms * 1000.0f / (float)reps;
    }
    }
    HIPCHK(c, zero_accumulators(fh, F, s));
    // restore
    for (int i = 0; i < F; ++i) {
        std::memcpy(c->pin + i * 16, &saved[i], sizeof(IterState));
        HIPCHK(c, hipMemcpyAsync(c->fh[i].st, c->pin + i * 16, sizeof(IterState), hipMemcpyHostToDevice, s));
    }
    HIPCHK(c, hipMemcpyAsync(c->fd, c->fh.data(), sizeof(FrameDev) * F, hipMemcpyHostToDevice, s));
    HIPCHK(c, hipStreamSynchronize(s));
    return TDLO_OK;
}

int tdlo_profile_iteration(tdlo_ctx *c, int reps, float *estep_us, float *mstep_us, float *iter_us, char *mstep_kernel, int name_cap) {
    if (!c || c->last_F < 1 || c->fh.empty()) return TDLO_E_INVALID;
    if (reps < 1 || reps > 256) return fail(c, TDLO_E_INVALID, "bad reps");
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t s = c->stream;
    const int F = c->last_F;
    std::vector<IterState> saved(F);
    for (int i = 0; i < F; ++i) HIPCHK(c, hipMemcpyAsync(&saved[i], c->fh[i].st, sizeof(IterState), hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    for (int i = 0; i < F; ++i) {
        IterState is = saved[i];
        is.done = 0; is.it = 0;
        std::memcpy(c->pin + i * 16, &is, sizeof is);
        HIPCHK(c, hipMemcpyAsync(c->fh[i].st, c->pin + i * 16, sizeof(IterState), hipMemcpyHostToDevice, s));
    }
    std::vector<FrameDev> fh = c->fh;
    for (auto &f : fh) { f.max_iter = 1 << 30; f.tol = -1.0; }
    HIPCHK(c, hipMemcpyAsync(c->fd, fh.data(), sizeof(FrameDev) * F, hipMemcpyHostToDevice, s));
    HIPCHK(c, zero_accumulators(fh, F, s));
    std::vector<hipEvent_t> evs(4 * (size_t)reps, nullptr);
    for (auto &e : evs) HIPCHK(c, hipEventCreate(&e));
    for (int w = 0; w < 3; ++w) HIPCHK(c, launch_iteration(c->fd, fh.data(), F, s, w));      // (the counter was set to 0 above)
    HIPCHK(c, hipEventRecord(c->ev[0], s));
    for (int r = 0; r < reps; ++r) HIPCHK(c, launch_iteration_timed(c->fd, fh.data(), F, s, evs[4 * r], evs[4 * r + 1], evs[4 * r + 2], evs[4 * r + 3], 3 + r));
    HIPCHK(c, hipEventRecord(c->ev[1], s));
    HIPCHK(c, hipStreamSynchronize(s));
    double te = 0, tm = 0;
    bool m_ok = true;
    for (int r = 0; r < reps; ++r) {
        float ms = 0;
        hipEventElapsedTime(&ms, evs[4 * r], evs[4 * r + 1]); te += ms;
        if (hipEventElapsedTime(&ms, evs[4 * r + 2], evs[4 * r + 3]) == hipSuccess) tm += ms; else m_ok = false;     // dispatch without events
    }
    (void)hipGetLastError();
    float tot = 0;
    hipEventElapsedTime(&tot, c->ev[0], c->ev[1]);
    for (auto &e : evs) hipEventDestroy(e);
    if (estep_us) *estep_us = (float)(te * 1000.0 / reps);
    if (mstep_us) *mstep_us = m_ok ? (float)(tm * 1000.0 / reps) : -1.0f;
    if (iter_us) *iter_us = tot * 1000.0f / (float)reps;
    if (mstep_kernel && name_cap > 0) { std::strncpy(mstep_kernel, mstep_kernel_name(fh.data(), F), (size_t)name_cap - 1); mstep_kernel[name_cap - 1] = 0; }
    for (int i = 0; i < F; ++i) {
        std::memcpy(c->pin + i * 16, &saved[i], sizeof(IterState));
        HIPCHK(c, hipMemcpyAsync(c->fh[i].st, c->pin + i * 16, sizeof(IterState), hipMemcpyHostToDevice, s));
    }
    HIPCHK(c, zero_accumulators(fh, F, s));
    HIPCHK(c, hipMemcpyAsync(c->fd, c->fh.data(), sizeof(FrameDev) * F, hipMemcpyHostToDevice, s));
    HIPCHK(c, hipStreamSynchronize(s));
    return TDLO_OK;
}

int tdlo_debug_read_cloud(tdlo_ctx *c, int slot, double *out, int max_points, double *ctr) {
    if (!c || slot < 0 || slot >= (int)c->slots.size() || !out || c->fh.empty()) return TDLO_E_INVALID;
    const FrameDev &f = c->fh[0];
    IterState is;
    HIPCHK(c, hipMemcpy(&is, f.st, sizeof is, hipMemcpyDeviceToHost));
    const int N = is.N;
    if (N > max_points) return TDLO_E_INVALID;
    const size_t es = f.precision == TDLO_PREC_F64 ? 8 : 4;
    std::vector<char> buf(es * (size_t)N);
    for (int d = 0; d < 3; ++d) {
        HIPCHK(c, hipMemcpy(buf.data(), (const char *)f.Xs + es * (size_t)f.ldx * d, es * (size_t)N, hipMemcpyDeviceToHost));
        for (int n = 0; n < N; ++n) out[(size_t)d * N + n] = es == 8 ? ((const double *)buf.data())[n] : (double)((const float *)buf.data())[n];
    }
    if (ctr) HIPCHK(c, hipMemcpy(ctr, f.ctr, 3 * sizeof(double), hipMemcpyDeviceToHost));
    return N;
}

int tdlo_debug_mstep_dense(int on) { return mstep_set_dense(on); }
int tdlo_debug_mstep_lle_dense(int on) { return mstep_set_lle_dense(on); }
long long tdlo_debug_band_retries(tdlo_ctx *c) { return c ? c->band_retries : -1; }

long long tdlo_debug_route_count(tdlo_ctx *c, int which) {
    if (!c || which < 0 || which > 13) return -1;
    if (which == 13) return c->batch_loop_fallbacks;
    if (which == 12) return c->batch_loop_calls;
    if (which == 11) return c->spin_calls;
    if (which == 10) return c->boost_retries;
    if (which == 9) return c->estep2_frames;
    if (which == 8) return c->cloud_vis_rides;
    return which < 6 ? c->route_count[which] : c->cloud_route[which - 6];
}

int tdlo_debug_lle_band_device(tdlo_ctx *c, const double *Y, int M, double *Hb) {
    if (!c) return TDLO_E_INVALID;
    if (!Y || !Hb || M < 1 || M > 256) return fail(c, TDLO_E_INVALID, "tdlo_debug_lle_band_device: 1 .. 256 nodes");
    HIPCHK(c, hipSetDevice(c->device));
    double *d = nullptr;
    HIPCHK(c, hipMalloc((void **)&d, sizeof(double) * 16 * (size_t)M));
    hipError_t e = hipMemcpyAsync(d, Y, sizeof(double) * 3 * M, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = launch_lle_band_debug(d, M, d + 3 * M, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(Hb, d + 3 * M, sizeof(double) * 13 * M, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(d);
    HIPCHK(c, e);
    return TDLO_OK;
}
int tdlo_debug_fail_hip(tdlo_ctx *c) {
    if (!c) return TDLO_E_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpy(nullptr, nullptr, 16, hipMemcpyDeviceToDevice));      // hipErrorInvalidValue
    return TDLO_OK;
}

int tdlo_pci_bus_id(tdlo_ctx *c, char *out, int len) {
    if (!c || !out || len < 16) return TDLO_E_INVALID;
    HIPCHK(c, hipDeviceGetPCIBusId(out, len, c->device));
    return TDLO_OK;
}

int tdlo_set_xch_self(tdlo_ctx *c, int on) {
    if (!c) return TDLO_E_INVALID;
    const int prev = c->xch_self ? 1 : 0;
    c->xch_self = on != 0;
    return prev;
}

int tdlo_set_sort_reuse(tdlo_ctx *c, int on) {
    if (!c) return TDLO_E_INVALID;
    const int prev = c->sort_reuse ? 1 : 0;
    c->sort_reuse = on != 0;
    return prev;
}

int tdlo_set_timing(tdlo_ctx *c, int on) {
    if (!c) return TDLO_E_INVALID;
    const int prev = c->timing ? 1 : 0;
    c->timing = on != 0;
    return prev;
}

int tdlo_debug_stamps(tdlo_ctx *c, int slot, unsigned long long *out, int n) {
    if (!c || slot < 0 || slot >= (int)c->slots.size() || !out || n < 1 || n > 64 || c->fh.empty()) return TDLO_E_INVALID;
    const int dbg_frame = getenv("TDLO_DEBUG_FRAME") ? atoi(getenv("TDLO_DEBUG_FRAME")) : 0;      // which frame of the last batch
    if (dbg_frame < 0 || dbg_frame >= (int)c->fh.size()) return TDLO_E_INVALID;
    HIPCHK(c, hipMemcpyAsync(c->pin, c->fh[dbg_frame].dbg, sizeof(unsigned long long) * n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    std::memcpy(out, c->pin, sizeof(unsigned long long) * n);
    return TDLO_OK;
}

int tdlo_debug_cloud_stamps(tdlo_ctx *c, unsigned long long *out, int n) {
    if (!c || !out || n < 0 || n > 24) return TDLO_E_INVALID;
    if (!c->cloud_fws) return fail(c, TDLO_E_INVALID, "no depth -> cloud call yet");
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(out, (const char *)c->cloud_fws + 64, sizeof(unsigned long long) * n, hipMemcpyDeviceToHost));      // (24 words behind the 16 state words)
    return TDLO_OK;
}

int tdlo_debug_exp2(tdlo_ctx *c, const double *x, double *y, int n) {
    if (!c || !x || !y || n < 1) return TDLO_E_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    double *d = nullptr;
    HIPCHK(c, hipMalloc((void **)&d, 2 * (size_t)n * sizeof(double)));
    hipError_t e = hipMemcpyAsync(d, x, (size_t)n * sizeof(double), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = launch_debug_exp2(d, d + n, n, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(y, d + n, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    hipFree(d);
    HIPCHK(c, e);
    return TDLO_OK;
}

// ---- host helpers --------------------------------------------------------------------------------
int tdlo_calc_lle_weights(int k, const double *Y, int M, double *L) {
    if (!Y || !L || M < 1 || k < 2 || k > 13) return TDLO_E_INVALID;
    lle_weights(k, Y, M, L);
    return TDLO_OK;
}

int tdlo_calc_lle_regulariser(const double *Y, int M, double *H, double *Hb) {
    if (!Y || M < 1 || (!H && !Hb)) return TDLO_E_INVALID;
    if (H) {
        std::vector<double> L((size_t)M * M);
        lle_weights(6, Y, M, L.data());
        lle_regulariser(L.data(), M, H);
    }
    if (Hb) lle_regulariser_band(Y, M, Hb);
    return TDLO_OK;
}

int tdlo_line_sphere_intersection(const double A[3], const double B[3], const double C[3], double radius, double out[6]) {
    Vec3 o[2];
    const int n = line_sphere(Vec3{A[0], A[1], A[2]}, Vec3{B[0], B[1], B[2]}, Vec3{C[0], C[1], C[2]}, radius, o);
    for (int i = 0; i < n; ++i) { out[3 * i] = o[i].x; out[3 * i + 1] = o[i].y; out[3 * i + 2] = o[i].z; }
    return n;
}

double tdlo_piecewise_error(const double *Y_track, int n_track, const double *Y_true, int n_true) {
    if (!Y_track || !Y_true || n_track < 1 || n_true < 2) return -1.0;
    return piecewise_error(Y_track, n_track, Y_true, n_true);
}

double tdlo_compute_error(const double *Y_track, int n_track, const double *Y_true, int n_true) {
    if (!Y_track || !Y_true || n_track < 2 || n_true < 2) return -1.0;
    return 0.5 * (piecewise_error(Y_track, n_track, Y_true, n_true) + piecewise_error(Y_true, n_true, Y_track, n_track));
}

int tdlo_traverse_euclidean(const double *coord, int n_coord, const double *guide, int Mg, const int *vis, int n_vis,
                            int alignment, int anchor, double *out) {
    if (!coord || !guide || !vis || !out) return TDLO_E_INVALID;
    std::vector<double> cv(coord, coord + n_coord), res;
    std::vector<int> vv(vis, vis + n_vis);
    const int n = traverse_euclidean(cv, guide, Mg, vv, alignment, anchor, res);
    if (n < 0) return TDLO_E_TRAVERSE;
    std::memcpy(out, res.data(), sizeof(double) * res.size());
    return n;
}

}  // extern "C"

// =================================================================================================
// class trackdlo (trackdlo/include/trackdlo.h:53-130)
// =================================================================================================
struct tdlo_tracker {
    tdlo_ctx *ctx;
    int slot;
    int M;
    int precision = TDLO_PREC_F32;
    // members of the reference class, trackdlo.h:104-121
    std::vector<double> Y;              // M x 3 column-major
    std::vector<double> guide_nodes;    // Mg x 3 column-major
    int Mg;
    double sigma2, beta, beta_pre_proc, lambda, lambda_pre_proc, alpha, k_vis, mu, tol, lle_weight;
    int max_iter;
    std::vector<double> geodesic_coord;
    std::vector<double> priors;         // K x 4 row-major
    double visibility_threshold;
    std::vector<double> trav1, trav2, trav2r;      // scratch of the priors' formation, kept across frames (their growth was a dozen reallocations per frame,
    std::vector<int> vis_ext;                      // on the host path between the two registrations)
    std::vector<int> frame_vis, frame_vis_ext;     // tdlo_tracker_frame_from_depth: the frame's visible sets
    std::vector<double> frame_dist;                // ... and the nodes' distances to the frame's cloud
    bool painter_on = false;                       // tdlo_tracker_set_self_occlusion: frame_from_depth applies trackdlo_node.cpp:279-343 (off by default)
    double painter_proj[12] = {0};
    int painter_width = 0;
    int last_iters[2] = {0, 0};         // iterations the two registrations of the previous frame took: how many are enqueued before the host looks (tdlo_ctx::iter_hint;
                                        // the larger of the last two frames' counts was tried instead: no difference)
};

extern "C" {

tdlo_tracker *tdlo_tracker_create(tdlo_ctx *ctx, int slot, int M, double visibility_threshold, double beta, double lambda,
                                  double alpha, double k_vis, double mu, int max_iter, double tol, double beta_pre_proc,
                                  double lambda_pre_proc, double lle_weight) {
    if (!ctx || slot < 0 || slot >= (int)ctx->slots.size() || M < 1) return nullptr;
    tdlo_tracker *t = new tdlo_tracker();
    t->ctx = ctx; t->slot = slot; t->M = M;
    t->Y.assign(3 * (size_t)M, 0.0);                       // trackdlo.cpp:43
    t->guide_nodes = t->Y; t->Mg = M;                      // :45
    t->sigma2 = 0.0;                                       // :46
    t->visibility_threshold = visibility_threshold; t->beta = beta; t->beta_pre_proc = beta_pre_proc;
    t->lambda = lambda; t->lambda_pre_proc = lambda_pre_proc; t->alpha = alpha; t->lle_weight = lle_weight;
    t->k_vis = k_vis; t->mu = mu; t->max_iter = max_iter; t->tol = tol;
    return t;
}

tdlo_tracker *tdlo_tracker_create_default(tdlo_ctx *ctx, int slot, int M) {
    // trackdlo.cpp:10-28
    return tdlo_tracker_create(ctx, slot, M, /*visibility_threshold*/ 0.02, /*beta*/ 5.0, /*lambda*/ 1.0, /*alpha*/ 0.0,
                               /*k_vis*/ 0.0, /*mu*/ 0.05, /*max_iter*/ 50, /*tol*/ 0.00001, /*beta_pre_proc*/ 3.0,
                               /*lambda_pre_proc*/ 1.0, /*lle_weight*/ 1.0);
}

void tdlo_tracker_destroy(tdlo_tracker *t) { g_prof.report(); delete t; }

int tdlo_tracker_set_precision(tdlo_tracker *t, int precision) {
    if (!t || (precision != TDLO_PREC_F32 && precision != TDLO_PREC_F64)) return TDLO_E_INVALID;
    t->precision = precision;
    return TDLO_OK;
}

int tdlo_tracker_initialize_nodes(tdlo_tracker *t, const double *Y_init) {
    if (!t || !Y_init) return TDLO_E_INVALID;
    t->Y.assign(Y_init, Y_init + 3 * (size_t)t->M);
    t->guide_nodes = t->Y; t->Mg = t->M;
    return TDLO_OK;
}

int tdlo_tracker_initialize_geodesic_coord(tdlo_tracker *t, const double *coord, int n) {
    if (!t || !coord || n < 0) return TDLO_E_INVALID;
    t->geodesic_coord.insert(t->geodesic_coord.end(), coord, coord + n);
    return TDLO_OK;
}

int tdlo_tracker_copy_state(tdlo_tracker *dst, const tdlo_tracker *src) {
    if (!dst || !src || dst->M != src->M) return TDLO_E_INVALID;
    if (dst == src) return TDLO_OK;
    tdlo_ctx *ctx = dst->ctx; const int slot = dst->slot;
    *dst = *src;                                       // every member of trackdlo.h:104-121 (and the precision switch)
    dst->ctx = ctx; dst->slot = slot;
    return TDLO_OK;
}

double tdlo_tracker_get_sigma2(const tdlo_tracker *t) { return t ? t->sigma2 : 0.0; }
void tdlo_tracker_set_sigma2(tdlo_tracker *t, double s) { if (t) t->sigma2 = s; }

int tdlo_tracker_get_tracking_result(const tdlo_tracker *t, double *out) {
    if (!t || !out) return TDLO_E_INVALID;
    std::memcpy(out, t->Y.data(), sizeof(double) * t->Y.size());
    return t->M;
}

int tdlo_tracker_get_guide_nodes(const tdlo_tracker *t, double *out, int max_rows) {
    if (!t || !out) return TDLO_E_INVALID;
    if (max_rows < t->Mg) return TDLO_E_INVALID;
    std::memcpy(out, t->guide_nodes.data(), sizeof(double) * 3 * (size_t)t->Mg);
    return t->Mg;
}

int tdlo_tracker_get_correspondence_pairs(const tdlo_tracker *t, double *out, int max_rows) {
    if (!t || !out) return TDLO_E_INVALID;
    const int K = (int)(t->priors.size() / 4);
    if (max_rows < K) return TDLO_E_INVALID;
    std::memcpy(out, t->priors.data(), sizeof(double) * t->priors.size());
    return K;
}

int tdlo_tracker_tracking_step(tdlo_tracker *t, const double *X, int N, const int *vis, int n_vis,
                               const int *vis_ext, int n_ext, const double *H_pre, tdlo_stats *stats) {
    if (!t || !vis_ext) return TDLO_E_INVALID;
    tdlo_ctx *c = t->ctx;
    const int M = t->M;
    if (n_ext <= 0 || n_ext > M) return fail(c, TDLO_E_INVALID, "visible_nodes_extended must hold 1..M indices (empty is undefined in the reference, trackdlo_node.cpp:351)");
    for (int i = 0; i < n_ext; ++i) if (vis_ext[i] < 0 || vis_ext[i] >= M) return fail(c, TDLO_E_INVALID, "visible_nodes_extended index out of range");
    for (int i = 0; i < n_vis; ++i) if (vis[i] < 0 || vis[i] >= M) return fail(c, TDLO_E_INVALID, "visible_nodes index out of range");
    t->priors.clear();                                                   // :908
    int rc = TDLO_OK;
    g_prof.start(); g_prof.base = 0;
    bool staged = false;          // this frame's cloud is in the pinned staging buffer
    if (X && N > 0 && c->cloud_direct_on && c->direct_in && c->fuse_on && N <= 16384) {
        // a cloud the fused prologue can take (up to 64 point workgroups): staged in pinned host memory, read from there by the prologue itself
        HIPCHK(c, hipSetDevice(c->device));
        Slot &s = c->slots[t->slot];
        if ((rc = flush_pending_cloud(c)) || (rc = ensure_points(c, s, N)) || (rc = ensure_cloud_pin(c, 3 * (size_t)N))) return rc;
        std::memcpy(c->cloud_pin, X, 3 * (size_t)N * sizeof(double));
        s.N0 = N; s.sorted_valid = false;
        c->cloud_pending = t->slot;
        staged = true;
    }
    else if (X) rc = set_cloud_impl(c, t->slot, X, N, false);            // X_orig by value: one upload for both registrations (X stays the caller's
                                                                         // until this function returns; the first registration's read-back waits for the copy)
    else if (c->slots[t->slot].N0 <= 0) rc = fail(c, TDLO_E_INVALID, "X is NULL and no cloud is resident in the tracker's slot");
    if (rc) return rc;
    g_prof.mark(0);

    // guide nodes = visible sub-chain (:913-921)
    const int Mg = n_ext;
    t->Mg = Mg;
    t->guide_nodes.assign(3 * (size_t)Mg, 0.0);
    if (Mg != M) { for (int i = 0; i < Mg; ++i) for (int d = 0; d < 3; ++d) t->guide_nodes[d * Mg + i] = t->Y[d * M + vis_ext[i]]; }
    else t->guide_nodes = t->Y;

    // pre-processing registration (:925-927): sigma2 copy, beta/lambda_pre_proc, include_lle = true
    tdlo_params pp{};
    pp.beta = t->beta_pre_proc; pp.lambda = t->lambda_pre_proc; pp.lle_weight = t->lle_weight; pp.mu = t->mu;
    pp.max_iter = t->max_iter; pp.tol = t->tol; pp.include_lle = 1; pp.alpha = 0; pp.k_vis = 0; pp.visibility_threshold = 0.01;
    pp.precision = t->precision;
    double sigma2_pre = t->sigma2;
    tdlo_stats st_pre{}, st_main{};
    // the main registration's parameters (:998): include_lle = false, priors, alpha, visible_nodes_extended, k_vis
    tdlo_params mp{};
    mp.beta = t->beta; mp.lambda = t->lambda; mp.lle_weight = t->lle_weight; mp.mu = t->mu; mp.max_iter = t->max_iter;
    mp.tol = t->tol; mp.include_lle = 0; mp.alpha = t->alpha; mp.k_vis = t->k_vis; mp.visibility_threshold = t->visibility_threshold;
    mp.precision = t->precision;
    // every node visible: both registrations start from t->Y, and the main one's node-side set-up depends on nothing the pre-processing one
    // produces -- the pre-processing registration's prologue is asked to do it as well (tdlo_ctx::PairNext; run_frames takes it up if it can)
    if (c->lle_next_on && M <= 256 && (rc = ensure_hb_next(c, c->slots[t->slot], M))) return rc;
    c->pair.state = 0; c->pair.ahead = false;
    if (Mg == M && c->pair_on && c->sort_reuse && c->late_on && c->direct_in && c->fuse_on && check_params(c, M, &mp) == 0) {
        tdlo_ctx::PairNext &pn = c->pair;
        pn.slot = t->slot; pn.M = M; pn.n_vis = n_ext; pn.sigma2 = t->sigma2; pn.p = mp; pn.Y = t->Y;
        pn.state = 1;
    } else if (Mg != M && staged && c->ahead_on && c->pair_on && c->late_on && c->direct_in && c->fuse_on && c->spec_on && c->mbox_on && !c->timing && M <= 256 &&
               mp.max_iter > 0 && check_params(c, M, &mp) == 0) {
        // hidden nodes: the registrations start from different node sets, but the main one's prologue, k_dmin and first E-step need nothing the
        // pre-processing one produces -- they are asked to run beside it (PairNext::ahead; the pre-processing registration's run_frames launches
        // them).  What they will point at is sized here, before anything is launched: the larger registration's mailbox and late-priors buffer
        const NodeCarve ncm(M);
        if ((rc = ensure_mbox(c, ncm.readback + 4)) || (rc = ensure_late(c, 4 * (size_t)M + 2))) return rc;
        tdlo_ctx::PairNext &pn = c->pair;
        pn.slot = t->slot; pn.M = M; pn.n_vis = n_ext; pn.sigma2 = t->sigma2; pn.p = mp; pn.Y = t->Y; pn.cloud_in_pin = true;
        pn.state = 3;
    }
    g_prof.mark(1);
    c->iter_hint = t->last_iters[0]; c->iter_hint_next = t->last_iters[1];
    rc = tdlo_cpd_lle_resident(c, t->slot, t->guide_nodes.data(), Mg, &sigma2_pre, &pp, nullptr, 0, nullptr, 0, H_pre, &st_pre);
    t->last_iters[0] = rc ? 0 : st_pre.iters;
    if (stats) stats[0] = st_pre;
    if (rc) {        // (an early error return may not have waited for the cloud's copy yet; what was launched ahead is told to leave and drained)
        c->pair.state = 0; c->pair.ahead = false; c->iter_hint = 0; c->iter_hint_next = 0; spec_abort(c); (void)flush_pending_cloud(c); (void)hipStreamSynchronize(c->stream);
        if (c->twin_busy && c->stream2[0]) { (void)hipStreamSynchronize(c->stream2[0]); c->twin_busy = false; }
        return rc;
    }

    std::vector<int> &ve = t->vis_ext;
    ve.assign(vis_ext, vis_ext + n_ext);
    std::vector<double> &p1 = t->trav1, &p2 = t->trav2;
    const double *guide = t->guide_nodes.data();
    auto trav = [&](int alignment, int anchor, std::vector<double> &out) {
        return traverse_euclidean(t->geodesic_coord, guide, Mg, ve, alignment, anchor, out);
    };
    const char *oob = "traverse_euclidean: the reference would index out of bounds for these visible nodes";
    // The priors of the main registration (:929-995), formed from the pre-processing registration's result.  As a function: the main registration's
    // set-up kernel does not need them, so run_frames launches it first and calls this while it runs (LatePriors).
    const LatePriors form_priors = [&](const double *&lp, int &lk) -> int {
    if (Mg == M) {                                                       // all visible / minor occlusion (:929-957)
        const int n1 = trav(0, -1, p1), n2 = trav(1, -1, p2);
        if (n1 <= 0 || n2 <= 0) return fail(c, TDLO_E_TRAVERSE, oob);
        // p2 runs tail -> head; bring it to ascending order (:942)
        std::vector<double> &r2 = t->trav2r;
        r2.resize(p2.size());
        for (int i = 0; i < n2; ++i) std::memcpy(&r2[4 * i], &p2[4 * (n2 - 1 - i)], 4 * sizeof(double));
        t->priors.reserve(t->priors.size() + 4 * (size_t)M);
        for (int i = 0; i < M; ++i) {
            const long long j2 = (long long)i - ((long long)M - n2);     // unsigned in the reference: negative == huge
            const bool j2_ok = j2 >= 0 && j2 < n2;
            if ((double)i < r2[0] && i < n1) {
                t->priors.insert(t->priors.end(), &p1[4 * i], &p1[4 * i] + 4);
            } else if ((double)i > p1[4 * (n1 - 1)] && j2_ok) {
                t->priors.insert(t->priors.end(), &r2[4 * j2], &r2[4 * j2] + 4);
            } else {
                if (i >= n1 || !j2_ok) return fail(c, TDLO_E_TRAVERSE, oob);
                for (int k = 0; k < 4; ++k) t->priors.push_back((p1[4 * i + k] + r2[4 * j2 + k]) / 2.0);    // :954
            }
        }
    } else if (ve.front() == 0 && ve.back() == M - 1) {                  // mid-section occluded (:958-967)
        if (trav(0, -1, p1) < 0 || trav(1, -1, p2) < 0) return fail(c, TDLO_E_TRAVERSE, oob);
        t->priors = p1;
        t->priors.insert(t->priors.end(), p2.begin(), p2.end());
    } else if (ve.front() == 0) {                                        // tail occluded (:968-973)
        if (trav(0, -1, p1) < 0) return fail(c, TDLO_E_TRAVERSE, oob);
        t->priors = p1;
    } else if (ve.back() == M - 1) {                                     // head occluded (:974-979)
        if (trav(1, -1, p1) < 0) return fail(c, TDLO_E_TRAVERSE, oob);
        t->priors = p1;
    } else {                                                             // both ends occluded (:980-995)
        int anchor = -1; double moved = 999999;
        for (int i = 0; i < n_vis && i < Mg; ++i) {                      // reference pairs visible_nodes[i] with guide row i
            double s = 0;
            for (int d = 0; d < 3; ++d) { const double e = t->Y[d * M + vis[i]] - t->guide_nodes[d * Mg + i]; s += e * e; }
            const double dd = std::sqrt(s);
            if (dd < moved) { moved = dd; anchor = i; }
        }
        if (trav(2, anchor, p1) < 0) return fail(c, TDLO_E_TRAVERSE, oob);
        t->priors = p1;
    }
    lp = t->priors.data(); lk = (int)(t->priors.size() / 4);
    return 0;
    };

    // main registration (:998): include_lle = false, priors, alpha, visible_nodes_extended, k_vis (mp, above)
    // prior indices may be fractional after the averaging at :954; the reference truncates (:247)
    g_prof.mark(13); g_prof.base = 6;
    HIPCHK(c, hipSetDevice(c->device));
    c->iter_hint = t->last_iters[1];
    rc = run_frames(c, 1, &t->slot, t->Y.data(), M, &t->sigma2, &mp, nullptr, 0, vis_ext, n_ext, nullptr, &st_main, &form_priors);
    if (rc == TDLO_E_FUSE) {
        // the main registration's own fused prologue (hidden nodes: in the slot, or launched ahead in the twin slot) was abandoned: Y_, sigma2_ and the
        // cloud in the slot are untouched -- once more on the three-kernel route, the priors formed again from the same guide nodes
        fuse_fallback(c);
        t->priors.clear();
        rc = run_frames(c, 1, &t->slot, t->Y.data(), M, &t->sigma2, &mp, nullptr, 0, vis_ext, n_ext, nullptr, &st_main, &form_priors);
        if (rc == TDLO_E_FUSE) rc = fail(c, TDLO_E_HIP, "the fused prologue reported a time-out on the route that does not use it");
    }
    t->last_iters[1] = rc ? 0 : st_main.iters;
    c->iter_hint = 0; c->iter_hint_next = 0;
    c->pair.state = 0; c->pair.ahead = false; spec_abort(c);
    if (c->cloud_pending >= 0) { const int frc = flush_pending_cloud(c); if (!rc) rc = frc; }
    if (stats) stats[1] = st_main;
    return rc;
}

// One frame of the ROS node's callback, images in, nodes out (trackdlo_node.cpp:195-369): depth image + mask -> cloud -> voxel grid and the visibility
// pre-pass of the tracker's nodes (tdlo_depth_to_cloud_visibility: one launch), then tracking_step on the cloud resident in the tracker's slot.
int tdlo_tracker_frame_from_depth(tdlo_tracker *t, const unsigned short *depth, const unsigned char *mask, int rows, int cols,
                                  double fx, double fy, double cx, double cy, double leaf_size, double d_vis,
                                  int *visible_nodes, int *n_vis, int *visible_nodes_extended, int *n_vis_ext,
                                  int *n_out, int *n_raw_out, tdlo_stats *stats) {
    if (!t) return TDLO_E_INVALID;
    tdlo_ctx *c = t->ctx;
    const int M = t->M;
    if (t->geodesic_coord.size() != (size_t)M) return fail(c, TDLO_E_INVALID, "the tracker has no nodes / geodesic coordinates yet (tdlo_tracker_initialize_*)");
    std::vector<int> &ve = t->frame_vis_ext, &v = t->frame_vis;
    v.resize(M); ve.resize(M);
    int nv = 0, ne = 0, n = 0;
    t->frame_dist.resize(M);
    int rc = tdlo_depth_to_cloud_visibility(c, t->slot, depth, mask, rows, cols, fx, fy, cx, cy, leaf_size, t->Y.data(), M, t->visibility_threshold, d_vis,
                                            t->geodesic_coord.data(), t->frame_dist.data(), v.data(), &nv, ve.data(), &ne, &n, n_raw_out);
    if (!rc && t->painter_on && n > 0) {
        // the callback's self-occlusion test (trackdlo_node.cpp:279-343) between the distance pre-pass and the gap fill, when the tracker was given a
        // projection matrix for it (tdlo_tracker_set_self_occlusion; off by default: parity against OpenCV's rasteriser is unpinned)
        std::vector<int> pv, pe;
        self_occlusion_visible(t->Y.data(), M, t->painter_proj, t->painter_width, t->frame_dist.data(), t->visibility_threshold, pv);
        fill_visible_gaps(pv, t->geodesic_coord.data(), d_vis, pe);
        nv = (int)pv.size(); ne = (int)pe.size();
        std::copy(pv.begin(), pv.end(), v.begin()); std::copy(pe.begin(), pe.end(), ve.begin());
    }
    if (n_out) *n_out = n;
    if (n_vis) *n_vis = nv;
    if (n_vis_ext) *n_vis_ext = ne;
    if (visible_nodes) std::copy(v.begin(), v.begin() + nv, visible_nodes);
    if (visible_nodes_extended) std::copy(ve.begin(), ve.begin() + ne, visible_nodes_extended);
    if (rc) return rc;
    // (the reference's callback indexes visible_nodes[size() - 1] without a test, trackdlo_node.cpp:351-361: a frame without a cloud or without a visible node is an error here)
    if (n == 0) return fail(c, TDLO_E_EMPTY, "the mask selects no pixel: no cloud for this frame");
    if (ne == 0) return fail(c, TDLO_E_EMPTY, "no node within the visibility threshold of the cloud (the reference's callback is undefined here, trackdlo_node.cpp:351)");
    return tdlo_tracker_tracking_step(t, nullptr, 0, v.data(), nv, ve.data(), ne, nullptr, stats);
}

// trackdlo_node.cpp:279-343 on the host (tdlo_host.cpp, self_occlusion_visible): O(M^2) integer tests, no kernel warranted
int tdlo_self_occlusion_visible(const double *Y, int M, const double proj[12], int dlo_pixel_width, const double *node_dist, double visibility_threshold,
                                int *visible_nodes, int *n_vis) {
    if (!Y || !proj || !node_dist || !visible_nodes || !n_vis || M < 1 || dlo_pixel_width < 1) return TDLO_E_INVALID;
    std::vector<int> v;
    self_occlusion_visible(Y, M, proj, dlo_pixel_width, node_dist, visibility_threshold, v);
    *n_vis = (int)v.size();
    std::copy(v.begin(), v.end(), visible_nodes);
    return TDLO_OK;
}

// trackdlo_node.cpp:345-360
int tdlo_extend_visible_nodes(const int *visible_nodes, int n_vis, const double *geodesic_coord, double d_vis, int *visible_nodes_extended, int *n_vis_ext) {
    if (n_vis < 0 || (n_vis > 0 && (!visible_nodes || !geodesic_coord)) || !visible_nodes_extended || !n_vis_ext) return TDLO_E_INVALID;
    std::vector<int> v(visible_nodes, visible_nodes + n_vis), e;
    std::sort(v.begin(), v.end());                                  // :346
    fill_visible_gaps(v, geodesic_coord, d_vis, e);
    *n_vis_ext = (int)e.size();
    std::copy(e.begin(), e.end(), visible_nodes_extended);
    return TDLO_OK;
}

int tdlo_tracker_set_self_occlusion(tdlo_tracker *t, const double *proj, int dlo_pixel_width) {
    if (!t) return TDLO_E_INVALID;
    if (!proj) { t->painter_on = false; return TDLO_OK; }
    if (dlo_pixel_width < 1) return fail(t->ctx, TDLO_E_INVALID, "tdlo_tracker_set_self_occlusion: dlo_pixel_width >= 1");
    std::copy(proj, proj + 12, t->painter_proj);
    t->painter_width = dlo_pixel_width; t->painter_on = true;
    return TDLO_OK;
}

}  // extern "C"
