// tdlo_internal.h -- structures shared by the host driver and the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace tdlo {

constexpr int kBlock = 256;          // workgroup size of the point-parallel kernels (4 wave64)
constexpr int kWave = 64;
constexpr int kChunk = 64;           // nodes handled per lane-transposed tile
constexpr int kPStride = 65;         // LDS row stride of the 64 x 64 transposition tile (conflict-free)
constexpr int kPtsStride = 68;       // E-step: a wave's 64 normalised points in LDS, one entry of padding after every 16
constexpr int kAccRows = 8;          // replica rows of the E-step's fixed-point accumulators (workgroup b adds to row b % 8: spreads the atomics)
constexpr int kTileRows = 24;        // rows of the E-step's transposition tile: the node window is processed in chunks of this many nodes ...
constexpr int kTileRows32 = 16;      // ... fp32 E-step, chains up to 64 nodes: 16 (4 KB of LDS per wave; with twice the workgroups the GPU holds more waves per
                                     // SIMD: 32-frame batch 25.6 -> 23.2 us per E-step, N = 2 000 000 27.9 -> 26.5 us; fp64 / long chains lose 5 % with it)
#ifndef TDLO_TILE_ROWS_LONG
#define TDLO_TILE_ROWS_LONG 24         // (chains beyond 64 nodes; -DTDLO_TILE_ROWS_LONG=n builds the variants scripts/gpu_c5.py compares)
#endif
template <typename T> __host__ __device__ constexpr int tile_rows(int nch) { return nch == 1 ? (sizeof(T) == 4 ? kTileRows32 : kTileRows) : TDLO_TILE_ROWS_LONG; }
constexpr int kMaxNodes = 1024;      // E-step template covers ceil(M/64) in {1,2,4,8,16}
constexpr int kChainLdsMaxNodes = 512;   // the four-direction chain smoother and the banded LLE solve keep a record per node in LDS (161 / 157 KB at 512 nodes);
                                         // longer chains: k_mstep_chain_long (one direction, compact records) / the one-workgroup dense elimination
constexpr int kLdsSolveMaxM = 128;   // M-step keeps [A|B] in LDS up to this M
constexpr int kMaxXchRanks = 8;      // ranks of the one-shot N-split exchange (one node: 8 GPUs)
// IterState::status of a registration whose fused prologue was abandoned at its grid barrier (tdlo_device.hip, fuse_wait).  Internal: the entry
// points repeat the call on the three-kernel route (tdlo_api.cpp, fuse_fallback); no caller ever sees the code.
#define TDLO_E_FUSE (-100)

// Mutable per-iteration state of one registration, device resident (trackdlo.cpp:275-438 loop state).
struct IterState {
    double sigma2;      // current sigma2
    double k2;          // -log2(e) / (2 sigma2): membership exponent scale (:354)
    double c_norm;      // denominator constant c (:300) or c' (:378) for the current sigma2 and N
    double crit;        // last value of sum_m |dY_m| / M (:424)
    double Np;          // last Np (:388)
    double sum_d2;      // sum over kept points and nodes of |Y0_m - x_n|^2 (:263-273)
    double Nc;          // N used in the denominator constant c (global N when the cloud is split)
    double rwin32, rwin64;  // E-step's node window, fp32 / fp64 mode: E / |k2| = E 2 ln2 sigma2 (a squared arc length), E = FrameDev::win_e32 / win_e64 bits --
                            // a membership is dropped once its exponent lies E below that of the point's nearest node (k_estep)
    int N;              // points kept by the prune (:195)
    int it;             // iterations completed
    int done;           // 1: remaining kernels of the loop are no-ops
    int converged;      // value cpd_lle returns (:440)
    int status;         // 0 or TDLO_E_*
    int retry_pending;  // set by the finishing workgroup of a multi-CU M-step after a timed-out hand-off: the one-workgroup kernel that follows redoes the iteration
    int retries;        // iterations whose multi-CU elimination hit a hand-off time limit and were redone in one workgroup
    int sh_boost;       // fp64 mode: extra binary digits of the E-step's fixed-point sums R (and twice as many of Q) in the iteration that follows -- set with sigma2
                        // (set_iter_consts), used by that E-step and by the M-step that reads its sums
};

// Immutable-after-setup description of one frame's registration.
struct FrameDev {
    // sizes / parameters (trackdlo.h:80-94)
    int N0, M, ldx, nblkE;
    int max_iter, include_lle, has_priors, vis_branch;
    int precision, nprune_blocks, eb, wide_tile;    // eb: E-step workgroup size (256 or 512); wide_tile: 64-row transposition tile (one frame of moderate size)
    double tol, beta, lambda, lle_weight, mu, alpha, k_vis, vis_thr, sigma2_in;
    // cloud
    const double *Xraw;     // N0 x 3 column-major as uploaded
    void *Xs;               // pruned, centred SoA in compute precision: x[ldx] y[ldx] z[ldx]
    unsigned short *bucket; // N0: nearest node of a kept point, 0xffff = pruned
    int *hist;              // nprune_blocks x M: kept points per (block, nearest node), written by k_prune_pass1
    int *hist_off;          // nprune_blocks x M: where each (block, node) run starts in the sorted cloud, written by k_setup (a separate array: the
                            // scan's loads need not wait for its own stores), read by k_prune_scatter
    double *blksum;         // per prune block sum of d2 over kept points
    // nodes
    const double *Yin;      // M x 3 as given by the caller
    double *ctr;            // 3: centring offset (centroid of Yin)
    double *Y;              // M x 3 centred, current (fp64)
    double *Y0;             // M x 3 centred, start of call
    void *nodes;            // M x {x,y,z,coord} in compute precision (float4 / double4)
    double *coord;          // M cumulative arc length (:214-223)
    double *G;              // M x M kernel (:233)
    double *chain;          // (M + 1) x 8: state-space form of G for the chain smoother (tdlo_mstep_chain.hip): [0] = {sf2, s^2 sf2}, [i] = link between nodes i-1 and i {Phi (4), Q (3)}
    const double *H;        // M x M LLE regulariser (host supplied; the dense LLE M-steps only) or nullptr
    const double *Hb;       // the banded LLE M-step: H's 13 diagonals, Hb[13 i + u] = H(i, i - 6 + u) (host supplied)
    double *HG;             // M x M  H*G   (include_lle)
    double *HY0;            // M x 3  H*Yin (include_lle)
    const double *aJ;       // M: alpha * J_mm (:240-260, :406)
    const double *aYd;      // M x 3: alpha * (Y_extended - Y0) (:407)
    unsigned long long *dminbits;  // M: per-node min squared distance, as ordered bits
    long long *acc;         // [2 (iteration parity)][kAccRows][4M+2]: the E-step's sums [P1 | Rx | Ry | Rz | Q] in 64-bit fixed point, added with
                            // integer atomics (order-independent, i.e. reproducible); the M-step of iteration `it` reads parity it & 1 and zeroes the other
    int acc_sh[3];          // binary exponents of the fixed point: value = integer * 2^-sh for P1 / R / Q
    double acc_lim[3];      // largest |value| one conversion may take for P1 / R / Q: keeps the conversion exact (|v 2^sh| < 2^51) and the totals
                            // of all batches below 2^62; a contribution beyond it (or NaN) ends the registration with TDLO_E_NUMERIC
    int force_timeout_it;   // test hook (environment TDLO_MCU_FORCE_TIMEOUT=k): iteration k of the multi-CU M-steps behaves as if a hand-off timed out; -1 = off
    int prune_tiles;        // 256-point tiles one prune workgroup handles (1 up to 262 144 points)
    int reuse_sorted;       // 1: the slot's pruned, centred, node-sorted cloud was made for exactly these nodes by the previous registration (the two
                            // registrations of one tracking_step when every node is visible, trackdlo.cpp:913-927 / :998): prune and sort are skipped
    double *keep;           // 2 doubles that outlive a registration: kept points and sum of d2 of the slot's sorted cloud (for reuse_sorted)
    int need_G;             // the M x M kernel matrix is built at setup (dense M-steps: dense LLE path, comparators); the chain smoother and the banded LLE M-step do not read it
    int mstep_dense;        // registrations without the LLE term: 0 the chain smoother, 1 the dense eliminations (comparators) -- decided when the frame is prepared
    int h_banded;           // the dense LLE M-steps: H is zero beyond +-6 nodes (the library's own always is; an H_override is checked by the host): H G and H Y0 are formed from the band
    int lle_band;           // registrations with the LLE term: 1 the banded L D L^T in the chain's state (tdlo_mstep_band.hip), 0 the dense pivoted eliminations
    double *band;           // lle_band: one 16-double column record per unknown of the state-space system (band_record_doubles(M)), written by k_setup
    double band_s2_max;     // lle_band, fp64 mode: above this sigma2 the banded M-step reports TDLO_E_NUMERIC and the call is repeated on the dense pivoted kernels
                            // (the state precision's entries, rounded to fp64, no longer hold the mode's 1e-9 m: prepare_frame); fp32 mode: never
    double *sums;           // 4M+2 reduced sums (N-split interface)
    double *Ascr;           // (M x (M+3)) scratch for the M-step when it does not fit LDS
    double *Yout;           // M x 3 uncentred result
    unsigned long long *dbg; // 64 shader-clock stamps written by the M-step (tdlo_debug_stamps)
    unsigned *sync;         // 256 words, zeroed when the slot is created: generation / arrivals / flags of the multi-CU M-steps' hand-offs (words 0..47, 128..255),
                            // k_dmin's ticket (8), the fused prologue's grid barrier (100: arrivals, 101: epoch flag)
    IterState *st;
    // Results straight into pinned host memory (one frame per call, the one-workgroup M-steps k_mstep_chain / k_mstep_band): the M-step that
    // finishes the registration copies [Yout | IterState] (the layout of the slot's read-back block) to host_out and then raises host_prog;
    // every other M-step only reports its progress there.  The host waits on that word instead of a device-to-host copy and a stream
    // synchronisation (a 4-6 us blit kernel, a dependent-dispatch gap and the wake-up of the blocking wait per registration).
    //   host_prog = epoch << 32 | done << 31 | iterations completed.     nullptr: off (batches, N-split, the dense M-steps)
    // The E-step's node window: node m is left out for a wave's 64 points when its membership is below 2^-E of EVERY point's largest one (that
    // of its nearest node).  E = 36 bits in fp32 mode, 66 in fp64 mode: 12 bits below the arithmetic's own rounding (24 / 53 mantissa bits), i.e.
    // what is dropped could not have changed a sum's last bit by more than 2^-12 of an ulp.  TDLO_WINDOW=exact: 154 / 1100 bits -- only memberships
    // that are exactly zero in the arithmetic (fp32 flushes below 2^-149, fp64 below 2^-1075) are left out, the rule of rounds 1-3 (comparator).
    double win_e32, win_e64;
    // Correspondence priors that arrive AFTER the set-up kernel was launched (tracking_step's second registration: the host forms them from the
    // first registration's result while the set-up kernel already runs): alpha J (M doubles) and alpha (Y_ext - Y0) (3 M doubles) in pinned host
    // memory; the E-step's workgroup 0 copies them to aJ / aYd in iteration 0, the M-step reads them there.  nullptr: they came with the upload block.
    // (a registration whose first iteration starts from given sums -- from_sums == 1 -- has no E-step in front of its first M-step: the chain
    //  smoother then reads them from here itself and keeps them in aJ / aYd for the iterations that follow)
    const double *late_aJ, *late_aYd;
    double *host_out;
    unsigned long long *host_prog;
    unsigned host_epoch;
    int host_report_it;     // an M-step that does not finish the registration reports its progress only when it has completed this iteration (0: never)
    // one-shot exchange of the N-split (tdlo_xch_*, tdlo_split_run without a communicator): every rank's inbox as a device
    // pointer valid on THIS device (own inbox included); xch_nranks == 0: no exchange
    unsigned long long *xch_inbox[kMaxXchRanks];
    int xch_rank, xch_nranks, xch_mcap;
    int xch_self;           // xch_nranks == 1: 0 (default) a lone rank has nobody to exchange with -- the kernels skip the exchange; 1 (TDLO_XCH_SELF=1): it writes to and
                            // reads from its own inbox like any rank of a larger group (the cost of the exchange proper on one GPU; tests)
    unsigned xch_epoch;     // tag of this registration: flags carry (epoch << 32 | iteration + 1)
    // ---- tracking_step's short cuts (round 4, second step).  At the END of the descriptor: the offsets of everything above are those of the
    //      kernels measured before (with fields inserted in the middle C2 came out 0.8 % slower in an A/B on one box, 13.80 against 13.62 us per iteration; with them here: level)
    // tracking_step with every node visible (tdlo_api.cpp, PairNext): the first E-step of the main registration would repeat the first E-step of
    // the pre-processing registration to the bit -- same cloud, same nodes, sigma2, mu, precision, no visibility term, and the sums are integers.
    // The pre-processing registration's M-step of iteration 0 stores the 4M + 1 sums it has read here (the main registration's `sums`), and that
    // registration starts with its M-step.  nullptr: off.
    double *pair_sums;
    // ... and that first M-step is LAUNCHED right behind the pre-processing registration's first iteration, before the host has the priors it
    // needs (they come out of the pre-processing registration's result through traverse_euclidean): it leaves at once unless that registration
    // (spec_prev) has finished without an error, and otherwise waits for the host to raise spec_flag (pinned host memory) to spec_epoch << 32 | 1
    // -- priors are in late_aJ / late_aYd -- or | 2 -- leave, nothing is touched.  Gives up after 2 s (the host then finds the stream drained
    // without a report and makes up for it: run_frames, mbox_done).  What it saves is the launch: enqueue + dispatch latency, ~6 us between the two registrations.  nullptr: off.
    const unsigned long long *spec_flag;
    const IterState *spec_prev;
    unsigned spec_epoch;
    // The M-step that finishes this registration (k_mstep_chain, chains of up to 256 nodes) then forms the 13 diagonals of the LLE regulariser
    // H = (I - L)^T (I - L) (trackdlo.cpp:236-237) of the nodes it leaves behind, here -- the host's values bit for bit (tdlo_lle_dev.h): the
    // next frame's pre-processing registration starts from exactly these nodes when every node is visible.  nullptr: off.
    double *lle_next;
    const double *Xhost;    // k_prologue only: the cloud still in PINNED HOST memory (N0 x 3 column-major) -- its point workgroups read it from there
                            // and put it in Xraw themselves: no host-to-device copy in front of the frame's first kernel.  nullptr: Xraw holds it
    // tracking_step with hidden nodes (tdlo_api.cpp, PairNext::ahead): the main registration's prologue, k_dmin and first E-step run on a second
    // stream BESIDE the pre-processing registration (they depend on nothing it produces), and its first M-step waits behind them for the priors
    // (spec_flag with spec_prev == nullptr: it waits for the host's word only).  That E-step ran before the priors existed and has not copied
    // them: 1 = the chain smoother reads late_aJ / late_aYd itself although it starts from the accumulators (from_sums == 0) and keeps them in
    // aJ / aYd for the iterations that follow; the E-step leaves them alone
    int late_mstep;
    // 0: k_estep (one point per lane, tdlo_device.hip); 8 / 16: k_estep2 (two points per lane, tdlo_estep2.hip) with that many tile rows -- clouds
    // and batches that fill the GPU, fp32 mode, chains of 8 .. 64 nodes (prepare_frame / run_frames; the same value in every frame of a launch)
    int estep2;
    // fp64 mode: 1 = the sums' resolution does NOT follow sigma (IterState::sh_boost stays 0): the call is the repeat of one whose E-step refused a share
    // under the finer limits (run_frames: a registration that the coarse limits of rounds 1-4 would have served must not fail on the boost)
    int acc_boost_off;
    // The spin-ahead loop of ONE frame (round 6 experiment, TDLO_SPIN_AHEAD=1; run_frames): the E-steps go to a second stream, the M-steps stay on the
    // first, everything is enqueued up front, and the kernels order themselves through two words of `sync` -- E-step k+1 is dispatched while M-step k still
    // runs, requests its first points and parks on word 110 until that M-step stores its tag there; M-step k+1 is dispatched behind M-step k, requests
    // everything that does not come from the E-step and parks on word 111 until the E-step's workgroups have all added themselves to it.  The tags are
    // the host's running counts (unsigned, compared for equality: no reset between registrations).  Every wait is bounded (2 s -> TDLO_E_EXCHANGE).
    // spin_on: 0 off; 1 this launch takes part.  E-step: waits for word 110 == spin_wait unless spin_first; adds 1 per workgroup to word 111.
    // M-step: waits for word 111 == spin_wait; stores spin_signal to word 110.
    int spin_on, spin_first;
    unsigned spin_wait, spin_signal;
    // fp64 E-step, chains beyond 64 nodes: a batch whose node window holds at least this many nodes (and at most 320) goes lane = node (tdlo_estep_wide.h;
    // TDLO_ESTEP_WIDE=0: never -- the comparator; =n: from n nodes on)
    int estep_wide_min;
    // how many of the kAccRows replica rows of the accumulators the E-step's workgroups spread their atomics over (workgroup b adds to row b & (acc_rows - 1);
    // 2, 4 or 8; 0 reads as 8): the unused rows stay zero, so a reader may add up all kAccRows (exact either way) -- the chain M-step, whose first memory
    // round trip is on every iteration's critical path, requests only the used ones (choose_acc_rows in tdlo_api.cpp has the measurements)
    int acc_rows;
};
constexpr int kSpinWordM = 110, kSpinWordE = 111;

// Inbox layout in 64-bit words (R ranks, node capacity Mc); rank r writes the [r] entries of every peer's inbox:
//   flags: init [R] | dmin [2][R] | sums [2][R]      (parity = iteration & 1)
//   init [R][2] (kept points, sum d2)  |  dmin [2][R][Mc]  |  sums [2][R][4 Mc + 2]
__host__ __device__ inline size_t xch_off_flag_init(int R) { (void)R; return 0; }
__host__ __device__ inline size_t xch_off_flag_dmin(int R) { return (size_t)R; }
__host__ __device__ inline size_t xch_off_flag_sums(int R) { return 3 * (size_t)R; }
__host__ __device__ inline size_t xch_off_init(int R) { return (5 * (size_t)R + 7) & ~(size_t)7; }
__host__ __device__ inline size_t xch_off_dmin(int R, int Mc) { (void)Mc; return xch_off_init(R) + 2 * (size_t)R; }
__host__ __device__ inline size_t xch_off_sums(int R, int Mc) { return xch_off_dmin(R, Mc) + 2 * (size_t)R * Mc; }
__host__ __device__ inline size_t xch_words(int R, int Mc) { return xch_off_sums(R, Mc) + 2 * (size_t)R * (4 * (size_t)Mc + 2); }

// Banded LLE M-step (tdlo_mstep_band.hip): how the 2M unknowns of the state-space system are dealt out.  One direction (short chains, and
// chains whose records would not fit the LDS twice): wave 0 eliminates unknowns 0, 1, 2, ... in whole chunks of 13 steps (identity
// unknowns pad the last chunk).  TWISTED (tw): wave 0 eliminates unknowns 0 .. mT-1 from the chain's head, wave 1 the unknowns from
// the tail backwards, in its own order u~ = nUp - 1 - u (D identity unknowns in front make its count a whole number of chunks as
// well); the 12 unknowns R = mT .. mT+11 between them (half-bandwidth 12: they separate the two sides) receive both sides' Schur
// complements, and BOTH waves finish with them (one more chunk each) -- no hand-over of results, each wave back-substitutes
// its own side.  A direction's records: one per step (+ 15: the columns entering behind the last pivot, and the requests two steps ahead).
constexpr int kBandSlots = 13;
// dynamic LDS k_mstep_band may ask for: the CU's 160 KB less 1 KB for the kernel's own static allocations (256 bytes today: a plan that
// used the last kilobyte -- chains of 477 .. 481 nodes -- was refused at launch with hipErrorInvalidValue)
constexpr size_t kBandLdsLimit = 160 * 1024 - 1024;
struct BandPlan {
    int nU, tw, cT, cB, D, mT, mB, nUp, limT, sT, sB, nRecT, nRecB;
    __host__ __device__ explicit BandPlan(int M, int allow_twisted = 1) {
        nU = 2 * M;
        tw = (allow_twisted && nU >= 38) ? 1 : 0;                    // (below: two directions of 13 + 13 steps are no shorter than one)
        for (;;) {
            if (!tw) { cT = (nU + kBandSlots - 1) / kBandSlots; cB = 0; D = 0; }
            else { const int q = nU - 12, ch = (q + kBandSlots - 1) / kBandSlots; D = kBandSlots * ch - q; cT = (ch + 1) / 2; cB = ch - cT; }
            mT = kBandSlots * cT; mB = kBandSlots * cB; nUp = nU + D;
            limT = tw ? mT + 12 : nU;                                   // top's records below limT are real columns, identity from there on
            sT = kBandSlots * (cT + tw); sB = tw ? kBandSlots * (cB + 1) : 0;
            nRecT = sT + 15; nRecB = tw ? sB + 15 : 0;
            if (!tw || lds_doubles(M) * 8 <= kBandLdsLimit) break;
            tw = 0;                                                     // both directions' records do not fit the LDS: one direction
        }
    }
    // LDS of k_mstep_band (doubles): sums | scratch | per direction: dump, zeros (28 records, right in front of the records), records | merge tiles
    __host__ __device__ size_t off_S() const { return 0; }
    __host__ __device__ size_t lds_doubles(int M) const {
        size_t o = (size_t)((4 * M + 2 + 1) & ~1) + 32;
        o += (size_t)(tw ? 2 : 1) * (16 * 28 + 64 + 16 * 28) + (size_t)16 * (nRecT + nRecB);
        o += tw ? 2 * 256 : 0;
        return o;
    }
};
__host__ __device__ inline int band_rec_pos(int q) { return (q & 3) * 4 + (q >> 2); }      // position of row slot q inside a 16-double column record

// launchers implemented in tdlo_device.hip
hipError_t launch_prune_and_setup(const FrameDev *frames_dev, const FrameDev *frames_host, int F, hipStream_t s);
bool prologue_direct_ok(const FrameDev &f);
bool prologue_pair_ok(const FrameDev &f);      // the fused form itself (not the reused-sort set-up): a second registration's set-up can ride along
hipError_t launch_prologue_direct(const FrameDev *fh, const double *host_up, double *dev_up, int up_doubles, int yin_off, unsigned epoch, hipStream_t s,
                                  const FrameDev *f2 = nullptr, const double *host_up2 = nullptr, double *dev_up2 = nullptr, int up_doubles2 = 0);
hipError_t launch_iteration(const FrameDev *frames_dev, const FrameDev *frames_host, int F, hipStream_t s, int iteration = -1);      // (iteration: mstep_parity_hint)
// one iteration of the spin-ahead loop (FrameDev::spin_on): the E-step on s_e, the M-step on s_m; fh[0]'s spin fields are set per launch (the one-frame kernels take
// the descriptor by value).  ecount / mtag: the slot's running counts behind the tags.
hipError_t launch_iteration_spin(const FrameDev *frames_dev, FrameDev *frames_host, hipStream_t s_e, hipStream_t s_m, bool first, unsigned *ecount, unsigned *mtag);
hipError_t launch_iteration_timed(const FrameDev *frames_dev, const FrameDev *frames_host, int F, hipStream_t s, hipEvent_t e_start, hipEvent_t e_stop,
                                  hipEvent_t m_start, hipEvent_t m_stop, int iteration = -1);
const char *mstep_kernel_name(const FrameDev *frames_host, int F);
// tdlo_estep2.hip: the E-step with two points per lane (FrameDev::estep2)
hipError_t launch_estep2(const FrameDev *frames_dev, const FrameDev *frames_host, int F, hipStream_t s, hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr);
size_t estep2_lds_bytes(int M, int tile_rows);
// tdlo_estep2.hip: a batch's whole loop in one launch (k_batch_loop): tickets (iteration, frame, chunk), per-frame dependency counters in `ctl`
size_t batch_loop_ctl_words(int F);
hipError_t launch_batch_loop(const FrameDev *frames_dev, const FrameDev *frames_host, int F, int iters, unsigned *ctl, hipStream_t s);
hipError_t launch_estep_only(const FrameDev *frames_dev, const FrameDev *frames_host, int F, int kind, hipStream_t s);
hipError_t launch_split_setup(const FrameDev *frames_dev, const FrameDev *frames_host, hipStream_t s);
hipError_t launch_split_set_global(const FrameDev *frames_dev, double Nglob, double Sglob, hipStream_t s);
hipError_t launch_split_init_pack(const FrameDev *frames_dev, double *init2, hipStream_t s);        // [kept points, sum d2] of the shard -> device buffer (for the all-reduce)
hipError_t launch_split_set_global_dev(const FrameDev *frames_dev, const double *init2, hipStream_t s);   // ... and back, reduced
hipError_t launch_split_poll_pack(const FrameDev *frames_dev, double *out2, hipStream_t s);             // [done without error, status] of the shard (for a MIN all-reduce)
hipError_t launch_xch_init(const FrameDev *frames_dev, hipStream_t s);                             // one-shot exchange of the same two numbers
hipError_t launch_split_dmin_xch(const FrameDev *frames_dev, const FrameDev *frames_host, double *xch, int import, hipStream_t s);
hipError_t launch_debug_exp2(const double *x, double *y, int n, hipStream_t s);       // test aid: Num<double>::exp2 on an array
hipError_t launch_node_min_dist(const double *X, int N, const double *Y, int M, unsigned long long *out_bits, hipStream_t s);
hipError_t launch_node_min_dist_direct(const double *X, int N, const double *Yhost, int M, unsigned long long *state, unsigned long long *res_host, unsigned epoch, hipStream_t s);
size_t mstep_lds_bytes(int M);
// tdlo_mstep_big.hip: M-step for 60 < M <= kMaxNodes without LLE (blocked Gauss-Jordan, tableau in global memory)
hipError_t launch_mstep_big(const FrameDev *frames_dev, const FrameDev *frames_host, int F, int from_sums, bool f64, hipStream_t s);
size_t mstep_big_scratch_doubles(int M);
// M-step with the LLE term for M > kLdsSolveMaxM: 16 rows per workgroup, partial pivoting across the workgroups
hipError_t launch_mstep_pivot_mcu(const FrameDev *frames_dev, const FrameDev *frames_host, int F, int from_sums, bool f64, hipStream_t s);
bool mstep_pivot_mcu_enabled();
// tdlo_mstep_chain.hip: M-step without the LLE term as a Kalman / Rauch-Tung-Striebel smoother along the chain, any M
hipError_t launch_mstep_chain(const FrameDev *frames_dev, const FrameDev *frames_host, int F, int from_sums, bool f64, hipStream_t s);
void mstep_parity_hint(int iteration);          // (tdlo_mstep_chain.hip) the iteration this thread's next chain M-step launches belong to; -1: unknown
hipError_t launch_lle_band_debug(const double *Y_dev, int M, double *Hb_dev, hipStream_t s);      // test aid: the device form of lle_regulariser_band (tdlo_lle_dev.h), M <= 256
bool mstep_chain_enabled();
int mstep_set_dense(int on);      // returns the previous setting
// tdlo_mstep_band.hip: M-step with the LLE term as a banded L D L^T in the chain's state (f, f'), any M
hipError_t launch_mstep_band(const FrameDev *frames_dev, const FrameDev *frames_host, int F, int from_sums, bool f64, hipStream_t s);
size_t band_record_doubles(int M);
size_t mstep_band_lds_bytes(int M);
bool mstep_band_enabled();
int mstep_set_lle_dense(int on);  // 1: registrations with the LLE term take the dense pivoted eliminations (comparators); returns the previous setting
// tdlo_reg.hip: plain GMM-EM `reg` (utils.cpp:21-82); ws layout: state (8) | Y (3 M) | block partials
size_t reg_ws_doubles(int M, int nblk);
int reg_max_nodes();        // the E-step's per-wave accumulators must fit 160 KB of LDS
hipError_t launch_reg(const double *X, int N, int M, double mu, int max_iter, int nblk, double *ws, hipStream_t s);
// tdlo_cloud.hip: depth image -> cloud -> voxel grid
size_t cloud_ws_bytes(int P);
hipError_t launch_cloud_bbox(const unsigned short *depth, const unsigned char *mask, int P, int cols, const double cam[4], unsigned *bbox, void *ws, hipStream_t s);
hipError_t launch_cloud_voxels(const unsigned short *depth, const unsigned char *mask, int P, int cols, const double cam[4],
                               const int min_b[3], int mul1, int mul2, float inv_leaf, int nodown, int passes, int n,
                               void *ws, int *total_dev, int cap, double *Xraw, hipStream_t s);
// the same in ONE launch (k_cloud_fused): up to cloud_fused_max_points() masked pixels, images of up to 4095 tiles of 4096 pixels
size_t cloud_fused_ws_bytes(int P);
int cloud_fused_max_points();
bool cloud_fused_ok(int P);
hipError_t launch_cloud_fused(const unsigned short *depth, const unsigned char *mask, int P, int cols, const double cam[4], float inv_leaf, void *ws, void *fws,
                              bool first, bool team, double *Xraw, int cap, unsigned long long *res_pinned, unsigned epoch, hipStream_t s,
                              const double *vis_nodes_pinned = nullptr, int vis_M = 0, unsigned long long *vis_state = nullptr, unsigned long long *vis_out_pinned = nullptr);
int check_device_image();

}  // namespace tdlo
