"""Randomised tracking_step sequences against the oracle tracker, held to the STATED tolerances (fp64 mode 1e-9 m / 1e-7, fp32 mode 1e-5 m / 1e-3; a frame
outside them passes only when the oracle itself is measured to be that uncertain on it: scripts/fuzz_adjudicate.py):
random chain length, cloud size (down to a few dozen points), noise, inter-frame motion, occlusion pattern per frame, six frames with the state
carried over; the oracle's H of the pre-processing registration injected on both sides.  usage: python scripts/gpu_fuzz_tracker.py [n_seq] [first_seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from trackdlo_amd import synth, binding as B
from oracle import ref_cpu as oracle
import fuzz_adjudicate as FA
P = synth.LAUNCH_PARAMS
def run(n, s0=0, PREC=1, ctx=None, verbose=True):
    """n sequences from seed s0; returns dict(frames, bad, errs, skipped, worst, outside_stated, adjudicated, unexplained).  PREC 1: fp64 mode;
    0: the default fp32 mode.  bad = unexplained deviations + errors on one side only."""
    GY, GS = FA.STATED[PREC]
    tally = FA.Tally(PREC)
    own = ctx is None
    if own: ctx = B.Context(device=0, max_points=1 << 14, max_nodes=64, timing=False)      # (no stream events: the product route of a C++ caller)
    frames = bad = errs = skipped = 0
    worst = (0.0, None)
    for seed in range(s0, s0 + n):
        rng = np.random.default_rng(52000 + seed)
        M = int(rng.integers(8, 61)); N = int(rng.choice([rng.integers(40, 300), rng.integers(300, 6000)]))
        noise = float(rng.choice([0.0005, 0.0015, 0.003])); step = float(rng.choice([0.001, 0.003, 0.008]))
        Y0 = synth.nodes(M); coord = synth.geodesic_coord(Y0)
        args = (M, P["visibility_threshold"], P["beta"], P["lambda_"], P["alpha"], P["k_vis"], P["mu"], int(rng.choice([5, 30, 50])), P["tol"], P["beta_pre_proc"], P["lambda_pre_proc"], P["lle_weight"])
        ref = oracle.Tracker(*args); ref.initialize_nodes(Y0); ref.initialize_geodesic_coord(coord)
        trk = B.trackdlo(*args, ctx=ctx); trk.initialize_nodes(Y0); trk.initialize_geodesic_coord(coord); trk.set_precision(B.PREC_F64 if PREC else B.PREC_F32)
        for frame in range(6):
            kind = int(rng.integers(0, 6)); occl = None
            if kind == 1: occl = (0.0, float(rng.uniform(0.1, 0.4)))
            elif kind == 2: occl = (float(rng.uniform(0.6, 0.9)), 1.0)
            elif kind == 3:
                a = float(rng.uniform(0.2, 0.6)); occl = (a, a + float(rng.uniform(0.05, 0.3)))
            elif kind == 4: occl = (float(rng.uniform(0.15, 0.3)), 1.0)
            X, _, _ = synth.scene(N, M, config=900 + seed, frame=frame, occlude=occl, noise=noise, outliers=int(rng.integers(0, 6)), shift=(0.0, step * (frame + 1), 0.0))
            if len(X) == 0: break
            Ycur = ref.get_tracking_result(); s2_pre = ref.get_sigma2()
            ctx.set_cloud(0, X)
            _, vis, vext = ctx.visibility_prepass(0, Ycur, P["visibility_threshold"], 0.06, coord)
            if len(vis) < 4: break
            Lg = oracle.calc_lle_weights(Ycur[vext], 6)
            Hpre = (np.eye(len(vext)) - Lg).T @ (np.eye(len(vext)) - Lg)
            if np.abs(Hpre).max() > 1e8: skipped += 1; break         # registrations no fp64 implementation pins
            try:
                ref.tracking_step(X, vis, vext, H_pre=Hpre)
                ref_ok = True
            except Exception:
                ref_ok = False
            try:
                trk.tracking_step(X, vis, vext, None, 0, 0, H_pre=Hpre)
                trk_ok = True
            except B.TdloError as e:
                trk_ok = False; terr = str(e)
            if not ref_ok or not trk_ok:
                errs += 1
                if ref_ok != trk_ok:
                    bad += 1; print(f"ERROR MISMATCH seed {seed} frame {frame} M {M} N {len(X)}: oracle {'ok' if ref_ok else 'failed'}, product {'ok' if trk_ok else terr}", flush=True)
                break
            frames += 1
            if not (ref.get_sigma2() > 1e-12): skipped += 1; break
            same = trk.last_stats[0]["iters"] == ref.stats_pre.iters and trk.last_stats[1]["iters"] == ref.stats_main.iters and trk.get_correspondence_pairs().shape == ref.get_correspondence_pairs().shape
            dy = float(np.abs(trk.get_tracking_result() - ref.get_tracking_result()).max())
            dg = float(np.abs(trk.get_guide_nodes() - ref.get_guide_nodes()).max()) if trk.get_guide_nodes().shape == ref.get_guide_nodes().shape else np.inf
            ds = abs(trk.get_sigma2() - ref.get_sigma2()) / ref.get_sigma2()
            if dy > worst[0]: worst = (dy, (seed, frame, M, len(X)))
            ok = FA.judge(tally, (seed, frame, M, len(X)), max(dy, dg), ds, same,
                          lambda: FA.frame_uncertainty(oracle, PREC, args, coord, Ycur, s2_pre, X, vis, vext, Hpre, ref))
            if not ok:
                bad += 1
                print(f"MISMATCH seed {seed} frame {frame} M {M} N {len(X)} visible {len(vis)}/{len(vext)} |H| {np.abs(Hpre).max():.1e} iters ref {ref.stats_pre.iters},{ref.stats_main.iters} "
                      f"product {trk.last_stats[0]['iters']},{trk.last_stats[1]['iters']} priors {ref.get_correspondence_pairs().shape[0]}/{trk.get_correspondence_pairs().shape[0]} "
                      f"dY {dy:.2e} dguide {dg:.2e} dsigma2 {ds:.2e} sigma2 {ref.get_sigma2():.3e}", flush=True)
            if not same or dy > GY or dg > GY or ds > GS:
                break                                                 # the states have parted (explained or not): the rest of the sequence compares nothing
    if own: ctx.close()
    if verbose:
        for line in tally.notes: print("   " + line)
        print(f"{n} sequences from seed {s0} ({('fp32', 'fp64')[PREC]} mode): {frames} frames, {tally.summary()}; {errs} ended by an error on either side ({bad - tally.unexplained} on one side only), "
              f"{skipped} left because |H| > 1e8 or sigma2 collapsed")
    return dict(frames=frames, bad=bad, errs=errs, skipped=skipped, worst=worst, **{k: v for k, v in tally.as_dict().items() if k != "worst"})


if __name__ == "__main__":
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 100, int(sys.argv[2]) if len(sys.argv) > 2 else 0, int(os.environ.get("FUZZ_PREC", "1")))
