"""One seed of tests/test_parity_gpu.py::test_randomised_tracking_sequences, frame by frame, without gates: the product's deviation from
the oracle in fp32 and in fp64 mode (same state before the frame) beside the oracle's own uncertainty.  usage: python scripts/gpu_seq_diag.py seed [...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from trackdlo_amd import synth, binding as B
from oracle import ref_cpu as oracle
P = synth.LAUNCH_PARAMS
for seed in [int(a) for a in sys.argv[1:]]:
    rng = np.random.default_rng(31000 + seed)
    M = int(rng.integers(12, 56)); N = int(rng.integers(800, 6000))
    Y0 = synth.nodes(M); coord = synth.geodesic_coord(Y0)
    args = (M, P["visibility_threshold"], P["beta"], P["lambda_"], P["alpha"], P["k_vis"], P["mu"], 30, P["tol"], P["beta_pre_proc"], P["lambda_pre_proc"], P["lle_weight"])
    ctx = B.Context(device=0, max_points=1 << 14, max_nodes=64)
    ref = oracle.Tracker(*args); ref.initialize_nodes(Y0); ref.initialize_geodesic_coord(coord)
    trk = B.trackdlo(*args, ctx=ctx); trk.initialize_nodes(Y0); trk.initialize_geodesic_coord(coord)
    for frame in range(4):
        kind = int(rng.integers(0, 5)); occl = None
        if kind == 1: occl = (0.0, float(rng.uniform(0.1, 0.4)))
        elif kind == 2: occl = (float(rng.uniform(0.6, 0.9)), 1.0)
        elif kind == 3:
            a = float(rng.uniform(0.2, 0.6)); occl = (a, a + float(rng.uniform(0.05, 0.3)))
        elif kind == 4: occl = (float(rng.uniform(0.15, 0.3)), 1.0)
        X, _, _ = synth.scene(N, M, config=70 + seed, frame=frame, occlude=occl, noise=0.0015, shift=(0.0, 0.002 * (frame + 1), 0.0))
        Ycur = ref.get_tracking_result()
        ctx.set_cloud(0, X)
        _, vis, vext = ctx.visibility_prepass(0, Ycur, P["visibility_threshold"], 0.06, coord)
        if len(vis) < 4: continue
        Lg = oracle.calc_lle_weights(Ycur[vext], 6)
        Hpre = (np.eye(len(vext)) - Lg).T @ (np.eye(len(vext)) - Lg)
        snap = B.trackdlo(*args, ctx=ctx); snap.copy_state_from(trk)
        ref.tracking_step(X, vis, vext, H_pre=Hpre)
        trk.tracking_step(X, vis, vext, None, 0, 0, H_pre=Hpre)
        t64 = B.trackdlo(*args, ctx=ctx); t64.copy_state_from(snap); t64.set_precision(B.PREC_F64)
        t64.tracking_step(X, vis, vext, None, 0, 0, H_pre=Hpre)
        dg = lambda t: float(np.abs(t.get_guide_nodes() - ref.get_guide_nodes()).max()) if t.get_guide_nodes().shape == ref.get_guide_nodes().shape else -1
        dy = lambda t: float(np.abs(t.get_tracking_result() - ref.get_tracking_result()).max())
        print(f"seed {seed} frame {frame} M={M} N={N} visible {len(vis)}/{len(vext)} |H| {np.abs(Hpre).max():.1e} iters ref {ref.stats_pre.iters},{ref.stats_main.iters} "
              f"f32 {trk.last_stats[0]['iters']},{trk.last_stats[1]['iters']} f64 {t64.last_stats[0]['iters']},{t64.last_stats[1]['iters']}: "
              f"guide f32 {dg(trk):.2e} f64 {dg(t64):.2e}  Y f32 {dy(trk):.2e} f64 {dy(t64):.2e}", flush=True)
        trk.copy_state_from(t64); trk.set_precision(B.PREC_F32)     # the next frame starts from (nearly) the oracle's state
    ctx.close()
