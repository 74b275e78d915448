"""Randomised tracking_step sequences on the product's two routes: every short cut of round 4 on (cloud read from pinned host memory, paired
set-up, first E-step's sums handed over, first M-step launched ahead of its priors, the next frame's LLE regulariser formed on the device, the iteration hint,
the main registration's first iteration beside the pre-processing registration when nodes are hidden) against
all of them off -- the library's own H on both sides (no H_pre: the device-formed regulariser is in play), random chain length, cloud size (down
to a few dozen points), noise, motion (frames that converge at once and frames that take many iterations), occlusion pattern per frame, both
precisions, eight frames with the state carried over.  Every result must be the same BITS, and an error must be the same error.
usage: python scripts/gpu_fuzz_routes.py [n_seq] [first_seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from trackdlo_amd import synth, binding as B
P = synth.LAUNCH_PARAMS
KEYS = ("TDLO_PAIR_SETUP", "TDLO_PAIR_SUMS", "TDLO_SPEC_MSTEP", "TDLO_LLE_NEXT", "TDLO_DIRECT_CLOUD", "TDLO_ITER_HINT", "TDLO_AHEAD")


def _ctx(on):
    old = {k: os.environ.get(k) for k in KEYS}
    try:
        for k in KEYS:
            os.environ.pop(k, None)
            if not on: os.environ[k] = "0"
        return B.Context(device=0, max_points=1 << 14, max_nodes=64, timing=False)
    finally:
        for k, v in old.items():
            if v is None: os.environ.pop(k, None)
            else: os.environ[k] = v


def run(n, s0=0, verbose=True):
    full, plain = _ctx(True), _ctx(False)
    frames = bad = errs = 0
    try:
        for seed in range(s0, s0 + n):
            rng = np.random.default_rng(77000 + seed)
            M = int(rng.integers(8, 61)); N = int(rng.choice([rng.integers(40, 300), rng.integers(300, 6000), rng.integers(6000, 16384)]))
            noise = float(rng.choice([0.0005, 0.0015, 0.003])); step = float(rng.choice([0.0, 0.0, 0.001, 0.004]))
            prec = int(rng.integers(0, 2)); max_iter = int(rng.choice([1, 5, 30, 50]))
            Y0 = synth.nodes(M); coord = synth.geodesic_coord(Y0)
            args = (M, P["visibility_threshold"], P["beta"], P["lambda_"], P["alpha"], P["k_vis"], P["mu"], max_iter, P["tol"], P["beta_pre_proc"], P["lambda_pre_proc"], P["lle_weight"])
            trks = []
            for ctx in (full, plain):
                t = B.trackdlo(*args, ctx=ctx, precision=prec); t.initialize_nodes(Y0); t.initialize_geodesic_coord(coord); trks.append(t)
            for frame in range(8):
                kind = int(rng.integers(0, 8)); occl = None
                if kind == 1: occl = (0.0, float(rng.uniform(0.1, 0.4)))
                elif kind == 2: occl = (float(rng.uniform(0.6, 0.9)), 1.0)
                elif kind == 3:
                    a = float(rng.uniform(0.2, 0.6)); occl = (a, a + float(rng.uniform(0.05, 0.3)))
                X, _, v = synth.scene(N, M, config=700 + seed, frame=frame if step else 0, occlude=occl, noise=noise, outliers=int(rng.integers(0, 6)),
                                      shift=(0.0, step * (frame + 1), 0.0))
                if kind == 7: X = X + np.array([5.0, 0.0, 0.0])          # a frame in which every point is pruned: both must fail alike and carry on
                if len(X) == 0: break
                v = np.arange(M, dtype=np.int32) if v is None else v
                vext = synth.extend_visible(v, M, coord)
                if len(vext) < 4: break
                res = []
                for t in trks:
                    try:
                        t.tracking_step(X, v, vext)
                        res.append(("ok", t.get_tracking_result(), t.get_sigma2(), [s["iters"] for s in t.last_stats], t.get_correspondence_pairs(), t.get_guide_nodes()))
                    except B.TdloError as e:
                        res.append(("err", e.code if hasattr(e, "code") else str(e)))
                a, b = res
                frames += 1
                if a[0] != b[0]:
                    bad += 1; print(f"ROUTE MISMATCH seed {seed} frame {frame} M {M} N {len(X)}: {a[0]} / {b[0]} ({a[1] if a[0] == 'err' else ''} {b[1] if b[0] == 'err' else ''})", flush=True); break
                if a[0] == "err":
                    errs += 1
                    if a[1] != b[1]: bad += 1; print(f"ERROR MISMATCH seed {seed} frame {frame}: {a[1]} / {b[1]}", flush=True); break
                    continue
                same = (np.array_equal(a[1], b[1]) and a[2] == b[2] and a[3] == b[3] and a[4].shape == b[4].shape and np.array_equal(a[4], b[4]) and np.array_equal(a[5], b[5]))
                if not same:
                    bad += 1
                    print(f"MISMATCH seed {seed} frame {frame} M {M} N {len(X)} prec {prec} max_iter {max_iter} visible {len(v)}/{len(vext)} iters {a[3]} / {b[3]} "
                          f"dY {np.abs(a[1] - b[1]).max():.2e} sigma2 {a[2]:.6e} / {b[2]:.6e}", flush=True)
                    break
        counts = full.route_counts()
    finally:
        full.close(); plain.close()
    if verbose: print(f"{n} sequences from seed {s0}: {frames} frames on both routes, {errs} ended by the same error, {bad} differing; routes taken {counts}")
    return dict(frames=frames, bad=bad, errs=errs, routes=counts)


if __name__ == "__main__":
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 100, int(sys.argv[2]) if len(sys.argv) > 2 else 0)
