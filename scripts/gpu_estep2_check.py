"""k_estep2 (two points per lane) against the oracle and against k_estep on the same inputs, then E-step times at C2 / C4 / one C3 batch.
usage: python scripts/gpu_estep2_check.py [parity] [time]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from trackdlo_amd import binding as B, synth
P = synth.LAUNCH_PARAMS
what = sys.argv[1:] or ["parity", "time"]


def ctx_with(mode, **kw):
    if mode is None: os.environ.pop("TDLO_ESTEP2", None)
    else: os.environ["TDLO_ESTEP2"] = str(mode)
    return B.Context(**kw)


if "parity" in what:
    from oracle import ref_cpu
    worst = 0.0
    for (N, M, iters, vis, s2) in ((2000, 30, 20, False, 0.0), (1999, 45, 10, False, 0.0), (5000, 50, 30, True, 0.0), (20000, 50, 25, False, 0.0), (130, 8, 6, False, 0.0),
                                   (7777, 64, 12, True, 1e-4), (3000, 50, 40, False, 1e-6), (64, 20, 5, False, 0.0), (129, 20, 5, False, 0.0), (40000, 50, 50, False, 0.0)):
        X, Y0, v = synth.scene(N, M, config=40 + M, occlude=(0.4, 0.6) if vis else None, outliers=5)
        vext = synth.extend_visible(v, M, synth.geodesic_coord(Y0)) if vis else None
        kw = dict(beta=P["beta"], lambda_=P["lambda_"], lle_weight=P["lle_weight"], mu=P["mu"], max_iter=iters, tol=0.0, include_lle=False, alpha=0.0,
                  k_vis=P["k_vis"] if vis else 0.0, visibility_threshold=P["visibility_threshold"])
        o = ref_cpu.cpd_lle(X, Y0, s2, visible_nodes=vext, **kw)
        res = {}
        for mode in (0, 1):
            ctx = ctx_with(mode)
            g = ctx.cpd_lle(X, Y0, s2, B.make_params(**kw), visible_nodes=vext)
            g2 = ctx.cpd_lle(X, Y0, s2, B.make_params(**kw), visible_nodes=vext)
            assert np.array_equal(g["Y"], g2["Y"]) and g["sigma2"] == g2["sigma2"], "not repeatable"
            res[mode] = g
            ctx.close()
        for mode in (0, 1):
            g = res[mode]
            dy = float(np.abs(g["Y"] - o["Y"]).max()); ds = abs(g["sigma2"] - o["sigma2"]) / o["sigma2"]
            ok = g["rc"] == 0 and g["iters"] == o["iters"] and g["n_kept"] == o["n_kept"] and dy <= 1e-5 and ds <= 1e-3
            print(f"N={N} M={M} it={iters} vis={vis} s2={s2}: estep2={mode} rc={g['rc']} iters={g['iters']}/{o['iters']} kept={g['n_kept']}/{o['n_kept']} dY={dy:.2e} ds={ds:.2e} {'ok' if ok else 'FAIL'}", flush=True)
            if mode == 1: worst = max(worst, dy)
        print(f"   two routes apart: {float(np.abs(res[0]['Y'] - res[1]['Y']).max()):.2e} m", flush=True)
    print(f"worst dY of k_estep2 against the oracle: {worst:.2e} m")

if "time" in what:
    pr = B.make_params(P['beta'], P['lambda_'], P['lle_weight'], P['mu'], 50, 0.0, False)
    for N, F in ((2000000, 1), (50000, 32), (262144, 1), (50000, 1)):
        for mode in (0, 1, 0, 1):
            ctx = ctx_with(mode, max_frames=F, max_points=N, max_nodes=50)
            Ys = []
            for fr in range(F):
                X, Y0, _ = synth.scene(N, 50, config=2 if N == 50000 else 4, frame=fr)
                ctx.set_cloud(fr, X); Ys.append(Y0)
            if F == 1:
                g = ctx.cpd_lle_resident(0, Ys[0], 0.0, pr); g = ctx.cpd_lle_resident(0, Ys[0], 0.0, pr)
                lm = g['loop_ms']
            else:
                g = ctx.cpd_lle_batch(Ys, [0.0] * F, pr); g = ctx.cpd_lle_batch(Ys, [0.0] * F, pr)
                lm = g['stats'][0]['loop_ms']
            e, m, it, name = ctx.profile_iteration(200)
            print(f"[estep2={mode}] N={N} F={F}: loop {lm:.3f} ms ({lm/50*1e3:.2f} us/iter)  in-situ estep {e:.2f} us mstep {m:.2f} us iter {it:.2f} us  b2b estep {ctx.profile_kernel(0, 200):.2f} us", flush=True)
            ctx.close()

if "time1" in what:      # the environment's own setting (TDLO_ESTEP2 / _ROWS / _BLOCKS), C4 and one C3 batch: scripts/gpu_estep2_sweep.sh
    pr = B.make_params(P['beta'], P['lambda_'], P['lle_weight'], P['mu'], 50, 0.0, False)
    for N, F in ((2000000, 1), (50000, 32)):
        ctx = B.Context(max_frames=F, max_points=N, max_nodes=50)
        Ys = []
        for fr in range(F):
            X, Y0, _ = synth.scene(N, 50, config=2 if N == 50000 else 4, frame=fr)
            ctx.set_cloud(fr, X); Ys.append(Y0)
        if F == 1:
            g = ctx.cpd_lle_resident(0, Ys[0], 0.0, pr); g = ctx.cpd_lle_resident(0, Ys[0], 0.0, pr)
            lm = g['loop_ms']
        else:
            g = ctx.cpd_lle_batch(Ys, [0.0] * F, pr); g = ctx.cpd_lle_batch(Ys, [0.0] * F, pr)
            lm = g['stats'][0]['loop_ms']
        e, m, it, name = ctx.profile_iteration(200)
        print(f"[{os.environ.get('TDLO_ESTEP2')} rows {os.environ.get('TDLO_ESTEP2_ROWS')} blocks {os.environ.get('TDLO_ESTEP2_BLOCKS')}] N={N} F={F}: loop {lm/50*1e3:.2f} us/iter  in-situ estep {e:.2f} us  b2b estep {ctx.profile_kernel(0, 200):.2f} us", flush=True)
        ctx.close()
