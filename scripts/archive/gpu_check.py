#!/usr/bin/env python3
"""Quick GPU parity / timing probe (development aid; the judged tests live in tests/)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import ref_cpu as R
from trackdlo_amd import binding as B, synth

P = synth.LAUNCH_PARAMS
ctx = B.Context(max_frames=4, max_points=1 << 16, max_nodes=64)


def run_case(name, N, M, iters, prec, vis=False, priors=False, lle=False, sigma2=0.0, config=1, tol=0.0):
    X, Y0, v = synth.scene(N, M, config=config, occlude=(0.4, 0.6) if vis else None)
    coord = synth.geodesic_coord(Y0)
    kw = dict(beta=P['beta'], lambda_=P['lambda_'], lle_weight=P['lle_weight'], mu=P['mu'], max_iter=iters, tol=tol,
              include_lle=lle, alpha=P['alpha'] if priors else 0.0, k_vis=P['k_vis'] if vis else 0.0,
              visibility_threshold=P['visibility_threshold'])
    pri = None
    if priors:
        idx = np.arange(0, M, 3)
        pri = np.concatenate([idx[:, None].astype(float), Y0[idx] + np.array([0, 0.004, 0.0])], axis=1)
    H = None
    if lle:
        L = R.calc_lle_weights(Y0, 6); H = (np.eye(M) - L).T @ (np.eye(M) - L)
        kw['beta'] = P['beta_pre_proc']; kw['lambda_'] = P['lambda_pre_proc']
    vext = synth.extend_visible(v, M, coord) if vis else None
    t0 = time.time()
    o = R.cpd_lle(X, Y0, sigma2, priors=pri, visible_nodes=vext, H=H, **kw)
    t_cpu = time.time() - t0
    pr = B.make_params(precision=prec, **kw)
    g = ctx.cpd_lle(X, Y0, sigma2, pr, priors=pri, visible_nodes=vext, H=H, check=False)
    g2 = ctx.cpd_lle_resident(0, Y0, sigma2, pr, priors=pri, visible_nodes=vext, H=H, check=False)
    dy = np.abs(g['Y'] - o['Y']).max(); ds = abs(g['sigma2'] - o['sigma2']) / o['sigma2']
    print(f"{name:28s} N={N} M={M} it={g['iters']}/{o['iters']} conv={g['converged']}/{o['converged']} kept={g['n_kept']}/{o['n_kept']} "
          f"rc={g['rc']} dY={dy:.3e} ds2={ds:.3e} rep={np.abs(g['Y']-g2['Y']).max():.1e} loop_ms={g2['loop_ms']:.3f} total_ms={g2['total_ms']:.3f} "
          f"host_ms={g2['host_ms']:.3f} cpu_loop_s={o['loop_seconds']:.3f} quirk={o['gap_quirk']}", flush=True)
    return g2, o


for prec, pn in ((B.PREC_F32, 'f32'), (B.PREC_F64, 'f64')):
    run_case(f'C1 {pn}', 2000, 30, 20, prec)
    run_case(f'C1 {pn} sigma2=1e-4', 2000, 30, 20, prec, sigma2=1e-4)
    run_case(f'C1 {pn} vis', 2000, 30, 20, prec, vis=True)
    run_case(f'C1 {pn} priors', 2000, 30, 20, prec, priors=True)
    run_case(f'C1 {pn} vis+priors', 2000, 30, 20, prec, vis=True, priors=True)
    run_case(f'C1 {pn} lle', 2000, 30, 20, prec, lle=True)
    run_case(f'C1 {pn} tol', 2000, 30, 50, prec, tol=2e-4)
    run_case(f'M=100 {pn}', 5000, 100, 10, prec)
g, o = run_case('C2 f32', 50000, 50, 50, B.PREC_F32)
print('C2 it/s GPU', 50 / (g['loop_ms'] * 1e-3), 'CPU', o['iters'] / o['loop_seconds'])
for kind, nm in ((0, 'estep'), (2, 'mstep')):
    print(nm, 'avg us', ctx.profile_kernel(kind, 200))
g, o = run_case('C2 f32 vis', 50000, 50, 50, B.PREC_F32, vis=True)
print('dmin avg us', ctx.profile_kernel(1, 200), 'estep(vis)', ctx.profile_kernel(0, 200))
g, o = run_case('C2 f64', 50000, 50, 50, B.PREC_F64)
print('f64 estep avg us', ctx.profile_kernel(0, 50))
run_case('M=300 f32', 20000, 300, 5, B.PREC_F32, config=5)
run_case('M=300 f64', 20000, 300, 5, B.PREC_F64, config=5)
