"""Multi-CU M-step (k_mstep_mcu) against the one-workgroup comparator (TDLO_MSTEP_BIG=1wg): same bits, and timings.
usage: gpu_mcu.py dump OUT.npz | gpu_mcu.py compare A.npz B.npz | gpu_mcu.py time"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from trackdlo_amd import binding as B, synth
P = synth.LAUNCH_PARAMS
CASES = ((3000, 61, B.PREC_F32, 5, False), (3000, 130, B.PREC_F64, 4, False), (3000, 200, B.PREC_F32, 3, True), (6000, 300, B.PREC_F64, 6, False),
         (20000, 512, B.PREC_F32, 3, False), (200000, 300, B.PREC_F64, 30, False), (5000, 77, B.PREC_F32, 25, False))

def run(N, M, prec, iters, priors):
    ctx = B.Context(max_points=N, max_nodes=M)
    X, Y0, _ = synth.scene(N, M, config=5)
    pr = B.make_params(P['beta'], P['lambda_'], P['lle_weight'], P['mu'], iters, 0.0, False, precision=prec, alpha=P['alpha'] if priors else 0.0)
    kw = {}
    if priors:
        idx = np.arange(0, M, 7)
        kw['priors'] = np.column_stack([idx, Y0[idx] + 0.003])
    ctx.set_cloud(0, X)
    out = []
    for rep in range(3):     # repeated calls on one slot: generation counter, flags re-armed
        g = ctx.cpd_lle_resident(0, Y0, 0.0, pr, **kw)
        out.append(np.concatenate([g['Y'].ravel(), [g['sigma2'], g['iters'], g['status']]]))
    t = (g['loop_ms'], ctx.profile_kernel(2, 5))
    ctx.close()
    return np.array(out), t

if sys.argv[1] == 'dump':
    res = {}
    for i, c in enumerate(CASES):
        o, t = run(*c)
        assert np.array_equal(o[0], o[1]) and np.array_equal(o[0], o[2]), f"case {c}: repeated calls differ"
        res[f"c{i}"] = o[0]
        print(f"N={c[0]} M={c[1]} prec={c[2]} iters={c[3]}: status={int(o[0][-1])} loop_ms={t[0]:.3f} ({t[0]/c[3]*1e3:.1f} us/iter) mstep_us={t[1]:.1f}", flush=True)
    np.savez(sys.argv[2], **res)
elif sys.argv[1] == 'compare':
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    for k in a.files:
        d = np.abs(a[k] - b[k]).max()
        print(k, "identical" if np.array_equal(a[k], b[k]) else f"DIFFERENT max abs {d:.3e}")
