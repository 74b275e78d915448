"""C5 (N = 200 000, M = 300, fp64) and other large-M timings: E-step / M-step kernel times, loop time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from trackdlo_amd import binding as B, synth
P = synth.LAUNCH_PARAMS
CASES = ((200000, 300, B.PREC_F64), (200000, 300, B.PREC_F32), (50000, 100, B.PREC_F32), (50000, 64, B.PREC_F32), (50000, 128, B.PREC_F64))
for N, M, prec in CASES[:int(os.environ.get("CASES", "5"))]:
    ctx = B.Context(max_points=N, max_nodes=M)
    X, Y0, _ = synth.scene(N, M, config=5)
    pr = B.make_params(P['beta'], P['lambda_'], P['lle_weight'], P['mu'], int(os.environ.get('ITERS', '5')), 0.0, False, precision=prec)
    ctx.set_cloud(0, X)
    g = ctx.cpd_lle_resident(0, Y0, 0.0, pr)
    g = ctx.cpd_lle_resident(0, Y0, 0.0, pr)
    print(f"N={N} M={M} prec={prec}: loop_ms={g['loop_ms']:.3f} ({g['loop_ms']/int(os.environ.get('ITERS','5'))*1e3:.1f} us/iter) total_ms={g['total_ms']:.3f} estep_us={ctx.profile_kernel(0, 20):.1f} mstep_us={ctx.profile_kernel(2, 5):.1f}", flush=True)
    ctx.close()
