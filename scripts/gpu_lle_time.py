"""M-step with the LLE term (include_lle = true, the pre-processing registration of tracking_step) over M: kernel times."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from trackdlo_amd import binding as B, synth
P = synth.LAUNCH_PARAMS
for M in (30, 50, 64, 65, 100, 128, 129, 200, 300):
    ctx = B.Context(max_points=1 << 16, max_nodes=M)
    X, Y0, _ = synth.scene(20000, M, config=5)
    H = np.eye(M) * 0.1 + 0.01 * np.diag(np.ones(M - 1), 1) + 0.01 * np.diag(np.ones(M - 1), -1)
    pr = B.make_params(3.0, 1.0, 10.0, 0.1, 5, 0.0, True)
    ctx.set_cloud(0, X)
    g = ctx.cpd_lle_resident(0, Y0, 1e-4, pr, H=H, check=False)
    g = ctx.cpd_lle_resident(0, Y0, 1e-4, pr, H=H, check=False)
    print(f"M={M} include_lle: status={g['status']} loop_ms={g['loop_ms']:.3f} ({g['loop_ms']/5*1e3:.1f} us/iter) mstep_us={ctx.profile_kernel(2, 5):.1f}", flush=True)
    ctx.close()
