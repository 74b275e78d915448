"""M-step with the LLE term (include_lle = true, the pre-processing registration of tracking_step) over M: kernel times;
`dump OUT.npz` also stores the results (run once plain and once with TDLO_MSTEP_LLE=1wg, then `compare A B`)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from trackdlo_amd import binding as B, synth
P = synth.LAUNCH_PARAMS
if len(sys.argv) > 1 and sys.argv[1] == 'compare':
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    for k in a.files: print(k, "identical" if np.array_equal(a[k], b[k]) else f"max abs diff {np.abs(a[k] - b[k]).max():.3e}")
    sys.exit(0)
res = {}
for M in (30, 50, 64, 65, 100, 128, 129, 160, 200, 300, 512):
    ctx = B.Context(max_points=1 << 16, max_nodes=M)
    X, Y0, _ = synth.scene(20000, M, config=5)
    H = np.eye(M) * 0.1 + 0.01 * np.diag(np.ones(M - 1), 1) + 0.01 * np.diag(np.ones(M - 1), -1)
    pr = B.make_params(3.0, 1.0, 10.0, 0.1, 5, 0.0, True, precision=1)
    ctx.set_cloud(0, X)
    g = ctx.cpd_lle_resident(0, Y0, 1e-4, pr, H=H, check=False)
    g2 = ctx.cpd_lle_resident(0, Y0, 1e-4, pr, H=H, check=False)
    assert np.array_equal(g['Y'], g2['Y'])
    res[f"M{M}"] = np.concatenate([g['Y'].ravel(), [g['sigma2'], g['status']]])
    print(f"M={M} include_lle: status={g['status']} loop_ms={g2['loop_ms']:.3f} ({g2['loop_ms']/5*1e3:.1f} us/iter) mstep_us={ctx.profile_kernel(2, 5):.1f}", flush=True)
    ctx.close()
if len(sys.argv) > 2 and sys.argv[1] == 'dump': np.savez(sys.argv[2], **res)
