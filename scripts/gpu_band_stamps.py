"""Phase stamps of k_mstep_band (shader clocks of thread 0) over chain lengths; needs the instrumented build:
bash scripts/build_variant.sh stamps -DTDLO_CHAIN_STAMPS"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from trackdlo_amd import binding as B, synth
_v = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tmp", "libtrackdlo_stamps.so")
if not os.environ.get("TDLO_LIBRARY") and os.path.exists(_v):
    B._lib = B.load_library(_v)
for M in (30, 50, 128, 300):
    ctx = B.Context(max_points=1 << 16, max_nodes=M)
    X, Y0, _ = synth.scene(20000, M, config=5)
    H = np.eye(M) * 0.1 + 0.01 * np.diag(np.ones(M - 1), 1) + 0.01 * np.diag(np.ones(M - 1), -1)
    pr = B.make_params(3.0, 1.0, 10.0, 0.1, 5, 0.0, True, precision=1)
    ctx.set_cloud(0, X)
    g = ctx.cpd_lle_resident(0, Y0, 1e-4, pr, H=H, check=False)
    st = ctx.debug_stamps(64).astype(np.int64)
    names = ["start", "fetch", "sums->LDS", "records", "window(8)", "eliminated(4)", "back-subst(5)", "barrier(6)", "T,sums(7)"]
    order = [0, 1, 2, 3, 8, 4, 5, 6, 7]
    v = [int(st[i] - st[0]) for i in order]
    print(f"M={M} " + "  ".join(f"{n}={x}" for n, x in zip(names, v)), " per-unknown: elim %.0f back %.0f" % ((v[5] - v[4]) / (2 * M), (v[6] - v[5]) / (2 * M)), flush=True)
    print("   mstep_us", ctx.profile_kernel(2, 50))
    ctx.close()
