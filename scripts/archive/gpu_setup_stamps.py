"""Phase stamps of k_setup (shader clocks of thread 0; instrumented build: bash scripts/build_variant.sh stamps -DTDLO_CHAIN_STAMPS -DTDLO_ESTEP_STAMPS):
0 start | 1 counts scanned (pass 1) | 2 serial sums | 3 offsets written (pass 2) | 4 centroid, coord | 5 nodes, accumulators cleared, chain links | 6 G / LLE records"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from trackdlo_amd import binding as B, synth
_v = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tmp", "libtrackdlo_stamps.so")
if not os.environ.get("TDLO_LIBRARY") and os.path.exists(_v):
    B._lib = B.load_library(_v)
P = synth.LAUNCH_PARAMS
for N, M, prec, lle in ((200000, 300, 1, False), (50000, 50, 0, False), (5000, 45, 0, True), (5000, 45, 0, False)):
    ctx = B.Context(max_points=N, max_nodes=M)
    X, Y0, _ = synth.scene(N, M, config=5)
    pr = B.make_params(3.0 if lle else P['beta'], 1.0 if lle else P['lambda_'], P['lle_weight'], P['mu'], 2, 0.0, lle, precision=prec)
    ctx.set_cloud(0, X)
    for rep in range(2):
        Yr = Y0 + 1e-6 * rep          # (new nodes: no reuse of the sorted cloud)
        g = ctx.cpd_lle_resident(0, Yr, 1e-4 if lle else 0.0, pr)
    st = ctx.debug_stamps(64).astype(np.int64)[16:23]
    print(f"N={N} M={M} lle={lle}: k_setup stamps (clocks)", (st - st[0]).tolist(), flush=True)
    ctx.close()
