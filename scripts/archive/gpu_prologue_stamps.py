"""Phase stamps of the fused prologue k_prologue at production size (shader clocks; instrumented build: bash scripts/build_variant.sh stamps -DTDLO_CHAIN_STAMPS):
node workgroup (k_setup's body): 0 start | 1 (scan: skipped) | 2 | 3 | 4 nodes from host, centroid, coord | 5 nodes, accumulators cleared, chain links | 6 LLE records, H Y0 | 7 end (the counts are summed by point workgroup 0 since the end of round 4)
point workgroup 0: 0 start | 1 nodes (host) + points loaded | 2 pruned, nearest node, block sum | 3 counts published, ticket | 4 grid barrier passed | 5 offsets | 6 scattered"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from trackdlo_amd import binding as B, synth
_v = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tmp", "libtrackdlo_stamps.so")
if not os.environ.get("TDLO_LIBRARY") and os.path.exists(_v):
    B._lib = B.load_library(_v)
P = synth.LAUNCH_PARAMS
for N, M, lle, reuse in ((5000, 45, True, False), (5000, 45, False, False), (5000, 45, False, True), (16000, 45, True, False), (1000, 45, True, False)):
    ctx = B.Context(max_points=N, max_nodes=64)
    X, Y0, _ = synth.scene(N, M, config=5)
    pr = B.make_params(3.0 if lle else P['beta'], 1.0 if lle else P['lambda_'], P['lle_weight'], P['mu'], 1, 0.0, lle)
    ctx.set_cloud(0, X)
    ctx.set_sort_reuse(reuse)
    rows = []
    for rep in range(6):
        g = ctx.cpd_lle_resident(0, Y0, 1e-4, pr)
        st = ctx.debug_stamps(64).astype(np.int64)
        rows.append(np.concatenate([st[16:24] - st[16], st[24:31] - st[24], [st[24] - st[16]]]))
    r = np.median(np.array(rows[2:]), axis=0).astype(int)
    print(f"N={N} M={M} lle={lle} reuse={reuse}: node wg {r[:8].tolist()}  point wg0 {r[8:15].tolist()}  (point wg0 started {r[15]} clocks after the node wg)", flush=True)
    ctx.close()
