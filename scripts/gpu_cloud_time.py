"""depth image -> cloud -> voxel grid (tdlo_depth_to_cloud): ms per call of the one-launch kernel (images copied from pageable memory / read in
place from the context's pinned buffers) and of the multi-launch form, at 640 x 480 and at the reference camera's 1280 x 720."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from trackdlo_amd import binding as B, synth

def ctx(fused):
    if fused: os.environ.pop("TDLO_CLOUD_FUSED", None)
    else: os.environ["TDLO_CLOUD_FUSED"] = "0"
    c = B.Context(device=0, timing=False)
    os.environ.pop("TDLO_CLOUD_FUSED", None)
    return c

def rate(fn, n=300):
    for _ in range(20): fn()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    return (time.perf_counter() - t0) * 1e3 / n

only = sys.argv[1] if len(sys.argv) > 1 else ""
for shape in ((480, 640), (720, 1280)):
    depth, mask, cam, _ = synth.depth_scene(50, config=9, frame=3, rows=shape[0], cols=shape[1])
    a = (cam["fx"], cam["fy"], cam["cx"], cam["cy"])
    nmask = int(np.count_nonzero(mask))
    for fused in (True, False):
        if only == "fused" and not fused: continue
        c = ctx(fused)
        _, n, _ = c.depth_to_cloud(0, depth, mask, *a, 0.008, fetch=False)
        ms = rate(lambda: c.depth_to_cloud(0, depth, mask, *a, 0.008, fetch=False))
        line = f"depth_to_cloud {shape[1]}x{shape[0]} ({nmask} masked px -> {n} points) {'one launch' if fused else 'multi-launch'}: pageable images {ms:.4f} ms"
        d, m = c.image_buffers(*shape); d[:] = depth; m[:] = mask
        ms2 = rate(lambda: c.depth_to_cloud(0, d, m, *a, 0.008, fetch=False))
        print(line + f", in the pinned buffers {ms2:.4f} ms   routes {c.cloud_route_counts()}", flush=True)
        c.close()
