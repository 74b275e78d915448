"""Batch of frames (C3): wall-clock timeline of iterations 20..27 from a -DTDLO_TIMELINE build (scripts/tmp/libtrackdlo_timeline.so):
begin of the first / end of the last E-step workgroup and begin / end of the M-step of one frame per stream group, 10 ns ticks.
usage: [TDLO_BATCH_STREAMS=n] python scripts/gpu_timeline.py [frames] [N]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from trackdlo_amd import binding as B, synth
lib = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tmp", "libtrackdlo_timeline.so")
B.load_library(lib); B._lib = B.load_library(lib)
P = synth.LAUNCH_PARAMS
F = int(sys.argv[1]) if len(sys.argv) > 1 else 32
N = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
M = 50
ctx = B.Context(max_frames=F, max_points=N, max_nodes=M)
Ys = []
for f in range(F):
    X, Y0, _ = synth.scene(N, M, config=2, frame=f)
    ctx.set_cloud(f, X); Ys.append(Y0)
pr = B.make_params(P['beta'], P['lambda_'], P['lle_weight'], P['mu'], 50, 0.0, False)
Ystack, s2 = np.asarray(Ys), np.zeros(F)
for _ in range(3):
    g = ctx.cpd_lle_batch(Ystack, s2, pr)
ev = []
for f in range(F):
    os.environ["TDLO_DEBUG_FRAME"] = str(f)
    st = ctx.debug_stamps(32).astype(np.int64).reshape(8, 4)
    for i in range(8):
        ev.append((st[i, 0], st[i, 1], "E", f, 20 + i)); ev.append((st[i, 2], st[i, 3], "M", f, 20 + i))
t0 = min(e[0] for e in ev)
# frames of one group share their E-step launch: print one line per distinct (start, kind)
seen = {}
for s, e, k, f, it in sorted(ev):
    key = (k, it, s if k == "E" else None, f if k == "M" else None)
    if k == "E":
        kk = (it, s // 20)      # frames of one launch start within 0.2 us of each other
        if kk in seen: seen[kk][1] = max(seen[kk][1], e); seen[kk][3].append(f); continue
        seen[kk] = [s, e, k, [f], it]
Es = sorted(seen.values())
print("E-step launches (start us, end us, dur, iteration, frames):")
for s, e, k, fs, it in Es:
    print(f"  {(s - t0) / 100:8.2f} {(e - t0) / 100:8.2f}  {(e - s) / 100:6.2f}  it {it}  frames {fs[0]}..{fs[-1]}")
print("M-steps of frames 0, 8, 16, 24 (start, end, dur, iteration):")
for s, e, k, f, it in sorted(ev):
    if k == "M" and f in (0, 8, 16, 24): print(f"  {(s - t0) / 100:8.2f} {(e - t0) / 100:8.2f}  {(e - s) / 100:6.2f}  it {it} frame {f}")
span = (max(e[1] for e in ev) - t0) / 100
print(f"8 iterations in {span:.1f} us = {span / 8:.2f} us per iteration; call loop_ms {g[0]['loop_ms'] if isinstance(g, list) else ''}")
ctx.close()
