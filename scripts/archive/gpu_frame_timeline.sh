#!/bin/bash
# One device-born frame (tdlo_tracker_frame_from_depth, 640 x 480, M = 30) on the GPU's timeline: kernels, durations, gaps (rocprofv3 --kernel-trace).
# usage (GPU box): bash scripts/gpu_frame_timeline.sh <tag>
tag=${1:-frametl}
R=$(pwd); O=$R/gpurun_out/$tag; mkdir -p $O
cat > /tmp/frame_run.py <<'PY'
import sys, os
sys.path.insert(0, os.environ["R"])
import numpy as np
from trackdlo_amd import binding as B, synth
P = synth.LAUNCH_PARAMS; M = 30
ctx = B.Context(device=0, timing=False)
depth, mask, cam, Y0 = synth.depth_scene(M, config=9, frame=3)
coord = synth.geodesic_coord(Y0)
trk = B.trackdlo(M, P["visibility_threshold"], P["beta"], P["lambda_"], P["alpha"], P["k_vis"], P["mu"], 50, P["tol"], P["beta_pre_proc"], P["lambda_pre_proc"], P["lle_weight"], ctx=ctx)
trk.initialize_nodes(Y0); trk.initialize_geodesic_coord(coord)
d, m = ctx.image_buffers(*depth.shape); d[:] = depth; m[:] = mask
a = (cam["fx"], cam["fy"], cam["cx"], cam["cy"])
for _ in range(400):
    trk.frame_from_depth(d, m, *a, 0.008, 0.06)
print("iters", [s["iters"] for s in trk.last_stats])
PY
cd /tmp && export TMPDIR=/tmp
R=$R PYTHONPATH=$R timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/ftl -- python /tmp/frame_run.py </dev/null > $O/run.log 2>&1
f=$(find /tmp/ftl -name "*kernel_trace.csv" | head -1)
python - "$f" > $O/timeline.txt <<'PY'
import csv, sys, statistics as st
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
import re
def nm(k):
    m = re.search(r"(k_[a-z0-9_]+(<[^>]*>)?)", k)
    return (m.group(1) if m else k)[:48]
ev=[(nm(r["Kernel_Name"]), int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
# frames start at k_cloud_team
idx=[i for i,e in enumerate(ev) if e[0].startswith("k_cloud_team")]
frames=[ev[idx[i]:idx[i+1]] for i in range(len(idx)-1)]
frames=frames[200:380]
span=[f[-1][2]-f[0][1] for f in frames]; period=[frames[i+1][0][1]-frames[i][0][1] for i in range(len(frames)-1)]
busy=[sum(e[2]-e[1] for e in f) for f in frames]
print("frames analysed %d: kernels per frame %s; GPU busy %.1f us, first kernel start -> last kernel end %.1f us, frame period %.1f us (medians)"%(len(frames), sorted(set(len(f) for f in frames)), st.median(busy)/1e3, st.median(span)/1e3, st.median(period)/1e3))
f=frames[len(frames)//2]
t0=f[0][1]; prev=None
for e in f:
    print("  +%7.2f us  dur %6.2f  gap %6.2f   %s"%((e[1]-t0)/1e3,(e[2]-e[1])/1e3,0.0 if prev is None else (e[1]-prev)/1e3,e[0])); prev=e[2]
PY
tail -2 $O/run.log | cut -c1-200; cat $O/timeline.txt
