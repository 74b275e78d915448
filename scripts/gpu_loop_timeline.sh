#!/bin/bash
# One frame's EM loop as the GPU runs it: per-dispatch start / end (rocprofv3 kernel trace) of a few iterations in the middle of ONE tdlo_cpd_lle call --
# kernel durations and the gaps E end -> M start, M end -> E start.   usage: bash scripts/gpu_loop_timeline.sh [N] [M] [precision 0|1]
export TMPDIR=/tmp
R=$PWD; N=${1:-2000000}; M=${2:-50}; PREC=${3:-0}
cat > /tmp/loop_run.py <<'PY'
import os, sys
sys.path.insert(0, os.environ["R"])
import numpy as np
from trackdlo_amd import binding as B, synth
P = synth.LAUNCH_PARAMS
N, M, prec = int(os.environ["N"]), int(os.environ["M"]), int(os.environ["PREC"])
pr = B.make_params(P['beta'], P['lambda_'], P['lle_weight'], P['mu'], 50, 0.0, False, precision=prec)
c = B.Context(max_points=N, max_nodes=M, timing=False); c.set_sort_reuse(False)
X, Y0, _ = synth.scene(N, M, config=4 if N > 100000 else 2)
c.set_cloud(0, X)
for i in range(4): c.cpd_lle_resident(0, Y0, 0.0, pr)
c.close()
PY
cd /tmp; rm -rf /tmp/tlL
R=$R N=$N M=$M PREC=$PREC rocprofv3 --kernel-trace --output-format csv -d /tmp/tlL -- python /tmp/loop_run.py > /dev/null 2>&1
f=$(find /tmp/tlL -name "*kernel_trace.csv" | head -1)
python3 - $f <<'PY'
import csv, sys
rows=[r for r in csv.DictReader(open(sys.argv[1]))]
ev=sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
idx=[i for i,e in enumerate(ev) if "k_prune_pass1" in e[2] or "k_prologue" in e[2]]
call=ev[idx[-1]:]
loop=[e for e in call if ("k_estep" in e[2] or "k_mstep" in e[2])]
E=[e for e in loop if "k_estep" in e[2]]; Mk=[e for e in loop if "k_mstep" in e[2]]
n=min(len(E),len(Mk))
ed=[(E[i][1]-E[i][0])/1e3 for i in range(20,n)]; md=[(Mk[i][1]-Mk[i][0])/1e3 for i in range(20,n)]
g1=[(Mk[i][0]-E[i][1])/1e3 for i in range(20,n)]; g2=[(E[i+1][0]-Mk[i][1])/1e3 for i in range(20,n-1)]
per=[(E[i+1][0]-E[i][0])/1e3 for i in range(20,n-1)]
mean=lambda v: sum(v)/len(v)
print(f"iterations 20..{n-1} of the last call (under the profiler): E-step {mean(ed):.2f} us, gap E end -> M start {mean(g1):.2f}, M-step {mean(md):.2f}, gap M end -> E start {mean(g2):.2f}; period {mean(per):.2f} us")
PY
