#!/bin/bash
# C3 batch (32 x 50 000 points, M = 50): per-dispatch timeline (rocprofv3 kernel trace) of a few iterations in the middle of ONE tdlo_cpd_lle_batch call,
# for TDLO_BATCH_STREAMS = $1 (default 3); E = k_estep / k_estep2, M = k_mstep_chain, q = hardware queue.
export TMPDIR=/tmp
R=$PWD; NS=${1:-3}
cat > /tmp/c3_run.py <<'PY'
import os, sys
sys.path.insert(0, os.environ["R"])
import numpy as np
from trackdlo_amd import binding as B, synth
P = synth.LAUNCH_PARAMS
F, N, M = 32, 50000, 50
pr = B.make_params(P['beta'], P['lambda_'], P['lle_weight'], P['mu'], 50, 0.0, False)
c = B.Context(max_frames=F, max_points=N, max_nodes=M, timing=False); c.set_sort_reuse(False)
Ys = []
for f in range(F):
    X, Y0, _ = synth.scene(N, M, config=2, frame=f); c.set_cloud(f, X); Ys.append(Y0)
for i in range(4): c.cpd_lle_batch(Ys, [0.0] * F, pr)
c.close()
PY
cd /tmp
R=$R TDLO_BATCH_STREAMS=$NS rocprofv3 --kernel-trace --output-format csv -d /tmp/tl$NS -- python /tmp/c3_run.py > /dev/null 2>&1
f=$(find /tmp/tl$NS -name "*kernel_trace.csv" | head -1)
python3 - $f <<'PY'
import csv, sys
rows=[r for r in csv.DictReader(open(sys.argv[1]))]
ev=sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id","?")) for r in rows)
idx=[i for i,e in enumerate(ev) if "k_prune_pass1" in e[2]]
i0=idx[-1]; t0=ev[i0][0]
call=ev[i0:]
def nm(k): return "E" if "k_estep" in k else ("M" if "k_mstep" in k else k.split("(")[0][-18:])
print("start_us  dur_us  kernel queue   (relative to the call's first kernel)")
for e in call[120:160]:
    print(f"{(e[0]-t0)/1e3:9.2f} {(e[1]-e[0])/1e3:7.2f}  {nm(e[2]):3s} q{e[3]}")
loop=[e for e in call if ("k_estep" in e[2] or "k_mstep" in e[2])]
T0=loop[0][0]; T1=max(e[1] for e in loop)
es=sorted((e[0],e[1]) for e in loop if "k_estep" in e[2])
busy=0; cs,ce=es[0]
for s,e in es[1:]:
    if s<=ce: ce=max(ce,e)
    else: busy+=ce-cs; cs,ce=s,e
busy+=ce-cs
print(f"loop span {(T1-T0)/1e3:.1f} us = {(T1-T0)/1e3/50:.2f} us per iteration; some E-step running {busy/1e3:.1f} us ({busy/(T1-T0):.2%}); sum of E durations {sum(e-s for s,e in es)/1e3:.1f} us; E launches {len(es)}")
PY
