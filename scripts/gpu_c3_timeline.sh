#!/bin/bash
# C3 batch: per-dispatch timeline (kernel trace) of a few iterations in the middle of a call, for TDLO_BATCH_STREAMS = $1
export TMPDIR=/tmp
R=$PWD; NS=${1:-2}
cd /tmp
TDLO_BATCH_STREAMS=$NS rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tl$NS -- python $R/bench.py --config c3 --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
f=$(find $R/gpurun_out/tl$NS -name "*kernel_trace.csv" | head -1)
python3 - $f <<'PY'
import csv, sys
rows=[r for r in csv.DictReader(open(sys.argv[1]))]
ev=[(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id","?")) for r in rows]
ev.sort()
# the last call: find the last k_prune_pass1 and show dispatches 600..680 after it
idx=[i for i,e in enumerate(ev) if "k_prune_pass1" in e[2]]
i0=idx[-1]
t0=ev[i0][0]
sel=ev[i0:i0+400]
def nm(k): return "E" if "k_estep" in k else ("M" if "k_mstep" in k else k.split("(")[0][-18:])
print("start_us  dur_us  kernel queue   (relative to the call's first kernel)")
for e in sel[200:260]:
    print(f"{(e[0]-t0)/1e3:9.2f} {(e[1]-e[0])/1e3:7.2f}  {nm(e[2]):3s} q{e[3]}")
# overall: time with >=1 E-step running / total in the loop part
import itertools
loop=[e for e in ev[i0:] if ("k_estep" in e[2] or "k_mstep" in e[2])]
T0=loop[0][0]; T1=max(e[1] for e in loop)
es=sorted((e[0],e[1]) for e in loop if "k_estep" in e[2])
busy=0; cur_s,cur_e=es[0]
for s,e in es[1:]:
    if s<=cur_e: cur_e=max(cur_e,e)
    else: busy+=cur_e-cur_s; cur_s,cur_e=s,e
busy+=cur_e-cur_s
print(f"loop span {(T1-T0)/1e3:.1f} us, some E-step running {busy/1e3:.1f} us ({busy/(T1-T0):.2%}); sum of E durations {sum(e-s for s,e in es)/1e3:.1f} us")
PY
rm -rf $R/gpurun_out/tl$NS
