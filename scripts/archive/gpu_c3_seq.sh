#!/bin/bash
# C3 (32 frames per launch): kernel durations in launch order inside one call (rocprofv3 --kernel-trace), per iteration index: where the batched E-step / M-step
# averages sit above their minima.   usage (GPU box): bash scripts/gpu_c3_seq.sh <tag> [config]
tag=${1:-c3seq}; cfg=${2:-c3}
R=$(pwd); O=$R/gpurun_out/$tag; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
PYTHONPATH=$R timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/c3seq -- python $R/bench.py --config $cfg --steps 12 --warmup 3 --no-cpu-baseline --no-legs --pmc off > $O/bench.log 2>&1
f=$(find /tmp/c3seq -name "*kernel_trace.csv" | head -1)
python - "$f" > $O/seq.txt <<'PY'
import csv, sys, collections
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
ev=[(r["Kernel_Name"].split("(")[0].replace("void tdlo::","").replace("tdlo::",""), int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
# split into calls at k_prune_pass1
calls=[]; cur=None
for e in ev:
    if e[0].startswith("k_prune_pass1") or e[0].startswith("k_prologue"):
        cur=[]; calls.append(cur)
    if cur is not None: cur.append(e)
calls=[c for c in calls if sum(1 for e in c if e[0].startswith("k_estep"))>=50][3:]
print("calls analysed:",len(calls))
E=collections.defaultdict(list); M=collections.defaultdict(list); G1=collections.defaultdict(list); G2=collections.defaultdict(list)
for c in calls:
    it=0; prev=None
    for e in c:
        if e[0].startswith("k_estep"):
            E[it].append(e[2]-e[1])
            if prev is not None: G1[it].append(e[1]-prev[2])
        elif e[0].startswith("k_mstep"):
            M[it].append(e[2]-e[1]); G2[it].append(e[1]-prev[2]); it+=1
        prev=e
    tot=c[-1][2]-c[0][1]
import statistics as st
print("it   E-step us (med)  gap  M-step us (med)  gap")
for it in sorted(E):
    if it<12 or it%8==0 or it>=48:
        print("%2d   %8.2f   %6.2f   %8.2f   %6.2f"%(it, st.median(E[it])/1e3, st.median(G2[it])/1e3 if G2[it] else 0, st.median(M[it])/1e3 if M[it] else 0, st.median(G1[it])/1e3 if G1[it] else 0))
print("per call: E total %.1f us, M total %.1f us, gaps E->M %.1f, M->E %.1f, call span %.1f us"%(
    sum(st.median(v) for v in E.values())/1e3, sum(st.median(v) for v in M.values())/1e3, sum(st.median(v) for v in G2.values())/1e3, sum(st.median(v) for v in G1.values())/1e3,
    st.median([c[-1][2]-c[0][1] for c in calls])/1e3))
pre=collections.defaultdict(list)
for c in calls:
    for e in c:
        if not (e[0].startswith("k_estep") or e[0].startswith("k_mstep")): pre[e[0]].append(e[2]-e[1])
for k,v in pre.items(): print("  %-40s %d per call-set, median %.1f us"%(k[:40],len(v)//len(calls),st.median(v)/1e3))
PY
tail -2 $O/bench.log | cut -c1-300; cat $O/seq.txt
