#!/bin/bash
# Per-iteration E-step durations of one C5 call (N = 200 000, M = 300, fp64) with the lane = node form for wide windows off (TDLO_ESTEP_WIDE=0) and on (from 129 nodes: the default): rocprofv3 kernel trace,
# the last call's first twelve and the converged E-steps.   usage: bash scripts/gpu_estep_wide_trace.sh <out dir>
O=$GRAFT_REPO_ROOT/${1:-gpurun_out/wide}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for mode in 0 129; do
  TDLO_ESTEP_WIDE=$mode rocprofv3 --kernel-trace --output-format csv -d $O/tr$mode -o t -- python $GRAFT_REPO_ROOT/scripts/gpu_c5.py > $O/run$mode.log 2>&1
  f=$(ls $O/tr$mode/*/t_kernel_trace.csv $O/tr$mode/t_kernel_trace.csv 2>/dev/null | head -1)
  python - "$f" $mode <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
es = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if r["Kernel_Name"].startswith("void tdlo::k_estep<double, 8")]
# gpu_c5.py's first configuration: two calls of ITERS (default 5) iterations -> use ITERS=50 via env
# gpu_c5.py with ITERS=50 CASES=1: two calls of 50 iterations, then the launches of its profile_kernel passes
print(f"TDLO_ESTEP_WIDE={sys.argv[2]}: second call's first 14 E-steps (us):", " ".join(f"{v:.0f}" for v in es[50:64]), "| the call's 50 E-steps together:", f"{sum(es[50:100]):.0f}", "us | converged:", f"{sum(es[70:100]) / 30:.1f}")
PY
done
