#!/bin/bash
# Kernel stats of the per-call kernels (prune / setup / scatter) at C5 and C2 sizes.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/setup_trace
rm -rf $O; mkdir -p $O
ITERS=2 rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python $R/scripts/gpu_c5.py > $O/run.log 2>&1
f=$(find $O -name "*kernel_stats.csv" | head -1); grep -E "k_setup|k_prune" $f
