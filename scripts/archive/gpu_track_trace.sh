#!/bin/bash
# Kernel + memory-copy trace of tracking_step at production size (N = 5000, M = 45): where a frame's 0.2 ms go.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/track_trace
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O -- python $R/scripts/gpu_track.py > $O/run.log 2>&1
tail -2 $O/run.log
find $O -name "*.csv" | head
