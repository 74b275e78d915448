#!/bin/bash
# Kernel + memory-copy trace of tracking_step at production size (N = 5000, M = 45) from the C++ caller: per-kernel averages and the
# timeline of one steady-state frame.   usage: gpu_track_trace.sh <tag> [ENV=VALUE ...]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=${1:-direct}; shift
O=$R/gpurun_out/track_trace_$tag
rm -rf $O; mkdir -p $O
(cd $R && g++ -O2 -std=c++17 scripts/ubench/track_cpp.cpp -o scripts/ubench/track_cpp -Ltrackdlo_amd -ltrackdlo_hip -Wl,-rpath,$R/trackdlo_amd) || exit 1
env "$@" rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $O -- $R/scripts/ubench/track_cpp > $O/run.log 2>&1
tail -1 $O/run.log
python3 $R/scripts/track_timeline.py $O | tee $O/timeline.txt
