#!/bin/bash
# kernel-trace stats of the throughput configurations (32-frame batch, N = 2M single frame); results under gpurun_out/<tag>/
tag=${1:-r01x}
export TMPDIR=/tmp
R=$PWD
mkdir -p $R/gpurun_out/$tag
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$tag/t32 -- python $R/bench.py --frames 32 --steps 10 --warmup 2 --no-cpu-baseline > $R/gpurun_out/$tag/bench32_trace.log 2>&1 </dev/null
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$tag/tc4 -- python $R/scripts/gpu_estep_pmc.py 2000000 50 0 > $R/gpurun_out/$tag/c4_trace.log 2>&1 </dev/null
cd $R
f=$(find gpurun_out/$tag/t32 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/$tag/kernel_stats_32frames.csv
f=$(find gpurun_out/$tag/tc4 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/$tag/kernel_stats_c4.csv
rm -rf gpurun_out/$tag/t32 gpurun_out/$tag/tc4
head -6 gpurun_out/$tag/kernel_stats_32frames.csv; head -8 gpurun_out/$tag/kernel_stats_c4.csv
