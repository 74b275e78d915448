#!/bin/bash
# Runs on the GPU box (through gpurun): kernel-trace stats and the two PMC passes for bench.py; results under gpurun_out/.
# usage: bash scripts/gpu_profile.sh <tag>
tag=${1:-r01x}
export TMPDIR=/tmp
R=$PWD
mkdir -p $R/gpurun_out/$tag
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$tag/trace -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $R/gpurun_out/$tag/bench_trace.log 2>&1 </dev/null
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/$tag/fetch -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $R/gpurun_out/$tag/bench_fetch.log 2>&1 </dev/null
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/$tag/write -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $R/gpurun_out/$tag/bench_write.log 2>&1 </dev/null
cd $R
f=$(find gpurun_out/$tag/trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/$tag/kernel_stats.csv
f=$(find gpurun_out/$tag/fetch -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/$tag/pmc_fetch.csv
f=$(find gpurun_out/$tag/write -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/$tag/pmc_write.csv
rm -rf gpurun_out/$tag/trace gpurun_out/$tag/fetch gpurun_out/$tag/write
ls -la gpurun_out/$tag
[ -f gpurun_out/$tag/kernel_stats.csv ] && head -8 gpurun_out/$tag/kernel_stats.csv
tail -1 gpurun_out/$tag/bench_trace.log | cut -c1-600
