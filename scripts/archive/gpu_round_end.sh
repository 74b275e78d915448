#!/bin/bash
# Runs on the GPU box (through gpurun): the round's measured-number log and profiles; results under gpurun_out/<tag>/.
# usage: bash scripts/gpu_round_end.sh <tag>
tag=${1:-r01x}
export TMPDIR=/tmp
R=$PWD
mkdir -p $R/gpurun_out/$tag
bash scripts/gpu_profile.sh $tag > $R/gpurun_out/$tag/profile.log 2>&1
bash scripts/gpu_profile_extra.sh $tag >> $R/gpurun_out/$tag/profile.log 2>&1
{
  echo "== c5_5it";  ITERS=5 timeout 200 python scripts/gpu_c5.py
  echo "== c5_30it"; ITERS=30 timeout 200 python scripts/gpu_c5.py
  echo "== mcu (multi-CU M-step, default)"; timeout 200 python scripts/gpu_mcu.py dump $R/gpurun_out/$tag/mcu.npz
  echo "== 1wg (TDLO_MSTEP_BIG=1wg comparator)"; TDLO_MSTEP_BIG=1wg timeout 200 python scripts/gpu_mcu.py dump $R/gpurun_out/$tag/onewg.npz
  python scripts/gpu_mcu.py compare $R/gpurun_out/$tag/mcu.npz $R/gpurun_out/$tag/onewg.npz
  echo "== lle (M-step with the LLE term over M)"; timeout 200 python scripts/gpu_lle_time.py
  echo "== lle accuracy against the oracle"; timeout 200 python scripts/gpu_lle_acc.py
  echo "== c4";   timeout 300 python scripts/gpu_c4.py
  echo "== prod"; timeout 200 python scripts/gpu_prod.py
  echo "== track"; timeout 200 python scripts/gpu_track.py
  echo "== cloud"; timeout 200 python scripts/gpu_cloud.py
  echo "== pcie"; timeout 200 python scripts/gpu_pcie.py
} 2>&1 | grep -v amdgpu.ids > $R/gpurun_out/$tag/measured.log
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$tag/tc5 -- env ITERS=30 python $R/scripts/gpu_c5.py > /dev/null 2>&1 </dev/null
cd $R
f=$(find gpurun_out/$tag/tc5 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/$tag/kernel_stats_c5.csv
rm -rf gpurun_out/$tag/tc5
timeout 300 python bench.py > gpurun_out/$tag/bench_line.json 2> gpurun_out/$tag/bench_stderr.log
timeout 300 python bench.py --frames 32 --steps 10 --no-cpu-baseline > gpurun_out/$tag/bench_line_32frames.json 2>> gpurun_out/$tag/bench_stderr.log
timeout 300 python bench.py --mode nsplit --no-cpu-baseline > gpurun_out/$tag/bench_line_nsplit_1rank.json 2>> gpurun_out/$tag/bench_stderr.log
tail -40 gpurun_out/$tag/measured.log
cat gpurun_out/$tag/bench_line.json
