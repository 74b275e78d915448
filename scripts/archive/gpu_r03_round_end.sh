#!/bin/bash
# Round 3, on the GPU box (through gpurun): the driver's own bench command (headline + legs), kernel-trace stats of the four configurations
# and of the pre-processing (LLE) registration, the HBM / VALU PMC passes, the E-step's SQ counters, the measured-number log.
# Results under gpurun_out/<tag>/; the summaries to keep are copied into profiles/ by hand.
# usage: bash scripts/gpu_r03_round_end.sh <tag>
tag=${1:-r03z}
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/$tag
mkdir -p $O
# 1. the driver's command: C2 headline + c3 / c4 / c5 legs + sustained + pre-processing leg
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_line_default.json 2> $O/bench_stderr.log
timeout 600 python bench.py --no-legs > $O/bench_line_c2.json 2>> $O/bench_stderr.log
for c in c3 c4 c5; do timeout 600 python bench.py --config $c > $O/bench_line_$c.json 2>> $O/bench_stderr.log; done
# 2. kernel-trace stats of the same workloads (no PMC child passes, no legs under the tracer)
cd /tmp
for c in c2 c3 c4 c5; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t_$c -- python $R/bench.py --config $c --no-cpu-baseline --no-legs --pmc off > $O/trace_$c.log 2>&1 </dev/null
  f=$(find $O/t_$c -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/kernel_stats_$c.csv
  rm -rf $O/t_$c
done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t_lle -- python $R/scripts/gpu_lle_time.py > $O/trace_lle.log 2>&1 </dev/null
f=$(find $O/t_lle -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/kernel_stats_lle_M30_to_512.csv
rm -rf $O/t_lle
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t_trk -- python $R/scripts/gpu_track.py > $O/trace_track.log 2>&1 </dev/null
f=$(find $O/t_trk -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/kernel_stats_tracking_step.csv
rm -rf $O/t_trk
# 3. HBM traffic: separate PMC passes (never together with a trace), C2
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/p_fetch -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-legs --pmc off > $O/pmc_fetch.log 2>&1 </dev/null
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/p_write -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-legs --pmc off > $O/pmc_write.log 2>&1 </dev/null
f=$(find $O/p_fetch -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp "$f" $O/pmc_fetch.csv
f=$(find $O/p_write -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp "$f" $O/pmc_write.csv
rm -rf $O/p_fetch $O/p_write
cd $R
[ -f $O/pmc_fetch.csv ] && [ -f $O/pmc_write.csv ] && python scripts/pmc_summary.py $O/pmc_fetch.csv $O/pmc_write.csv 50000 $O/pmc_hbm.json > $O/pmc_summary.log 2>&1
# 4. the E-step's SQ counters with the GPU full (N = 2 000 000)
bash scripts/gpu_estep_pmc.sh $tag 2000000 50 0 2>&1 | grep -v amdgpu.ids > $O/estep_sq_counters_c4.txt
# 5. measured numbers quoted in DESIGN.md
{
  echo "== band stamps (k_mstep_band phases, shader clocks; instrumented build)"; [ -f scripts/tmp/libtrackdlo_stamps.so ] || bash scripts/build_variant.sh stamps -DTDLO_ESTEP_STAMPS -DTDLO_CHAIN_STAMPS > /dev/null 2>&1; timeout 200 python scripts/gpu_band_stamps.py
  echo "== stamps (C2 chain M-step phases)"; timeout 200 python scripts/gpu_stamps.py
  echo "== lle (M-step with the LLE term over M: banded L D L^T)"; timeout 200 python scripts/gpu_lle_time.py
  echo "== lle, dense pivoted comparators (TDLO_MSTEP_LLE=dense)"; TDLO_MSTEP_LLE=dense timeout 300 python scripts/gpu_lle_time.py
  echo "== track"; timeout 200 python scripts/gpu_track.py
  echo "== track without the sorted-cloud reuse (TDLO_REUSE_SORT=0) / without spin-polled waits (TDLO_SPIN_US=0)"; TDLO_REUSE_SORT=0 timeout 200 python scripts/gpu_track.py; TDLO_SPIN_US=0 timeout 200 python scripts/gpu_track.py
  echo "== c5_5it";  ITERS=5 timeout 200 python scripts/gpu_c5.py
  echo "== pcie"; timeout 200 python scripts/gpu_pcie.py
  echo "== f64 MFMA and friends on a lone wave (scripts/ubench/mfma64.hip)"; [ -x scripts/ubench/mfma64 ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 scripts/ubench/mfma64.hip -o scripts/ubench/mfma64 2>/dev/null; timeout 60 ./scripts/ubench/mfma64
  echo "== instruction latencies of a lone wave"; [ -x scripts/ubench/lat ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 scripts/ubench/lat.hip -o scripts/ubench/lat 2>/dev/null; timeout 60 ./scripts/ubench/lat
} 2>&1 | grep -v amdgpu.ids > $O/measured.log
ls -la $O
cat $O/pmc_summary.log | head -12
tail -5 $O/bench_stderr.log
