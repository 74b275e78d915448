#!/bin/bash
# Round 2, on the GPU box (through gpurun): bench lines of the four configurations, kernel-trace stats, the HBM PMC passes, the E-step's
# SQ counters, the measured-number log.  Results under gpurun_out/<tag>/; the summaries to keep are copied into profiles/ by hand.
# usage: bash scripts/gpu_r02_round_end.sh <tag>
tag=${1:-r02z}
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/$tag
mkdir -p $O
# 1. bench lines (c2 with the CPU baseline: the default command the driver runs)
timeout 600 python bench.py > $O/bench_line_c2.json 2> $O/bench_stderr.log
for c in c3 c4 c5; do timeout 600 python bench.py --config $c --no-cpu-baseline > $O/bench_line_$c.json 2>> $O/bench_stderr.log; done
# 2. kernel-trace stats of the same commands
cd /tmp
for c in c2 c3 c4 c5; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t_$c -- python $R/bench.py --config $c --no-cpu-baseline --pmc off > $O/trace_$c.log 2>&1 </dev/null
  f=$(find $O/t_$c -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/kernel_stats_$c.csv
  rm -rf $O/t_$c
done
# 3. HBM traffic: separate PMC passes (never together with a trace), C2
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/p_fetch -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $O/pmc_fetch.log 2>&1 </dev/null
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/p_write -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $O/pmc_write.log 2>&1 </dev/null
f=$(find $O/p_fetch -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp "$f" $O/pmc_fetch.csv
f=$(find $O/p_write -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp "$f" $O/pmc_write.csv
rm -rf $O/p_fetch $O/p_write
cd $R
[ -f $O/pmc_fetch.csv ] && [ -f $O/pmc_write.csv ] && python scripts/pmc_summary.py $O/pmc_fetch.csv $O/pmc_write.csv 50000 $O/pmc_hbm.json > $O/pmc_summary.log 2>&1
# 4. the E-step's SQ counters with the GPU full (N = 2 000 000)
bash scripts/gpu_estep_pmc.sh $tag 2000000 50 0 2>&1 | grep -v amdgpu.ids > $O/estep_sq_counters_c4.txt
# 5. measured numbers quoted in DESIGN.md
{
  echo "== stamps (C2 M-step phases, shader clocks; instrumented build)"; [ -f scripts/tmp/libtrackdlo_stamps.so ] || bash scripts/build_variant.sh stamps -DTDLO_ESTEP_STAMPS -DTDLO_CHAIN_STAMPS > /dev/null 2>&1; timeout 200 python scripts/gpu_stamps.py
  echo "== c5_5it";  ITERS=5 timeout 200 python scripts/gpu_c5.py
  echo "== c5_30it"; ITERS=30 timeout 200 python scripts/gpu_c5.py
  echo "== dense comparators (TDLO_MSTEP=dense)"; TDLO_MSTEP=dense ITERS=5 timeout 200 python scripts/gpu_c5.py
  echo "== dense comparator at C2"; TDLO_MSTEP=dense timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | cut -c1-400
  echo "== lle (M-step with the LLE term over M)"; timeout 200 python scripts/gpu_lle_time.py
  echo "== track"; timeout 200 python scripts/gpu_track.py
  echo "== pcie"; timeout 200 python scripts/gpu_pcie.py
  echo "== instruction latencies of a lone wave"; [ -x scripts/ubench/lat ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 scripts/ubench/lat.hip -o scripts/ubench/lat 2>/dev/null; timeout 60 ./scripts/ubench/lat
} 2>&1 | grep -v amdgpu.ids > $O/measured.log
ls -la $O
head -5 $O/kernel_stats_c2.csv
cat $O/pmc_summary.log
tail -30 $O/measured.log
python3 - $O/bench_line_c2.json <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith("{"):
        d=json.loads(l); print(json.dumps({k: d[k] for k in ("metric","value","n_gpus","ms_per_step","roofline","cpu_baseline") if k in d})[:1500])
PY
