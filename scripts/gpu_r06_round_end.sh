#!/bin/bash
# Round 6, on the GPU box (through gpurun): everything profiles/r06_* is made from.  The driver's own bench command (headline + legs), every configuration on
# its own, kernel-trace stats of the four configurations / the LLE registration / tracking_step / depth -> cloud, the HBM PMC passes, the E-step's SQ counters with
# the GPU full for BOTH E-step kernels (k_estep2: the default there; k_estep: TDLO_ESTEP2=0), and the measured-number log DESIGN.md quotes.
# Instrumented libraries are rebuilt first; every section that fails is named at the end and makes the script exit non-zero.
# Results under gpurun_out/<tag>/; scripts/collect_profiles.sh <tag> r06 copies the summaries into profiles/.
# usage: bash scripts/gpu_r06_round_end.sh <tag>
tag=${1:-r06z}
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/$tag
mkdir -p $O
FAILED=""
run() {   # run <name> <log or -> <command...>
  local name=$1 log=$2; shift 2
  if [ "$log" = "-" ]; then "$@"; else "$@" > "$log" 2>&1; fi
  local rc=$?
  if [ $rc -ne 0 ]; then FAILED="$FAILED $name(rc=$rc)"; echo "!! section $name FAILED with rc $rc" >&2; fi
  return $rc
}
# 0. instrumented builds, always fresh
run build_stamps $O/build_stamps.log bash scripts/build_variant.sh stamps -DTDLO_ESTEP_STAMPS -DTDLO_CHAIN_STAMPS
run build_phases $O/build_phases.log bash scripts/build_variant.sh phases -DTDLO_ESTEP_PHASES
run build_cstamps $O/build_cstamps.log bash scripts/build_variant.sh cstamps -DTDLO_CLOUD_STAMPS
g++ -O2 -std=c++17 scripts/ubench/track_cpp.cpp -o scripts/ubench/track_cpp -Ltrackdlo_amd -ltrackdlo_hip -Wl,-rpath,$R/trackdlo_amd || FAILED="$FAILED build_track_cpp"
# 1. the driver's command: C2 headline + c3 / c4 / c5 legs + sustained + pre-processing + frame-from-depth legs; then every configuration on its own
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_line_default.json 2> $O/bench_stderr.log || FAILED="$FAILED bench_default"
cp bench_detail.json $O/bench_detail_default.json
timeout 600 python bench.py --no-legs > $O/bench_line_c2.json 2>> $O/bench_stderr.log || FAILED="$FAILED bench_c2"; cp bench_detail.json $O/bench_detail_c2.json
for c in c3 c4 c5; do timeout 600 python bench.py --config $c > $O/bench_line_$c.json 2>> $O/bench_stderr.log || FAILED="$FAILED bench_$c"; cp bench_detail.json $O/bench_detail_$c.json; done
# 2. kernel-trace stats of the same workloads (no PMC child passes, no legs under the tracer)
cd /tmp
for c in c2 c3 c4 c5; do
  run trace_$c $O/trace_$c.log timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t_$c -- python $R/bench.py --config $c --no-cpu-baseline --no-legs --pmc off </dev/null
  f=$(find $O/t_$c -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/kernel_stats_$c.csv || FAILED="$FAILED stats_$c"
  rm -rf $O/t_$c
done
run trace_lle $O/trace_lle.log timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t_lle -- python $R/scripts/gpu_lle_time.py </dev/null
f=$(find $O/t_lle -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/kernel_stats_lle_M30_to_512.csv; rm -rf $O/t_lle
run trace_track $O/trace_track.log timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t_trk -- $R/scripts/ubench/track_cpp </dev/null
f=$(find $O/t_trk -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/kernel_stats_tracking_step.csv || FAILED="$FAILED stats_track"; rm -rf $O/t_trk
run trace_cloud $O/trace_cloud.log env PYTHONPATH=$R timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t_cloud -- python $R/scripts/gpu_cloud_time.py </dev/null
f=$(find $O/t_cloud -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/kernel_stats_depth_to_cloud.csv || FAILED="$FAILED stats_cloud"; rm -rf $O/t_cloud
cd $R
run track_timeline $O/track_timeline_run.log bash scripts/gpu_track_trace.sh ${tag}_tl
cp gpurun_out/track_trace_${tag}_tl/timeline.txt $O/tracking_step_timeline.txt 2>/dev/null || FAILED="$FAILED timeline_copy"
run c3_timeline $O/c3_timeline.txt bash scripts/gpu_c3_timeline.sh 3
# 3. HBM traffic: separate PMC passes (never together with a trace), C2
cd /tmp
run pmc_fetch $O/pmc_fetch.log timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/p_fetch -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-legs --pmc off </dev/null
run pmc_write $O/pmc_write.log timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/p_write -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-legs --pmc off </dev/null
f=$(find $O/p_fetch -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp "$f" $O/pmc_fetch.csv
f=$(find $O/p_write -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp "$f" $O/pmc_write.csv
rm -rf $O/p_fetch $O/p_write
cd $R
run pmc_summary $O/pmc_summary.log python scripts/pmc_summary.py $O/pmc_fetch.csv $O/pmc_write.csv 50000 $O/pmc_hbm.json
# 4. the E-step's SQ counters with the GPU full (N = 2 000 000): the default kernel there (k_estep2), and k_estep (TDLO_ESTEP2=0) beside it
bash scripts/gpu_estep_pmc.sh $tag 2000000 50 0 2>&1 | grep -v amdgpu.ids > $O/estep_sq_counters_c4.txt
grep -q SQ_INSTS_VALU $O/estep_sq_counters_c4.txt || FAILED="$FAILED estep_sq_counters"
TDLO_ESTEP2=0 bash scripts/gpu_estep_pmc.sh ${tag}_e1 2000000 50 0 2>&1 | grep -v amdgpu.ids > $O/estep_sq_counters_c4_k_estep.txt
# 5. measured numbers quoted in DESIGN.md
sec() { echo "== $1"; shift; "$@" 2>&1 | grep -v amdgpu.ids; local rc=${PIPESTATUS[0]}; if [ $rc -ne 0 ]; then echo "!! FAILED (rc $rc): $*"; FAILED="$FAILED measured:$1"; fi; }
{
  sec "E-step kernels: k_estep2 against the oracle on forced small inputs, then k_estep / k_estep2 at C4, one C3 batch, 262 144 and 50 000 points" timeout 600 python scripts/gpu_estep2_check.py parity time
  sec "E-step phases at N = 2 000 000, k_estep2 (clocks of one wave; the stamps wait for the prefetch, so 'x loads' is an artefact of the instrumentation)" env TDLO_ESTEP2=1 timeout 200 python scripts/gpu_ephases.py 2000000 50
  sec "E-step phases at N = 2 000 000, k_estep (TDLO_ESTEP2=0)" env TDLO_ESTEP2=0 timeout 200 python scripts/gpu_ephases.py 2000000 50
  sec "k_estep2 over workgroup counts at C4 (TDLO_ESTEP2_BLOCKS)" bash scripts/gpu_estep2_sweep.sh "8" "782 977 1024 1280 1536 1954"
  sec "C3 over stream groups and TDLO_BATCH_CHAIN" bash scripts/gpu_c3_ns.sh "2 3 4" "0 1 2"
  sec "spin-ahead loop of one frame against the ordinary loop (TDLO_SPIN_AHEAD)" timeout 300 python scripts/gpu_spin_ab.py 3
  sec "stamps (C2: E-step / chain M-step phases and the iteration's timeline)" timeout 200 python scripts/gpu_stamps.py
  sec "chain stamps (k_mstep_chain phases over chain lengths; instrumented build)" env TDLO_ALT_LIB=scripts/tmp/libtrackdlo_stamps.so timeout 200 python scripts/gpu_chain_stamps.py
  sec "band stamps (k_mstep_band phases, shader clocks; instrumented build)" timeout 200 python scripts/gpu_band_stamps.py
  sec "lle (M-step with the LLE term over M: banded L D L^T)" timeout 200 python scripts/gpu_lle_time.py
  sec "track (Python caller, stream markers on)" timeout 200 python scripts/gpu_track.py
  sec "track C++ caller: fast path" scripts/ubench/track_cpp
  sec "track C++ caller: copy route (TDLO_HOST_MAILBOX=0 TDLO_DIRECT_UPLOAD=0)" env TDLO_HOST_MAILBOX=0 TDLO_DIRECT_UPLOAD=0 scripts/ubench/track_cpp
  sec "track C++ caller: every frame a new cloud of a moving rope (MOVE=10)" env MOVE=10 scripts/ubench/track_cpp
  sec "track C++ caller: nodes 18-24 hidden" env OCCL=1 scripts/ubench/track_cpp
  sec "depth -> cloud, ms per call (one launch / multi-launch; pageable / pinned images)" timeout 300 python scripts/gpu_cloud_time.py
  sec "device-born frame, its three parts" timeout 200 python scripts/gpu_frame_parts.py
  sec "E-step phases at C5 (N = 200 000, M = 300, fp64) over the iterations" timeout 300 python scripts/gpu_ephases.py 200000 300 1 5
  sec "c5_5it" env ITERS=5 timeout 200 python scripts/gpu_c5.py
  sec "pcie" timeout 200 python scripts/gpu_pcie.py
  sec "fp64 E-step of long chains: wide windows lane = node (default) against thread = point (TDLO_ESTEP_WIDE=0), C5" env MODES=0,129 timeout 400 python scripts/gpu_estep_wide_ab.py 200000 300 3
  sec "the same, per-iteration E-step durations of one call (rocprofv3 kernel trace)" env ITERS=50 CASES=1 timeout 400 bash scripts/gpu_estep_wide_trace.sh gpurun_out/$tag/wide_trace
  sec "a batch's loop as one launch (TDLO_BATCH_PERSIST=1, experiment) against the launch-per-step loop, C3" timeout 300 python scripts/gpu_batch_loop_ab.py 32 50000 2 50
  sec "loop timeline under the profiler: C2" bash scripts/gpu_loop_timeline.sh 50000 50 0
  sec "loop timeline under the profiler: C4" bash scripts/gpu_loop_timeline.sh 2000000 50 0
  sec "loop timeline under the profiler: C5" bash scripts/gpu_loop_timeline.sh 200000 300 1
} > $O/measured.log 2>&1
ls -la $O
head -12 $O/pmc_summary.log
tail -3 $O/bench_stderr.log
tail -c 2500 $O/bench_line_default.json
if [ -n "$FAILED" ]; then echo "FAILED SECTIONS:$FAILED"; exit 1; fi
echo "all sections OK"
