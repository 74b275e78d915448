#!/bin/bash
# Round 5, on the GPU box (through gpurun): the driver's own bench command (headline + legs), kernel-trace stats of the four configurations, of the
# pre-processing (LLE) registration and of tracking_step (Python and C++ callers, with a one-frame timeline), the HBM PMC passes, the E-step's SQ
# counters with the GPU full, and the measured-number log (phase stamps of both one-workgroup M-steps, of the fused prologue, of the E-step).
# The instrumented libraries are REBUILT first (round 3's log held two tracebacks from a stale one), and every section that fails is named at the
# end and makes the script exit non-zero.  Results under gpurun_out/<tag>/; scripts/collect_profiles.sh copies the summaries into profiles/.
# New in round 5: kernel stats and phase stamps of the one-launch depth -> cloud kernel, the parts of a device-born frame, the workgroup ubench.
# usage: bash scripts/gpu_r05_round_end.sh <tag>
tag=${1:-r05z}
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/$tag
mkdir -p $O
FAILED=""
run() {   # run <name> <log or -> <command...>: a section; its failure is recorded, not swallowed
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
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o scripts/ubench/wg1024 scripts/ubench/wg1024.hip > /dev/null 2>&1 || FAILED="$FAILED build_wg1024"
g++ -O2 -std=c++17 scripts/ubench/track_cpp.cpp -o scripts/ubench/track_cpp -Ltrackdlo_amd -ltrackdlo_hip -Wl,-rpath,$R/trackdlo_amd || FAILED="$FAILED build_track_cpp"
# 1. the driver's command: C2 headline + c3 / c4 / c5 legs + sustained + pre-processing leg; then every configuration on its own
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
run track_timeline $O/track_timeline_run.log bash scripts/gpu_r04_track_trace.sh ${tag}_tl
cp gpurun_out/r04_track_trace_${tag}_tl/timeline.txt $O/tracking_step_timeline.txt 2>/dev/null || FAILED="$FAILED timeline_copy"
run track_timeline_classic $O/track_timeline_classic_run.log bash scripts/gpu_r04_track_trace.sh ${tag}_tlc TDLO_HOST_MAILBOX=0 TDLO_DIRECT_UPLOAD=0
cp gpurun_out/r04_track_trace_${tag}_tlc/timeline.txt $O/tracking_step_timeline_copy_route.txt 2>/dev/null
run track_timeline_hidden $O/track_timeline_hidden_run.log bash scripts/gpu_r04_track_trace.sh ${tag}_tlh OCCL=1
cp gpurun_out/r04_track_trace_${tag}_tlh/timeline.txt $O/tracking_step_timeline_hidden_nodes.txt 2>/dev/null || FAILED="$FAILED timeline_hidden_copy"
run track_timeline_hidden_after $O/track_timeline_hidden_after_run.log bash scripts/gpu_r04_track_trace.sh ${tag}_tlha OCCL=1 TDLO_AHEAD=0
cp gpurun_out/r04_track_trace_${tag}_tlha/timeline.txt $O/tracking_step_timeline_hidden_nodes_ahead_off.txt 2>/dev/null
# 3. HBM traffic: separate PMC passes (never together with a trace), C2
cd /tmp
run pmc_fetch $O/pmc_fetch.log timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/p_fetch -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-legs --pmc off </dev/null
run pmc_write $O/pmc_write.log timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/p_write -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-legs --pmc off </dev/null
f=$(find $O/p_fetch -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp "$f" $O/pmc_fetch.csv
f=$(find $O/p_write -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp "$f" $O/pmc_write.csv
rm -rf $O/p_fetch $O/p_write
cd $R
run pmc_summary $O/pmc_summary.log python scripts/pmc_summary.py $O/pmc_fetch.csv $O/pmc_write.csv 50000 $O/pmc_hbm.json
# 4. the E-step's SQ counters with the GPU full (N = 2 000 000)
bash scripts/gpu_estep_pmc.sh $tag 2000000 50 0 2>&1 | grep -v amdgpu.ids > $O/estep_sq_counters_c4.txt
grep -q SQ_INSTS_VALU $O/estep_sq_counters_c4.txt || FAILED="$FAILED estep_sq_counters"
# 5. measured numbers quoted in DESIGN.md: every sub-section reports its own failure
sec() { echo "== $1"; shift; "$@" 2>&1 | grep -v amdgpu.ids; local rc=${PIPESTATUS[0]}; if [ $rc -ne 0 ]; then echo "!! FAILED (rc $rc): $*"; FAILED="$FAILED measured:$1"; fi; }
{
  sec "band stamps (k_mstep_band phases, shader clocks; instrumented build)" timeout 200 python scripts/gpu_band_stamps.py
  sec "chain stamps (k_mstep_chain phases over chain lengths; instrumented build)" env TDLO_ALT_LIB=scripts/tmp/libtrackdlo_stamps.so timeout 200 python scripts/gpu_chain_stamps.py
  sec "stamps (C2: E-step / chain M-step phases and the iteration's timeline)" timeout 200 python scripts/gpu_stamps.py
  sec "prologue stamps (k_prologue: node workgroup / point workgroup 0)" timeout 200 python scripts/gpu_prologue_stamps.py
  sec "E-step phases at N = 2 000 000 (clocks per wave of four batches)" timeout 200 python scripts/gpu_ephases.py 2000000 50
  sec "E-step phases at N = 2 000 000, TDLO_WINDOW=exact" env TDLO_WINDOW=exact timeout 200 python scripts/gpu_ephases.py 2000000 50
  sec "lle (M-step with the LLE term over M: banded L D L^T)" timeout 200 python scripts/gpu_lle_time.py
  sec "track (Python caller, stream markers on)" timeout 200 python scripts/gpu_track.py
  sec "track C++ caller: fast path" scripts/ubench/track_cpp
  sec "track C++ caller: fast path, host profile" env TDLO_TRACK_PROFILE=1 scripts/ubench/track_cpp
  sec "track C++ caller: copy route (TDLO_HOST_MAILBOX=0 TDLO_DIRECT_UPLOAD=0)" env TDLO_HOST_MAILBOX=0 TDLO_DIRECT_UPLOAD=0 scripts/ubench/track_cpp
  sec "track C++ caller: mailbox only (TDLO_DIRECT_UPLOAD=0)" env TDLO_DIRECT_UPLOAD=0 scripts/ubench/track_cpp
  sec "track C++ caller: fused prologue only (TDLO_HOST_MAILBOX=0)" env TDLO_HOST_MAILBOX=0 scripts/ubench/track_cpp
  sec "track C++ caller: without the sorted-cloud reuse (TDLO_REUSE_SORT=0)" env TDLO_REUSE_SORT=0 scripts/ubench/track_cpp
  sec "track C++ caller: cloud copied in front of the prologue (TDLO_DIRECT_CLOUD=0)" env TDLO_DIRECT_CLOUD=0 scripts/ubench/track_cpp
  sec "track C++ caller: LLE regulariser on the host (TDLO_LLE_NEXT=0)" env TDLO_LLE_NEXT=0 scripts/ubench/track_cpp
  sec "track C++ caller: main registration's M-step launched with its priors (TDLO_SPEC_MSTEP=0)" env TDLO_SPEC_MSTEP=0 scripts/ubench/track_cpp
  sec "track C++ caller: main registration runs its own first E-step (TDLO_PAIR_SUMS=0)" env TDLO_PAIR_SUMS=0 scripts/ubench/track_cpp
  sec "track C++ caller: main registration launches its own set-up (TDLO_PAIR_SETUP=0)" env TDLO_PAIR_SETUP=0 scripts/ubench/track_cpp
  sec "track C++ caller: round 4's first form (TDLO_PAIR_SETUP=0 TDLO_LLE_NEXT=0 TDLO_DIRECT_CLOUD=0)" env TDLO_PAIR_SETUP=0 TDLO_LLE_NEXT=0 TDLO_DIRECT_CLOUD=0 scripts/ubench/track_cpp
  sec "track C++ caller: every frame a new cloud of a moving rope (MOVE=10: 1 mm per frame, fresh noise)" env MOVE=10 scripts/ubench/track_cpp
  sec "track C++ caller: the same, round 4's first form (TDLO_PAIR_SETUP=0 TDLO_LLE_NEXT=0 TDLO_DIRECT_CLOUD=0)" env MOVE=10 TDLO_PAIR_SETUP=0 TDLO_LLE_NEXT=0 TDLO_DIRECT_CLOUD=0 scripts/ubench/track_cpp
  sec "track C++ caller: the same, copy route (TDLO_HOST_MAILBOX=0 TDLO_DIRECT_UPLOAD=0)" env MOVE=10 TDLO_HOST_MAILBOX=0 TDLO_DIRECT_UPLOAD=0 scripts/ubench/track_cpp
  sec "track C++ caller: nodes 18-24 hidden (the registrations start from different node sets: no pairing; the pre-processing one takes ~6 iterations; the main one's first iteration runs beside it on the second stream)" env OCCL=1 scripts/ubench/track_cpp
  sec "track C++ caller: the same, main registration launched when the pre-processing one has returned (TDLO_AHEAD=0)" env OCCL=1 TDLO_AHEAD=0 scripts/ubench/track_cpp
  sec "track C++ caller: nodes 18-24 hidden, host profile" env OCCL=1 TDLO_TRACK_PROFILE=1 scripts/ubench/track_cpp
  sec "track C++ caller: nodes 18-24 hidden with a moving rope" env OCCL=1 MOVE=10 scripts/ubench/track_cpp
  sec "track C++ caller: the same, TDLO_AHEAD=0" env OCCL=1 MOVE=10 TDLO_AHEAD=0 scripts/ubench/track_cpp
  sec "track C++ caller: the same, copy route (TDLO_HOST_MAILBOX=0 TDLO_DIRECT_UPLOAD=0)" env OCCL=1 MOVE=10 TDLO_HOST_MAILBOX=0 TDLO_DIRECT_UPLOAD=0 scripts/ubench/track_cpp
  sec "depth -> cloud, ms per call (one launch / multi-launch; pageable / pinned images)" timeout 300 python scripts/gpu_cloud_time.py
  sec "depth -> cloud: phases of k_cloud_fused's finishing workgroup (instrumented build)" env TDLO_LIBRARY=scripts/tmp/libtrackdlo_cstamps.so timeout 200 python scripts/gpu_cloud_stamps.py
  sec "device-born frame, its three parts" timeout 200 python scripts/gpu_frame_parts.py
  sec "one 1024-thread workgroup: what its basic steps cost (scripts/ubench/wg1024.hip)" scripts/ubench/wg1024
  sec "E-step phases at C5 (N = 200 000, M = 300, fp64) over the iterations" timeout 300 python scripts/gpu_ephases.py 200000 300 1 5
  sec "c5_5it" env ITERS=5 timeout 200 python scripts/gpu_c5.py
  sec "pcie" timeout 200 python scripts/gpu_pcie.py
} > $O/measured.log 2>&1
ls -la $O
head -12 $O/pmc_summary.log
tail -3 $O/bench_stderr.log
tail -c 2500 $O/bench_line_default.json
if [ -n "$FAILED" ]; then echo "FAILED SECTIONS:$FAILED"; exit 1; fi
echo "all sections OK"
