#!/bin/bash
# First contact with a multi-GPU MI355X box.  Nothing of the multi-GPU path has crossed xGMI between two physical GPUs yet
# (gpurun boxes have one GPU; the driver's 8-GPU tier was skipped in rounds 1-4): this is the one command for whoever gets such a node.
#   usage: bash scripts/gpu_multi_first_contact.sh [max_gpus]          (default: all visible, at most 8)
# What it runs, each step logged under gpurun_out/first_contact/ and summarised at the end:
#   1. the C++ N-split driver (tests/cpp/split_run_test): two ranks on devices 0 / 1 -- the one-shot exchange's peer stores cross xGMI
#   2. bench.py --config c3 (frames sharded, barrier only) on 2, 4, 8 GPUs
#      (the driver's own scaling command, `bench.py --gpus N` with the default config, runs c3 and both c4 forms as legs behind its headline as well: configs.c3 / c4 / c4_rccl)
#   3. bench.py --config c4 (N = 2 000 000 split; per iteration the 4M+2 sums exchanged) on 2, 4, 8 GPUs in BOTH exchange forms:
#      default (one-shot exchange: peer-written inboxes, IPC handles) and TDLO_BENCH_FORCE_RCCL=1 (the library's ncclAllReduce calls)
#   From every line: n_gpus, value, us_per_iteration, the form that ran, ranks[].rccl_size, xch_can_access (rank x rank peer-mapping matrix),
#   ranks_agree and ranks[].y_sha1 (every rank's nodes + sigma2 hashed: the N-split solves the same system on every rank -- same bits or it is wrong).
# TDLO_FIRST_CONTACT_DRYRUN=1 (tests/test_bench_launch.py): the same steps with the stand-in context under gloo on a box without GPUs (step 1 skipped).
cd "$(dirname "$0")/.."
O=gpurun_out/first_contact; mkdir -p $O
DRY=${TDLO_FIRST_CONTACT_DRYRUN:-0}
if [ "$DRY" = "1" ]; then
  NG=${1:-2}; STEPS="--steps 2 --warmup 1"
  export TDLO_BENCH_BACKEND=gloo TDLO_BENCH_STUB=bench_stub:StubContext TDLO_HIP_RUNTIME=system PYTHONPATH=$(pwd)/tests:$PYTHONPATH
else
  NG=$(python -c "import torch; print(min(8, torch.cuda.device_count()))")
  [ -n "$1" ] && NG=$(( $1 < NG ? $1 : NG ))
  STEPS=""
  export HSA_ENABLE_IPC_MODE_LEGACY=0
  if [ "$NG" -lt 2 ]; then echo "first contact needs at least 2 GPUs (found $NG)"; exit 2; fi
  echo "== 1. tests/cpp/split_run_test (two ranks, devices 0 / 1)" | tee $O/summary.txt
  python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
  timeout 600 tests/cpp/split_run_test > $O/split_run_test.log 2>&1; echo "   rc=$? $(tail -1 $O/split_run_test.log)" | tee -a $O/summary.txt
fi
port=29700
fail=0
for cfgname in c3 c4; do
  for n in 2 4 8; do
    [ $n -gt $NG ] && continue
    for form in oneshot rccl; do
      [ $cfgname = c3 ] && [ $form = rccl ] && continue
      port=$((port + 1))
      tag=${cfgname}_${n}gpu_${form}
      echo "== bench.py --config $cfgname --gpus $n ($form)" | tee -a $O/summary.txt
      if [ $form = rccl ]; then export TDLO_BENCH_FORCE_RCCL=1; else unset TDLO_BENCH_FORCE_RCCL; fi
      TDLO_BENCH_PORT=$port timeout 1800 python bench.py --config $cfgname --gpus $n --no-cpu-baseline $STEPS > $O/$tag.log 2> $O/$tag.err
      rc=$?
      python - "$O/$tag.log" $rc <<'PY' | tee -a $O/summary.txt
import json, sys
rc = int(sys.argv[2])
lines = [l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")]
if rc != 0 or not lines:
    print(f"   FAILED rc={rc}: no JSON line (see the .err file)"); sys.exit(3)
d = json.loads(lines[-1])
out = f"   {'STUB (stand-in context, NOT a measurement)  ' if d.get('data') == 'stub' else ''}n_gpus {d['n_gpus']}  value {d['value']} {d['unit']}  ms_per_step {d['ms_per_step']}"
if "us_per_iteration" in d:
    out += f"  us_per_iteration {d['us_per_iteration']}  form: {d['config']['parallelism'].split('per iteration: ')[-1][:60]}"
    out += f"\n   rccl_size {[r.get('rccl_size') for r in d['ranks']]}  xch_can_access {d.get('xch_can_access')}"
    out += f"\n   ranks_agree {d.get('ranks_agree')}  y_sha1 {[r.get('y_sha1') for r in d['ranks']]}"
print(out)
sys.exit(0 if d.get("ranks_agree", True) else 4)
PY
      [ ${PIPESTATUS[0]} -ne 0 ] && fail=1
    done
  done
done
unset TDLO_BENCH_FORCE_RCCL
echo "== done: $([ $fail = 0 ] && echo 'every run produced a line and the ranks agree' || echo 'SOMETHING FAILED, see above')" | tee -a $O/summary.txt
exit $fail
