#!/bin/bash
# A/B of the results mailbox (TDLO_HOST_MAILBOX=0: read-back copy + stream wait of rounds 1-3) on the production-size tracking_step and on C2
set -u
out=gpurun_out/r04_mbox; mkdir -p $out
g++ -O2 -std=c++17 scripts/ubench/track_cpp.cpp -o scripts/ubench/track_cpp -Ltrackdlo_amd -ltrackdlo_hip -Wl,-rpath,$PWD/trackdlo_amd || exit 1
for rep in 1 2 3; do
  for v in 0 1; do echo -n "mailbox=$v  "; TDLO_HOST_MAILBOX=$v scripts/ubench/track_cpp; done
done 2>&1 | tee $out/track_cpp.txt
for v in 0 1; do
  echo "== bench c2 mailbox=$v"; TDLO_HOST_MAILBOX=$v python bench.py --no-legs --pmc off --no-cpu-baseline --steps 300 --warmup 20 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['iteration_us'])"
done 2>&1 | tee $out/bench_c2.txt
python -m pytest tests -m gpu -x -q > $out/gpu_suite.log 2>&1; echo "suite rc=$?"; tail -4 $out/gpu_suite.log
