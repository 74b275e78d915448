#!/bin/bash
# A/B of the one-frame fast path (fused prologue reading pinned host memory + results mailbox) against the copy route
set -u
out=gpurun_out/r04_direct; mkdir -p $out
g++ -O2 -std=c++17 scripts/ubench/track_cpp.cpp -o scripts/ubench/track_cpp -Ltrackdlo_amd -ltrackdlo_hip -Wl,-rpath,$PWD/trackdlo_amd || exit 1
python -m pytest tests/test_direct_path_gpu.py tests/test_sort_reuse_gpu.py -m gpu -x -q > $out/new_tests.log 2>&1; echo "new tests rc=$?"; tail -15 $out/new_tests.log
for rep in 1 2 3; do
  echo -n "classic          "; TDLO_HOST_MAILBOX=0 TDLO_DIRECT_UPLOAD=0 scripts/ubench/track_cpp
  echo -n "mailbox          "; TDLO_DIRECT_UPLOAD=0 scripts/ubench/track_cpp
  echo -n "direct           "; TDLO_HOST_MAILBOX=0 scripts/ubench/track_cpp
  echo -n "direct + mailbox "; scripts/ubench/track_cpp
done 2>&1 | tee $out/track_cpp.txt
for v in 0 1; do
  echo "== bench c2 mailbox=$v"; TDLO_HOST_MAILBOX=$v python bench.py --no-legs --pmc off --no-cpu-baseline --steps 300 --warmup 20 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['iteration_us'])"
done 2>&1 | tee $out/bench_c2.txt
python -m pytest tests -m gpu -x -q > $out/gpu_suite.log 2>&1; echo "suite rc=$?"; tail -4 $out/gpu_suite.log
