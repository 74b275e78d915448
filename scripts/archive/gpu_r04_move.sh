#!/bin/bash
# tracking_step with a rope that keeps moving (registrations of several iterations), full route against the comparators
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
g++ -O2 -std=c++17 scripts/ubench/track_cpp.cpp -o scripts/ubench/track_cpp -Ltrackdlo_amd -ltrackdlo_hip -Wl,-rpath,$R/trackdlo_amd || exit 1
for mv in 2 10 30; do
  echo "full:      $(MOVE=$mv scripts/ubench/track_cpp 2>&1 | tail -1)"
  echo "no hint:   $(MOVE=$mv TDLO_ITER_HINT=0 scripts/ubench/track_cpp 2>&1 | tail -1)"
  echo "no spec:   $(MOVE=$mv TDLO_SPEC_MSTEP=0 scripts/ubench/track_cpp 2>&1 | tail -1)"
  echo "r4 first:  $(MOVE=$mv TDLO_PAIR_SETUP=0 TDLO_LLE_NEXT=0 TDLO_DIRECT_CLOUD=0 scripts/ubench/track_cpp 2>&1 | tail -1)"
done
echo "steady full:    $(scripts/ubench/track_cpp 2>&1 | tail -1)"
echo "steady no hint: $(TDLO_ITER_HINT=0 scripts/ubench/track_cpp 2>&1 | tail -1)"
timeout 600 python -m pytest tests/test_direct_path_gpu.py tests/test_fuzz_gpu.py -m gpu -q 2>&1 | tail -3
