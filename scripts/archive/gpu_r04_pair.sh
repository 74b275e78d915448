#!/bin/bash
# The paired set-up of tracking_step (round 4): parity tests of the route, then the per-frame time with and without it from the C++ caller.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04_pair
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_direct_path_gpu.py tests/test_sort_reuse_gpu.py tests/test_fuzz_gpu.py tests/test_lle_device_gpu.py -m gpu -q 2>&1 | tail -5 | tee $O/tests.log
g++ -O2 -std=c++17 scripts/ubench/track_cpp.cpp -o scripts/ubench/track_cpp -Ltrackdlo_amd -ltrackdlo_hip -Wl,-rpath,$R/trackdlo_amd || exit 1
for rep in 1 2 3; do
  echo "pair on:  $(scripts/ubench/track_cpp 2>&1 | tail -1)" | tee -a $O/time.log
  echo "pair on, cloud copied: $(TDLO_DIRECT_CLOUD=0 scripts/ubench/track_cpp 2>&1 | tail -1)" | tee -a $O/time.log
  echo "pair on, host LLE: $(TDLO_LLE_NEXT=0 scripts/ubench/track_cpp 2>&1 | tail -1)" | tee -a $O/time.log
  echo "pair on, M-step launched with its priors: $(TDLO_SPEC_MSTEP=0 scripts/ubench/track_cpp 2>&1 | tail -1)" | tee -a $O/time.log
  echo "pair on, own first E-step: $(TDLO_PAIR_SUMS=0 scripts/ubench/track_cpp 2>&1 | tail -1)" | tee -a $O/time.log
  echo "pair off: $(TDLO_PAIR_SETUP=0 scripts/ubench/track_cpp 2>&1 | tail -1)" | tee -a $O/time.log
done
TDLO_TRACK_PROFILE=1 scripts/ubench/track_cpp 2>&1 | tail -20 | tee $O/hostprof.log
