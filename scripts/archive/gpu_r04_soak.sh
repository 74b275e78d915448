#!/bin/bash
# Soak runs of tracking_step from the C++ caller (the route with a kernel that waits on the host, the LLE tail behind the results, the cloud read
# from pinned memory): half a million steady frames, then moving and partly hidden ropes; every return code is checked.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
g++ -O2 -std=c++17 scripts/ubench/track_cpp.cpp -o scripts/ubench/track_cpp -Ltrackdlo_amd -ltrackdlo_hip -Wl,-rpath,$R/trackdlo_amd || exit 1
mkdir -p gpurun_out/r04_soak
{
  FRAMES=500000 timeout 300 scripts/ubench/track_cpp 2>&1 | tail -2; echo "exit ${PIPESTATUS[0]}"
  FRAMES=200000 MOVE=10 timeout 300 scripts/ubench/track_cpp 2>&1 | tail -2; echo "exit ${PIPESTATUS[0]}"
  FRAMES=100000 MOVE=30 OCCL=1 timeout 300 scripts/ubench/track_cpp 2>&1 | tail -2; echo "exit ${PIPESTATUS[0]}"
} | tee gpurun_out/r04_soak/soak.log
