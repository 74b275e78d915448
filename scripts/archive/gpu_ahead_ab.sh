#!/bin/bash
# tracking_step with hidden nodes from the C++ caller: the main registration's first iteration beside the pre-processing registration (default)
# against TDLO_AHEAD=0, at rest and with a moving rope, alternating.   usage: bash scripts/gpu_ahead_ab.sh [rounds]
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
g++ -O2 -std=c++17 scripts/ubench/track_cpp.cpp -o scripts/ubench/track_cpp -Ltrackdlo_amd -ltrackdlo_hip -Wl,-rpath,$R/trackdlo_amd || exit 1
n=${1:-3}
for i in $(seq $n); do
  echo "-- round $i"
  echo -n "at rest, beside:   "; OCCL=1 scripts/ubench/track_cpp | tail -1
  echo -n "at rest, after:    "; OCCL=1 TDLO_AHEAD=0 scripts/ubench/track_cpp | tail -1
  echo -n "moving, beside:    "; OCCL=1 MOVE=10 scripts/ubench/track_cpp | tail -1
  echo -n "moving, after:     "; OCCL=1 MOVE=10 TDLO_AHEAD=0 scripts/ubench/track_cpp | tail -1
done
echo -n "every node visible: "; scripts/ubench/track_cpp | tail -1
