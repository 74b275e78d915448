#!/bin/bash
# Round 5's sweeps at the STATED tolerances, outliers adjudicated in place (scripts/fuzz_adjudicate.py): three counts per sweep.   usage: bash scripts/gpu_fuzz_sweeps.sh <tag> [scale]
tag=${1:-r05fuzz}; scale=${2:-10}
O=gpurun_out/$tag; mkdir -p $O
{
  echo "== the suite's fuzz tests, TDLO_SWEEP_SCALE=1"
  timeout 1500 python -m pytest tests/test_fuzz_gpu.py -m gpu -q -s 2>&1 | grep -E "passed|failed|outside_stated|UNEXPLAINED|MISMATCH|Error" | head -40
  for prec in 1 0; do
    echo "== tracker fuzz, $((50 * scale)) sequences, FUZZ_PREC=$prec"; FUZZ_PREC=$prec timeout 2400 python scripts/gpu_fuzz_tracker.py $((50 * scale)) 0 2>&1 | tail -25
    echo "== chain smoother sweep, 400 cases, FUZZ_PREC=$prec"; FUZZ_PREC=$prec timeout 1500 python scripts/gpu_fuzz_chain.py 400 2>&1 | tail -25
  done
  echo "== banded LLE M-step sweep, 1500 cases (fp64 mode)"; timeout 2400 python scripts/gpu_fuzz_band.py 1500 2>&1 | tail -40
  echo "== banded LLE M-step sweep, 400 cases, fp32 mode"; FUZZ_PREC=0 timeout 1500 python scripts/gpu_fuzz_band.py 400 2>&1 | tail -25
  echo "== tracking_step's short cuts on against off, bit for bit"; timeout 900 python scripts/gpu_fuzz_routes.py $((25 * scale)) 2>&1 | tail -4
} 2>&1 | grep -v amdgpu.ids | tee $O/fuzz.log
