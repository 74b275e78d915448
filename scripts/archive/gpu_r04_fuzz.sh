#!/bin/bash
# Round 4's long sweeps (on the GPU box): the tracker fuzz of the suite scaled up, and round 3's harnesses again on the new one-frame route
# (fused prologue, results mailbox, late priors) and the relative node window.   usage: bash scripts/gpu_r04_fuzz.sh <tag> [scale]
tag=${1:-r04fuzz}; scale=${2:-20}
O=gpurun_out/$tag; mkdir -p $O
{
  echo "== tracker fuzz, TDLO_SWEEP_SCALE=$scale ($((50 * scale)) sequences per precision)"
  TDLO_SWEEP_SCALE=$scale timeout 1500 python -m pytest tests/test_fuzz_gpu.py -m gpu -q 2>&1 | grep -E "passed|failed|MISMATCH|Error" | head -20
  echo "== tracking_step's short cuts on against off, bit for bit"; timeout 900 python scripts/gpu_fuzz_routes.py $((25 * scale)) 2>&1 | tail -8
  echo "== chain smoother sweep (fp64 mode)"; timeout 900 python scripts/gpu_fuzz_chain.py 400 2>&1 | tail -8
  echo "== banded LLE M-step sweep"; timeout 900 python scripts/gpu_fuzz_band.py 1500 2>&1 | tail -8
  echo "== batches against single calls, bit for bit"; timeout 900 python scripts/gpu_fuzz_batch.py 200 2>&1 | tail -5
  echo "== N-split against the unsplit call"; GPU_MAX_HW_QUEUES=16 timeout 900 python scripts/gpu_fuzz_split.py 100 2>&1 | tail -5
  echo "== randomised sweeps of the suite x 8"; TDLO_SWEEP_SCALE=8 timeout 1500 python -m pytest tests/test_parity_gpu.py -m gpu -q -k "randomised" 2>&1 | grep -E "passed|failed|Error" | head
} 2>&1 | grep -v amdgpu.ids | tee $O/fuzz.log
