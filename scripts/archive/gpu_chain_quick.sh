#!/bin/bash
# quick look at the chain M-step: a few parity tests, stamps, C2 / C5 timings
export TMPDIR=/tmp
mkdir -p gpurun_out/quick
timeout 900 python -m pytest tests/test_parity_gpu.py -m gpu -q -x -k "${1:-not lle and not redoes}" > gpurun_out/quick/pytest.log 2>&1; tail -3 gpurun_out/quick/pytest.log
python scripts/gpu_stamps.py 2>&1 | grep -E "loop_ms|^stamps|estep "
ITERS=5 python scripts/gpu_c5.py 2>&1 | grep "N="
