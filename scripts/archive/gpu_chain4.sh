#!/bin/bash
# four-direction chain M-step: chain tests, then the whole GPU suite, C2 / C5 timings
export TMPDIR=/tmp
O=gpurun_out/${1:-c4}; mkdir -p $O
timeout 900 python -m pytest tests/test_mstep_chain.py -m gpu -q -x > $O/chain.log 2>&1; tail -4 $O/chain.log
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; tail -8 $O/pytest.log
python scripts/gpu_stamps.py 2>&1 | grep -E "loop_ms|^stamps|estep "
ITERS=5 python scripts/gpu_c5.py 2>&1 | grep "N="
timeout 600 python bench.py --no-cpu-baseline 2>/dev/null | cut -c1-300
