#!/bin/bash
tag=${1:-r02d}
export TMPDIR=/tmp
R=$PWD
mkdir -p $R/gpurun_out/$tag
timeout 1200 python -m pytest tests/test_parity_gpu.py -m gpu -q -k "lle or timed_out or one_workgroup or randomised_conf" > $R/gpurun_out/$tag/pytest.log 2>&1
tail -5 $R/gpurun_out/$tag/pytest.log
timeout 300 python scripts/gpu_lle_time.py 2>&1 | grep -v amdgpu.ids | tail -20
