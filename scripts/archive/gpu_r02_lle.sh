#!/bin/bash
tag=${1:-r02b}
export TMPDIR=/tmp
R=$PWD
mkdir -p $R/gpurun_out/$tag
timeout 900 python scripts/gpu_lle_gates.py 192 > $R/gpurun_out/$tag/lle_gates.log 2>&1
cat $R/gpurun_out/$tag/lle_gates.log | grep -v amdgpu.ids
timeout 1200 python -m pytest tests/test_parity_gpu.py -m gpu -q -k "lle or tracking or randomised or timed_out or one_workgroup or cpp_drop_in" > $R/gpurun_out/$tag/pytest.log 2>&1
tail -40 $R/gpurun_out/$tag/pytest.log
TDLO_LLE_TIME=1 timeout 300 python scripts/gpu_lle_time.py 2>&1 | grep -v amdgpu.ids | tail -20
