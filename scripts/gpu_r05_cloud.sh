#!/bin/bash
R=$(pwd); O=$R/gpurun_out/r05b; mkdir -p $O
timeout 900 python -m pytest tests/test_cloud_gpu.py -x -q -m gpu > $O/cloud_tests.log 2>&1; echo "rc=$?" >> $O/cloud_tests.log
tail -15 $O/cloud_tests.log | cut -c1-250
echo "== team"; timeout 300 python scripts/gpu_cloud_time.py fused 2>&1 | grep -v amdgpu.ids
echo "== one finishing workgroup (TDLO_CLOUD_TEAM=0)"; TDLO_CLOUD_TEAM=0 timeout 300 python scripts/gpu_cloud_time.py fused 2>&1 | grep -v amdgpu.ids
TDLO_LIBRARY=scripts/tmp/libtrackdlo_cstamps.so timeout 300 python scripts/gpu_cloud_stamps.py 2>&1 | grep -v amdgpu.ids
