#!/bin/bash
R=$(pwd); O=$R/gpurun_out/r05b; mkdir -p $O
timeout 900 python -m pytest tests/test_cloud_gpu.py -x -q -m gpu > $O/cloud_tests.log 2>&1; echo "rc=$?" >> $O/cloud_tests.log
tail -25 $O/cloud_tests.log | cut -c1-250
timeout 300 python scripts/gpu_cloud_time.py fused 2>&1 | grep -v amdgpu.ids | tee $O/cloud_time.log
TDLO_LIBRARY=scripts/tmp/libtrackdlo_cstamps.so timeout 300 python scripts/gpu_cloud_stamps.py 2>&1 | grep -v amdgpu.ids | tee $O/cloud_stamps.log
cd /tmp && export TMPDIR=/tmp && PYTHONPATH=$R timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python $R/scripts/gpu_cloud_time.py fused > $O/prof.log 2>&1 </dev/null
cd $R; find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'head -8 {} | cut -c1-200'
