#!/bin/bash
R=$(pwd); O=$R/gpurun_out/r05e; mkdir -p $O
timeout 1800 python -m pytest tests -x -q -m gpu -p no:cacheprovider > $O/gpu_suite.log 2>&1; echo "rc=$?" >> $O/gpu_suite.log
grep -E "passed|failed|rc=|Error|assert" $O/gpu_suite.log | tail -6
for c in c2 c3 c4; do timeout 600 python bench.py --config $c --no-cpu-baseline --pmc off $( [ $c = c2 ] && echo --no-legs ) 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$c', d['value'], d['ms_per_step'], [ (k['kernel'],k['avg_launch_us']) for k in d['roofline_kernels']])"; done
