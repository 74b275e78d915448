#!/bin/bash
# C3 (32 frames per GPU) with 1..4 stream groups
for n in 1 2 3 4; do
  echo "== TDLO_BATCH_STREAMS=$n"
  TDLO_BATCH_STREAMS=$n timeout 600 python bench.py --config c3 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['value'], d.get('em_loop_only_iters_per_s'), [(o['kernel'],o['avg_launch_us']) for o in d['roofline_kernels']])"
done
