#!/bin/bash
# C3 (32 frames per GPU) and C2 with a cap on the E-step workgroups per frame (each wave then loops over several 64-point batches)
for n in 0 98 49 28; do
  echo "== TDLO_ESTEP_BLOCKS=$n"
  for c in c3 c2; do
  TDLO_ESTEP_BLOCKS=$n timeout 600 python bench.py --config $c --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['config']['workload'][:3], d['value'], d.get('em_loop_only_iters_per_s'), [(o['kernel'],o['avg_launch_us']) for o in d['roofline_kernels']])"
  done
done
