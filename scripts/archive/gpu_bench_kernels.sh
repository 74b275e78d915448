#!/bin/bash
# bench lines (value + per-kernel durations), twice per configuration
export TMPDIR=/tmp
for c in ${@:-c2}; do
  for rep in 1 2; do
  timeout 600 python bench.py --config $c --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$c', d['value'], d['ms_per_step'], {k['kernel']: k['avg_launch_us'] for k in d['roofline_kernels']}, d.get('iteration_us'))
"
  done
done
