#!/bin/bash
# E-step workgroup size experiment: C2 / C3 / C4 lines with 512 (default) and 256 threads per workgroup
for eb in 512 256; do
  echo "== TDLO_ESTEP_EB=$eb"
  for c in c2 c3 c4; do
    TDLO_ESTEP_EB=$eb timeout 600 python bench.py --config $c --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['config']['workload'][:30], d['value'], [(o['kernel'],o['avg_launch_us']) for o in d['roofline_kernels']], d.get('em_loop_only_iters_per_s'))"
  done
done
