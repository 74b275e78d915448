#!/bin/bash
# C4 (N = 2 000 000): E-step launch geometry sweep
for eb in 512 256; do for n in 256 512 768 1024; do
  echo "== EB=$eb BLOCKS=$n"
  TDLO_ESTEP_EB=$eb TDLO_ESTEP_BLOCKS=$n timeout 600 python bench.py --config c4 --no-cpu-baseline --steps 40 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['value'], d.get('unsplit_iters_per_s'), [(o['kernel'],o['avg_launch_us']) for o in d['roofline_kernels']])"
done; done
