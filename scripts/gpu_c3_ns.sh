#!/bin/bash
# C3 (32 frames per GPU) over the number of stream groups and the E-step kernel: value and the E-step's launch time from bench.py's own line
for e2 in 0 1; do for ns in 2 3 4; do
  TDLO_ESTEP2=$e2 TDLO_BATCH_STREAMS=$ns python bench.py --config c3 --no-cpu-baseline --pmc off 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('estep2=$e2 streams=$ns', d['value'], [(k['kernel'],k['avg_launch_us']) for k in d['roofline_kernels']])"
done; done
