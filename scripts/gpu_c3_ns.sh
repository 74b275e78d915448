#!/bin/bash
# C3 (32 frames per GPU) over the number of stream groups and TDLO_BATCH_CHAIN (0: the groups free-running; 1: every iteration's E-steps chained group after
# group; 2, the default: chained at four iterations of the call only -- the phase persists): value, loop-only rate and the kernels' launch times from bench.py's line.
#   usage: bash scripts/gpu_c3_ns.sh ["<streams list>"] ["<chain list>"]
for ns in ${1:-2 3 4}; do for ch in ${2:-0 2}; do
  TDLO_BATCH_CHAIN=$ch TDLO_BATCH_STREAMS=$ns python bench.py --config c3 --no-cpu-baseline --pmc off 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('streams=$ns chain=$ch', d['value'], d.get('em_loop_only_iters_per_s'), [(k['kernel'],k['avg_launch_us']) for k in d['roofline_kernels']])"
done; done
