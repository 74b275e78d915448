#!/bin/bash
# A/B on ONE box: the tree's library against scripts/tmp/libtrackdlo_<name>.so (scripts/build_variant.sh), alternating; boxes differ by ~2 %.
# usage: bash scripts/gpu_ab.sh <name> [config] [rounds]
name=$1; cfg=${2:-c2}; n=${3:-3}
one() { python bench.py --config $cfg --no-cpu-baseline --pmc off 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], [(k['kernel'],k['avg_launch_us']) for k in d['roofline_kernels']], d['roofline'].get('iteration_us'))"; }
for i in $(seq $n); do one tree; TDLO_LIBRARY=$PWD/scripts/tmp/libtrackdlo_$name.so one $name; done
