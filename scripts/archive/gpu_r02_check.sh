#!/bin/bash
# Runs on the GPU box (through gpurun): the whole -m gpu suite, then one bench line per config.  usage: bash scripts/gpu_r02_check.sh <tag> [pytest args]
tag=${1:-r02a}; shift
export TMPDIR=/tmp
R=$PWD
mkdir -p $R/gpurun_out/$tag
timeout 2400 python -m pytest tests -m gpu -q -x "$@" > $R/gpurun_out/$tag/pytest.log 2>&1
echo "pytest rc=$?" >> $R/gpurun_out/$tag/pytest.log
tail -30 $R/gpurun_out/$tag/pytest.log
for c in c2 c3 c5 c4; do
  timeout 600 python bench.py --config $c > $R/gpurun_out/$tag/bench_$c.json 2> $R/gpurun_out/$tag/bench_$c.err
  echo "bench $c rc=$?"; cut -c1-1500 $R/gpurun_out/$tag/bench_$c.json; tail -3 $R/gpurun_out/$tag/bench_$c.err | cut -c1-300
done
