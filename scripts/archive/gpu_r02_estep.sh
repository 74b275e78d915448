#!/bin/bash
# E-step work of round 2: parity suite, bench lines, SQ counters at N = 2 000 000.  usage: bash scripts/gpu_r02_estep.sh <tag>
tag=${1:-r02g}
export TMPDIR=/tmp
R=$PWD
mkdir -p $R/gpurun_out/$tag
timeout 2400 python -m pytest tests -m gpu -q -x > $R/gpurun_out/$tag/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $R/gpurun_out/$tag/pytest.log
for c in c2 c3 c4; do
  timeout 600 python bench.py --config $c --no-cpu-baseline > $R/gpurun_out/$tag/bench_$c.json 2> $R/gpurun_out/$tag/bench_$c.err
  python3 - $R/gpurun_out/$tag/bench_$c.json <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith("{"):
        d=json.loads(l); print(d["config"]["workload"][:40], d["value"], [(o["kernel"],o["avg_launch_us"]) for o in d["roofline_kernels"]], d.get("unsplit_iters_per_s"))
PY
done
bash scripts/gpu_estep_pmc.sh $tag 2000000 50 0 2>&1 | grep -v amdgpu.ids
