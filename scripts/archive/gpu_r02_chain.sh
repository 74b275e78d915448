#!/bin/bash
# Chain-smoother M-step (round 2): parity suite, bench lines, M-step stamps.  usage: bash scripts/gpu_r02_chain.sh <tag> [pytest args]
tag=${1:-r02j}
export TMPDIR=/tmp
R=$PWD
mkdir -p $R/gpurun_out/$tag
timeout 2400 python -m pytest tests -m gpu -q -x ${2:-} > $R/gpurun_out/$tag/pytest.log 2>&1; echo "pytest rc=$?"; tail -15 $R/gpurun_out/$tag/pytest.log
for c in c2 c3 c4 c5; do
  timeout 600 python bench.py --config $c --no-cpu-baseline > $R/gpurun_out/$tag/bench_$c.json 2> $R/gpurun_out/$tag/bench_$c.err
  python3 - $R/gpurun_out/$tag/bench_$c.json <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith("{"):
        d=json.loads(l); print(d["config"]["workload"][:40], d["value"], [(o["kernel"],o["avg_launch_us"]) for o in d["roofline_kernels"]], d.get("unsplit_iters_per_s"))
PY
done
python scripts/gpu_stamps.py 2>&1 | tail -20
