#!/bin/bash
# E-step work: the whole GPU suite, then the bench lines of C2 / C3 / C4 (value + the E-step's duration)
# usage: bash scripts/gpu_estep_quick.sh <tag> [pytest -k expression]
tag=${1:-eq}
export TMPDIR=/tmp
O=gpurun_out/$tag
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x ${2:+-k "$2"} > $O/pytest.log 2>&1; tail -5 $O/pytest.log
for c in c2 c3 c4; do
  timeout 600 python bench.py --config $c --no-cpu-baseline > $O/bench_$c.json 2> $O/bench_$c.err
  python3 - $O/bench_$c.json <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith("{"):
        d=json.loads(l); rk=d.get("roofline_kernels",{})
        print(d["config"]["workload"][:40], "value", d["value"], "ms/step", d["ms_per_step"], {k:(v.get("duration_us"), v.get("frac")) for k,v in rk.items()} if isinstance(rk,dict) else rk)
PY
done
