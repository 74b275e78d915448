#!/bin/bash
# A/B of the E-step's node window: 2^-36 / 2^-66 of a point's largest membership (default) against exact zeros only (TDLO_WINDOW=exact)
set -u
out=gpurun_out/r04_window; mkdir -p $out
python -m pytest tests -m gpu -x -q > $out/gpu_suite.log 2>&1; echo "suite rc=$?"; grep -E "passed|failed" $out/gpu_suite.log | tail -2
for mode in exact tight; do
  if [ $mode = exact ]; then export TDLO_WINDOW=exact; else unset TDLO_WINDOW; fi
  python bench.py --gpus 1 --pmc off --no-cpu-baseline > $out/bench_$mode.out 2>/dev/null
  cp bench_detail.json $out/bench_detail_$mode.json
  python - $mode <<'PY'
import json,sys
d=json.load(open("bench_detail.json"))
print(sys.argv[1], "c2", d["value"], "f64", d.get("em_iters_per_s_f64"), "E", d["roofline_kernels"][1]["avg_launch_us"] if len(d["roofline_kernels"])>1 else None, [ (k, v.get("value"), v["roofline"]["kernel"], v["roofline"]["avg_launch_us"]) for k,v in d["configs"].items()], "track", d["preproc"].get("tracking_step_ms_per_frame"))
PY
done
