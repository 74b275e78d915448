#!/bin/bash
# round 5, after the one-launch depth -> cloud kernel: its tests, the whole GPU suite, the default bench (with the frame_from_depth leg), kernel stats of the cloud step
R=$(pwd); O=$R/gpurun_out/r05c; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu -p no:cacheprovider > $O/gpu_suite.log 2>&1; echo "rc=$?" >> $O/gpu_suite.log
grep -E "passed|failed|rc=" $O/gpu_suite.log | tail -3
timeout 900 python bench.py > $O/bench_default.log 2> $O/bench_default.err; echo "rc=$?" >> $O/bench_default.log
cp bench_detail.json $O/bench_detail_default.json
tail -c 900 $O/bench_default.log
python - <<'PY'
import json
d = json.load(open("bench_detail.json"))
print(json.dumps(d.get("frame_from_depth"), indent=1))
PY
cd /tmp && export TMPDIR=/tmp && PYTHONPATH=$R timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_cloud -- python $R/scripts/gpu_cloud_time.py > $O/prof_cloud.log 2>&1 </dev/null
cd $R; find $O/prof_cloud -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'head -20 {} | cut -c1-220'
