#!/bin/bash
# the whole GPU suite on the final tree, then the round-end measurements
R=$(pwd); O=$R/gpurun_out/r05final; mkdir -p $O
( time timeout 1800 python -m pytest tests -q -m gpu -p no:cacheprovider ) > $O/gpu_suite.log 2>&1; echo "rc=$?" >> $O/gpu_suite.log
grep -E "passed|failed|rc=|real" $O/gpu_suite.log | tail -4
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; grep "smoke ok" $O/smoke.log | cut -c1-160
bash scripts/gpu_r05_round_end.sh r05end7 2>&1 | tail -4
