#!/bin/bash
# round 5, first GPU call: the bounded grid barrier's tests, the whole GPU suite, the default bench line (with its fatal parity gate)
set -x
O=gpurun_out/r05a; mkdir -p $O
timeout 900 python -m pytest tests/test_direct_path_gpu.py -x -q -m gpu -k "abandoned" > $O/fuse_tests.log 2>&1; echo "rc=$?" >> $O/fuse_tests.log
timeout 1500 python -m pytest tests -x -q -m gpu > $O/gpu_suite.log 2>&1; echo "rc=$?" >> $O/gpu_suite.log
timeout 900 python bench.py > $O/bench_default.log 2> $O/bench_default.err; echo "rc=$?" >> $O/bench_default.log
cp bench_detail.json $O/bench_detail_default.json
tail -3 $O/fuse_tests.log $O/gpu_suite.log; tail -c 1500 $O/bench_default.log
