#!/bin/bash
# k_setup / prune kernel times at the C5 and C2 sizes for the tree's library and build variants (scripts/build_variant.sh): kernel-trace stats.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for v in tree "$@"; do
  O=$R/gpurun_out/setup_ab_$v; rm -rf $O; mkdir -p $O
  if [ $v = tree ]; then unset TDLO_LIBRARY; else export TDLO_LIBRARY=$R/scripts/tmp/libtrackdlo_$v.so; fi
  ITERS=2 rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python $R/scripts/gpu_c5.py > $O/run.log 2>&1
  f=$(find $O -name "*kernel_stats.csv" | head -1); echo "== $v"; grep -E "k_setup" $f | cut -d, -f1-8
done
