#!/bin/bash
# C3 / C4 of round 4's tree (55dfc03), round 5's final tree (95f8f87) and this tree on ONE box, alternating (the driver's r05 figures for the
# GPU-filling configurations were 15 - 19 % below DESIGN's, and r05 had touched those kernels in 7570a6b -- a slow box or a regression?).  The old trees live as git
# worktrees under scripts/tmp/ (git worktree add scripts/tmp/wt_r04 55dfc03; wt_r05 95f8f87; each built with make -C trackdlo_amd/csrc and make -C oracle).
#   usage (on the GPU box): bash scripts/gpu_trees_ab.sh [rounds]
R=$PWD; n=${1:-2}
one() {   # tree dir, label, config
  ( cd $1 && python bench.py --config $3 --no-cpu-baseline --pmc off 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$2 $3', d['value'], [(k['kernel'],k['avg_launch_us']) for k in d.get('roofline_kernels', [])], 'sclk', d.get('sclk_mhz_mean'), 'W', d.get('power_w_mean'))" )
}
for i in $(seq $n); do for cfg in c3 c4; do
  [ -d $R/scripts/tmp/wt_r04 ] && one $R/scripts/tmp/wt_r04 r04 $cfg
  [ -d $R/scripts/tmp/wt_r05 ] && one $R/scripts/tmp/wt_r05 r05 $cfg
  one $R HEAD $cfg
  TDLO_ESTEP2=0 one $R HEAD-k_estep $cfg
done; done
