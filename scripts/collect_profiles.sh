#!/bin/bash
# Copies the summaries of a gpu_r0N_round_end.sh run into profiles/ under the round's prefix.   usage: bash scripts/collect_profiles.sh <tag> <prefix>
tag=$1; pre=${2:-r04}
O=gpurun_out/$tag
for c in default c2 c3 c4 c5; do
  [ -s $O/bench_line_$c.json ] && tail -1 $O/bench_line_$c.json > profiles/${pre}_bench_line_$c.json
  [ -s $O/bench_detail_$c.json ] && cp $O/bench_detail_$c.json profiles/${pre}_bench_detail_$c.json
done
for f in kernel_stats_c2.csv kernel_stats_c3.csv kernel_stats_c4.csv kernel_stats_c5.csv kernel_stats_lle_M30_to_512.csv kernel_stats_tracking_step.csv kernel_stats_depth_to_cloud.csv \
         pmc_hbm.json estep_sq_counters_c4.txt estep_sq_counters_c4_k_estep.txt measured.log c3_timeline.txt tracking_step_timeline.txt tracking_step_timeline_copy_route.txt \
         tracking_step_timeline_hidden_nodes.txt tracking_step_timeline_hidden_nodes_ahead_off.txt; do
  [ -s $O/$f ] && cp $O/$f profiles/${pre}_$f
done
ls -la profiles/${pre}_*
