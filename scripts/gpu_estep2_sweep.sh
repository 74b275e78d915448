#!/bin/bash
# E-step time of k_estep2 over tile rows x workgroup counts at C4 (N = 2 000 000) and one C3 batch (32 x 50 000); k_estep as the reference line.
# usage (on the GPU box): bash scripts/gpu_estep2_sweep.sh "<rows list>" "<blocks list>"
rows=${1:-"8 16"}; blocks=${2:-"512 768 1024 1280 2048"}
echo "== k_estep"; TDLO_ESTEP2=0 python scripts/gpu_estep2_check.py time1 2>&1 | grep "^\["
for r in $rows; do for b in $blocks; do
  echo "== rows $r blocks $b"; TDLO_ESTEP2=1 TDLO_ESTEP2_ROWS=$r TDLO_ESTEP2_BLOCKS=$b python scripts/gpu_estep2_check.py time1 2>&1 | grep "^\["
done; done
