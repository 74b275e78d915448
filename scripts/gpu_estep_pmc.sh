#!/bin/bash
# usage: bash scripts/gpu_estep_pmc.sh <tag> [N] [M] [blocks]   (on the GPU box, through gpurun)
tag=${1:-r01x}; N=${2:-2000000}; M=${3:-50}; BL=${4:-0}
export TMPDIR=/tmp
R=$PWD
mkdir -p $R/gpurun_out/$tag
cd /tmp
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" \
           "GRBM_GUI_ACTIVE SQ_INSTS_VALU_TRANS SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES" \
           "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_BRANCH" \
           "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_ATOMIC"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/$tag/p$i -- python $R/scripts/gpu_estep_pmc.py $N $M $BL > $R/gpurun_out/$tag/pmc$i.log 2>&1 </dev/null
  f=$(find $R/gpurun_out/$tag/p$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" <<'PY'
import csv, sys
from collections import defaultdict
acc = defaultdict(lambda: [0.0, 0])
for row in csv.DictReader(open(sys.argv[1])):
    if "k_estep" in row["Kernel_Name"]:
        a = acc[row["Counter_Name"]]; a[0] += float(row["Counter_Value"]); a[1] += 1
        meta = (row["Kernel_Name"][:60], row["Grid_Size"], row["Workgroup_Size"], row["LDS_Block_Size"], row["VGPR_Count"], row["Accum_VGPR_Count"], row["SGPR_Count"])
print(meta)
for k, v in acc.items():
    print(f"{k:28s} {v[0]/v[1]:16.1f}  per launch ({v[1]} launches)")
PY
  rm -rf $R/gpurun_out/$tag/p$i
  tail -2 $R/gpurun_out/$tag/pmc$i.log
done
