#!/bin/bash
# SQ counter passes over three 1080p frames of the shipped build (tools/frame_loop.py), aggregation kernels.
# usage (on the GPU box): bash tools/pmc_sq_agg.sh <outdir> [variant]
set -e
OUT=${1:-gpurun_out/pmc_sq}
VAR=${2:-0}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set -d $OUT/p$i -o f --output-format csv -- python3 tools/frame_loop.py 3 $VAR > $OUT/p$i.log 2>&1 || { echo "pass $i failed"; tail -5 $OUT/p$i.log; }
  echo "pass $i done"
done
python3 tools/pmc_summary.py $(find $OUT -name '*counter_collection.csv') > $OUT/summary.txt
echo ok
