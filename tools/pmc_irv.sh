#!/bin/bash
# SQ counter passes over three REAL-content 1080p frames (tools/real_frame_loop.py): what bounds stm_k_irv_vote?
# usage (on the GPU box): bash tools/pmc_irv.sh <outdir>
set -e
OUT=${1:-gpurun_out/pmc_irv}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS SQ_LDS_BANK_CONFLICT" \
           "SQ_LDS_IDX_ACTIVE SQ_LDS_ATOMIC_RETURN SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_LDS_ADDR_CONFLICT GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set -d $OUT/p$i -o f --output-format csv -- python3 tools/real_frame_loop.py 2 > $OUT/p$i.log 2>&1 || { echo "pass $i failed"; tail -5 $OUT/p$i.log; }
  echo "pass $i done"
done
python3 tools/pmc_summary.py $(find $OUT -name '*counter_collection.csv') > $OUT/summary.txt
grep -A30 "stm_k_irv_vote" $OUT/summary.txt | head -40
