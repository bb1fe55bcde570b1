#!/bin/bash
# usage: tools/prof_kernel.sh <outdir under gpurun_out> <cmd...>   -- kernel trace + separate PMC passes (never combined)
# ALWAYS pass --output-format csv: without it rocprofv3 7.x writes a rocpd database and derives the statistics from it after
# the run, which took > 5 minutes of box time for a 3-forward LatteT2V trace (round 2b: one such call ate the GPU budget).
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- "$@" > $OUT/trace.log 2>&1
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAVES GRBM_GUI_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $OUT/pmc$i -o p -- "$@" > $OUT/pmc$i.log 2>&1
done
ls -R $OUT | head -40
