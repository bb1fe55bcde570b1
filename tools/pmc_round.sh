#!/bin/bash
# rocprofv3 evidence for profiles/: (1) kernel trace + stats of the bench command, (2) PMC passes over the XL/2 forward at the
# bench batch -- counters in their own runs (no sys/hip/hsa trace domains beside --pmc), FETCH_SIZE and WRITE_SIZE in separate
# passes (TCC slots), (3) kernel stats of a forward + one 16-frame VAE decode.  Run from the repo root on the GPU box:
#   bash tools/pmc_round.sh r2      -> gpurun_out/<tag>_*   then   python tools/pmc_collect.py r2
set -u
TAG=${1:-r4}
REPO=$(pwd)
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats_bench -o s -- python $REPO/bench.py --steps 20 --no-cpu-baseline --no-vae --no-side > $OUT/${TAG}_stats_bench.json 2> $OUT/${TAG}_stats_bench.log
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats_fwd_vae -o s -- python $REPO/tools/pmc_workload.py 8 2 vae > $OUT/${TAG}_stats_fwd_vae.log 2>&1
i=0
for CTRS in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_F16 GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_WAVES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $CTRS --output-format csv -d $OUT/${TAG}_pmc$i -o p -- python $REPO/tools/pmc_workload.py 8 1 > $OUT/${TAG}_pmc$i.log 2>&1
done
# (4) round 4: the VAE decoder's kernels (two 16-frame decodes, the second one is evaluated): SQ group, FETCH_SIZE, WRITE_SIZE
for CTRS in "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $CTRS --output-format csv -d $OUT/${TAG}_pmc$i -o p -- python $REPO/tools/pmc_workload.py 8 0 vaeonly > $OUT/${TAG}_pmc$i.log 2>&1
done
ls -R $OUT | grep -c csv
