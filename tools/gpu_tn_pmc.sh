set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
export LATTE_TRAIN_DTYPE=f16 LATTE_TRAIN_STEPS=2
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/tn_pmc1 -o p -- python $R/tools/train_bench.py > $O/tn_pmc1.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/tn_pmc2 -o p -- python $R/tools/train_bench.py > $O/tn_pmc2.log 2>&1
