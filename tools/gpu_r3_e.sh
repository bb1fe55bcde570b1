#!/bin/bash
# Round 3, call E: training step with f16 operands + loss scaling (gradient parity vs the reference's fp32 gradients), T2V f16-only
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
timeout 900 python -m pytest tests/test_training_step.py -q -m gpu -s 2>&1 | grep -v "^$" | tail -30 > $O/e_train.log
timeout 600 python -m pytest tests/test_t2v.py tests/test_checkpoints.py -q -m gpu 2>&1 | tail -8 > $O/e_t2v.log
timeout 300 python tools/train_bench.py > $O/e_train_bench.log 2>&1
LATTE_TRAIN_DTYPE=f16 timeout 300 python tools/train_bench.py >> $O/e_train_bench.log 2>&1
