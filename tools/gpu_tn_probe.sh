set -u
cd $GRAFT_REPO_ROOT; O=gpurun_out
timeout 600 python -m pytest tests/test_training_step.py -q -m gpu 2>&1 | grep -E "passed|failed" > $O/tn_tests.log
LATTE_TRAIN_DTYPE=f16 timeout 300 python tools/train_bench.py > $O/tn_bench.log 2>&1
LATTE_TN_KERNEL=4 LATTE_TRAIN_DTYPE=f16 timeout 300 python tools/train_bench.py >> $O/tn_bench.log 2>&1
LATTE_TRAIN_DTYPE=f16 timeout 300 python tools/train_bench.py >> $O/tn_bench.log 2>&1
LATTE_TRAIN_MODEL=Latte-XL/2 LATTE_TRAIN_DTYPE=f16 timeout 300 python tools/train_bench.py >> $O/tn_bench.log 2>&1
cd /tmp && export TMPDIR=/tmp && LATTE_TRAIN_DTYPE=f16 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3_stats_train -o s -- python $GRAFT_REPO_ROOT/tools/train_bench.py > $GRAFT_REPO_ROOT/gpurun_out/r3_stats_train.log 2>&1
