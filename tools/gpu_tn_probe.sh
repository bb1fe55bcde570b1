set -u
cd $GRAFT_REPO_ROOT; O=gpurun_out
timeout 600 python -m pytest tests/test_training_step.py -q -m gpu -s 2>&1 | grep -v "^$" | tail -12 > $O/tn_tests.log
LATTE_TRAIN_DTYPE=f16 timeout 300 python tools/train_bench.py > $O/tn_bench.log 2>&1
LATTE_TN_KERNEL=4 LATTE_TRAIN_DTYPE=f16 timeout 300 python tools/train_bench.py >> $O/tn_bench.log 2>&1
LATTE_TRAIN_DTYPE=f16 timeout 300 python tools/train_bench.py >> $O/tn_bench.log 2>&1
LATTE_TN_KERNEL=4 LATTE_TRAIN_DTYPE=f16 timeout 300 python tools/train_bench.py >> $O/tn_bench.log 2>&1
LATTE_TRAIN_DTYPE=bf16 timeout 300 python tools/train_bench.py >> $O/tn_bench.log 2>&1
LATTE_TRAIN_MODEL=Latte-XL/2 LATTE_TRAIN_DTYPE=f16 timeout 300 python tools/train_bench.py >> $O/tn_bench.log 2>&1
LATTE_TN_KERNEL=4 LATTE_TRAIN_MODEL=Latte-XL/2 LATTE_TRAIN_DTYPE=f16 timeout 300 python tools/train_bench.py >> $O/tn_bench.log 2>&1
