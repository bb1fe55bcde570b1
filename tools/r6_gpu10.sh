set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "gelu_training" 2>&1 | tail -5
timeout 900 python -m pytest tests/test_training_step.py -q -m gpu -x 2>&1 | tail -5
for o in fuse_gelu=0 fuse_gelu=1 fuse_gelu=0 fuse_gelu=1; do LATTE_TRAIN_OPTIONS=$o python tools/train_bench.py 2>&1 | tail -1 | cut -c1-200; done | tee gpurun_out/r6_train_fuse_gelu_ab.log
