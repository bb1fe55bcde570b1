set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_lo4.py -q -m gpu -x -s 2>&1 | grep -vE "amdgpu.ids" | tail -40 | tee gpurun_out/r6_lo4_tests.log
timeout 600 python -m pytest tests/test_lo8.py -q -m gpu -x 2>&1 | tail -3
