set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_lo4.py tests/test_lo8.py -q -m gpu -x 2>&1 | tail -30
timeout 1500 python -m pytest tests/test_chain250.py -q -m gpu -x -k "b2_guided-ddim" 2>&1 | tail -30
