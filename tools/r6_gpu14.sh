set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_lo4.py tests/test_lo8.py -q -m gpu -x 2>&1 | tail -3
timeout 600 python tools/guided_split_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6_guided_split_cost_fp4.log
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "guided_split_contract or guided_forward_latte" 2>&1 | tail -3
