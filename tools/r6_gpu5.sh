set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_lo8.py tests/test_gpu_kernels.py -q -m gpu -x -k "lo8 or remainder or fused or fp8" 2>&1 | tail -5
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "fused or guided_split_contract" 2>&1 | tail -5
timeout 600 python tools/guided_split_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6_guided_split_cost_b.log
