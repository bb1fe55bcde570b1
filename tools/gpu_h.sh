set -u
cd $GRAFT_REPO_ROOT; O=gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "fused_qkv" 2>&1 | grep -E "passed|failed" > $O/h_kernel.log
timeout 600 python -m pytest tests/test_training_step.py tests/test_gpu_parity.py -q -m gpu -k "fused_qkv or training or engine_train or staged or xl_width" 2>&1 | grep -E "passed|failed" >> $O/h_kernel.log
timeout 300 python tools/fused_probe.py --trace > $O/h_trace.log 2>&1
timeout 600 python tools/fused_probe.py --options 3,7,3,7,3,7 > $O/h_probe.log 2>&1
