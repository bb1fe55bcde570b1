set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_lo8.py -q -m gpu -x 2>&1 | tail -4
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "attention" 2>&1 | tail -3
timeout 1500 python -m pytest tests/test_chain250.py -q -m gpu -x -k "b2_guided or s2_guided or xl_uncond_g03" -s 2>&1 | grep -E "^(b2|s2|xl)|passed|failed" | tail -20
cp gpurun_out/chain250_drift.json gpurun_out/r6_chain250_drift_partial.json
