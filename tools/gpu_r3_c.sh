#!/bin/bash
# Round 3, call D: fused kernel v2 (LDS-staged bias, stage 0 behind the images + early fill, attention-phase priority)
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "fused_qkv" 2>&1 | tail -8 > $O/d_kernel.log
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "fused_qkv" 2>&1 | tail -8 > $O/d_fused_parity.log
timeout 300 python tools/fused_probe.py --trace > $O/d_trace.log 2>&1
timeout 600 python tools/fused_probe.py --options 3,15,7,3,15,7,0 > $O/d_probe.log 2>&1
