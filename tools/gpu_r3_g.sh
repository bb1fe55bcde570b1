#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "fused_qkv" 2>&1 | tail -8 > $O/g_kernel.log
timeout 600 python tools/fused_probe.py --options 3,35,3,35,3,35 > $O/g_probe.log 2>&1
