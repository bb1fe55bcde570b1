#!/bin/bash
# Round 3, call B: fused kernel after the explicit-softmax fix (bit-identity tests), phase trace, T2V with fused temporal blocks.
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "fused_qkv or attention" 2>&1 | tail -8 > $O/b_kernel.log
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "fused_qkv" 2>&1 | tail -8 > $O/b_fused_parity.log
timeout 600 python -m pytest tests/test_t2v.py -q -m gpu 2>&1 | tail -8 > $O/b_t2v.log
timeout 300 python tools/fused_probe.py --trace > $O/b_trace.log 2>&1
timeout 300 python tools/t2v_bench.py --dtype f16 --steps 6 > $O/b_t2v_bench.log 2>&1
