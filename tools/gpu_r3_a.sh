#!/bin/bash
# Round 3, call A: the new tests first (fused qkv + attention kernel, 250-step chains, bench launcher), then A/B measurements.
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "fused_qkv" 2>&1 | tail -25 > $O/a_fused_kernel.log
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "fused_qkv or (oracle and (XL or S/2-32))" 2>&1 | tail -25 > $O/a_fused_parity.log
timeout 900 python -m pytest tests/test_chain250.py -q -m gpu -s 2>&1 | tail -40 > $O/a_chain250.log
timeout 600 python tools/fused_probe.py > $O/a_fused_probe.log 2>&1
timeout 300 python tools/fused_probe.py --batch 1 --steps 20 --options 0,3,15,0,3 > $O/a_fused_probe_b1.log 2>&1
timeout 300 python tools/fused_probe.py --batch 2 --steps 20 --options 0,3,15,0,3 > $O/a_fused_probe_b2.log 2>&1
timeout 400 python bench.py --steps 40 --warmup 3 --no-cpu-baseline > $O/a_bench.json 2> $O/a_bench.err
