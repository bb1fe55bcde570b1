#!/bin/bash
# Round 3, call A: the new tests first (fused qkv + attention kernel, 250-step chains, bench launcher), then A/B bench lines.
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "fused_qkv" -x 2>&1 | tail -15 > $O/a_fused_kernel.log
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "fused_qkv or (oracle and (XL or S/2-32))" 2>&1 | tail -15 > $O/a_fused_parity.log
timeout 900 python -m pytest tests/test_chain250.py -q -m gpu -s 2>&1 | tail -40 > $O/a_chain250.log
timeout 400 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-vae --no-side > $O/a_bench_fused.json 2> $O/a_bench_fused.err
timeout 400 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-vae --no-side --engine-option fuse_qkv_attn=0 > $O/a_bench_unfused.json 2> $O/a_bench_unfused.err
timeout 400 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-vae --no-side --engine-option fuse_qkv_attn=2 > $O/a_bench_fused_t.json 2> $O/a_bench_fused_t.err
timeout 400 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-vae --no-side --engine-option fuse_qkv_attn=1 > $O/a_bench_fused_s.json 2> $O/a_bench_fused_s.err
