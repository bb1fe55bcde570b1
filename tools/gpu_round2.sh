#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
python -m pytest tests -q -m gpu -x 2>&1 | tail -3 > $O/pytest_gpu.log
python bench.py --steps 100 --warmup 3 > $O/bench_default.json 2> $O/bench_default.err
# N = 2 code path (barrier, table broadcast, max-over-ranks) on ONE GPU through gloo: functional check only
LATTE_BENCH_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 \
   bench.py --gpus 2 --steps 6 --warmup 1 --batch 2 --no-vae > $O/bench_n2_gloo.json 2> $O/bench_n2_gloo.err
python tools/sample_ddp.py --config configs/tiny_sample.yaml --out $O/tiny_videos --num-samples 4 > $O/sample_tiny.log 2>&1
