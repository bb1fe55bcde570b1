#!/bin/bash
# Round-1 closing evidence: full GPU tests, smoke, default bench, rocprofv3 kernel stats of the bench command.
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
timeout 600 python -m pytest tests -q -m gpu 2>&1 | tail -3 > $O/pytest_gpu.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
timeout 300 python bench.py > $O/bench_default.json 2> $O/bench_default.err
export TMPDIR=/tmp; cd /tmp
rm -rf $O/prof_bench
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -o bench -- python $R/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-vae > $O/prof_bench.log 2>&1
