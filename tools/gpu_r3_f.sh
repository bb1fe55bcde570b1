#!/bin/bash
# Round 3, call F: full GPU test suite, smoke, default bench, rocprofv3 kernel stats of the bench command + PMC passes.
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
timeout 900 python -m pytest tests -q -m gpu -x 2>&1 | tail -6 > $O/f_pytest_gpu.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/f_smoke.log 2>&1
timeout 400 python bench.py > $O/f_bench_default.json 2> $O/f_bench_default.err
timeout 900 bash tools/pmc_round.sh r3 > $O/f_pmc_round.log 2>&1
