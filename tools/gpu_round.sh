#!/bin/bash
# One GPU session: tests, driver smoke, bench, kernel-trace profile of the bench, PMC of the dominant GEMM.
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
python -m pytest tests -q -m gpu -x 2>&1 | tail -4 > $O/pytest_gpu.log
python tools/sample.py --config configs/tiny_sample.yaml --out $O/tiny_videos > $O/sample_tiny.log 2>&1
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --batch 2 --steps 100 --no-cpu-baseline > $O/bench_b2.json 2> $O/bench_b2.err
export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -o bench -- python $R/bench.py --steps 20 --warmup 2 --no-cpu-baseline > $O/prof_bench.log 2>&1
for shape in "32768 1152 4608 2 8 fc2" "32768 4608 1152 1 9 fc1"; do
  set -- $shape
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d $O/pmc_gemm/$6_$c -o p -- python $R/tools/gemm_pmc.py $1 $2 $3 $4 $5 5 > $O/pmc_gemm_$6_$c.log 2>&1
  done
done
ls -R $O/prof_bench $O/pmc_gemm | head -40
