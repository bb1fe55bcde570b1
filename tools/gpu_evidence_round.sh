#!/bin/bash
# The closing GPU evidence of a round, one command on the GPU box (from the repo root; results under gpurun_out/):
#   full GPU test suite, smoke, the driver's default bench line, then rocprofv3 kernel stats of the bench command and the PMC
#   passes (tools/pmc_round.sh <tag>); fold the PMC output into profiles/ afterwards with  python tools/pmc_collect.py <tag>.
# Usage:  bash tools/gpu_evidence_round.sh [tag]        (e.g. through gpurun: /usr/local/graft/bin/gpurun -- 'bash tools/gpu_evidence_round.sh r5')
set -u
TAG=${1:-r5}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -5 > $O/${TAG}_pytest_gpu.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/${TAG}_smoke.log 2>&1
timeout 400 python bench.py > $O/${TAG}_bench_default.json 2> $O/${TAG}_bench_default.err
timeout 900 bash tools/pmc_round.sh $TAG > $O/${TAG}_pmc_round.log 2>&1
# the N = 2 code path of the bench INCLUDING the side measurements, both ranks on this GPU over gloo (numbers are not reported)
LATTE_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 4 --warmup 1 --no-vae --no-cpu-baseline > $O/${TAG}_bench_two_ranks_gloo.json 2> $O/${TAG}_bench_two_ranks_gloo.err
# what bounds the MFMA kernels: the same launches on random / quiet operands with rocm-smi power and clock beside them
timeout 200 python tools/operand_power_probe.py 1.2 > $O/${TAG}_operand_power_probe_gemm.log 2>&1
timeout 200 python tools/operand_power_probe.py 1.5 fused > $O/${TAG}_operand_power_probe_fused_and_forward.log 2>&1
# round 6: rocprofv3 kernel statistics of the side configurations (configs 5 and 4) on the closing code
export TMPDIR=/tmp
for pair in "train_step_B2_batch5:tools/train_bench.py" "temporal_decoder:tools/t2v_decode_bench.py" "t2v_forward:tools/t2v_bench.py --steps 4"; do
  n=${pair%%:*}; c=${pair#*:}
  (cd /tmp && LATTE_DECODE_PROFILE=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$n -o $n -- python $R/$c > $O/${TAG}_prof_$n.log 2>&1)
  cp $(find /tmp/prof_$n -name "*kernel_stats.csv" | head -1) $O/${TAG}_kernel_stats_$n.csv
done
