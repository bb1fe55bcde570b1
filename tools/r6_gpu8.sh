set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
prof() { # name, cmd...
  n=$1; shift
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$n -o $n -- "$@" > /tmp/prof_$n.log 2>&1)
  cp $(find /tmp/prof_$n -name "*kernel_stats.csv" | head -1) $R/gpurun_out/r6_kernel_stats_$n.csv
}
prof train_step_B2_batch5_before python $R/tools/train_bench.py
LATTE_DECODE_PROFILE=0 prof temporal_decoder_before python $R/tools/t2v_decode_bench.py
prof t2v_forward_before python $R/tools/t2v_bench.py --steps 4
tail -3 /tmp/prof_t2v_forward_before.log
