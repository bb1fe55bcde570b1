set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 2400 python -m pytest tests/test_chain250.py -q -m gpu -s 2>&1 | grep -E "^(xl_|b2_guided_g03b)|passed|failed" | tail -20
cp gpurun_out/chain250_drift.json gpurun_out/r6_chain250_drift.json
