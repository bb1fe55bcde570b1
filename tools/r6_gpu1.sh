set -x
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
hipcc -O3 --offload-arch=gfx950 tools/mx_probe.hip -o /tmp/mx_probe && timeout 300 /tmp/mx_probe 3 > gpurun_out/r6_mx_probe.log 2>&1
tail -40 gpurun_out/r6_mx_probe.log
timeout 900 python -m pytest tests/test_lo8.py -x -q -m gpu 2>&1 | tail -30 > gpurun_out/r6_lo8_tests.log; cat gpurun_out/r6_lo8_tests.log
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "guided" 2>&1 | tail -30 > gpurun_out/r6_guided_tests.log; cat gpurun_out/r6_guided_tests.log
timeout 600 python tools/guided_split_probe.py > gpurun_out/r6_guided_split_cost.log 2>&1; cat gpurun_out/r6_guided_split_cost.log
