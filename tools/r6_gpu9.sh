set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_vae_temporal.py tests/test_vae.py -q -m gpu -x -s -k "sweep" 2>&1 | grep -E "^0x|^default|passed|failed|Error" | tee gpurun_out/r6_vae_split_sweep.log
timeout 1200 python -m pytest tests/test_vae_temporal.py tests/test_vae.py tests/test_vae_operand_budget.py tests/test_kernel_choice.py -q -m gpu -x 2>&1 | tail -3
python tools/t2v_decode_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6_temporal_decoder_after.log
python -c "
import bench, torch, json
print(json.dumps(bench.vae_decode_rate(torch.device('cuda'))))" 2>&1 | tail -1 | tee gpurun_out/r6_vae_decode_rate_after.json
