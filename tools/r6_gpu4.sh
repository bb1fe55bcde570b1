set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "gemm" 2>&1 | tail -5
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "batch or small" 2>&1 | tail -5
LATTE_FL_B=1,2 LATTE_FL_VARIANTS=0,9,11,13,18,19 timeout 600 python tools/gpu_first_light.py gemm_in_model 2>&1 | grep -v amdgpu.ids | tail -12 | tee gpurun_out/r6_gemm_tile144_inmodel.log
timeout 600 python bench.py --steps 20 --no-side 2>/dev/null | tail -1 > gpurun_out/b.json
python - <<'PY'
import json
d=json.loads(open('gpurun_out/b.json').read().strip().splitlines()[-1]); print('value',d['value'])
PY
python - <<'PY'
import torch, time, latte_amd
from bench import build_model, timed_steps
from latte_amd._lib import load_library
lib=load_library()
dev=torch.device('cuda')
diff=latte_amd.create_diffusion("250")
for B in (1,2,4):
    m=build_model(dev,"f16",B)
    x=torch.randn(B,16,4,32,32,device=dev)
    for rep in range(2):
        dt=timed_steps(lib,m,diff,x.clone(),20,"ddim",B)
    print("B",B,"ms/step",round(dt*1e3,3),"sample-steps/s",round(B/dt,1),flush=True)
    del m
PY
