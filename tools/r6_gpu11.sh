set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cat > /tmp/b1.py <<'PY'
import sys, torch, time
sys.path.insert(0, "/root/repo")
import latte_amd
from bench import build_model, timed_steps
from latte_amd._lib import load_library
lib = load_library(); dev = torch.device('cuda'); diff = latte_amd.create_diffusion("250")
B = int(sys.argv[1])
m = build_model(dev, "f16", B)
x = torch.randn(B, 16, 4, 32, 32, device=dev)
for rep in range(2):
    dt = timed_steps(lib, m, diff, x.clone(), 40, "ddim", B)
print("B", B, "ms/step", round(dt * 1e3, 4), flush=True)
PY
for B in 1 2; do
python /tmp/b1.py $B 2>&1 | tail -1
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pb$B -o s -- python /tmp/b1.py $B > /tmp/pb$B.log 2>&1)
tail -1 /tmp/pb$B.log
cp /tmp/pb$B/s_kernel_stats.csv gpurun_out/r6_kernel_stats_B${B}_80steps.csv
done
for o in fuse_gelu=1 fuse_gelu=1; do LATTE_TRAIN_OPTIONS=$o python tools/train_bench.py 2>&1 | tail -1 | cut -c1-200; done | tee gpurun_out/r6_train_after_optimizer.log
timeout 600 python -m pytest tests/test_training_step.py -q -m gpu -x 2>&1 | grep -E "passed|failed" 
