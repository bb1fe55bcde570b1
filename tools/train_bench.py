"""Training-step throughput on the engine (BASELINE config 5: Latte-B/2 16x256x256 synthetic latents; per-GPU micro-batch as
configs/ffs/ffs_train.yaml's local_batch_size = 5).  Prints ms per step and algorithmic TFLOP/s (3 x forward FLOPs)."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import latte_amd

name = os.environ.get("LATTE_TRAIN_MODEL", "Latte-B/2")
B = int(os.environ.get("LATTE_TRAIN_BATCH", "5"))
steps = int(os.environ.get("LATTE_TRAIN_STEPS", "10"))
dtype = os.environ.get("LATTE_TRAIN_DTYPE", "f16")
model = latte_amd.Latte_models[name](input_size=32, num_frames=16, extras=1, max_batch=B).to("cuda")
with torch.no_grad():
    for p in model.parameters():
        if p.requires_grad and float(p.abs().max()) == 0.0:
            p.normal_(0, 0.02)
tr = latte_amd.LatteTrainer(model, latte_amd.create_diffusion(""), max_batch=B, compute_dtype=dtype)
for kv in os.environ.get("LATTE_TRAIN_OPTIONS", "").split(","):   # e.g. LATTE_TRAIN_OPTIONS=fuse_gelu=0
    if kv:
        k, v = kv.split("=")
        tr.set_option(k, float(v))
g = torch.Generator("cpu").manual_seed(0)
x = torch.randn(B, 16, 4, 32, 32, generator=g).cuda()
for _ in range(2):
    out = tr.train_step(x)
torch.cuda.synchronize()
t0 = time.time()
for _ in range(steps):
    out = tr.train_step(x)
torch.cuda.synchronize()
dt = (time.time() - t0) / steps
D, depth, Hm = model.hidden_size, model.depth, model.mlp_hidden
M = B * 16 * 256
lin = depth * 2.0 * M * (4 * D * D + 2 * D * Hm)
attn = (depth // 2) * (4.0 * B * 16 * 256 * 256 * D + 4.0 * B * 256 * 16 * 16 * D)
fwd = lin + attn
print(json.dumps({"model": name, "operands": dtype, "local_batch": B, "ms_per_step": round(dt * 1e3, 3), "samples_per_s": round(B / dt, 2),
                  "algorithmic_tflops": round(3 * fwd / dt / 1e12, 1), "forward_gflop": round(fwd / 1e9, 1),
                  "loss": float(out["loss"].mean()), "grad_norm": float(out["grad_norm"])}))
