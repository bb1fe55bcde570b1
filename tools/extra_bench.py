"""Side measurements for DESIGN.md (not the headline): B = 1 latency, and BASELINE config 3's per-GPU share
(UCF101 class-conditional, CFG 7.0: 8 samples = 16 sequences per GPU)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import latte_amd  # noqa: E402
from latte_amd._lib import check, load_library, ptr, stream_ptr  # noqa: E402

lib = load_library()
dev = torch.device("cuda")


def model(extras, max_batch, num_classes=101):
    m = latte_amd.Latte_models["Latte-XL/2"](input_size=32, num_frames=16, extras=extras, num_classes=num_classes,
                                             compute_dtype="bf16", max_batch=max_batch)
    g = torch.Generator("cpu").manual_seed(1)
    with torch.no_grad():
        for _, p in m.named_parameters():
            if p.requires_grad and float(p.detach().abs().max()) == 0.0:
                p.copy_(torch.randn(p.shape, generator=g) * 0.02)
    return m.to(dev).eval()


def run(m, d, x, y, cfg, steps, method=1):
    B = x.shape[0]
    eng = m.engine(B)
    T = d.num_timesteps
    check(lib.latte_sample_loop(eng, d._h, method, 0.0, 0, cfg, ptr(x), ptr(y), B, T - 1, T - 3, None, None, None, stream_ptr()))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    check(lib.latte_sample_loop(eng, d._h, method, 0.0, 0, cfg, ptr(x), ptr(y), B, T - 1, T - steps, None, None, None, stream_ptr()))
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


d = latte_amd.create_diffusion("250")
m = model(1, 1)
for meth, name in ((1, "ddim"), (0, "ddpm")):
    dt = run(m, d, torch.randn(1, 16, 4, 32, 32, device=dev), None, 1.0, 100, meth)
    print(f"B=1 uncond {name}: {dt*1e3:.2f} ms/step -> {1/dt:.1f} steps/s, one 250-step video in {250*dt:.2f} s")
del m
torch.cuda.empty_cache()
m = model(2, 16)
y = torch.cat([torch.randint(0, 101, (8,)), torch.full((8,), 101)]).to(dev)
z = torch.randn(8, 16, 4, 32, 32, device=dev)
x = torch.cat([z, z]).contiguous()
dt = run(m, d, x, y, 7.0, 30)
print(f"config-3 share (class-cond, CFG 7.0, 8 samples = 16 sequences): {dt*1e3:.2f} ms/step -> {8/dt:.1f} guided sample-steps/s "
      f"({16/dt:.1f} sequence-steps/s, {16/dt*3.726e12/2.5e15:.3f} of MFMA peak)")
