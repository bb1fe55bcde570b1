"""Side measurement: one LatteT2V (Latte-1 geometry) denoiser forward on the engine, random weights.

  python tools/t2v_bench.py [--batch 2] [--layers 28] [--steps 10]

Latte-1: 16 frames of a 64x64 latent (512 px), patch 2 -> 1024 tokens per frame, D = 1152, 28 spatial + 28 temporal
blocks, 120 T5 tokens; batch 2 = the classifier-free-guidance pair of sample/pipeline_latte.py:735-746."""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import latte_amd  # noqa: E402
from latte_amd.random_init import t2v_state_dict  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--layers", type=int, default=28)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--dtype", default="f16", choices=["f16"])
    ap.add_argument("--attn-variant", type=int, default=0, help="latte_debug_set_choice(\"attn_variant\", v) for the whole run (12 / 13: the round-6c L > 256 kernels)")
    a = ap.parse_args()
    if a.attn_variant:
        from latte_amd import _lib
        assert _lib.load_library().latte_debug_set_choice(b"attn_variant", a.attn_variant) == 0
    sd = t2v_state_dict(0, num_layers=a.layers)
    m = latte_amd.LatteT2V(num_layers=a.layers, compute_dtype=a.dtype, max_batch=a.batch).load_state_dict(sd).to("cuda")
    B = a.batch
    x = torch.randn(B, 4, 16, 64, 64, device="cuda")
    t = torch.full((B,), 500, device="cuda", dtype=torch.int64)
    enc = torch.randn(B, 120, 4096, device="cuda")
    mask = torch.ones(B, 120, device="cuda")
    mask[:, 40:] = 0
    for _ in range(2):
        out = m(x, t, enc, encoder_attention_mask=mask).sample
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(a.steps):
        out = m(x, t, enc, encoder_attention_mask=mask).sample
    torch.cuda.synchronize()
    dt = (time.time() - t0) / a.steps
    D, T, F, L = 1152, 1024, 16, a.layers
    M = B * F * T
    lin = 2.0 * M * D * (3 * D + D + 4 * D + 4 * D) * 2 * L + 2.0 * M * D * (D + D) * L      # self blocks (x2) + cross q / out
    attn = L * (4.0 * B * F * T * T * D + 4.0 * B * T * F * F * D + 4.0 * B * F * T * 120 * D)
    print(f"LatteT2V {a.dtype} B={B} layers={L}+{L}: {dt*1e3:.2f} ms per forward, finite={bool(torch.isfinite(out).all())}; "
          f"algorithmic {(lin + attn)/1e12:.2f} TFLOP -> {(lin + attn)/dt/1e12:.0f} TF/s ({(lin + attn)/dt/2.5e15*100:.1f} % of the dense bf16 peak)")
    if B % 2 == 0:
        # the guided DDIM loop inside the engine (text context once, one fused update kernel per step)
        from latte_amd.schedulers import DDIMScheduler
        sch = DDIMScheduler()
        sch.set_timesteps(50)
        pipe = latte_amd.LattePipeline(transformer=m, scheduler=sch)
        n = max(a.steps, 4)
        ts = [int(v) for v in sch.timesteps[:n]]
        ratio = sch.num_train_timesteps // sch.num_inference_steps
        at = [float(sch.alphas_cumprod[v]) for v in ts]
        ap = [float(sch.alphas_cumprod[v - ratio]) if v >= ratio else 1.0 for v in ts]
        lat = torch.randn(B // 2, 4, 16, 64, 64, device="cuda")
        m.set_text(enc)
        m.guided_ddim_loop(lat, ts[:2], at[:2], ap[:2], 7.5)
        torch.cuda.synchronize()
        t0 = time.time()
        outl = m.guided_ddim_loop(lat, ts, at, ap, 7.5)
        torch.cuda.synchronize()
        dl = (time.time() - t0) / n
        print(f"guided DDIM loop in the engine ({B // 2} video(s), guidance pair): {dl*1e3:.2f} ms per step "
              f"({dl / dt:.3f} x one forward), finite={bool(torch.isfinite(outl).all())}; 50 steps = {50*dl:.2f} s")
        del pipe


if __name__ == "__main__":
    main()
