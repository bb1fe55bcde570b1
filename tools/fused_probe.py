"""Measurement: the fused QKV + attention kernel (csrc/qkv_attn.hip) against the un-fused pair inside the XL/2 forward.

  python tools/fused_probe.py [--batch 8] [--steps 10]

For every setting of the engine option fuse_qkv_attn (bit 0 spatial, bit 1 temporal, bits 2-3 schedule variant) prints the
per-class kernel times of one eager forward (HIP events per launch) and the rate of a short DDIM loop, interleaved A/B/A."""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402
import latte_amd  # noqa: E402
from latte_amd._lib import load_library  # noqa: E402


def trace(lib, dev, B=8, F=16, T=256, D=1152, H=16):
    """Stand-alone launches of the fused kernel at the XL/2 shape with the phase trace of workgroup 0 (shader-clock ticks)."""
    from latte_amd._lib import check, ptr, stream_ptr
    rows = B * F * T
    g = torch.Generator("cpu").manual_seed(7)
    xn = torch.randn(rows, D, generator=g).to(dev).to(torch.bfloat16)
    W = (torch.randn(3 * D, D, generator=g) * D ** -0.5).to(dev).to(torch.bfloat16)
    bias = torch.zeros(3 * D, device=dev)
    out = torch.empty(rows, D, dtype=torch.bfloat16, device=dev)
    tr = torch.zeros(8, 4, dtype=torch.int64, device=dev)
    for mode in (0, 1):
        for flags in ((0, 1, 2, 3) if mode == 0 else (0, 1)):
            times = []
            for rep in range(6):
                tr.zero_()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                check(lib.latte_debug_qkv_attention_trace(ptr(xn), ptr(W), ptr(bias), ptr(out), None, ptr(tr), B, F, T, D, H, mode, flags,
                                                          0, stream_ptr()))
                e1.record()
                torch.cuda.synchronize()
                times.append(e0.elapsed_time(e1) * 1e3)
            t = tr.cpu().tolist()
            units = t[0][3]
            per = [[round(v / max(units, 1)) for v in w[:3]] for w in t]
            print(f"trace mode={mode} flags={flags}: launch us {sorted(round(v, 1) for v in times)}; units of workgroup 0: {units}; "
                  f"ticks per unit [projection loop, image write, attention] wave 0: {per[0]}, wave 4: {per[4]}, wave 7: {per[7]}", flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--trace", action="store_true")
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--options", default="0,3,7,11,15,1,2,0,3,15")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    lib = load_library()
    if a.trace:
        return trace(lib, dev)
    B = a.batch
    model = bench.build_model(dev, "bf16", B)
    d = latte_amd.create_diffusion("250")
    x = torch.randn(B, 16, 4, 32, 32, generator=torch.Generator("cpu").manual_seed(1)).to(dev)
    t = torch.full((B,), 500, device=dev, dtype=torch.int64)
    keys = ["gemm_qkv", "attn_spatial", "attn_temporal", "qkv_attn_spatial", "qkv_attn_temporal", "gemm_proj", "gemm_fc1", "gemm_fc2",
            "ln_modulate"]
    for opt in [int(v) for v in a.options.split(",")]:
        model.set_engine_option("fuse_qkv_attn", opt, B)
        model.profile_forward(x, t)
        prof = model.profile_forward(x, t)
        dt = bench.timed_steps(lib, model, d, x.clone(), a.steps, "ddim", B)
        per = {k: (round(prof[k][0] / prof[k][1] * 1e3, 1) if prof[k][1] else None) for k in keys}
        tot = sum(v[0] for v in prof.values())
        print(f"fuse_qkv_attn={opt:2d}: {B / dt:7.2f} sample-steps/s ({dt * 1e3:.2f} ms/step), forward by events {tot:.2f} ms; us per launch: {per}",
              flush=True)


if __name__ == "__main__":
    main()
