"""Run the spatial / temporal attention kernel on the XL/2 shape in a loop (for rocprofv3 --pmc / --kernel-trace)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from latte_amd._lib import check, load_library, ptr, stream_ptr

lib = load_library()
B, iters = int(sys.argv[1]), int(sys.argv[2])
mode = sys.argv[3] if len(sys.argv) > 3 else "spatial"
F, T, H, hd = 16, 256, 16, 72
D, rows = H * hd, B * F * T
qkv = torch.randn(rows, 3 * D, device="cuda").bfloat16()
out = torch.zeros(rows, D, dtype=torch.bfloat16, device="cuda")
args = (B * F, T, H, hd, F, F * T, T, 1) if mode == "spatial" else (B * T, F, H, hd, T, F * T, 1, T)
for _ in range(3):
    check(lib.latte_debug_attention(ptr(qkv), ptr(out), *args, 0, stream_ptr()))
torch.cuda.synchronize()
t0 = time.time()
for _ in range(iters):
    check(lib.latte_debug_attention(ptr(qkv), ptr(out), *args, 0, stream_ptr()))
torch.cuda.synchronize()
dt = (time.time() - t0) / iters
flops = 4.0 * B * F * T * T * D if mode == "spatial" else 4.0 * B * T * F * F * D
print(f"attn {mode} B={B}: {dt*1e6:.1f} us/launch, {flops/dt/1e12:.0f} TF/s algorithmic, {(rows*4*D*2)/dt/1e12:.2f} TB/s algorithmic bytes")
