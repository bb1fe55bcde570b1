#!/usr/bin/env python
"""Launch one GEMM variant a few times at one shape (for rocprofv3 --pmc / --kernel-trace):
  python tools/gemm_one.py M N K epi variant [iters]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from latte_amd._lib import check, load_library, ptr, stream_ptr  # noqa: E402

M, N, K, epi, variant = [int(v) for v in sys.argv[1:6]]
iters = int(sys.argv[6]) if len(sys.argv) > 6 else 5
lib = load_library()
Mp = (M + 255) // 256 * 256
A = torch.randn(Mp, K, device="cuda").bfloat16()
W = (torch.randn(N, K, device="cuda") / K ** 0.5).bfloat16()
bias = torch.randn(N, device="cuda")
gate = torch.randn(2 * N, device="cuda")
out = torch.zeros(Mp, N, device="cuda", dtype=torch.float32 if epi >= 2 else torch.bfloat16)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for it in range(iters):
    if it == iters - 1:
        e0.record()
    check(lib.latte_debug_gemm(ptr(A), ptr(W), ptr(bias), ptr(out), ptr(gate), M, N, K, 0, M, epi, 0, variant, stream_ptr()))
e1.record()
torch.cuda.synchronize()
print(f"gemm M={M} N={N} K={K} epi={epi} variant={variant}: {e0.elapsed_time(e1)*1e3:.1f} us (last launch)")
