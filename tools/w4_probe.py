"""One consumer wave per SIMD (gemm_w4_kernel, variant 14 of the measurement build) against the ping-pong kernel (variant 9) on the
fc1 shape: same bits?  launch time on random and on all-zero operands.  LATTE_AMD_LIB=latte_amd/lib/liblatte_amd_dbg.so."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from latte_amd._lib import check, load_library, ptr, stream_ptr  # noqa: E402

lib = load_library()
M, N, K = 32768, 4608, 1152
dt, tdt = 1, torch.float16
g = torch.Generator(device="cuda").manual_seed(0)
bias = torch.randn(N, generator=g, device="cuda")
gate = torch.zeros(2 * N, device="cuda")


def run(A, W, variant, epi, n):
    out = torch.zeros(M, N, device="cuda", dtype=tdt)
    for _ in range(3):
        rc = lib.latte_debug_gemm(ptr(A), ptr(W), ptr(bias), ptr(out), ptr(gate), M, N, K, 0, M, epi, dt, variant, stream_ptr())
        if rc != 0:
            return None, None
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        check(lib.latte_debug_gemm(ptr(A), ptr(W), ptr(bias), ptr(out), ptr(gate), M, N, K, 0, M, epi, dt, variant, stream_ptr()))
    e1.record()
    torch.cuda.synchronize()
    return out, e0.elapsed_time(e1) * 1e3 / n


for pattern in ("random", "zeros"):
    if pattern == "random":
        A = torch.randn(M, K, generator=g, device="cuda").to(tdt)
        W = (torch.randn(N, K, generator=g, device="cuda") / K ** 0.5).to(tdt)
    else:
        A = torch.zeros(M, K, device="cuda", dtype=tdt)
        W = torch.zeros(N, K, device="cuda", dtype=tdt)
    for epi in (1, 0):
        ref, us9 = run(A, W, 9, epi, 300)
        for v in (14, 15, 3):
            out, us = run(A, W, v, epi, 300)
            if out is None:
                print(f"{pattern} epi {epi} variant {v}: not available in this build")
                continue
            same = bool(torch.equal(out, ref))
            diff = float((out.float() - ref.float()).abs().max())
            print(f"{pattern:7s} epi {epi} variant {v:2d}: {us:7.1f} us ({2.0 * M * N * K / us * 1e-6 / 2500:.3f} of peak)   variant 9: {us9:7.1f} us   "
                  f"bit-identical to variant 9: {same} (max abs diff {diff:.3e})", flush=True)
