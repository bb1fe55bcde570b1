"""Round-6 debugging aid: the fp8 correction pass on XL/2 shapes with REAL remainders (not random codes), stand-alone and in the engine."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import latte_amd  # noqa: E402
from latte_amd._lib import check, load_library, ptr, stream_ptr  # noqa: E402

lib = load_library()
dev = torch.device("cuda")
F8 = torch.float8_e4m3fn
g = torch.Generator("cpu").manual_seed(0)
for (M, N, K, amp) in [(8192, 1152, 1152, 0.05), (8192, 1152, 1152, 1.0), (8192, 384, 384, 0.05), (8192, 4608, 1152, 1.0)]:
    X = (torch.randn(M, K, generator=g) * amp).to(dev)
    Wf = ((torch.rand(N, K, generator=g) * 2 - 1) * (6.0 / (N + K)) ** 0.5).to(dev)
    A = X.to(torch.float16)
    W = Wf.to(torch.float16)
    A8 = ((X - A.float()) * 4096.0).clamp(-448, 448).to(F8).view(torch.uint8)
    W8 = torch.zeros(N, K, dtype=torch.uint8, device=dev)
    check(lib.latte_debug_pack_w8(ptr(W), ptr(W8), W.numel(), 1, stream_ptr()))
    bias = torch.zeros(N, device=dev)
    gate = torch.ones(1, N, device=dev)
    ref = X.double() @ W.double().t()
    out_p = torch.zeros(M, N, device=dev)
    check(lib.latte_debug_gemm(ptr(A), ptr(W), ptr(bias), ptr(out_p), ptr(gate), M, N, K, N, M, 2, 1, 11, stream_ptr()))
    out_l = torch.zeros(M, N, device=dev)
    check(lib.latte_debug_gemm_lo8(ptr(A), ptr(W), ptr(A8), ptr(W8), ptr(bias), ptr(out_l), ptr(gate), M, N, K, N, M, 2, 1, stream_ptr()))
    torch.cuda.synchronize()
    lo_true = (X.double() - A.double()) @ W.double().t()
    print(f"M {M} N {N} K {K} amp {amp}: plain err {float((out_p - ref).norm() / ref.norm()):.3e}  with correction {float((out_l - ref).norm() / ref.norm()):.3e}"
          f"   (kernel's correction vs true remainder product: {float(((out_l - out_p).double() - lo_true).norm() / lo_true.norm()):.3e})", flush=True)

kw = dict(input_size=32, num_frames=16, num_classes=101, extras=2)
for name in ("Latte-S/2", "Latte-XL/2"):
    m = latte_amd.Latte_models[name](max_batch=2, **kw)
    with torch.no_grad():
        gg = torch.Generator().manual_seed(0)
        for _, p in m.named_parameters():
            if float(p.abs().max()) == 0.0:
                p.copy_(torch.randn(p.shape, generator=gg) * 0.3)
    m = m.cuda()
    z = torch.randn(1, 16, 4, 32, 32, device="cuda")
    x = torch.cat([z, z])
    t = torch.full((2,), 500, device="cuda", dtype=torch.int64)
    y = torch.tensor([5, 101], device="cuda")
    outs = {}
    for gs in (0, 1, 4, 2, 8, 3, 12):
        m.set_engine_option("guided_split", gs, 2, guided=True)
        outs[gs] = m.forward_with_cfg(x, t, y=y, cfg_scale=7.0).clone()
        print(name, gs, "active", m.get_engine_option("guided_split_active", 2, guided=True))
    r = lambda a, b: float((a - b).norm() / b.norm())
    print(name, "|1-0|", r(outs[1], outs[0]), "|4-0|", r(outs[4], outs[0]), "|4-1|", r(outs[4], outs[1]), "|8-2|", r(outs[8], outs[2]), "|2-0|", r(outs[2], outs[0]),
          "|12-3|", r(outs[12], outs[3]), "|3-0|", r(outs[3], outs[0]), flush=True)
