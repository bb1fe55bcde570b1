#!/usr/bin/env python
"""Structured-input checks of one self-attention variant (1 sequence, 1 head): which stage of a new kernel is wrong."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from latte_amd import _lib
lib = _lib.load_library()
v = int(sys.argv[1]) if len(sys.argv) > 1 else 11
L = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
hd = 72
st = torch.cuda.current_stream().cuda_stream
def run(qkv):
    out = torch.zeros(L, hd, device="cuda", dtype=torch.half)
    assert lib.latte_debug_set_choice(b"attn_variant", v) == 0
    assert lib.latte_debug_attention(qkv.data_ptr(), out.data_ptr(), 1, L, 1, hd, 1, L, L, 1, 1, st) == 0
    torch.cuda.synchronize()
    return out.float()
def ref(qkv):
    q, k, vv = qkv.float()[:, :hd], qkv.float()[:, hd:2 * hd], qkv.float()[:, 2 * hd:]
    return torch.softmax((q @ k.t()) * hd ** -0.5, dim=-1) @ vv
def report(name, qkv):
    o, w = run(qkv), ref(qkv)
    err = (o - w).abs()
    print(f"{name}: rel {float((o - w).norm() / w.norm()):.3e} max {float(err.max()):.3e} nan {int(torch.isnan(o).sum())}")
    if float(err.max()) > 1e-2 or torch.isnan(o).any():
        bad = (err > 1e-2) | torch.isnan(o)
        rows = bad.any(1).nonzero().flatten()
        cols = bad.any(0).nonzero().flatten()
        print("   bad rows", rows[:16].tolist(), "n", len(rows), " bad cols", cols[:80].tolist())
        r = int(rows[0])
        print("   row", r, "got", o[r, :12].tolist(), "\n          want", w[r, :12].tolist())
g = torch.Generator("cpu").manual_seed(1)
z = torch.zeros(L, 3 * hd)
a = z.clone(); a[:, 2 * hd:] = 1.0
report("q=0 k=0 v=1", a.cuda().half())
a = z.clone(); a[:, 2 * hd:] = torch.arange(hd).float()[None, :] / 64
report("q=0 k=0 v=d/64", a.cuda().half())
a = z.clone(); a[:, 2 * hd:] = (torch.arange(L).float()[:, None] / L).expand(L, hd)
report("q=0 k=0 v=key/L", a.cuda().half())
a = z.clone(); a[:, 2 * hd:] = torch.randn(L, hd, generator=g)
report("q=0 k=0 v=rand", a.cuda().half())
a = torch.randn(L, 3 * hd, generator=g); a[:, 2 * hd:] = 1.0
report("q,k rand v=1", a.cuda().half())
a = torch.randn(L, 3 * hd, generator=g); a[:, 2 * hd:] = (torch.arange(L).float()[:, None] / L).expand(L, hd)
report("q,k rand v=key/L", a.cuda().half())
a = torch.randn(L, 3 * hd, generator=g)
report("all rand", a.cuda().half())
