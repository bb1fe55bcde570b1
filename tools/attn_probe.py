#!/usr/bin/env python
"""Interleaved A/B of the self-attention kernels on one shape (both in ONE process, alternating, so clock drift and box variance
cancel): latte_debug_set_choice("attn_variant", v) switches per launch.  Default shape = Latte-1 T2V spatial attention (32 sequences x 1024 tokens,
16 heads x 72).  Variants: 0 default choice, 1 generic flash kernel, 4 256-key block kernel, 5 streaming kernel also for
128 < L <= 256, 7 / 8 / 9 streaming kernel without DMA issue in the loop / without softmax / without barrier (results garbage).
--sync: synchronise after every warm-up launch (the FIRST launch of a kernel that needs scratch memory, e.g. the block kernel,
faulted on this pool when it was queued behind a running 120 KB-LDS kernel; with a synchronise in between it never did)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from latte_amd import _lib  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seqs", type=int, default=32)
    ap.add_argument("--L", type=int, default=1024)
    ap.add_argument("--heads", type=int, default=16)
    ap.add_argument("--hd", type=int, default=72)
    ap.add_argument("--variants", default="0,4")
    ap.add_argument("--rounds", type=int, default=6)
    ap.add_argument("--iters", type=int, default=40)
    ap.add_argument("--sync", action="store_true", help="synchronize after every warm-up launch (fault localisation)")
    ap.add_argument("--check", action="store_true", help="relative L2 error of every variant against fp32 torch on the same half operands")
    a = ap.parse_args()
    lib = _lib.load_library()
    D = a.heads * a.hd
    qkv = (torch.randn(a.seqs * a.L, 3 * D, device="cuda") * 0.5).half()
    out = torch.zeros(a.seqs * a.L, D, device="cuda", dtype=torch.half)
    st = torch.cuda.current_stream().cuda_stream
    flop = 4.0 * a.seqs * a.heads * a.L * a.L * a.hd

    def run(v):
        assert lib.latte_debug_set_choice(b"attn_variant", int(v)) == 0, f"variant {v} is not offered by this build (7-9: LATTE_DEBUG_BUILD=1)"
        rc = lib.latte_debug_attention(qkv.data_ptr(), out.data_ptr(), a.seqs, a.L, a.heads, a.hd, 1, a.L, a.L, 1, 1, st)
        assert rc == 0, rc

    vs = [int(v) for v in a.variants.split(",")]
    if a.check:
        q, k, v = [qkv.float().view(a.seqs, a.L, 3, a.heads, a.hd)[:, :, i].permute(0, 2, 1, 3) for i in range(3)]
        want = torch.cat([(torch.softmax((q[i:i + 4] @ k[i:i + 4].transpose(-1, -2)) * a.hd ** -0.5, dim=-1) @ v[i:i + 4])
                          for i in range(0, a.seqs, 4)]).permute(0, 2, 1, 3).reshape(a.seqs * a.L, D)
        for v_ in vs:
            out.zero_()
            run(v_)
            torch.cuda.synchronize()
            print(f"variant {v_}: rel L2 error {float((out.float() - want).norm() / want.norm()):.3e}, max abs {float((out.float() - want).abs().max()):.3e}", flush=True)
    best = {v: 1e9 for v in vs}
    for v in vs:
        run(v)
        if a.sync:
            torch.cuda.synchronize()
            print("warm-up of variant", v, "done", flush=True)
    torch.cuda.synchronize()
    for _ in range(a.rounds):
        for v in vs:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                run(v)
            e1.record()
            torch.cuda.synchronize()
            best[v] = min(best[v], e0.elapsed_time(e1) / a.iters * 1e3)
    for v in vs:
        print(f"variant {v}: {best[v]:8.1f} us  {flop / best[v] * 1e-6:7.1f} TF/s ({flop / best[v] * 1e-6 / 2500 * 100:.1f} % of dense peak)")


if __name__ == "__main__":
    main()
