"""Gated GEMMs with the epilogue burst hidden under the K loop (variant 17, the two-accumulator-set kernel of gemm_pw.hip; round 5
also ran variant 16, the alternating-groups form, with this script) against the rolling 12-wave kernel (variant 11) on the MI355X:
bit-identity, stand-alone launch time on random operands, per-launch time INSIDE the XL/2 forward (HIP events of
latte_profile_forward), interleaved settings in one process.  Needs the measurement build:
    LATTE_DEBUG_BUILD=1 python -m latte_amd.build;  LATTE_AMD_LIB=latte_amd/lib/liblatte_amd_dbg.so python tools/alt_probe.py [check]
    [standalone] [inmodel] [trace] [B=8,2] [dtype=f16] [variants=11,17]
Result (profiles/r5_gated_overlap_*.log): same bits, the burst is hidden, the launch is slower -- a halved tile moves 1.43 x the operand
bytes per MFMA through the LDS-DMA path, which is what bounds these kernels."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from latte_amd._lib import check, load_library, ptr, stream_ptr  # noqa: E402

lib = load_library()
dev = torch.device("cuda:0")
args = sys.argv[1:]
what = [a for a in args if "=" not in a] or ["check", "standalone", "inmodel"]
kv = dict(a.split("=") for a in args if "=" in a)
DT = {"bf16": 0, "f16": 1}[kv.get("dtype", "f16")]
TD = {0: torch.bfloat16, 1: torch.float16}[DT]
VARIANTS = [int(v) for v in kv.get("variants", "11,17").split(",")]


def log(*a):
    print(*a, flush=True)


def do_check():
    for (M, N, K, rps) in [(4096, 1152, 1152, 4096), (8192, 1152, 4608, 4096), (33000, 1152, 1152, 128), (65536, 1152, 1152, 4096),
                           (2304, 384, 1152, 256), (32768, 1152, 4608, 4096), (300, 192, 1216, 128)]:
        g = torch.Generator("cpu").manual_seed(M + K)
        Mp = (M + 255) // 256 * 256
        A = torch.randn(Mp, K, generator=g).to(dev).to(TD)
        W = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev).to(TD)
        bias = torch.randn(N, generator=g).to(dev)
        gate = torch.randn((M + rps - 1) // rps, 2 * N, generator=g).to(dev)
        out0 = torch.randn(Mp, N, generator=g).to(dev)
        outs = {}
        for variant in (11, 17, 1017):
            out = out0.clone()
            check(lib.latte_debug_gemm(ptr(A), ptr(W), ptr(bias), ptr(out), ptr(gate), M, N, K, 2 * N, rps, 2, DT, variant, stream_ptr()))
            torch.cuda.synchronize()
            outs[variant] = out
        want = out0[:M] + gate[torch.arange(M, device=dev) // rps, :N] * (A.float()[:M] @ W.float().t() + bias)
        for v in (17,):
            err = float((outs[v][:M] - want).norm() / want.norm())
            same = torch.equal(outs[v], outs[11]) and torch.equal(outs[1000 + v], outs[11])
            pad_ok = torch.equal(outs[v][M:], out0[M:])
            log(f"check v{v} M={M} N={N} K={K} rps={rps}: rel err {err:.2e}, bit-identical to v11: {same}, pad rows untouched: {pad_ok}")
        # second launch on the result (stale-state screen): res += again
        out = outs[17].clone()
        ref = outs[11].clone()
        check(lib.latte_debug_gemm(ptr(A), ptr(W), ptr(bias), ptr(out), ptr(gate), M, N, K, 2 * N, rps, 2, DT, 17, stream_ptr()))
        check(lib.latte_debug_gemm(ptr(A), ptr(W), ptr(bias), ptr(ref), ptr(gate), M, N, K, 2 * N, rps, 2, DT, 11, stream_ptr()))
        torch.cuda.synchronize()
        log(f"      second launch identical: {torch.equal(out, ref)}")


def do_standalone():
    import ctypes
    ms = ctypes.c_float()
    for (M, N, K, name) in [(32768, 1152, 1152, "proj"), (32768, 1152, 4608, "fc2"), (65536, 1152, 1152, "proj B16"), (65536, 1152, 4608, "fc2 B16"),
                            (8192, 1152, 1152, "proj B2"), (8192, 1152, 4608, "fc2 B2")]:
        rows = {v: [] for v in VARIANTS}
        for rep in range(4):
            for v in VARIANTS:
                check(lib.latte_bench_gemm(M, N, K, 2, DT, v + (1000 if K > N else 0), 30, ctypes.byref(ms), stream_ptr()))
                rows[v].append(ms.value * 1e3)
        fl = 2.0 * M * N * K
        log(f"standalone {name:9s} M={M} K={K}: " + " | ".join(
            f"v{v}: med {sorted(r)[len(r) // 2]:6.1f} us min {min(r):6.1f} ({fl / min(r) / 1e6 / 2.5e3:.3f} of peak)" for v, r in rows.items()))


def do_inmodel():
    from latte_amd.models import Latte_models
    for B in [int(v) for v in kv.get("B", "8").split(",")]:
        m = Latte_models["Latte-XL/2"](compute_dtype=kv.get("dtype", "f16"), max_batch=B, input_size=32, num_frames=16, extras=1)
        with torch.no_grad():
            gcpu = torch.Generator().manual_seed(0)
            for n_, p_ in m.named_parameters():
                if float(p_.abs().max()) == 0.0:
                    p_.copy_(torch.randn(p_.shape, generator=gcpu) * 0.02)
        m = m.to(dev)
        x = torch.randn(B, 16, 4, 32, 32, device=dev)
        t = torch.full((B,), 500, device=dev, dtype=torch.int64)
        res = {}
        outs = {}
        settings = [("v11/v11", 0, 0), ("v17/v11", 17, 0), ("v11/v17", 0, 17), ("v17/v17", 17, 17)]
        for rep in range(4):
            for name, vp, vf in settings:
                m.set_engine_option("gemm_variant_proj", vp, B)
                m.set_engine_option("gemm_variant_fc2", vf, B)
                m.profile_forward(x, t)
                pr = m.profile_forward(x, t)
                # whole forward, un-profiled: 5 back-to-back forwards between events
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                o = m(x, t)
                e0.record()
                for _ in range(5):
                    o = m(x, t)
                e1.record()
                torch.cuda.synchronize()
                res.setdefault(name, []).append((pr["gemm_proj"][0] / 28 * 1e3, pr["gemm_fc2"][0] / 28 * 1e3, pr["ln_modulate"][0] / 56 * 1e3,
                                                 pr["gemm_fc1"][0] / 28 * 1e3, e0.elapsed_time(e1) / 5))
                outs[name] = o
        for name, rows in res.items():
            best = [min(r[i] for r in rows) for i in range(5)]
            med = [sorted(r[i] for r in rows)[len(rows) // 2] for i in range(5)]
            log(f"in-model B={B} {name}: proj {best[0]:6.1f} us  fc2 {best[1]:6.1f} us  ln {best[2]:5.1f}  fc1 {best[3]:6.1f} | forward min {best[4]:.3f} ms med {med[4]:.3f} ms"
                f"  same bits as default: {torch.equal(outs[name], outs['v11/v11'])}")
        m.set_engine_option("gemm_variant_proj", 0, B)
        m.set_engine_option("gemm_variant_fc2", 0, B)
        del m
        torch.cuda.empty_cache()


def do_trace():
    """Measurement build (LATTE_AMD_LIB=latte_amd/lib/liblatte_amd_dbg.so): per-wave phase times of workgroup 0, shader clocks per K step.
    v11 consumers: h0 | h1 head + lgkm | barrier | h1 rest | epilogue | refill;  v16 consumers: the same four K-step phases, then
    epilogue-slice work | (barrier waits of the epilogue phase + fragment fill);  producers: DMA issue | vmcnt wait | barrier | - | - | bookkeeping."""
    for (M, N, K, var) in [(32768, 1152, 1152, 11), (32768, 1152, 4608, 11)]:
        A = torch.randn(M, K, device=dev).to(TD)
        W = (torch.randn(N, K, device=dev) / K ** 0.5).to(TD)
        bias = torch.randn(N, device=dev)
        out = torch.zeros(M * N, dtype=torch.float32, device=dev)
        gate = torch.randn(2 * N, device=dev)
        for rep in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            check(lib.latte_debug_gemm(ptr(A), ptr(W), ptr(bias), ptr(out), ptr(gate), M, N, K, 0, M, 12, DT, var, stream_ptr()))
            e1.record()
            torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3
        t = out.view(torch.int64)[:96].cpu().view(12, 8)
        log(f"trace v{var} M={M} N={N} K={K}: launch {us:.1f} us (no stores); ticks per K step of the wave's own count")
        for w in range(12):
            kt = max(int(t[w, 7]), 1)
            row = " ".join(f"[{i}] {float(t[w, i]) / kt:7.1f}" for i in range(6))
            log(f"   wave {w:2d} ({'grp %d' % (w >> 2) if w < 8 else 'producer'}): {row} | {int(t[w, 6]) / us:.0f} ticks/us, {kt} K steps, {float(t[w, 6]) / kt:.1f} ticks/K step")


t0 = time.time()
for w in what:
    {"check": do_check, "standalone": do_standalone, "inmodel": do_inmodel, "trace": do_trace}[w]()
log(f"alt_probe done in {time.time() - t0:.1f} s")
