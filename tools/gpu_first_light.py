"""First-light / measurement report on a real MI355X (run through gpurun; writes gpurun_out/first_light.log).

Not a test: every section is independent and prints errors + timings so one run answers as many
questions as possible (GPU minutes are scarce)."""
import ctypes
import os
import sys
import time
import traceback

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
OUT = os.path.join(ROOT, "gpurun_out")
os.makedirs(OUT, exist_ok=True)
LOG = open(os.path.join(OUT, "first_light.log"), "a")


def log(*a):
    s = " ".join(str(x) for x in a)
    print(s, flush=True)
    LOG.write(s + "\n")
    LOG.flush()


def section(fn):
    log(f"\n===== {fn.__name__} =====")
    t0 = time.time()
    try:
        fn()
    except Exception:
        log("SECTION FAILED:\n" + traceback.format_exc())
    log(f"[{fn.__name__}: {time.time() - t0:.1f}s]")


from latte_amd import _lib  # noqa: E402
from latte_amd._lib import check, load_library, ptr, stream_ptr  # noqa: E402

lib = load_library()
dev = torch.device("cuda")
TD = {0: torch.bfloat16, 1: torch.float16}


def to_h16(x, dt):
    return x.to(TD[dt]).contiguous()


def env():
    log("torch", torch.__version__, "hip", torch.version.hip, "dev", torch.cuda.get_device_name(0))
    p = torch.cuda.get_device_properties(0)
    log("CUs", p.multi_processor_count, "mem GB", p.total_memory / 2**30, "cpus", os.cpu_count())
    log(lib.latte_version().decode())


def tr16_probe():
    out = torch.zeros(256, dtype=torch.int16, device=dev)
    check(lib.latte_debug_tr16_probe(ptr(out), stream_ptr()))
    torch.cuda.synchronize()
    o = out.cpu().numpy().astype(np.int64).reshape(64, 4)
    for l in range(0, 64, 1):
        if l < 20 or l % 16 == 0:
            log(f"lane {l:2d}: {o[l].tolist()}")
    # hypothesis: out[l][j] = lds[ 4*(4*j + (l&15)>>2) + (l&3) + 64*(l>>4) ]
    ok = True
    for l in range(64):
        for j in range(4):
            i = l & 15
            want = 64 * (l >> 4) + 4 * (4 * j + (i >> 2)) + (i & 3)
            ok &= (o[l, j] == want)
    log("hypothesis out[l][j] = lds[64*(l>>4) + 16*j + 4*((l&15)>>2) + (l&3)]... ->", ok)
    ok2 = all(o[l, j] == 64 * (l >> 4) + 16 * j + (l & 15) for l in range(64) for j in range(4))
    log("hypothesis out[l][j] = lds[64*(l>>4) + 16*j + (l&15)] ->", ok2)


def gemm_checks():
    g = torch.Generator(device="cpu").manual_seed(0)
    for dt in (0, 1):
        for variant in (1, 2, 3):
            for (M, N, K) in [(256, 256, 128), (512, 768, 1152), (300, 512, 256)]:
                if variant == 3 and N % 256:
                    continue
                Mp = (M + 255) // 256 * 256
                A = torch.randn(Mp, K, generator=g).to(dev)
                W = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev)
                bias = torch.randn(N, generator=g).to(dev)
                Ah, Wh = to_h16(A, dt), to_h16(W, dt)
                ref = Ah.float()[:M] @ Wh.float().t() + bias
                for epi in (0, 1, 2, 3):
                    rps = 64
                    nsamp = (M + rps - 1) // rps
                    gate = torch.randn(nsamp, 2 * N, generator=g).to(dev)
                    if epi in (0, 1):
                        out = torch.zeros(Mp, N, dtype=TD[dt], device=dev)
                    else:
                        out = torch.randn(Mp, N, generator=g).to(dev)
                    out0 = out.clone()
                    check(lib.latte_debug_gemm(ptr(Ah), ptr(Wh), ptr(bias), ptr(out), ptr(gate), M, N, K,
                                               2 * N, rps, epi, dt, variant, stream_ptr()))
                    torch.cuda.synchronize()
                    if epi == 0:
                        want = ref
                    elif epi == 1:
                        want = torch.nn.functional.gelu(ref, approximate="tanh")
                    elif epi == 2:
                        gi = torch.arange(M, device=dev) // rps
                        want = out0[:M] + gate[gi, :N] * ref
                    else:
                        want = ref
                    got = out[:M].float()
                    err = (got - want).abs().max().item()
                    rel = ((got - want).norm() / want.norm()).item()
                    pad_ok = bool((out[M:] == out0[M:]).all()) if M < Mp else True
                    flag = "" if rel < (6e-3 if epi in (0, 1) else 1e-4) and pad_ok else "  <<<<<< BAD"
                    log(f"gemm dt={dt} var={variant} M={M} N={N} K={K} epi={epi}: max|d|={err:.3e} rel={rel:.3e} pad_untouched={pad_ok}{flag}")


def attention_checks():
    g = torch.Generator(device="cpu").manual_seed(1)
    for dt in (0, 1):
        for (B, F, T, H, hd) in [(1, 4, 16, 2, 64), (2, 16, 256, 16, 72), (1, 4, 64, 6, 64), (1, 3, 100, 2, 72), (1, 16, 1024, 6, 64)]:
            D = H * hd
            rows = B * F * T
            qkv = torch.randn(rows, 3 * D, generator=g).to(dev)
            qh = to_h16(qkv, dt)
            q5 = qh.float().view(B, F, T, 3, H, hd)
            for mode in ("spatial", "temporal"):
                if mode == "spatial":
                    q, k, v = [q5[:, :, :, i].permute(0, 1, 3, 2, 4) for i in range(3)]      # [B,F,H,T,hd]
                    a = torch.softmax((q @ k.transpose(-1, -2)) * hd ** -0.5, dim=-1) @ v
                    want = a.permute(0, 1, 3, 2, 4).reshape(rows, D)
                    args = (B * F, T, H, hd, F, F * T, T, 1)
                else:
                    q, k, v = [q5[:, :, :, i].permute(0, 2, 3, 1, 4) for i in range(3)]      # [B,T,H,F,hd]
                    a = torch.softmax((q @ k.transpose(-1, -2)) * hd ** -0.5, dim=-1) @ v
                    want = a.permute(0, 3, 1, 2, 4).reshape(rows, D)
                    args = (B * T, F, H, hd, T, F * T, 1, T)
                out = torch.zeros(rows, D, dtype=TD[dt], device=dev)
                check(lib.latte_debug_attention(ptr(qh), ptr(out), *args, dt, stream_ptr()))
                torch.cuda.synchronize()
                got = out.float()
                rel = ((got - want).norm() / want.norm()).item()
                flag = "" if rel < 8e-3 else "  <<<<<< BAD"
                log(f"attn dt={dt} {mode} B={B} F={F} T={T} H={H} hd={hd}: max|d|={(got - want).abs().max().item():.3e} rel={rel:.3e}{flag}")


def ln_checks():
    g = torch.Generator(device="cpu").manual_seed(2)
    for dt in (0, 1):
        for D in (128, 384, 1152):
            B, F, T = 2, 4, 16
            M = B * F * T
            x = (torch.randn(M, D, generator=g) * 3 + 0.5).to(dev)
            mod = torch.randn(B, 6 * D, generator=g).to(dev)
            te = torch.randn(F, D, generator=g).to(dev)
            for use_te in (False, True):
                xin = x.clone()
                y = torch.zeros(M, D, dtype=TD[dt], device=dev)
                check(lib.latte_debug_ln_modulate(ptr(xin), ptr(y), ptr(mod), ptr(mod[:, D:]), 6 * D, M, D, F * T,
                                                  ptr(te) if use_te else None, T, F, dt, stream_ptr()))
                torch.cuda.synchronize()
                xr = x.clone()
                if use_te:
                    xr = (xr.view(B, F, T, D) + te.view(1, F, 1, D)).view(M, D)
                s = torch.arange(M, device=dev) // (F * T)
                want = torch.nn.functional.layer_norm(xr, (D,), eps=1e-6) * (1 + mod[s, D:2 * D]) + mod[s, :D]
                rel = ((y.float() - want).norm() / want.norm()).item()
                xerr = (xin - xr).abs().max().item()
                log(f"ln dt={dt} D={D} te={use_te}: rel={rel:.3e} x_writeback_err={xerr:.2e}")


def normal_check():
    n = 1 << 22
    out = torch.empty(n, device=dev)
    check(lib.latte_debug_fill_normal(ptr(out), n, 123, 0, stream_ptr()))
    torch.cuda.synchronize()
    log(f"fill_normal: mean={out.mean().item():.4f} std={out.std().item():.4f} kurt={(out**4).mean().item():.3f} finite={bool(torch.isfinite(out).all())}")


def gemm_bench():
    ms = _lib.c_f32()
    for M in (4096, 8192, 32768):
        for (N, K, epi, nm) in [(3456, 1152, 0, "qkv"), (1152, 1152, 2, "proj"), (4608, 1152, 1, "fc1"), (1152, 4608, 2, "fc2")]:
            row = []
            for variant in (1, 5, 6, 7, 8, 9, 10, 11):
                if N % {1: 128, 5: 192, 6: 256, 7: 128, 8: 192, 9: 256, 10: 192, 11: 192}[variant]:
                    continue
                check(lib.latte_bench_gemm(M, N, K, epi, 0, variant, 20, ctypes.byref(ms), stream_ptr()))
                tf = 2.0 * M * N * K / (ms.value * 1e-3) / 1e12
                row.append(f"v{variant}: {ms.value*1e3:6.1f}us {tf:5.0f}TF")
            log(f"gemm_bench M={M:5d} {nm:4s} N={N} K={K}: " + " | ".join(row))
    for (M, N, K, epi) in [(8192, 4096, 4096, 0), (4096, 4096, 4096, 0), (8192, 8192, 8192, 0)]:
        row = []
        for variant in (6, 9):
            check(lib.latte_bench_gemm(M, N, K, epi, 0, variant, 10, ctypes.byref(ms), stream_ptr()))
            tf = 2.0 * M * N * K / (ms.value * 1e-3) / 1e12
            row.append(f"v{variant}: {ms.value*1e3:6.1f}us {tf:5.0f}TF")
        log(f"gemm_bench M={M} N={N} K={K}: " + " | ".join(row))



def gemm_epi_ablation():
    """Same shape, different epilogues on the persistent kernel: 0 bf16 out, 1 GELU bf16, 2 gated fp32 RMW, 3 fp32 out, 4 no store."""
    ms = _lib.c_f32()
    for (M, N, K, variant) in [(32768, 4608, 1152, 9), (32768, 4608, 1152, 8), (32768, 3456, 1152, 8), (32768, 1152, 4608, 8),
                               (32768, 1152, 1152, 8)]:
        row = []
        for epi in (0, 1, 2, 3, 4):
            check(lib.latte_bench_gemm(M, N, K, epi, 0, variant, 20, ctypes.byref(ms), stream_ptr()))
            tf = 2.0 * M * N * K / (ms.value * 1e-3) / 1e12
            row.append(f"epi{epi}: {ms.value*1e3:6.1f}us {tf:5.0f}TF")
        log(f"epi_ablation v{variant} M={M} N={N} K={K}: " + " | ".join(row))


def gemm_loop_ablation():
    """Main-loop ablations of the persistent kernel (no stores in all of them): 4 = full loop, 6 = no DMA in the loop,
    7 = no LDS fragment reads, 8 = no MFMA, 9 = DMA from a cache-hot source, 10 / 11 = only A / only B DMA'd."""
    ms = _lib.c_f32()
    for (M, N, K, variant) in [(8192, 4096, 4096, 9), (32768, 4608, 1152, 9), (32768, 1152, 4608, 8), (8192, 4608, 4096, 8)]:
        row = []
        check(lib.latte_bench_gemm(M, N, K, 4, 0, variant, 20, ctypes.byref(ms), stream_ptr()))   # warm-up (clocks)
        for epi in (4, 6, 7, 8, 9, 10, 11):
            check(lib.latte_bench_gemm(M, N, K, epi, 0, variant, 20, ctypes.byref(ms), stream_ptr()))
            tf = 2.0 * M * N * K / (ms.value * 1e-3) / 1e12
            row.append(f"epi{epi}: {ms.value*1e3:6.1f}us {tf:5.0f}TF")
        log(f"loop_ablation v{variant} M={M} N={N} K={K}: " + " | ".join(row))


def gemm_trace():
    """Per-wave phase times of the persistent kernel's main loop (workgroup 0), s_memtime ticks."""
    names = ["L(reads)", "bar1 wait", "C issue", "vmcnt", "bar2 wait", "DMA issue"]
    for (M, N, K, v) in [(32768, 4608, 1152, 9), (32768, 1152, 4608, 8)]:
        Mp = (M + 255) // 256 * 256
        A = torch.randn(Mp, K, device=dev).to(torch.bfloat16)
        W = (torch.randn(N, K, device=dev) / K ** 0.5).to(torch.bfloat16)
        bias = torch.randn(N, device=dev)
        out = torch.zeros(Mp * N, dtype=torch.float32, device=dev)
        gate = torch.randn(2 * N, device=dev)
        for rep in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            check(lib.latte_debug_gemm(ptr(A), ptr(W), ptr(bias), ptr(out), ptr(gate), M, N, K, 0, M, 12, 0, v, stream_ptr()))
            e1.record()
            torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3
        t = out.view(torch.int64)[:64].cpu().view(8, 8)
        log(f"trace v{v} M={M} N={N} K={K}: launch {us:.1f} us; wave rows: per-K-tile ticks; total ticks, K tiles")
        for w in range(8):
            kt = int(t[w, 7])
            row = " ".join(f"{names[i]} {float(t[w, i]) / kt:7.1f}" for i in range(6))
            log(f"   wave {w} (grp {w >> 2}): {row} | total {int(t[w, 6])} ticks = {int(t[w, 6]) / us:.1f} ticks/us, {kt} K tiles, {float(t[w, 6]) / kt:.1f} ticks/K tile")


def gemm_trace_pw():
    """Measurement build: per-wave phase times of the producer-wave kernel (variant 10, workgroup 0), s_memtime ticks.
    Consumers: L(reads) | barrier-1 wait | C (MFMAs + second-half reads) | epilogue g1 | barrier-2 wait | epilogue g0.
    Producers: DMA issue | vmcnt wait (even interval) | barrier | vmcnt wait (odd, incl. issue) | barrier | bookkeeping."""
    names = {10: (["L(reads)", "bar1 wait", "C", "epi(g1)", "bar2 wait", "epi(g0)"],
                  ["DMA issue", "vmcnt even", "bar even", "vmcnt odd", "bar odd", "bookkeep"]),
             11: (["h0", "h1 head+lgkm", "barrier", "h1 rest", "epilogue", "refill"],
                  ["DMA issue", "vmcnt wait", "barrier", "-", "-", "bookkeep"])}
    for (M, N, K, var) in [(32768, 1152, 4608, 10), (32768, 1152, 4608, 11), (32768, 1152, 1152, 11), (32768, 4608, 1152, 11)]:
        cn, pn = names[var]
        A = torch.randn(M, K, device=dev).to(torch.bfloat16)
        W = (torch.randn(N, K, device=dev) / K ** 0.5).to(torch.bfloat16)
        bias = torch.randn(N, device=dev)
        out = torch.zeros(M * N, dtype=torch.float32, device=dev)
        gate = torch.randn(2 * N, device=dev)
        for rep in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            check(lib.latte_debug_gemm(ptr(A), ptr(W), ptr(bias), ptr(out), ptr(gate), M, N, K, 0, M, 12, 0, var, stream_ptr()))
            e1.record()
            torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3
        t = out.view(torch.int64)[:96].cpu().view(12, 8)
        log(f"trace v{var} M={M} N={N} K={K}: launch {us:.1f} us; per-K-tile ticks")
        for w in range(12):
            kt = max(int(t[w, 7]), 1)
            nm = cn if w < 8 else pn
            row = " ".join(f"{nm[i]} {float(t[w, i]) / kt:7.1f}" for i in range(6))
            log(f"   wave {w:2d} ({'grp %d' % (w >> 2) if w < 8 else 'producer'}): {row} | {int(t[w, 6]) / us:.0f} ticks/us, {kt} K tiles, {float(t[w, 6]) / kt:.1f} ticks/K tile")


def dma_probe():
    """Issue / completion cost of the operand paths (cache-hot source), ticks per 1 KB (or 256 B) wave instruction."""
    src = torch.randn(16 << 20, device=dev)           # 64 MiB window
    out = torch.zeros(128, dtype=torch.int64, device=dev)
    reps = 200
    names = {0: "buffer_load x4 -> lds", 1: "buffer_load x1 -> lds", 2: "global_load x4 -> vgpr", 3: "global_load x4 + ds_write_b128",
             4: "buffer_load x4 -> lds, one M0", 5: "buffer_load x4 -> lds, imm offsets", 6: "global_load x4 -> lds"}
    for mode in (7, 8, 9):   # 4 DMA waves + 4 MFMA companion waves
        for _ in range(2):
            check(lib.latte_debug_dma_probe(ptr(src), ptr(out), mode, 8, reps, stream_ptr()))
        torch.cuda.synchronize()
        t = out[:16].cpu().view(8, 2).double()
        log(f"dma_probe mode {mode} (4 DMA waves next to 4 MFMA waves; 7: MFMA at prio 1, 8: no prio, 9: DMA at prio 3): "
            f"DMA issue {(t[:4, 0] / (reps * 16)).mean():6.1f} landed {(t[:4, 1] / (reps * 16)).mean():6.1f} ticks/instr/wave; "
            f"MFMA {(t[4:, 0] / t[4:, 1]).mean():5.1f} ticks per MFMA (total MFMA ticks {t[4:, 0].mean():.0f}, DMA ticks {t[:4, 1].mean():.0f})")
    names[10] = "GEMM A pattern, own panels (8 instr/burst)"
    names[11] = "GEMM A pattern, shared panels (8 instr/burst)"
    src = torch.randn(160 << 20, device=dev)          # 640 MiB: 1024 panels of 256 rows x 2304 B
    for mode in (10, 11):
        for waves in (4, 8):
            for _ in range(2):
                check(lib.latte_debug_dma_probe(ptr(src), ptr(out), mode, waves, reps, stream_ptr()))
            torch.cuda.synchronize()
            t = out[:16].cpu().view(8, 2).double() / (reps * (8 if mode >= 10 else 16))
            log(f"dma_probe {names[mode]:32s} waves/CU {waves}: issue {t[:waves, 0].mean():7.1f} ticks/instr/wave, "
                f"landed {t[:waves, 1].mean():7.1f} ticks/instr/wave -> CU-wide {t[:waves, 1].mean() / waves:6.1f} ticks/instr")


def gemm_in_model():
    """Per-GEMM tile variants measured INSIDE the XL/2 forward (cache state of the real pipeline), B = 8 and 2."""
    from latte_amd.models import Latte_models
    for B in [int(v) for v in os.environ.get("LATTE_FL_B", "8,2").split(",")]:
        m = Latte_models["Latte-XL/2"](compute_dtype="bf16", max_batch=B, input_size=32, num_frames=16, extras=1)
        with torch.no_grad():
            for n_, p_ in m.named_parameters():
                if float(p_.abs().max()) == 0.0:
                    p_.normal_(0, 0.02)
        m = m.to(dev)
        x = torch.randn(B, 16, 4, 32, 32, device=dev)
        t = torch.full((B,), 500, device=dev, dtype=torch.int64)
        for gname, key in (("qkv", "gemm_qkv"), ("proj", "gemm_proj"), ("fc1", "gemm_fc1"), ("fc2", "gemm_fc2")):
            row = []
            for v in [int(s) for s in os.environ.get('LATTE_FL_VARIANTS', '0,1,5,6,7,8,9,10,11,12').split(',')]:
                try:
                    m.set_engine_option("gemm_variant_" + gname, v, B)
                    m.profile_forward(x, t)
                    ms = min(m.profile_forward(x, t)[key][0] for _ in range(3))
                    row.append(f"v{v}: {ms/28*1e3:6.1f}us")
                except Exception as ex:
                    row.append(f"v{v}: n/a")
            m.set_engine_option("gemm_variant_" + gname, 0, B)
            log(f"in-model B={B} {gname}: " + " | ".join(row))
        del m
        torch.cuda.empty_cache()


def rmw_ahead():
    """Measurement build (LATTE_DEBUG_BUILD=1): look-ahead depth / non-temporal loads of the synchronous read-modify-write
    epilogue (LATTE_RMW_MODE = depth + 16 * nt), stand-alone with an output larger than the Infinity Cache (M = 65536:
    the residual is HBM-cold, as inside the model) and inside the XL/2 forward at B = 8."""
    ms = _lib.c_f32()
    modes = [0, 3, 4, 6, 8, 12, 18, 20, 22, 24]
    for (N, K, nm, tag) in [(1152, 1152, "proj", 0), (1152, 4608, "fc2", 1)]:
        for M in (65536, 32768):
            row = []
            for mode in modes + [0]:
                os.environ["LATTE_RMW_MODE"] = str(mode)
                check(lib.latte_bench_gemm(M, N, K, 2, 0, 8 + 1000 * tag, 20, ctypes.byref(ms), stream_ptr()))
                row.append(f"m{mode}: {ms.value*1e3:6.1f}")
            log(f"rmw_ahead {nm} M={M} (us): " + " | ".join(row))
    from latte_amd.models import Latte_models
    B = 8
    m = Latte_models["Latte-XL/2"](compute_dtype="bf16", max_batch=B, input_size=32, num_frames=16, extras=1)
    with torch.no_grad():
        for n_, p_ in m.named_parameters():
            if float(p_.abs().max()) == 0.0:
                p_.normal_(0, 0.02)
    m = m.to(dev)
    x = torch.randn(B, 16, 4, 32, 32, device=dev)
    t = torch.full((B,), 500, device=dev, dtype=torch.int64)
    os.environ["LATTE_RMW_VARIANT"] = "8"
    for mode in modes + [0]:
        os.environ["LATTE_RMW_MODE"] = str(mode)
        m.profile_forward(x, t)
        pr = [m.profile_forward(x, t) for _ in range(3)]
        row = " ".join(f"{k}: {min(p[k][0] for p in pr) / max(pr[0][k][1], 1) * 1e3:6.1f}us" for k in
                       ("gemm_proj", "gemm_fc2", "gemm_qkv", "gemm_fc1", "ln_modulate"))
        tot = min(sum(v_[0] for v_ in p.values()) for p in pr)
        log(f"rmw_ahead in-model B={B} mode {mode}: {row} | forward {tot:.3f} ms")
    os.environ["LATTE_RMW_MODE"] = "0"
    del os.environ["LATTE_RMW_VARIANT"]


def attn_variants():
    """Attention kernels on the model shapes: XL/2 spatial (L = 256) through attn_full / the block kernel / the generic flash
    kernel, Latte-1 spatial (L = 1024, B = 2) through the block kernel / flash, temporal (L = 16)."""
    def run(B, F, T, mode, env):
        H, hd = 16, 72
        D, rows = H * hd, B * F * T
        qkv = torch.randn(rows, 3 * D, device=dev).bfloat16()
        out = torch.zeros(rows, D, dtype=torch.bfloat16, device=dev)
        args = (B * F, T, H, hd, F, F * T, T, 1) if mode == "spatial" else (B * T, F, H, hd, T, F * T, 1, T)
        if mode == "temporal_adjacent":   # the 16 frames of a token as 16 ADJACENT rows (a (b, t, f) row order): layout probe
            args = (B * T, F, H, hd, T, F * T, F, 1)
        check(lib.latte_debug_set_choice(b"attn_variant", 0 if env is None else int(env)))
        for _ in range(3):
            check(lib.latte_debug_attention(ptr(qkv), ptr(out), *args, 0, stream_ptr()))
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            check(lib.latte_debug_attention(ptr(qkv), ptr(out), *args, 0, stream_ptr()))
        e1.record()
        torch.cuda.synchronize()
        check(lib.latte_debug_set_choice(b"attn_variant", 0))
        us = e0.elapsed_time(e1) * 1e3 / 20
        fl = 4.0 * B * F * T * T * D if mode == "spatial" else 4.0 * B * T * F * F * D
        return us, fl / us / 1e6, out.float()
    for (B, F, T, nm) in [(8, 16, 256, "XL/2 spatial L=256 B=8"), (2, 16, 1024, "Latte-1 spatial L=1024 B=2")]:
        row, ref = [], None
        for env, name in ((None, "default"), (5, "stream"), (1, "flash")):
            us, tf, o = run(B, F, T, "spatial", env)
            ref = o if ref is None else ref
            row.append(f"{name}: {us:7.1f}us {tf:5.0f}TF d={float((o - ref).abs().max()):.1e}")
        log(f"attn_variants {nm}: " + " | ".join(row))
    us, tf, _ = run(8, 16, 256, "temporal", None)
    log(f"attn_variants XL/2 temporal L=16 B=8: {us:.1f}us, {8*16*256*4*1152*2/us/1e6:.2f} TB/s algorithmic")
    us, tf, _ = run(8, 16, 256, "temporal_adjacent", None)
    log(f"attn_variants XL/2 temporal L=16 B=8, frames of a token in adjacent rows: {us:.1f}us, {8*16*256*4*1152*2/us/1e6:.2f} TB/s algorithmic")


def group_m_sweep():
    """Measurement build: tile rows walked together by the persistent GEMM (LATTE_GROUP_M=epi:value), inside the XL/2 forward."""
    from latte_amd.models import Latte_models
    B = 8
    m = Latte_models["Latte-XL/2"](compute_dtype="bf16", max_batch=B, input_size=32, num_frames=16, extras=1)
    with torch.no_grad():
        for n_, p_ in m.named_parameters():
            if float(p_.abs().max()) == 0.0:
                p_.normal_(0, 0.02)
    m = m.to(dev)
    x = torch.randn(B, 16, 4, 32, 32, device=dev)
    t = torch.full((B,), 500, device=dev, dtype=torch.int64)
    for spec in ("", "2:5", "2:6", "2:4", "2:3", "2:16", "0:4,1:4", "0:6,1:6", "0:16,1:16", "0:2,1:2", ""):
        if spec:
            os.environ["LATTE_GROUP_M"] = spec
        else:
            os.environ.pop("LATTE_GROUP_M", None)
        m.profile_forward(x, t)
        pr = [m.profile_forward(x, t) for _ in range(3)]
        row = " ".join(f"{k}: {min(p[k][0] for p in pr) / max(pr[0][k][1], 1) * 1e3:6.1f}us" for k in
                       ("gemm_qkv", "gemm_proj", "gemm_fc1", "gemm_fc2"))
        log(f"group_m '{spec}': {row} | forward {min(sum(v_[0] for v_ in p.values()) for p in pr):.3f} ms")
    os.environ.pop("LATTE_GROUP_M", None)


def gemm_stagger():
    ms = _lib.c_f32()
    for (M, N, K, epi, v, nm) in [(32768, 4608, 1152, 1, 9, "fc1"), (32768, 3456, 1152, 0, 9, "qkv"), (32768, 1152, 4608, 2, 8, "fc2"),
                                  (32768, 1152, 1152, 2, 8, "proj"), (8192, 4608, 1152, 1, 8, "fc1 B=2")]:
        row = []
        for st in (0, 2, 4, 8):
            check(lib.latte_bench_gemm(M, N, K, epi, 0, v + 100 * st, 20, ctypes.byref(ms), stream_ptr()))
            row.append(f"st{st}: {ms.value*1e3:6.1f}us {2.0*M*N*K/(ms.value*1e-3)/1e12:5.0f}TF")
        log(f"stagger {nm} v{v}: " + " | ".join(row))


def xl_profile():
    from latte_amd.models import Latte_models
    import latte_amd
    kw = dict(input_size=32, num_frames=16, extras=1)
    for B in [int(v) for v in os.environ.get("LATTE_FL_B", "2,8").split(",")]:
        m = Latte_models["Latte-XL/2"](compute_dtype="bf16", max_batch=B, **kw)
        with torch.no_grad():
            for n_, p_ in m.named_parameters():
                if float(p_.abs().max()) == 0.0:
                    p_.normal_(0, 0.02)
        m = m.to(dev)
        x = torch.randn(B, 16, 4, 32, 32, device=dev)
        t = torch.full((B,), 500, device=dev, dtype=torch.int64)
        for variant in (0,):
            m.set_engine_option("gemm_variant", variant, B)
            m(x, t)
            prof = m.profile_forward(x, t)
            tot = sum(v[0] for v in prof.values())
            log(f"XL/2 B={B} variant={variant} profile (ms, launches): total {tot:.3f} ms")
            for k, v in prof.items():
                log(f"    {k:14s} {v[0]:8.3f} ms  {v[1]:4d}  {100*v[0]/tot:5.1f}%")
            d = latte_amd.create_diffusion("250")
            torch.cuda.synchronize()
            for steps in (10,):
                xx = x.clone()
                eng = m.engine(B)
                check(lib.latte_sample_loop(eng, d._h, 1, 0.0, 0, 1.0, ptr(xx), None, B, 249, 249 - 2, None, None, None, stream_ptr()))
                torch.cuda.synchronize()
                t0 = time.time()
                check(lib.latte_sample_loop(eng, d._h, 1, 0.0, 0, 1.0, ptr(xx), None, B, 246, 246 - steps + 1, None, None, None, stream_ptr()))
                torch.cuda.synchronize()
                dtm = time.time() - t0
                log(f"XL/2 B={B} variant={variant}: {steps} DDIM steps {dtm*1e3:.1f} ms -> {steps/dtm:.2f} it/s, {B*steps/dtm:.2f} sample-steps/s, "
                    f"MFMA frac {B*steps/dtm*3.726e12/2.5e15:.3f}; finite={bool(torch.isfinite(xx).all())}")
        del m
        torch.cuda.empty_cache()


def small_batch():
    """B = 1, 2 at XL/2: split-K gated GEMMs (engine option gated_split_k: 1 = off, 0 = rule, 2..4 forced) -- per-kernel times
    inside the forward and the 10-step DDIM loop."""
    import latte_amd
    from latte_amd.models import Latte_models
    for B in [int(v) for v in os.environ.get("LATTE_FL_B", "1,2").split(",")]:
        m = Latte_models["Latte-XL/2"](compute_dtype="bf16", max_batch=B, input_size=32, num_frames=16, extras=1)
        with torch.no_grad():
            for n_, p_ in m.named_parameters():
                if float(p_.abs().max()) == 0.0:
                    p_.normal_(0, 0.02)
        m = m.to(dev)
        x = torch.randn(B, 16, 4, 32, 32, device=dev)
        t = torch.full((B,), 500, device=dev, dtype=torch.int64)
        d = latte_amd.create_diffusion("250")
        eng = m.engine(B)
        for mode in (1, 0, 2, 1, 0):
            m.set_engine_option("gated_split_k", mode, B)
            m.profile_forward(x, t)
            pr = [m.profile_forward(x, t) for _ in range(3)]
            row = " ".join(f"{k[5:] if k.startswith('gemm_') else k}: {min(p[k][0] for p in pr) / max(pr[0][k][1], 1) * 1e3:5.1f}" for k in
                           ("gemm_proj", "gemm_fc2", "gemm_qkv", "gemm_fc1", "ln_modulate", "attn_spatial", "attn_temporal"))
            xx = x.clone()
            check(lib.latte_sample_loop(eng, d._h, 1, 0.0, 0, 1.0, ptr(xx), None, B, 249, 249 - 2, None, None, None, stream_ptr()))
            torch.cuda.synchronize()
            best = 1e9
            for _ in range(3):
                t0 = time.time()
                check(lib.latte_sample_loop(eng, d._h, 1, 0.0, 0, 1.0, ptr(xx), None, B, 246, 246 - 20 + 1, None, None, None, stream_ptr()))
                torch.cuda.synchronize()
                best = min(best, time.time() - t0)
            log(f"small_batch B={B} gated_split_k={mode} (us/launch): {row} | 20 DDIM steps {best*1e3:.1f} ms -> {B*20/best:.2f} sample-steps/s")
        m.set_engine_option("gated_split_k", 0, B)
        del m
        torch.cuda.empty_cache()


if __name__ == "__main__":
    which = sys.argv[1:] or ["env", "tr16_probe", "gemm_checks", "attention_checks", "ln_checks", "normal_check",
                             "gemm_bench", "xl_profile"]
    for w in which:
        section(globals()[w])
