"""Every HIP kernel against the PyTorch op sequence it replaces (through the C-ABI test hooks)."""
import pytest
import torch

from latte_amd._lib import check, ptr, stream_ptr

pytestmark = pytest.mark.gpu
TD = {0: torch.bfloat16, 1: torch.float16}
# half-precision operand rounding: unit roundoff 2^-9 (bf16) / 2^-11 (f16)
OUT_TOL = {0: 6e-3, 1: 1e-3}


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda")


@pytest.fixture
def kernel_choice(lib):
    """latte_debug_set_choice for the duration of a test (include/latte_amd_debug.h: another implementation of the same function)."""
    used = []

    def choose(name, value):
        check(lib.latte_debug_set_choice(name.encode(), int(value)))
        used.append(name)
    yield choose
    for name in used:
        check(lib.latte_debug_set_choice(name.encode(), 0))


@pytest.mark.parametrize("dt", [0, 1])
@pytest.mark.parametrize("variant", [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 18, 19])
@pytest.mark.parametrize("shape", [(256, 256, 128), (512, 768, 1152), (300, 512, 256), (1024, 1152, 4608), (700, 384, 64),
                                   (8292, 2304, 192), (8292, 2432, 192),  # > 256 tiles, odd K-tile count (persistent kernel)
                                   (300, 288, 64), (130, 144, 128), (4096, 1152, 1152)])  # 144-wide tiles: 1 / 2 K tiles, B = 1 shape
def test_gemm_epilogues(lib, dev, dt, variant, shape):
    M, N, K = shape
    tile_n = {0: 64, 1: 128, 2: 128, 3: 256, 4: 128, 5: 192, 6: 256, 7: 32, 8: 48, 9: 64, 10: 192, 11: 192, 12: 144, 13: 144, 18: 144, 19: 144}[variant]   # 7-9: wave width
    if N % tile_n:
        pytest.skip(f"tile width {tile_n} does not divide N")
    if 7 <= variant <= 11 and K < 128 and (variant >= 10 or N % (4 * tile_n)):
        pytest.skip("single K tile: the persistent kernel defers to the plain one, which needs whole tile columns")
    g = torch.Generator("cpu").manual_seed(M + N + K)
    Mp = (M + 255) // 256 * 256
    A = torch.randn(Mp, K, generator=g).to(dev).to(TD[dt])
    W = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev).to(TD[dt])
    bias = torch.randn(N, generator=g).to(dev)
    ref = A.float()[:M] @ W.float().t() + bias            # same rounded operands, fp32 accumulate
    rps = 64
    gate = torch.randn((M + rps - 1) // rps, 2 * N, generator=g).to(dev)
    for epi in (0, 1, 2, 3):
        if epi in (0, 1):
            out = torch.zeros(Mp, N, dtype=TD[dt], device=dev)
        else:
            out = torch.randn(Mp, N, generator=g).to(dev)
        out0 = out.clone()
        check(lib.latte_debug_gemm(ptr(A), ptr(W), ptr(bias), ptr(out), ptr(gate), M, N, K, 2 * N, rps, epi, dt,
                                   variant, stream_ptr()))
        torch.cuda.synchronize()
        if epi == 0:
            want, tol = ref, OUT_TOL[dt]
        elif epi == 1:
            want, tol = torch.nn.functional.gelu(ref, approximate="tanh"), OUT_TOL[dt]
        elif epi == 2:
            gi = torch.arange(M, device=dev) // rps
            want, tol = out0[:M] + gate[gi, :N] * ref, 2e-5
        else:
            want, tol = ref, 2e-5
        got = out[:M].float()
        rel = float((got - want).norm() / want.norm())
        assert rel < tol, (epi, rel)
        if M < Mp:
            assert torch.equal(out[M:], out0[M:]), "rows beyond M must not be written"


@pytest.mark.parametrize("dt", [0, 1])
def test_gemm_gated_residual_call_site_tags(lib, dev, dt):
    """The two gated-residual call sites (attention out-projection, fc2) run separate instantiations of the same kernel
    (distinct symbols in a kernel trace): bit-identical results, exact fp32 read-modify-write."""
    M, N, K, rps = 4096, 1152, 1152, 4096
    g = torch.Generator("cpu").manual_seed(5)
    A = torch.randn(M, K, generator=g).to(dev).to(TD[dt])
    W = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev).to(TD[dt])
    bias = torch.randn(N, generator=g).to(dev)
    gate = torch.randn(M // rps, 2 * N, generator=g).to(dev)
    out0 = torch.randn(M, N, generator=g).to(dev)
    want = out0 + gate[torch.arange(M, device=dev) // rps, :N] * (A.float() @ W.float().t() + bias)
    outs = []
    for tag in (0, 1):
        out = out0.clone()
        check(lib.latte_debug_gemm(ptr(A), ptr(W), ptr(bias), ptr(out), ptr(gate), M, N, K, 2 * N, rps, 2, dt,
                                   1000 * tag, stream_ptr()))
        torch.cuda.synchronize()
        assert float((out - want).norm() / want.norm()) < 2e-5
        outs.append(out)
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("dt", [0, 1])
@pytest.mark.parametrize("shape", [(4096, 1152, 1152, 4096), (8192, 1152, 4608, 4096), (33000, 1152, 1152, 256), (2304, 384, 256, 256)])
def test_gemm_producer_wave_kernel(lib, dev, dt, shape):
    """Variants 10 / 11 (12 waves: 8 MFMA waves + 4 DMA waves, gemm_pw.hip; two-segment and rolling schedule) on the gated
    read-modify-write epilogue's fast path (whole tiles inside one sample), > 256 tiles, a partial last tile row; bit-identical
    to the 8-wave kernel of the same tile (same MFMA order per output element)."""
    M, N, K, rps = shape
    g = torch.Generator("cpu").manual_seed(M + K)
    Mp = (M + 255) // 256 * 256
    A = torch.randn(Mp, K, generator=g).to(dev).to(TD[dt])
    W = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev).to(TD[dt])
    bias = torch.randn(N, generator=g).to(dev)
    gate = torch.randn((M + rps - 1) // rps, 2 * N, generator=g).to(dev)
    out0 = torch.randn(Mp, N, generator=g).to(dev)
    want = out0[:M] + gate[torch.arange(M, device=dev) // rps, :N] * (A.float()[:M] @ W.float().t() + bias)
    outs = {}
    for variant in (10, 1010, 11, 1011, 8):
        out = out0.clone()
        check(lib.latte_debug_gemm(ptr(A), ptr(W), ptr(bias), ptr(out), ptr(gate), M, N, K, 2 * N, rps, 2, dt, variant,
                                   stream_ptr()))
        torch.cuda.synchronize()
        assert float((out[:M] - want).norm() / want.norm()) < 2e-5, variant
        assert torch.equal(out[M:], out0[M:])
        outs[variant] = out
    assert all(torch.equal(outs[v], outs[8]) for v in (10, 1010, 11, 1011))


@pytest.mark.parametrize("dt", [0, 1])
@pytest.mark.parametrize("shape", [(4096, 3072, 768), (20480, 1536, 384), (33000, 384, 1536), (300, 192, 128)])
def test_gemm_gelu_training_epilogues(lib, dev, dt, shape):
    """Round 6: the MLP's GELU passes inside the rolling 12-wave GEMM (gemm_pw.hip; train_engine.cpp uses them for fc1's forward and fc2's
    input gradient).  epi 13: out = the plain launch's half output bit for bit, aux = gelu_tanh of THAT half (the separate pass's result up
    to the last half bit of the fast exp2 / rcp forms on both sides: the same expressions -> equal); epi 14: (A W^T) * gelu'(u) against
    fp32 torch on the same half operands.  Partial last tile rows, > 256 and < 256 tiles."""
    M, N, K = shape
    g = torch.Generator("cpu").manual_seed(M + N)
    Mp = (M + 255) // 256 * 256
    A = torch.randn(Mp, K, generator=g).to(dev).to(TD[dt])
    W = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev).to(TD[dt])
    bias = torch.randn(N, generator=g).to(dev)
    plain = torch.zeros(Mp, N, device=dev, dtype=TD[dt])
    check(lib.latte_debug_gemm(ptr(A), ptr(W), ptr(bias), ptr(plain), None, M, N, K, 0, M, 0, dt, 11, stream_ptr()))
    u = torch.full((Mp, N), 7.0, device=dev, dtype=TD[dt])
    h = torch.full((Mp, N), 7.0, device=dev, dtype=TD[dt])
    check(lib.latte_debug_gemm_gelu(ptr(A), ptr(W), ptr(bias), ptr(u), ptr(h), M, N, K, 13, dt, stream_ptr()))
    torch.cuda.synchronize()
    assert torch.equal(u[:M], plain[:M])
    assert bool((u[M:] == 7.0).all()) and bool((h[M:] == 7.0).all())          # rows beyond M are never written
    want_h = torch.nn.functional.gelu(u[:M].float(), approximate="tanh")
    assert float((h[:M].float() - want_h).norm() / want_h.norm()) < (3e-3 if dt == 0 else 4e-4)
    # the in-engine separate pass on the same u: identical halves
    sep = torch.nn.functional.gelu(u[:M].float(), approximate="tanh").to(TD[dt])
    assert float((h[:M].float() - sep.float()).abs().max()) <= float(sep.float().abs().max()) * (2 ** -7 if dt == 0 else 2 ** -10)
    # epi 14: du = (A W^T + bias) * gelu'(u)
    du = torch.full((Mp, N), 7.0, device=dev, dtype=TD[dt])
    check(lib.latte_debug_gemm_gelu(ptr(A), ptr(W), ptr(bias), ptr(du), ptr(u), M, N, K, 14, dt, stream_ptr()))
    torch.cuda.synchronize()
    uf = u[:M].float().requires_grad_(True)
    torch.nn.functional.gelu(uf, approximate="tanh").sum().backward()
    want = (A.float()[:M] @ W.float().t() + bias) * uf.grad
    assert float((du[:M].float() - want).norm() / want.norm()) < (4e-3 if dt == 0 else 5e-4)
    assert bool((du[M:] == 7.0).all())


CASES = [(1, 4, 16, 2, 64), (2, 16, 256, 16, 72), (1, 4, 64, 6, 64), (1, 3, 100, 2, 72), (1, 16, 1024, 6, 64),
         (2, 4, 4, 2, 64), (1, 2, 200, 2, 72), (1, 3, 144, 3, 64), (2, 2, 256, 4, 64), (1, 2, 129, 1, 72),
         # L > 256: the 256-key-block kernel with the online softmax (whole blocks, ragged last block, partial query block)
         (1, 2, 1024, 2, 72), (1, 1, 300, 2, 72), (2, 1, 700, 4, 64), (1, 1, 513, 1, 72), (1, 8, 512, 3, 64)]


@pytest.mark.parametrize("dt", [0, 1])
@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("mode", ["spatial", "temporal"])
def test_attention(lib, dev, dt, case, mode):
    _attention_case(lib, dev, dt, case, mode)


@pytest.mark.parametrize("dt", [0, 1])
@pytest.mark.parametrize("case", [c for c in CASES if c[2] > 128])
@pytest.mark.parametrize("variant", [1, 5, 12, 13])
def test_attention_long_sequence_kernels_forced(lib, dev, dt, case, variant, kernel_choice):
    """The spatial cases with more than 128 tokens through the kernels the default choice does not take for them: 1 = the generic
    flash kernel, 5 = the streaming kernel (also for 128 < L <= 256, where the single-block kernel is the default), 12 / 13 = the
    round-6c forms of the streaming kernel on the 32 x 32 x 16 MFMA shape for head dim 72 and L > 256 (12: 4 waves x
    64 queries, one wave per SIMD, software-pipelined, lazy reference maximum; 13 = that pipeline on 8 waves x 32 queries) -- other shapes take
    the default kernel under them."""
    kernel_choice("attn_variant", variant)
    _attention_case(lib, dev, dt, case, "spatial")


def test_debug_choice_refuses_what_is_not_offered(lib):
    """Round-3 advisor finding: a stray environment variable could route production launches to ablation kernels with garbage
    results.  The overrides are an explicit debug entry now, and the production library refuses the ablation values outright."""
    for name, v in (("attn_variant", 4), ("attn_variant", 7), ("attn_variant", 9), ("no_such_choice", 1), ("tn_kernel", 3)):
        assert lib.latte_debug_set_choice(name.encode(), v) != 0, (name, v)
    assert lib.latte_debug_set_choice(b"attn_variant", 5) == 0 and lib.latte_debug_set_choice(b"attn_variant", 0) == 0


def _attention_case(lib, dev, dt, case, mode):
    B, F, T, H, hd = case
    D, rows = H * hd, B * F * T
    g = torch.Generator("cpu").manual_seed(rows + hd)
    qh = torch.randn(rows, 3 * D, generator=g).to(dev).to(TD[dt])
    q5 = qh.float().view(B, F, T, 3, H, hd)
    if mode == "spatial":
        q, k, v = [q5[:, :, :, i].permute(0, 1, 3, 2, 4) for i in range(3)]
        a = torch.softmax((q @ k.transpose(-1, -2)) * hd ** -0.5, dim=-1) @ v
        want = a.permute(0, 1, 3, 2, 4).reshape(rows, D)
        args = (B * F, T, H, hd, F, F * T, T, 1)
    else:
        q, k, v = [q5[:, :, :, i].permute(0, 2, 3, 1, 4) for i in range(3)]
        a = torch.softmax((q @ k.transpose(-1, -2)) * hd ** -0.5, dim=-1) @ v
        want = a.permute(0, 3, 1, 2, 4).reshape(rows, D)
        args = (B * T, F, H, hd, T, F * T, 1, T)
    out = torch.zeros(rows, D, dtype=TD[dt], device=dev)
    check(lib.latte_debug_attention(ptr(qh), ptr(out), *args, dt, stream_ptr()))
    torch.cuda.synchronize()
    rel = float((out.float() - want).norm() / want.norm())
    assert rel < (8e-3 if dt == 0 else 1.5e-3), rel


def test_attention_forced_rescale(lib, dev):
    """Online-softmax rescale branch: one key dominates late in the sequence (guide §5.4 rule 26)."""
    B, F, T, H, hd, dt = 1, 1, 256, 1, 64, 1
    g = torch.Generator("cpu").manual_seed(7)
    qkv = torch.randn(T, 3 * hd, generator=g)
    qkv[200, hd:2 * hd] = qkv[5, :hd] * 6.0          # key 200 (4th tile) spikes against query 5
    qh = qkv.to(dev).to(TD[dt])
    q, k, v = qh.float()[:, :hd], qh.float()[:, hd:2 * hd], qh.float()[:, 2 * hd:]
    want = torch.softmax((q @ k.t()).double() * hd ** -0.5, dim=-1).float() @ v
    out = torch.zeros(T, hd, dtype=TD[dt], device=dev)
    torch.cuda.synchronize()
    check(lib.latte_debug_attention(ptr(qh), ptr(out), 1, T, 1, hd, 1, T, T, 1, dt, stream_ptr()))
    torch.cuda.synchronize()
    assert float((out.float() - want).norm() / want.norm()) < 2e-3


@pytest.mark.parametrize("hd,spike_key", [(72, 900), (64, 300), (72, 1023)])
@pytest.mark.parametrize("kernel", ["stream", "flash", "stream64", "stream64_8w"])
def test_attention_blocks_forced_rescale(lib, dev, hd, spike_key, kernel, kernel_choice):
    """The online-softmax kernels for L > 256 (L = 1024: the streaming kernel with its ring of 128-key blocks, and the generic
    64-key-tile flash kernel it falls back to): a key in a LATE block dominates one query, so the running maximum jumps and
    the accumulated output / sum of the earlier blocks must be rescaled (guide section 5.4 rule 26); fp64 reference."""
    if kernel == "flash":
        kernel_choice("attn_variant", 1)
    if kernel in ("stream64", "stream64_8w"):   # (stream64 raises its reference maximum lazily: the spikes are what forces the raise)
        if hd != 72:
            pytest.skip("the 32 x 32 x 16 forms are instantiated for head dim 72")
        kernel_choice("attn_variant", {"stream64": 12, "stream64_8w": 13}[kernel])
    T, dt = 1024, 1
    g = torch.Generator("cpu").manual_seed(spike_key)
    qkv = torch.randn(T, 3 * hd, generator=g)
    qkv[spike_key, hd:2 * hd] = qkv[37, :hd] * 6.0        # key spike_key spikes against query 37
    qkv[10, hd:2 * hd] = qkv[700, :hd] * 5.0              # and an EARLY key dominates a query of a later query block
    qh = qkv.to(dev).to(TD[dt])
    q, k, v = qh.float()[:, :hd], qh.float()[:, hd:2 * hd], qh.float()[:, 2 * hd:]
    want = (torch.softmax((q.double() @ k.double().t()) * hd ** -0.5, dim=-1) @ v.double()).float()
    out = torch.zeros(T, hd, dtype=TD[dt], device=dev)
    check(lib.latte_debug_attention(ptr(qh), ptr(out), 1, T, 1, hd, 1, T, T, 1, dt, stream_ptr()))
    torch.cuda.synchronize()
    assert float((out.float() - want).norm() / want.norm()) < 2e-3
    assert float((out.float()[37] - want[37]).norm() / want[37].norm()) < 4e-3


@pytest.mark.parametrize("dt", [0, 1])
@pytest.mark.parametrize("D", [128, 384, 768, 1024, 1152])
@pytest.mark.parametrize("use_te", [False, True])
def test_ln_modulate(lib, dev, dt, D, use_te):
    B, F, T = 2, 4, 16
    M = B * F * T
    g = torch.Generator("cpu").manual_seed(D)
    x = (torch.randn(M, D, generator=g) * 3 + 0.5).to(dev)
    mod = torch.randn(B, 6 * D, generator=g).to(dev)
    te = torch.randn(F, D, generator=g).to(dev)
    xin = x.clone()
    y = torch.zeros(M, D, dtype=TD[dt], device=dev)
    check(lib.latte_debug_ln_modulate(ptr(xin), ptr(y), ptr(mod), ptr(mod[:, D:]), 6 * D, M, D, F * T,
                                      ptr(te) if use_te else None, T, F, dt, stream_ptr()))
    torch.cuda.synchronize()
    xr = (x.view(B, F, T, D) + te.view(1, F, 1, D)).view(M, D) if use_te else x
    s = torch.arange(M, device=dev) // (F * T)
    want = torch.nn.functional.layer_norm(xr, (D,), eps=1e-6) * (1 + mod[s, D:2 * D]) + mod[s, :D]
    assert float((y.float() - want).norm() / want.norm()) < (3e-3 if dt == 0 else 4e-4)
    assert torch.allclose(xin, xr, rtol=0, atol=1e-6)


def test_fill_normal_moments(lib, dev):
    n = 1 << 22
    out = torch.empty(n, device=dev)
    check(lib.latte_debug_fill_normal(ptr(out), n, 123, 0, stream_ptr()))
    torch.cuda.synchronize()
    assert torch.isfinite(out).all()
    assert abs(float(out.mean())) < 3e-3 and abs(float(out.std()) - 1) < 3e-3
    assert abs(float((out ** 4).mean()) - 3.0) < 0.05
    out2 = torch.empty(n, device=dev)
    check(lib.latte_debug_fill_normal(ptr(out2), n, 123, 0, stream_ptr()))
    assert torch.equal(out, out2)                      # counter-based: reproducible


@pytest.mark.parametrize("dt", [0, 1])
def test_gemm_bias_residual_half_epilogue(lib, dev, dt):
    """epi 5 (VAE attention out-projection): out(half) = A W^T + bias + res(half), in place."""
    M, N, K = 512, 512, 512
    g = torch.Generator("cpu").manual_seed(11)
    A = torch.randn(M, K, generator=g).to(dev).to(TD[dt])
    W = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev).to(TD[dt])
    bias = torch.randn(N, generator=g).to(dev)
    res = torch.randn(M, N, generator=g).to(dev).to(TD[dt])
    want = A.float() @ W.float().t() + bias + res.float()
    out = res.clone()
    check(lib.latte_debug_gemm(ptr(A), ptr(W), ptr(bias), ptr(out), ptr(out), M, N, K, 0, M, 5, dt, 1, stream_ptr()))
    torch.cuda.synchronize()
    assert float((out.float() - want).norm() / want.norm()) < OUT_TOL[dt]


# (B, F, T, heads, hd): spatial needs T == 256, temporal F == 16; more than 256 units (persistent walk), fewer than 256, a
# sequence-group count that is not a multiple of 8 (plain unit order), hd 64 (192 columns per head) and 72 (216 -> padded 224)
FUSED_CASES = [(2, 16, 256, 16, 72), (1, 3, 256, 6, 64), (1, 16, 256, 8, 72), (3, 16, 64, 4, 64), (1, 16, 16, 8, 72),
               (1, 16, 256, 16, 64)]   # Latte-L width: 16 heads of 64 (the four-heads-per-XCD unit order at hd = 64)


@pytest.mark.parametrize("dt", [0, 1])
@pytest.mark.parametrize("case", FUSED_CASES)
@pytest.mark.parametrize("mode", [0, 1], ids=["spatial", "temporal"])
def test_fused_qkv_attention_is_the_unfused_pair(lib, dev, dt, case, mode):
    """csrc/qkv_attn.hip (QKV projection + attention in one kernel, q / k / v only in LDS) against the two kernels it replaces:
    the in-LDS q | k | v must be the qkv GEMM's output bit for bit (same K order, same rounding), the attention output must be the
    stand-alone attention kernel's bit for bit, and both must match the fp32 torch reference of latte.py:50-70."""
    B, F, T, H, hd = case
    if (mode == 0 and T != 256) or (mode == 1 and F != 16):
        pytest.skip("shape belongs to the other mode")
    D, rows = H * hd, B * F * T
    rows_pad = (rows + 255) // 256 * 256
    g = torch.Generator("cpu").manual_seed(rows + hd + mode)
    xn = torch.zeros(rows_pad, D, dtype=TD[dt], device=dev)
    xn[:rows] = torch.randn(rows, D, generator=g).to(dev).to(TD[dt])
    W = (torch.randn(3 * D, D, generator=g) * D ** -0.5).to(dev).to(TD[dt])
    bias = (torch.randn(3 * D, generator=g) * 0.1).to(dev)
    qkv = torch.zeros(rows_pad, 3 * D, dtype=TD[dt], device=dev)
    check(lib.latte_debug_gemm(ptr(xn), ptr(W), ptr(bias), ptr(qkv), None, rows, 3 * D, D, 0, F * T, 0, dt, 0, stream_ptr()))
    want = torch.zeros(rows, D, dtype=TD[dt], device=dev)
    args = (B * F, T, H, hd, F, F * T, T, 1) if mode == 0 else (B * T, F, H, hd, T, F * T, 1, T)
    check(lib.latte_debug_attention(ptr(qkv), ptr(want), *args, dt, stream_ptr()))
    out = torch.full((rows, D), float("nan"), dtype=TD[dt], device=dev)
    dbg = torch.full((rows, 3 * D), float("nan"), dtype=TD[dt], device=dev)
    for flags in ((0, 0, 1, 2, 3, 4, 7) if mode == 0 else (0, 0, 1, 4, 5)):   # bit 2: the four-heads-per-XCD unit order   # flags 0 twice: same result with warm LDS / caches (stale-image screen)
        check(lib.latte_debug_qkv_attention(ptr(xn), ptr(W), ptr(bias), ptr(out), ptr(dbg), B, F, T, D, H, mode, flags, dt, stream_ptr()))
        torch.cuda.synchronize()
        assert torch.equal(dbg.view(torch.int16), qkv[:rows].view(torch.int16)), f"flags {flags}: in-LDS q | k | v != qkv GEMM output"
        assert torch.equal(out.view(torch.int16), want.view(torch.int16)), f"flags {flags}: fused attention != stand-alone attention kernel"
        out.fill_(float("nan"))
        dbg.fill_(float("nan"))
    check(lib.latte_debug_qkv_attention(ptr(xn), ptr(W), ptr(bias), ptr(out), None, B, F, T, D, H, mode, 0, dt, stream_ptr()))
    torch.cuda.synchronize()
    # fp32 reference of latte.py:50-70 on the half q | k | v
    q5 = (xn[:rows].float() @ W.float().t() + bias).to(TD[dt]).float().view(B, F, T, 3, H, hd)
    if mode == 0:
        q, k, v = [q5[:, :, :, i].permute(0, 1, 3, 2, 4) for i in range(3)]
        ref = (torch.softmax((q @ k.transpose(-1, -2)) * hd ** -0.5, dim=-1) @ v).permute(0, 1, 3, 2, 4).reshape(rows, D)
    else:
        q, k, v = [q5[:, :, :, i].permute(0, 2, 3, 1, 4) for i in range(3)]
        ref = (torch.softmax((q @ k.transpose(-1, -2)) * hd ** -0.5, dim=-1) @ v).permute(0, 3, 1, 2, 4).reshape(rows, D)
    rel = float((out.float() - ref).norm() / ref.norm())
    assert rel < (8e-3 if dt == 0 else 1.5e-3), rel


@pytest.mark.parametrize("dt", [0, 1])
@pytest.mark.parametrize("case", [(2, 16, 256, 16, 72), (1, 16, 256, 6, 64)])
@pytest.mark.parametrize("mode", [0, 1], ids=["spatial", "temporal"])
def test_fused_qkv_attention_split_output(lib, dev, dt, case, mode):
    """Guided calls (engine option guided_split, round 5) take the attention output as a split pair [hi | lo]: hi must be the plain
    kernel's output bit for bit (the half nearest to the value), and hi + lo must carry the fp32 attention output to ~2^-20 -- i.e.
    the out-projection's K-concatenated operand [hi | lo] . [W | W]^T no longer rounds its activation."""
    B, F, T, H, hd = case
    D, rows = H * hd, B * F * T
    g = torch.Generator("cpu").manual_seed(rows + hd + mode)
    xn = torch.randn(rows, D, generator=g).to(dev).to(TD[dt])
    W = (torch.randn(3 * D, D, generator=g) * D ** -0.5).to(dev).to(TD[dt])
    bias = (torch.randn(3 * D, generator=g) * 0.1).to(dev)
    plain = torch.full((rows, D), float("nan"), dtype=TD[dt], device=dev)
    dbg = torch.zeros(rows, 3 * D, dtype=TD[dt], device=dev)
    flags = 3 if mode == 0 else 1
    check(lib.latte_debug_qkv_attention(ptr(xn), ptr(W), ptr(bias), ptr(plain), ptr(dbg), B, F, T, D, H, mode, flags, dt, stream_ptr()))
    pair = torch.full((rows, 2 * D), float("nan"), dtype=TD[dt], device=dev)
    check(lib.latte_debug_qkv_attention(ptr(xn), ptr(W), ptr(bias), ptr(pair), None, B, F, T, D, H, mode, flags | 256, dt, stream_ptr()))
    torch.cuda.synchronize()
    assert torch.equal(pair[:, :D].view(torch.int16), plain.view(torch.int16))
    # fp32 attention on the half q | k | v the kernel holds (latte.py:61-70)
    q, k, v = dbg.float().reshape(B, F, T, 3, H, hd).unbind(3)
    if mode == 0:
        q, k, v = (t_.permute(0, 1, 3, 2, 4) for t_ in (q, k, v))          # [B, F, H, T, hd]
        ref = (torch.softmax((q @ k.transpose(-1, -2)) * hd ** -0.5, dim=-1) @ v).permute(0, 1, 3, 2, 4).reshape(rows, D)
    else:
        q, k, v = (t_.permute(0, 2, 3, 1, 4) for t_ in (q, k, v))          # [B, T, H, F, hd]
        ref = (torch.softmax((q @ k.transpose(-1, -2)) * hd ** -0.5, dim=-1) @ v).permute(0, 3, 1, 2, 4).reshape(rows, D)
    e_plain = float((plain.float() - ref).norm() / ref.norm())
    e_pair = float((pair[:, :D].float() + pair[:, D:].float() - ref).norm() / ref.norm())
    print(dt, case, mode, e_plain, e_pair)
    # the pair removes the OUTPUT rounding; what is left is the softmax probabilities' half rounding inside the kernel (P feeds the MFMA)
    assert e_pair < 0.75 * e_plain and e_pair < (4e-3 if dt == 0 else 6e-4)


def test_fused_qkv_attention_rejects_other_shapes(lib, dev):
    x = torch.zeros(256, 128, dtype=torch.bfloat16, device=dev)
    w = torch.zeros(384, 128, dtype=torch.bfloat16, device=dev)
    b = torch.zeros(384, device=dev)
    o = torch.zeros(256, 128, dtype=torch.bfloat16, device=dev)
    assert lib.latte_debug_qkv_attention(ptr(x), ptr(w), ptr(b), ptr(o), None, 1, 4, 64, 128, 2, 0, 0, 0, stream_ptr()) != 0   # T != 256
    assert lib.latte_debug_qkv_attention(ptr(x), ptr(w), ptr(b), ptr(o), None, 1, 4, 64, 128, 2, 1, 0, 0, stream_ptr()) != 0   # F != 16


# (M, N, K): the Latte-B/2 linears at a short contraction, XL width (1152 = 4.5 tiles: half-tile edges in n and k), a contraction
# that is not a multiple of 64 and an N that is not a multiple of 128 (both fall back to the 4-wave kernel)
TN_CASES = [(1024, 768, 768), (2048, 2304, 768), (1536, 768, 3072), (1280, 1152, 1152), (640, 3456, 1152), (1000, 768, 768),
            (1024, 200, 256)]


@pytest.mark.parametrize("dt", [0, 1])
@pytest.mark.parametrize("shape", TN_CASES)
def test_weight_gradient_gemm_tn(lib, dev, dt, shape, kernel_choice):
    """dW = dY^T X (csrc/gemm_tn.hip: the 8-wave LDS-DMA kernel where the shape allows it, else the 4-wave one) against fp32
    torch on the same half operands, and the two kernels against each other (debug choice tn_kernel = 4 forces the 4-wave kernel)."""
    M, N, K = shape
    g = torch.Generator("cpu").manual_seed(M + N + K)
    dY = torch.randn(M, N, generator=g).to(dev).to(TD[dt])
    X = torch.randn(M, K, generator=g).to(dev).to(TD[dt])
    ws = torch.empty(64 * 1024 * 1024, device=dev)
    want = dY.float().t() @ X.float()
    outs = []
    for force4 in (False, True):
        if force4:
            kernel_choice("tn_kernel", 4)
        dW = torch.full((N, K), float("nan"), device=dev)
        check(lib.latte_debug_gemm_tn(ptr(dY), ptr(X), ptr(dW), ptr(ws), ws.numel(), M, N, K, dt, stream_ptr()))
        torch.cuda.synchronize()
        assert float((dW - want).norm() / want.norm()) < 2e-5, force4       # fp32 accumulation of exact half products
        outs.append(dW)
    assert float((outs[0] - outs[1]).abs().max()) <= 1e-3 * float(want.abs().max())


@pytest.mark.parametrize("dt", [0, 1])
@pytest.mark.parametrize("shape", [(1024, 768, 768), (2048, 2304, 768), (1536, 3072, 768), (640, 3456, 1152), (1280, 1152, 1152), (256, 128, 128)])
def test_weight_gradient_gemm_tn_with_bias_column_sums(lib, dev, dt, shape):
    """Round 6b: the linear's bias gradient (column sums of dY) on the weight-gradient launch -- wave 0 of each group of the k-tile-0
    workgroups adds the dY fragments it holds with v_dot2c against (1, 1).  The column sums against fp32 torch on the same half
    operand; the weight gradient BIT-IDENTICAL to the launch without them (same MFMA stream); shapes with half-tile edges in n
    (1152 = 4.5 tiles, 3456 = 13.5) and a single-tile case.  A shape the 8-wave kernel does not take is refused."""
    M, N, K = shape
    g = torch.Generator("cpu").manual_seed(M + N + K + 1)
    dY = torch.randn(M, N, generator=g).to(dev).to(TD[dt])
    X = torch.randn(M, K, generator=g).to(dev).to(TD[dt])
    ws = torch.empty(64 * 1024 * 1024, device=dev)
    dW0 = torch.full((N, K), float("nan"), device=dev)
    check(lib.latte_debug_gemm_tn(ptr(dY), ptr(X), ptr(dW0), ptr(ws), ws.numel(), M, N, K, dt, stream_ptr()))
    dW1 = torch.full((N, K), float("nan"), device=dev)
    cs = torch.full((N,), float("nan"), device=dev)
    check(lib.latte_debug_gemm_tn_colsum(ptr(dY), ptr(X), ptr(dW1), ptr(cs), ptr(ws), ws.numel(), M, N, K, dt, stream_ptr()))
    torch.cuda.synchronize()
    assert torch.equal(dW0, dW1)
    want = dY.double().sum(0)
    assert float((cs.double() - want).norm() / want.norm()) < 2e-6
    assert lib.latte_debug_gemm_tn_colsum(ptr(dY), ptr(X), ptr(dW1), ptr(cs), ptr(ws), ws.numel(), 1000, N, K, dt, stream_ptr()) != 0


# (B, F, T, heads, hd): temporal sequences of 16 / 8 / 5 frames (the one-wave-per-problem kernel, ragged L), spatial 64 and 256 tokens
ATTN_BWD_CASES = [(2, 16, 32, 4, 64), (1, 8, 24, 2, 72), (3, 5, 16, 3, 64), (1, 2, 64, 2, 72), (1, 2, 256, 2, 64), (1, 2, 200, 2, 72),
                  (1, 3, 128, 2, 64), (2, 2, 256, 3, 72)]


@pytest.mark.parametrize("dt", [0, 1])
@pytest.mark.parametrize("case", ATTN_BWD_CASES)
@pytest.mark.parametrize("mode", ["spatial", "temporal"])
def test_attention_backward(lib, dev, dt, case, mode, kernel_choice):
    """dq, dk, dv of the attention core (csrc/train_attn.hip) against torch autograd on the same half q / k / v / dout, for the
    strided sequence layouts of both block kinds; where L <= 16 the one-wave kernel AND the tile passes (forced) are checked, where 64 < L <= 256 the resident-image
    kernels AND the tile passes."""
    B, F, T, H, hd = case
    D, rows = H * hd, B * F * T
    g = torch.Generator("cpu").manual_seed(rows + hd)
    qkv = (torch.randn(rows, 3 * D, generator=g) * 0.7).to(dev).to(TD[dt])
    dout = torch.randn(rows, D, generator=g).to(dev).to(TD[dt])
    q5 = qkv.float().view(B, F, T, 3, H, hd).detach().requires_grad_(True)
    if mode == "spatial":
        q, k, v = [q5[:, :, :, i].permute(0, 1, 3, 2, 4) for i in range(3)]
        o = (torch.softmax((q @ k.transpose(-1, -2)) * hd ** -0.5, dim=-1) @ v).permute(0, 1, 3, 2, 4).reshape(rows, D)
        args, L = (B * F, T, H, hd, F, F * T, T, 1), T
    else:
        q, k, v = [q5[:, :, :, i].permute(0, 2, 3, 1, 4) for i in range(3)]
        o = (torch.softmax((q @ k.transpose(-1, -2)) * hd ** -0.5, dim=-1) @ v).permute(0, 3, 1, 2, 4).reshape(rows, D)
        args, L = (B * T, F, H, hd, T, F * T, 1, T), F
    o.backward(dout.float())
    want = q5.grad.reshape(rows, 3 * D)
    oh = o.detach().to(TD[dt])
    stats = torch.zeros(args[0] * H * L * 3 + 16, device=dev)
    # L <= 16: the one-wave kernel and the tile passes (forced, 1); 64 < L <= 256: the resident-image kernels (round 6b) and the
    # tile passes (forced, 2) -- the same MFMA products in the same order per own row
    outs = []
    for force in ([0, 1] if L <= 16 else [0, 2] if 64 < L <= 256 else [0]):
        if force:
            kernel_choice("attn_bwd_tiles", force)
        got = torch.full((rows, 3 * D), float("nan"), dtype=TD[dt], device=dev)
        check(lib.latte_debug_attention_bwd(ptr(qkv), ptr(oh), ptr(dout), ptr(got), ptr(stats), *args, dt, stream_ptr()))
        torch.cuda.synchronize()
        rel = float((got.float() - want).norm() / want.norm())
        assert rel < (1.2e-2 if dt == 0 else 2e-3), (force, rel)
        outs.append(got)
    if 64 < L <= 256:   # same MFMA products; the elementwise chain between them is a shorter form of the same arithmetic
        d = float((outs[0].float() - outs[1].float()).norm() / outs[1].float().norm())
        assert d < (8e-3 if dt == 0 else 1e-3), d
