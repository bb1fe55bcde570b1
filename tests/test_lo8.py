"""The FP8 correction pass of split operands (round 6; engine option guided_split bits 2 / 3): the pieces through
the C-ABI test hooks, each against the PyTorch op sequence it replaces.

A guided call carries the attention output and fc1's operand as  hi (f16) + lo8 (one byte: e4m3 of (value - hi) 2^12)  and the two
linears collect  hi . W^T + dec(lo8) 2^-12 . (dec(W8) 2^-6)^T  in ONE launch: the f16 K loop, then K / 128 steps of the block-scaled fp8
MFMA (v_mfma_scale_f32_16x16x128_f8f6f4; operand maps: tools/mx_probe.hip), then the epilogue.  The fp8 decode of the reference side is
torch's own float8_e4m3fn (the OCP format gfx950 implements).
"""
import pytest
import torch

from latte_amd._lib import check, ptr, stream_ptr

pytestmark = pytest.mark.gpu
F8 = torch.float8_e4m3fn
A_SHIFT, W_SHIFT = 12, 6          # csrc/common.h: LO8_A_SHIFT, LO8_W_SHIFT


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda")


def _codes(shape, g, dev):
    """random finite e4m3 codes (no 0x7f / 0xff = NaN)"""
    c = torch.randint(0, 256, shape, generator=g, dtype=torch.int32)
    c = torch.where((c & 0x7f) == 0x7f, c ^ 1, c)
    return c.to(torch.uint8).to(dev)


def test_pack_w8_is_round_to_nearest_even_e4m3(lib, dev):
    g = torch.Generator("cpu").manual_seed(1)
    w = torch.cat([torch.randn(4096, generator=g) * 0.03, torch.randn(1024, generator=g) * 3.0, torch.tensor([0.0, -0.0, 7.0, -7.0, 9.0, 1e-4, 6e-5, -2e-5])])
    w = w.to(torch.float16).to(dev)
    out = torch.zeros(w.numel(), dtype=torch.uint8, device=dev)
    check(lib.latte_debug_pack_w8(ptr(w), ptr(out), w.numel(), 1, stream_ptr()))
    torch.cuda.synchronize()
    want = (w.float() * 2.0 ** W_SHIFT).clamp(-448, 448).cpu().to(F8).view(torch.uint8)
    got = out.cpu()
    # +-0 may differ in the sign bit only
    same = (got == want) | (((got & 0x7f) == 0) & ((want & 0x7f) == 0))
    assert bool(same.all()), (got[~same][:8], want[~same][:8])


@pytest.mark.parametrize("D", [384, 1152])
def test_ln_modulate_fp8_remainder(lib, dev, D):
    M, rps = 520, 130
    g = torch.Generator("cpu").manual_seed(D)
    x = (torch.randn(M, D, generator=g) * 2 + 0.3).to(dev)
    mod = (torch.randn(M // rps, 2 * D, generator=g) * 0.3).to(dev)
    plain = torch.zeros(M, D, dtype=torch.float16, device=dev)
    check(lib.latte_debug_ln_modulate(ptr(x), ptr(plain), ptr(mod), ptr(mod[:, D:]), 2 * D, M, D, rps, None, 1, 1, 1, stream_ptr()))
    hi = torch.zeros(M, D, dtype=torch.float16, device=dev)
    lo8 = torch.zeros(M, D, dtype=torch.uint8, device=dev)
    check(lib.latte_debug_ln_modulate_split8(ptr(x), ptr(hi), ptr(lo8), ptr(mod), ptr(mod[:, D:]), 2 * D, M, D, rps, 1, stream_ptr()))
    torch.cuda.synchronize()
    assert torch.equal(hi.view(torch.int16), plain.view(torch.int16))          # the f16 half IS the plain kernel's output
    xd = x.double()
    ref = (xd - xd.mean(1, keepdim=True)) / (xd.var(1, unbiased=False, keepdim=True) + 1e-6).sqrt()
    smp = torch.arange(M, device=dev) // rps
    ref = ref * (1 + mod[smp, D:].double()) + mod[smp, :D].double()
    lo = lo8.view(F8).double() * 2.0 ** -A_SHIFT
    e_plain = float((plain.double() - ref).norm() / ref.norm())
    e_pair = float((hi.double() + lo - ref).norm() / ref.norm())
    print(D, e_plain, e_pair)
    # e4m3 keeps 4 significant bits of the remainder: the pair is >= 10 x closer than the f16 value alone (fp32 LayerNorm noise included)
    assert e_pair < 0.1 * e_plain and e_pair < 3e-5


@pytest.mark.parametrize("case", [(2, 16, 256, 16, 72), (1, 16, 256, 6, 64)])
@pytest.mark.parametrize("mode", [0, 1], ids=["spatial", "temporal"])
def test_fused_qkv_attention_fp8_remainder(lib, dev, case, mode):
    B, F, T, H, hd = case
    D, rows = H * hd, B * F * T
    g = torch.Generator("cpu").manual_seed(rows + hd + mode)
    xn = torch.randn(rows, D, generator=g).to(dev).to(torch.float16)
    W = (torch.randn(3 * D, D, generator=g) * D ** -0.5).to(dev).to(torch.float16)
    bias = (torch.randn(3 * D, generator=g) * 0.1).to(dev)
    flags = 3 if mode == 0 else 1
    plain = torch.zeros(rows, D, dtype=torch.float16, device=dev)
    pair = torch.zeros(rows, 2 * D, dtype=torch.float16, device=dev)
    check(lib.latte_debug_qkv_attention(ptr(xn), ptr(W), ptr(bias), ptr(plain), None, B, F, T, D, H, mode, flags, 1, stream_ptr()))
    check(lib.latte_debug_qkv_attention(ptr(xn), ptr(W), ptr(bias), ptr(pair), None, B, F, T, D, H, mode, flags | 256, 1, stream_ptr()))
    hi = torch.zeros(rows, D, dtype=torch.float16, device=dev)
    lo8 = torch.zeros(rows, D, dtype=torch.uint8, device=dev)
    check(lib.latte_debug_qkv_attention_split8(ptr(xn), ptr(W), ptr(bias), ptr(hi), ptr(lo8), B, F, T, D, H, mode, 1, stream_ptr()))
    torch.cuda.synchronize()
    assert torch.equal(hi.view(torch.int16), plain.view(torch.int16))
    # the f16 pair's lo is the SAME remainder at 11 bits: the fp8 code must be its e4m3 rounding (the remainder is computed identically)
    lo16 = pair[:, D:].float()
    want = (lo16 * 2.0 ** A_SHIFT).clamp(-448, 448)
    got = lo8.view(F8).float()
    # lo16 is itself rounded to f16 before we see it, the kernel converts the fp32 remainder: allow one e4m3 step where they disagree
    step = torch.maximum(want.abs() * 2.0 ** -3, torch.tensor(2.0 ** -9, device=dev))
    assert bool(((got - want).abs() <= step).all())
    rel = float((got - want).norm() / want.norm())
    print(case, mode, rel)
    assert rel < 0.05


@pytest.mark.parametrize("case", [(2, 4, 64, 6, 64), (1, 3, 200, 2, 72), (1, 2, 300, 2, 72), (2, 16, 1024, 2, 64), (2, 4, 16, 2, 64), (1, 3, 5, 3, 72)])
@pytest.mark.parametrize("mode", ["spatial", "temporal"])
def test_unfused_attention_fp8_remainder(lib, dev, case, mode):
    """The un-fused attention kernels (generic flash, 128 < L <= 256, L > 256 streaming, L <= 16) with the fp8-remainder output: the f16
    half is the plain call's output bit for bit, and f16 + remainder is >= 8 x closer to the fp32 attention on the same half q | k | v
    than the f16 output alone (what the out-projection's correction pass then removes)."""
    B, F, T, H, hd = case
    D, rows = H * hd, B * F * T
    g = torch.Generator("cpu").manual_seed(rows + hd)
    qkv = torch.randn(rows, 3 * D, generator=g).to(dev).to(torch.float16)
    args = (B * F, T, H, hd, F, F * T, T, 1) if mode == "spatial" else (B * T, F, H, hd, T, F * T, 1, T)
    L = T if mode == "spatial" else F
    plain = torch.zeros(rows, D, dtype=torch.float16, device=dev)
    check(lib.latte_debug_attention(ptr(qkv), ptr(plain), *args, 1, stream_ptr()))
    hi = torch.zeros(rows, D, dtype=torch.float16, device=dev)
    lo8 = torch.zeros(rows, D, dtype=torch.uint8, device=dev)
    check(lib.latte_debug_attention_split8(ptr(qkv), ptr(hi), ptr(lo8), *args, 1, stream_ptr()))
    torch.cuda.synchronize()
    assert torch.equal(hi.view(torch.int16), plain.view(torch.int16))
    q, k, v = qkv.float().reshape(B, F, T, 3, H, hd).unbind(3)
    if mode == "spatial":
        q, k, v = (t_.permute(0, 1, 3, 2, 4) for t_ in (q, k, v))
        ref = (torch.softmax((q @ k.transpose(-1, -2)) * hd ** -0.5, dim=-1) @ v).permute(0, 1, 3, 2, 4).reshape(rows, D)
    else:
        q, k, v = (t_.permute(0, 2, 3, 1, 4) for t_ in (q, k, v))
        ref = (torch.softmax((q @ k.transpose(-1, -2)) * hd ** -0.5, dim=-1) @ v).permute(0, 3, 1, 2, 4).reshape(rows, D)
    e_plain = float((plain.float() - ref).norm() / ref.norm())
    e_pair = float((hi.float() + lo8.view(F8).float() * 2.0 ** -A_SHIFT - ref).norm() / ref.norm())
    print(case, mode, L, e_plain, e_pair)
    # what is left with the remainder is the kernel's own P rounding (the probabilities feed the MFMA as f16), as for the f16 pair of round 5
    assert e_pair < 0.75 * e_plain and e_pair < 6e-4


LO_SHAPES = [(512, 384, 1152, 256), (300, 192, 256, 100), (256, 192, 128, 256), (11520, 1152, 1152, 256), (4096, 4608, 1152, 4096)]


@pytest.mark.parametrize("shape", LO_SHAPES)
@pytest.mark.parametrize("epi", [2, 1], ids=["gated_rmw", "gelu_half"])
def test_gemm_fp8_correction_pass(lib, dev, shape, epi):
    """hi . W^T + dec(A8) 2^-12 . (dec(W8) 2^-6)^T against fp64 on the same operands; with all-zero fp8 operands the launch must give
    the plain rolling kernel's result BIT FOR BIT (the f16 K loop and the epilogue are unchanged); more than one output tile per
    workgroup (11520 x 1152 = 270 tiles), a ragged last tile row, a single correction K tile (K = 128)."""
    M, N, K, rps = shape
    g = torch.Generator("cpu").manual_seed(M + N + K)
    Mp = (M + 255) // 256 * 256
    A = torch.randn(Mp, K, generator=g).to(dev).to(torch.float16)
    W = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev).to(torch.float16)
    A8, W8 = _codes((Mp, K), g, dev), _codes((N, K), g, dev)
    # magnitudes as in the engine: |lo 2^12| up to a few units, |w 2^6| up to a few units -- keep the random codes below 16
    A8 = torch.where((A8 & 0x7f) > 0x57, A8 & 0xbf, A8)
    W8 = torch.where((W8 & 0x7f) > 0x57, W8 & 0xbf, W8)
    bias = torch.randn(N, generator=g).to(dev)
    gate = torch.randn((M + rps - 1) // rps, 2 * N, generator=g).to(dev)
    hi = A.double()[:M] @ W.double().t()
    lo = (A8.view(F8).double()[:M] * 2.0 ** -A_SHIFT) @ (W8.view(F8).double() * 2.0 ** -W_SHIFT).t()
    smp = torch.arange(M, device=dev) // rps

    def run(a8, w8):
        if epi == 2:
            out = out0.clone()
        else:
            out = torch.zeros(Mp, N, dtype=torch.float16, device=dev)
        check(lib.latte_debug_gemm_lo8(ptr(A), ptr(W), ptr(a8), ptr(w8), ptr(bias), ptr(out), ptr(gate), M, N, K, 2 * N, rps, epi, 1, stream_ptr()))
        torch.cuda.synchronize()
        return out

    out0 = torch.randn(Mp, N, generator=g).to(dev)
    out = run(A8, W8)
    zero = run(torch.zeros_like(A8), torch.zeros_like(W8))
    if epi == 2:
        plain = out0.clone()
        check(lib.latte_debug_gemm(ptr(A), ptr(W), ptr(bias), ptr(plain), ptr(gate), M, N, K, 2 * N, rps, 2, 1, 11, stream_ptr()))
        want = out0.double()[:M] + gate.double()[smp, :N] * (hi + lo + bias.double())
        base = out0.double()[:M] + gate.double()[smp, :N] * (hi + bias.double())
        assert torch.equal(out[M:], out0[M:])
        tol = 3e-6
    else:
        plain = torch.zeros(Mp, N, dtype=torch.float16, device=dev)
        check(lib.latte_debug_gemm(ptr(A), ptr(W), ptr(bias), ptr(plain), ptr(gate), M, N, K, 2 * N, rps, 1, 1, 11, stream_ptr()))
        want = torch.nn.functional.gelu(hi + lo + bias.double(), approximate="tanh")
        base = torch.nn.functional.gelu(hi + bias.double(), approximate="tanh")
        tol = 6e-4       # half output
    torch.cuda.synchronize()
    assert torch.equal(zero, plain)
    e = float((out.double()[:M] - want).norm() / want.norm())
    moved = float((want - base).norm() / want.norm())
    print(shape, epi, "err", e, "the correction term is", moved, "of the result")
    assert e < tol
    if epi == 2:
        assert moved > 30 * e          # the term the pass adds is far above what the comparison resolves


def test_gemm_fp8_correction_pass_rejects_other_shapes(lib, dev):
    a = torch.zeros(256, 192, dtype=torch.float16, device=dev)
    a8 = torch.zeros(256, 192, dtype=torch.uint8, device=dev)
    o = torch.zeros(256, 192, device=dev)
    b = torch.zeros(192, device=dev)
    # K = 192 is not a multiple of 128; N = 128 has no whole 192-wide tile column; bf16 has no correction pass
    assert lib.latte_debug_gemm_lo8(ptr(a), ptr(a), ptr(a8), ptr(a8), ptr(b), ptr(o), ptr(o), 256, 192, 192, 192, 256, 2, 1, stream_ptr()) != 0
    assert lib.latte_debug_gemm_lo8(ptr(a), ptr(a), ptr(a8), ptr(a8), ptr(b), ptr(o), ptr(o), 256, 128, 128, 128, 256, 2, 1, stream_ptr()) != 0
    assert lib.latte_debug_gemm_lo8(ptr(a), ptr(a), ptr(a8), ptr(a8), ptr(b), ptr(o), ptr(o), 256, 192, 128, 192, 256, 2, 0, stream_ptr()) != 0
