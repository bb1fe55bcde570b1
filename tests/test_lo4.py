"""The FP4 form of the correction pass of split operands (round 6, second form): OCP e2m1 codes with one E8M0 scale
per ROW, multiplied by v_mfma_scale_f32_16x16x128_f8f6f4 (cbsz = blgp = 4: twice the fp8 rate) behind the f16 K loop of the same GEMM
launch.  Pieces through the C-ABI test hooks against a CPU emulation of the format: the quantiser (nearest-even onto the e2m1 grid,
scale = one binade below the smallest power of two that brings the row's largest magnitude to <= 6), the LayerNorm-modulate producer, the GEMM.
"""
import pytest
import torch

from latte_amd._lib import check, ptr, stream_ptr

pytestmark = pytest.mark.gpu
GRID = torch.tensor([0.0, 0.5, 1.0, 1.5, 2.0, 3.0, 4.0, 6.0])


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda")


def pitch(K):
    return (K + 255) // 256 * 128


def decode4(codes, scales, K):
    """codes [R, pitch] uint8 (two e2m1 codes per byte, low nibble first), scales [R] uint8 (E8M0) -> [R, K] float64"""
    c = codes.cpu().to(torch.int64)
    nib = torch.stack([c & 15, c >> 4], dim=2).reshape(c.shape[0], -1)[:, :K]
    val = GRID.double()[nib & 7] * torch.where((nib & 8) != 0, -1.0, 1.0)
    return val * torch.pow(2.0, scales.cpu().double() - 127.0)[:, None]


def quant4_ref(v):
    """CPU restatement: per row e = ceil(log2(amax / 6)) - 1 (the top binade saturates, everything else gains a bit), nearest grid value of
    v / 2^e, saturating at +-6"""
    v = v.double().cpu()
    amax = v.abs().amax(1)
    e = torch.ceil(torch.log2(torch.where(amax > 0, amax, torch.ones_like(amax)) / 6.0)) - 1
    s = torch.pow(2.0, e)[:, None]
    t = (v / s).clamp(-6, 6)
    d = (t.abs()[..., None] - GRID.double()).abs()
    idx = d.argmin(-1)
    return GRID.double()[idx] * torch.sign(t) * s, e


def test_pack_w4_quantiser(lib, dev):
    g = torch.Generator("cpu").manual_seed(3)
    N, K = 96, 1152
    w = (torch.randn(N, K, generator=g) * 0.03)
    w[5] *= 40.0
    w[7] = 0.0
    w = w.to(torch.float16).to(dev)
    c4 = torch.full((N, pitch(K)), 0xAA, dtype=torch.uint8, device=dev)
    sc = torch.zeros(N, dtype=torch.uint8, device=dev)
    check(lib.latte_debug_pack_w4(ptr(w), ptr(c4), ptr(sc), N, K, 1, stream_ptr()))
    torch.cuda.synchronize()
    got = decode4(c4, sc, K)
    want, e = quant4_ref(w.float())
    assert bool((c4.cpu()[:, K // 2:] == 0).all())                       # the padding to K % 256 == 0 is zero codes
    nz = w.float().abs().amax(1).cpu() > 0
    assert torch.equal((sc.cpu().double() - 127.0)[nz], e[nz])
    # ties between two grid values may go either way in the reference's argmin; everything else is the same value
    diff = (got - want).abs()
    step = torch.pow(2.0, e)[:, None]
    assert bool((diff <= 1.0 * step + 1e-30).all())
    assert float((diff > 0).double().mean()) < 0.02
    rel = float((got - w.float().cpu().double()).norm() / w.float().cpu().double().norm())
    print("W4 relative rms error", rel)
    assert rel < 0.2


@pytest.mark.parametrize("D", [384, 1152])
def test_ln_modulate_fp4_remainder(lib, dev, D):
    M, rps = 520, 130
    g = torch.Generator("cpu").manual_seed(D)
    x = (torch.randn(M, D, generator=g) * 2 + 0.3).to(dev)
    mod = (torch.randn(M // rps, 2 * D, generator=g) * 0.3).to(dev)
    plain = torch.zeros(M, D, dtype=torch.float16, device=dev)
    check(lib.latte_debug_ln_modulate(ptr(x), ptr(plain), ptr(mod), ptr(mod[:, D:]), 2 * D, M, D, rps, None, 1, 1, 1, stream_ptr()))
    hi = torch.zeros(M, D, dtype=torch.float16, device=dev)
    c4 = torch.zeros(M, pitch(D), dtype=torch.uint8, device=dev)
    sc = torch.zeros(M, dtype=torch.uint8, device=dev)
    check(lib.latte_debug_ln_modulate_split4(ptr(x), ptr(hi), ptr(c4), ptr(sc), ptr(mod), ptr(mod[:, D:]), 2 * D, M, D, rps, 1, stream_ptr()))
    torch.cuda.synchronize()
    assert torch.equal(hi.view(torch.int16), plain.view(torch.int16))          # the f16 half IS the plain kernel's output
    xd = x.double()
    ref = (xd - xd.mean(1, keepdim=True)) / (xd.var(1, unbiased=False, keepdim=True) + 1e-6).sqrt()
    smp = torch.arange(M, device=dev) // rps
    ref = ref * (1 + mod[smp, D:].double()) + mod[smp, :D].double()
    lo_true = (ref - hi.double()).cpu()
    lo4 = decode4(c4, sc, D)
    e_plain = float(lo_true.norm() / ref.cpu().norm())
    e_pair = float((lo_true - lo4).norm() / ref.cpu().norm())
    print(D, "f16 alone", e_plain, "f16 + fp4 remainder", e_pair)
    assert e_pair < 0.22 * e_plain       # amplitude: <= 5 % of the remainder's variance is left


LO_SHAPES = [(512, 384, 1152, 256), (300, 192, 256, 100), (256, 192, 128, 256), (11520, 1152, 1152, 256), (4096, 4608, 1152, 4096), (1024, 192, 320, 256)]


@pytest.mark.parametrize("shape", LO_SHAPES)
@pytest.mark.parametrize("epi", [1, 2])
def test_gemm_fp4_correction_pass(lib, dev, shape, epi):
    """out = epilogue(A W^T + dec(A4) dec(W4)^T + bias): the plain launch plus EXACTLY the decoded remainder product (fp32 accumulate),
    for K % 256 == 0 and K % 256 != 0 (zero-padded code rows), > 256 and < 256 tiles, partial last tile row."""
    M, N, K, rps = shape
    g = torch.Generator("cpu").manual_seed(M + N + K)
    Mp = (M + 255) // 256 * 256
    A = torch.randn(Mp, K, generator=g).to(dev).to(torch.float16)
    W = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev).to(torch.float16)
    bias = torch.randn(N, generator=g).to(dev)
    lo = (torch.randn(Mp, K, generator=g) * 2.0 ** -12 * A.float().abs().cpu()).to(dev)     # a remainder-sized operand
    # quantise lo and W through the library's own producers: W by pack_w4, lo by a CPU restatement written as codes
    W4 = torch.zeros(N, pitch(K), dtype=torch.uint8, device=dev)
    W4s = torch.zeros(N, dtype=torch.uint8, device=dev)
    check(lib.latte_debug_pack_w4(ptr(W), ptr(W4), ptr(W4s), N, K, 1, stream_ptr()))
    loq, e = quant4_ref(lo.float())
    t = (loq / torch.pow(2.0, e)[:, None])
    code = (t.abs()[..., None] - GRID.double()).abs().argmin(-1) | torch.where(t < 0, 8, 0)
    code = torch.nn.functional.pad(code, (0, pitch(K) * 2 - K))
    A4 = (code[:, 0::2] | (code[:, 1::2] << 4)).to(torch.uint8).to(dev)
    A4s = (e + 127).clamp(0, 254).to(torch.uint8).to(dev)
    torch.cuda.synchronize()
    corr = decode4(A4, A4s, K)[:M].float().to(dev) @ decode4(W4, W4s, K).float().to(dev).t()
    base = A.float()[:M] @ W.float().t() + bias
    gate = torch.randn((M + rps - 1) // rps, N, generator=g).to(dev)
    if epi == 2:
        out0 = torch.randn(Mp, N, generator=g).to(dev)
        gr = gate[torch.arange(M, device=dev) // rps]
        want, want_plain = out0[:M] + gr * (base + corr), out0[:M] + gr * base
        out = out0.clone()
        check(lib.latte_debug_gemm_lo4(ptr(A), ptr(W), ptr(A4), ptr(A4s), ptr(W4), ptr(W4s), ptr(bias), ptr(out), ptr(gate), M, N, K, N, rps, 2, 1,
                                       stream_ptr()))
        torch.cuda.synchronize()
        got = out[:M]
        assert torch.equal(out[M:], out0[M:])
    else:
        want = torch.nn.functional.gelu(base + corr, approximate="tanh")
        want_plain = torch.nn.functional.gelu(base, approximate="tanh")
        out = torch.zeros(Mp, N, dtype=torch.float16, device=dev)
        check(lib.latte_debug_gemm_lo4(ptr(A), ptr(W), ptr(A4), ptr(A4s), ptr(W4), ptr(W4s), ptr(bias), ptr(out), None, M, N, K, 0, M, 1, 1,
                                       stream_ptr()))
        torch.cuda.synchronize()
        got = out[:M].float()
    err = float((got - want).norm() / want.norm())
    moved = float((want - want_plain).norm() / want.norm())
    print(shape, epi, "err", err, "size of the correction", moved)
    assert err < (2e-5 if epi == 2 else 4e-4)
    if epi == 2:
        # the correction itself is reproduced to fp32 accumulation accuracy
        assert float(((got - want_plain) - (want - want_plain)).norm() / (want - want_plain).norm()) < 1e-2   # (fp32 rounding of the O(1) outputs is 1e-3 of this O(1e-4) term)
