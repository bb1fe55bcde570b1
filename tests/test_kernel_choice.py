"""Host logic of the GEMM launcher (csrc/gemm.hip: gemm_resolve_variant) -- which tile / kernel a linear of the Latte block runs
on when no variant is forced.  Pure host code behind the C-ABI debug entry latte_debug_gemm_choice: runs without a GPU.
Epilogues: 0 bias -> half (qkv), 1 bias + GELU -> half (fc1), 2 gated fp32 read-modify-write (out-projection, fc2; latte.py:179-180)."""
import pytest

from latte_amd._lib import load_library

TILE_N = {1: 128, 2: 128, 3: 256, 4: 128, 5: 192, 6: 256, 7: 128, 8: 192, 9: 256, 10: 192, 11: 192, 12: 144, 13: 144, 18: 144, 19: 144}
ROWS_PER_VIDEO = 16 * 256          # Latte-XL/2 at 256 px: 16 frames x 256 tokens


@pytest.mark.parametrize("B", [1, 2, 4, 8, 16])
def test_xl2_block_choices(B):
    lib = load_library()
    M, D = B * ROWS_PER_VIDEO, 1152
    proj = lib.latte_debug_gemm_choice(M, D, D, 2)
    fc2 = lib.latte_debug_gemm_choice(M, D, 4 * D, 2)
    fc1 = lib.latte_debug_gemm_choice(M, 4 * D, D, 1)
    qkv = lib.latte_debug_gemm_choice(M, 3 * D, D, 0)
    if B == 1:
        # 96 tiles of 256 x 192 for 256 CUs: the 128 x 144 tile (32 x 8 = 256 tiles, one per CU) takes the gated GEMMs
        assert proj == 13 and fc2 == 13
    else:
        # the 12-wave producer / consumer kernel on 256 x 192 tiles
        assert proj == 11 and fc2 == 11
    if B >= 8:
        assert fc1 == 9 and qkv == 9        # persistent ping-pong kernel, 256 x 256 tiles (full-line epilogue for half outputs)
    assert fc1 in (9, 11) and qkv in (9, 11)     # (the 256 x 144 tile of round 6, variants 18 / 19, ties them and is not picked)


@pytest.mark.parametrize("shape", [(1024, 384, 384), (1024, 1536, 384), (20480, 768, 3072), (4096, 1024, 1024), (8292, 2432, 192),
                                   (300, 288, 64), (64, 128, 64), (4096, 1152, 64), (33000, 1152, 1152), (700, 384, 64)])
@pytest.mark.parametrize("epi", [0, 1, 2, 3])
def test_choice_is_launchable(shape, epi):
    """Whatever is picked must be able to run the shape: whole tile columns (whole wave widths for the persistent kernels 7-9),
    the 12-wave kernel only with >= 2 K tiles, the small tile only for the gated epilogue with at most one tile per CU."""
    M, N, K = shape
    v = load_library().latte_debug_gemm_choice(M, N, K, epi)
    assert v in TILE_N
    width = TILE_N[v] // 4 if 7 <= v <= 9 else TILE_N[v]
    assert N % width == 0, (v, N)
    if v in (10, 11):
        assert K >= 128 and N % 192 == 0
    if v in (12, 13):
        assert epi == 2 and ((M + 127) // 128) * (N // 144) <= 256
    if v in (18, 19):
        assert K >= 128 and ((M + 255) // 256) * (N // 144) <= 512


def test_fused_qkv_attention_shape_rule():
    """csrc/qkv_attn.hip: which attention blocks run as ONE projection + attention kernel (latte.py:48-70 without the qkv round
    trip through HBM): 256 tokens per frame (spatial) / 16 frames with a multiple of 16 tokens (temporal), head_dim 64 | 72."""
    fus = load_library().latte_debug_qkv_attention_fusable
    rows = 8 * ROWS_PER_VIDEO
    assert fus(1152, 16, 16, 256, 0, rows) == 1 and fus(1152, 16, 16, 256, 1, rows) == 1      # Latte-XL/2 at 256 px, both block kinds
    assert fus(1152, 16, 16, 1024, 0, rows) == 0 and fus(1152, 16, 16, 1024, 1, rows) == 1    # Latte-1 at 512 px: temporal blocks only
    assert fus(768, 12, 16, 256, 0, rows) == 1 and fus(384, 6, 16, 256, 1, rows) == 1         # B/2, S/2 (head_dim 64)
    assert fus(384, 6, 4, 64, 0, 256) == 0 and fus(384, 6, 4, 64, 1, 256) == 0                # 64 tokens, 4 frames: separate kernels
    assert fus(1152, 12, 16, 256, 0, rows) == 0                                                 # head_dim 96
    assert fus(1152, 16, 16, 256, 0, 2 ** 21) == 0                                              # operand beyond the 4 GiB offset range


@pytest.mark.parametrize("shape", [(20480, 768, 768), (20480, 3072, 768), (20480, 768, 3072), (128, 384, 384), (64, 1152, 1152),
                                   (8192, 1152, 4608), (100, 384, 384)])
def test_weight_gradient_split_plan(shape):
    """csrc/gemm_tn.hip: the contraction of dW = dY^T X is cut into whole multiples of 64 rows, the cuts cover M exactly once,
    and the partial products fill the chip without exceeding the trainer's workspace rule (splits * N * K floats)."""
    import ctypes
    M, N, K = shape
    rows = ctypes.c_int(0)
    splits = load_library().latte_debug_gemm_tn_plan(M, N, K, ctypes.byref(rows))
    assert splits >= 1 and rows.value % 64 == 0 and rows.value > 0
    assert (splits - 1) * rows.value < M <= splits * rows.value
    assert splits * ((N + 255) // 256) * ((K + 255) // 256) <= 768


def test_debug_choice_offers_only_implementations_of_the_same_function(lib):
    """Round-3 advisor finding: the launchers read LATTE_* environment variables per launch, among them ablation variants with
    garbage results.  The overrides are an explicit debug entry now (include/latte_amd_debug.h); the product library refuses the
    ablation values (attention variants 7-9, the removed block kernel's 4) and unknown names."""
    for name, v in (("attn_variant", 4), ("attn_variant", 7), ("attn_variant", 9), ("no_such_choice", 1), ("tn_kernel", 3), ("conv_kernel", 5)):
        assert lib.latte_debug_set_choice(name.encode(), v) != 0, (name, v)
    for name, v in (("attn_variant", 1), ("attn_variant", 5), ("xattn_flash", 1), ("tn_kernel", 4), ("tn_wn", 4), ("attn_bwd_tiles", 1), ("attn_bwd_tiles", 2),
                    ("conv_kernel", 1), ("conv_kernel", 2), ("conv_kernel", 3), ("conv_kernel", 4)):
        assert lib.latte_debug_set_choice(name.encode(), v) == 0, (name, v)
        assert lib.latte_debug_set_choice(name.encode(), 0) == 0


def test_no_kernel_reads_the_environment_in_the_product_build():
    """No getenv outside the measurement build (LATTE_GEMM_ABLATE) in the kernels' launchers."""
    import os
    import re
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "latte_amd", "csrc")
    for fn in sorted(os.listdir(root)):
        src = open(os.path.join(root, fn)).read()
        # drop the #ifdef LATTE_GEMM_ABLATE ... #endif regions, then look for getenv
        kept = re.sub(r"#ifdef LATTE_GEMM_ABLATE.*?#endif", "", src, flags=re.S)
        assert "getenv" not in kept, fn
