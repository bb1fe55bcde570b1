"""Host logic of the GEMM launcher (csrc/gemm.hip: gemm_resolve_variant) -- which tile / kernel a linear of the Latte block runs
on when no variant is forced.  Pure host code behind the C-ABI debug entry latte_debug_gemm_choice: runs without a GPU.
Epilogues: 0 bias -> half (qkv), 1 bias + GELU -> half (fc1), 2 gated fp32 read-modify-write (out-projection, fc2; latte.py:179-180)."""
import pytest

from latte_amd._lib import load_library

TILE_N = {1: 128, 2: 128, 3: 256, 4: 128, 5: 192, 6: 256, 7: 128, 8: 192, 9: 256, 10: 192, 11: 192, 12: 144, 13: 144}
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
    assert fc1 in (9, 11) and qkv in (9, 11)


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
