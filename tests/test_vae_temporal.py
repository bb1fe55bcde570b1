"""AutoencoderKLTemporalDecoder (BASELINE config 4's decoder; sample_t2x.py:31-32, pipeline_latte.py:779-798): oracle
self-checks on CPU -- parity with real diffusers is UNPINNED (oracle/vae_temporal_oracle.py) -- and the engine against the
oracle on the GPU."""
import pytest
import torch

from _util import rel_l2
from latte_amd._lib import check
from oracle import vae_oracle as vo
from oracle import vae_temporal_oracle as vt

TOL = 1e-3   # relative L2 on the decoded frames against the fp32 restatement (f16 MFMA operands, fp32 residual stream)


def test_temporal_oracle_keys_and_structure():
    ks = vt.decoder_keys()
    assert "post_quant_conv.weight" not in ks
    assert ks["decoder.up_blocks.2.resnets.0.spatial_res_block.conv_shortcut.weight"] == (256, 512, 1, 1)
    assert ks["decoder.up_blocks.2.resnets.0.temporal_res_block.conv1.weight"] == (256, 256, 3, 1, 1)
    assert ks["decoder.up_blocks.2.resnets.0.temporal_res_block.norm1.weight"] == (256,)      # the temporal block works on out_channels
    assert ks["decoder.mid_block.resnets.1.time_mixer.mix_factor"] == (1,)
    assert ks["decoder.time_conv_out.weight"] == (3, 3, 3, 1, 1)
    from latte_amd.random_init import vae_temporal_decoder_keys
    assert vae_temporal_decoder_keys() == ks


def test_temporal_oracle_reduces_to_the_spatial_decoder_when_the_temporal_paths_are_closed():
    """mix_factor -> -inf closes every temporal branch (AlphaBlender: out = x_spatial + sigmoid(mix) * branch) and an identity
    time_conv_out leaves conv_out's frames: the temporal decoder then IS the per-frame SD-VAE decoder (without post_quant_conv)
    -- ties the restatement to oracle/vae_oracle.py's blocks."""
    sd = vt.init_state_dict(seed=1, mix=-40.0)
    w = torch.zeros(3, 3, 3, 1, 1)
    for c in range(3):
        w[c, c, 1, 0, 0] = 1.0
    sd["decoder.time_conv_out.weight"], sd["decoder.time_conv_out.bias"] = w, torch.zeros(3)
    spatial = {k.replace("spatial_res_block.", ""): v for k, v in sd.items() if "temporal_res_block" not in k and "time_" not in k}
    spatial["post_quant_conv.weight"] = torch.eye(4).view(4, 4, 1, 1)
    spatial["post_quant_conv.bias"] = torch.zeros(4)
    z = torch.randn(3, 4, 8, 8, generator=torch.Generator().manual_seed(0))
    assert rel_l2(vt.decode(sd, z, num_frames=3), vo.decode(spatial, z)) < 1e-6
    # and with open temporal paths the frames of a chunk interact, different chunks do not
    sd2 = vt.init_state_dict(seed=1)
    a = vt.decode(sd2, z, num_frames=3)
    z2 = z.clone()
    z2[2] += 1.0
    b = vt.decode(sd2, z2, num_frames=3)
    assert rel_l2(b[0], a[0]) > 1e-4                              # frame 0 sees the change of frame 2 (GroupNorm over the chunk)
    c = vt.decode(sd2, torch.cat([z, z2]), num_frames=3)
    assert rel_l2(c[:3], a) < 2e-5 and rel_l2(c[3:], b) < 2e-5            # (fp32 batching noise)


def test_host_shim_refuses_without_gpu_and_filters_keys():
    from latte_amd import AutoencoderKLTemporalDecoder
    vae = AutoencoderKLTemporalDecoder()
    vae.load_state_dict({**vt.init_state_dict(0), "encoder.conv_in.weight": torch.zeros(1), "quant_conv.bias": torch.zeros(1)})
    assert set(vae.state_dict()) == set(vt.decoder_keys())
    with pytest.raises(Exception):
        vae.decode(torch.zeros(2, 4, 16, 16), num_frames=2)      # not on a GPU: must raise, never fall back


# ------------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
@pytest.mark.parametrize("frames,latent", [(3, 16), (14, 16), (2, 32)])
def test_temporal_decoder_vs_oracle(lib, frames, latent):
    from latte_amd import AutoencoderKLTemporalDecoder
    sd = vt.init_state_dict(seed=5)
    z = torch.randn(frames, 4, latent, latent, generator=torch.Generator().manual_seed(frames))
    want = vt.decode(sd, z, num_frames=frames)
    vae = AutoencoderKLTemporalDecoder(latent_size=latent, max_frames=frames)
    vae.load_state_dict(sd)
    vae.to("cuda")
    got = vae.decode(z.cuda(), num_frames=frames).sample
    torch.cuda.synchronize()
    assert got.shape == want.shape and torch.isfinite(got).all()
    err = rel_l2(got, want)
    print(f"temporal decoder {frames} x {latent}x{latent} rel-L2 vs oracle: {err:.3e}")
    assert err < TOL


@pytest.mark.gpu
def test_temporal_decoder_chunks_and_pipeline_path(lib):
    """Two chunks in one call = each chunk alone (pipeline_latte.py:785-793 decodes 14 + 2 frames of a 16-frame video)."""
    from latte_amd import AutoencoderKLTemporalDecoder
    sd = vt.init_state_dict(seed=2)
    z = torch.randn(6, 4, 16, 16, generator=torch.Generator().manual_seed(9)).cuda()
    vae = AutoencoderKLTemporalDecoder(latent_size=16, max_frames=3)
    vae.load_state_dict(sd)
    vae.to("cuda")
    both = vae.decode(z, num_frames=3).sample
    assert torch.equal(both[:3], vae.decode(z[:3], num_frames=3).sample)
    assert torch.equal(both[3:], vae.decode(z[3:], num_frames=3).sample)
    import latte_amd
    from types import SimpleNamespace
    lat = torch.randn(1, 4, 16, 16, 16, generator=torch.Generator().manual_seed(1)).cuda() * 0.18215
    vid = latte_amd.LattePipeline.decode_latents_with_temporal_decoder(SimpleNamespace(vae=vae), lat)   # 14 + 2 frames
    assert vid.shape == (1, 16, 128, 128, 3) and vid.dtype == torch.uint8


SPLIT_MASKS = [0x3ff, 0x0, 0x319c03, 0x31bc63, 0x318c03, 0x301c00, 0xffffff]


@pytest.mark.gpu
def test_temporal_decoder_split_mask_sweep(lib):
    """Which split-operand passes the 1e-3 bar needs (round 6; csrc/vae_engine.cpp: vae_split_mask): the decode error against the fp32
    restatement for stage masks of the spatial residual pass (bits 0..4: mid block, up blocks 0..3) and of the three-pass temporal
    convolutions (bits 5..9), over four (weights, latent) draws, with the time of one full-size 14-frame chunk beside it.  RECORDED in
    gpurun_out/vae_split_sweep.json (-> profiles/); asserted: the library's default mask holds the bar on every draw."""
    import json
    import os
    import time
    from _util import ROOT
    from latte_amd import AutoencoderKLTemporalDecoder
    from latte_amd.random_init import vae_temporal_decoder_state_dict
    draws = [(5, 3, 16), (5, 14, 16), (5, 2, 32), (6, 4, 16), (7, 2, 32)]
    table = {f"{m:#08x}": {} for m in SPLIT_MASKS}
    table["default"] = {}
    try:
        for seed, frames, latent in draws:
            sd = vt.init_state_dict(seed=seed)
            z = torch.randn(frames, 4, latent, latent, generator=torch.Generator().manual_seed(frames + seed))
            want = vt.decode(sd, z, num_frames=frames)
            vae = AutoencoderKLTemporalDecoder(latent_size=latent, max_frames=frames)
            vae.load_state_dict(sd)
            vae.to("cuda")
            for m in SPLIT_MASKS + [None]:
                check(lib.latte_debug_set_choice(b"vae_split", 0 if m is None else (1 << 24) | m))
                got = vae.decode(z.cuda(), num_frames=frames).sample
                torch.cuda.synchronize()
                table["default" if m is None else f"{m:#08x}"][f"seed{seed}_{frames}x{latent}"] = rel_l2(got, want)
            del vae
        big = AutoencoderKLTemporalDecoder(latent_size=64, max_frames=14)
        big.load_state_dict(vae_temporal_decoder_state_dict(0))
        big.to("cuda")
        zb = torch.randn(14, 4, 64, 64, generator=torch.Generator().manual_seed(1)).cuda()
        for m in SPLIT_MASKS + [None]:
            check(lib.latte_debug_set_choice(b"vae_split", 0 if m is None else (1 << 24) | m))
            big.decode(zb, num_frames=14)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(2):
                big.decode(zb, num_frames=14)
            torch.cuda.synchronize()
            table["default" if m is None else f"{m:#08x}"]["ms_per_14_frame_chunk_64x64"] = round((time.perf_counter() - t0) / 2 * 1e3, 2)
    finally:
        check(lib.latte_debug_set_choice(b"vae_split", 0))
    for k, row in table.items():
        errs = [v for kk, v in row.items() if kk.startswith("seed")]
        row["max"] = max(errs)
        print(k, f"max {row['max']:.3e}", " ".join(f"{e:.2e}" for e in errs), row["ms_per_14_frame_chunk_64x64"], "ms")
    assert table["default"]["max"] < TOL
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "vae_split_sweep.json"), "w") as f:
        json.dump(table, f, indent=1)
