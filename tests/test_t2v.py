"""LatteT2V denoiser on the HIP engine vs the oracle restatement of
/root/reference/models/latte_t2v.py (pinned against the reference's own file; the diffusers leaves are memory-derived, see
oracle/latte_t2v_oracle.py).  Tolerance: the sampling path's 1e-3 relative L2."""
import json
import os

import numpy as np
import pytest
import torch

import latte_amd
from _util import GOLDEN, rel_l2

TOL = 1e-3


def _fixture():
    from oracle import latte_t2v_oracle as to
    z = np.load(os.path.join(GOLDEN, "tiny_t2v.npz"))
    cfg = to.T2VConfig(**json.loads(bytes(z["cfg_json"]).decode()))
    sd = {k[4:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd::")}
    return cfg, sd, z


def _model(cfg, sd, cd, **kw):
    m = latte_amd.LatteT2V(num_attention_heads=cfg.num_attention_heads, attention_head_dim=cfg.attention_head_dim,
                           in_channels=cfg.in_channels, out_channels=cfg.out_channels, num_layers=cfg.num_layers,
                           sample_size=cfg.sample_size, patch_size=cfg.patch_size, cross_attention_dim=cfg.cross_attention_dim,
                           caption_channels=cfg.caption_channels, video_length=cfg.video_length,
                           **({} if cd is None else {"compute_dtype": cd}), **kw)
    m.load_state_dict(sd)
    return m


def test_t2v_shim_surface_on_cpu():
    cfg, sd, z = _fixture()
    m = _model(cfg, sd, "f16")
    assert set(m.state_dict()) == set(sd)
    with pytest.raises(latte_amd.LatteError):
        latte_amd.LatteT2V(norm_type="layer_norm")
    with pytest.raises(latte_amd.LatteError):      # f16 operands only (the reference's own type, sample_t2x.py:29)
        _model(cfg, sd, "bf16")
    with pytest.raises(latte_amd.LatteError):
        m.to(torch.bfloat16)
    if not torch.cuda.is_available():
        with pytest.raises(latte_amd.LatteError):
            m(torch.from_numpy(z["x"]), torch.from_numpy(z["t"]), torch.from_numpy(z["encoder_hidden_states"]))


@pytest.mark.gpu
@pytest.mark.parametrize("cd", [None, "f16"])
def test_t2v_forward_matches_reference_fixture(cd):
    cfg, sd, z = _fixture()
    m = _model(cfg, sd, cd).to("cuda")
    x, t = torch.from_numpy(z["x"]), torch.from_numpy(z["t"])
    enc, mask = torch.from_numpy(z["encoder_hidden_states"]), torch.from_numpy(z["encoder_attention_mask"])
    out = m(x.cuda(), timestep=t.cuda(), encoder_hidden_states=enc.cuda(), encoder_attention_mask=mask.cuda(),
            added_cond_kwargs={"resolution": None, "aspect_ratio": None}, return_dict=False)[0]
    assert out.shape == (x.shape[0], cfg.out_channels, *x.shape[2:])
    assert rel_l2(out, torch.from_numpy(z["forward"])) < TOL
    out = m(x.cuda(), t.cuda(), enc.cuda(), enable_temporal_attentions=False).sample
    assert rel_l2(out, torch.from_numpy(z["forward_spatial_only"])) < TOL
    # a padded caption token is ignored
    enc2 = enc.clone()
    enc2[-1, -1] += 3.0
    a = m(x.cuda(), t.cuda(), enc.cuda(), encoder_attention_mask=mask.cuda()).sample
    b = m(x.cuda(), t.cuda(), enc2.cuda(), encoder_attention_mask=mask.cuda()).sample
    assert rel_l2(b[-1], a[-1]) < 1e-5


T2V_CASES = [
    # (heads, hd, layers, sample_size, frames, text tokens, caption channels, batch): Latte-1 width at reduced depth
    (16, 72, 1, 16, 16, 20, 128, 1),     # T = 64: flash self-attention, temporal L = 16
    (16, 72, 1, 32, 4, 120, 256, 2),     # T = 256: the full-sequence self-attention kernel, 120 text tokens (two key tiles)
    (4, 64, 2, 8, 8, 7, 64, 2),          # hd = 64, ragged text length
    (4, 64, 1, 32, 2, 33, 64, 2),        # T = 256, hd = 64, 33 text tokens: the whole-panel cross-attention kernel at hd = 64
    (16, 72, 1, 24, 2, 50, 64, 1),       # T = 144: a partial 256-query block in the whole-panel cross-attention kernel
]


@pytest.mark.gpu
@pytest.mark.parametrize("case", T2V_CASES, ids=lambda c: "h%dx%d_L%d_s%d_f%d_k%d" % c[:6])
def test_t2v_forward_matches_oracle(case, monkeypatch):
    from oracle import latte_t2v_oracle as to
    heads, hd, layers, ss, fr, lk, cc, B = case
    cfg = to.T2VConfig(num_attention_heads=heads, attention_head_dim=hd, num_layers=layers, sample_size=ss,
                       cross_attention_dim=heads * hd, caption_channels=cc, video_length=fr)
    sd = to.init_state_dict(cfg, seed=5)
    g = torch.Generator("cpu").manual_seed(6)
    x = torch.randn(B, 4, fr, ss, ss, generator=g)
    t = torch.tensor([981, 44][:B])
    enc = torch.randn(B, lk, cc, generator=g)
    mask = torch.ones(B, lk)
    mask[0, lk - max(lk // 3, 1):] = 0
    with torch.no_grad():
        want = to.latte_t2v_forward(sd, cfg, x, t, enc, mask)
    m = _model(cfg, sd, "f16", max_batch=B).to("cuda")
    got = m(x.cuda(), t.cuda(), enc.cuda(), encoder_attention_mask=mask.cuda()).sample
    assert rel_l2(got, want) < TOL
    # attn1 of the blocks with 256-token / 16-frame sequences ran as the fused projection + attention kernel (csrc/qkv_attn.hip):
    # the separate qkv GEMM + attention kernels give the same bits
    m.set_engine_option("fuse_qkv_attn", 0)
    assert torch.equal(m(x.cuda(), t.cuda(), enc.cuda(), encoder_attention_mask=mask.cuda()).sample, got)
    m.set_engine_option("fuse_qkv_attn", 3)
    # text cross-attention: sequences of >= 128 tokens with <= 128 text tokens run the whole-panel kernel (attn_cross_kernel);
    # the generic flash kernel agrees with it to rounding (exact softmax against two key tiles with an online rescale)
    from latte_amd._lib import check, load_library
    check(load_library().latte_debug_set_choice(b"xattn_flash", 1))
    try:
        flash = m(x.cuda(), t.cuda(), enc.cuda(), encoder_attention_mask=mask.cuda()).sample
    finally:
        check(load_library().latte_debug_set_choice(b"xattn_flash", 0))
    assert rel_l2(flash, want) < TOL and rel_l2(flash, got) < 2e-4
    # default operand type of LatteT2V = f16, the type the reference runs this transformer in (sample_t2x.py:29)
    md = _model(cfg, sd, None, max_batch=B).to("cuda")
    assert md.compute_dtype == "f16"
    assert torch.equal(md(x.cuda(), t.cuda(), enc.cuda(), encoder_attention_mask=mask.cuda()).sample, got)


def test_pipeline_surface_on_cpu():
    """LattePipeline keeps the reference's constructor / call surface (pipeline_latte.py:100-115,516-542)."""
    from latte_amd.schedulers import DDIMScheduler
    with pytest.raises(latte_amd.LatteError):
        latte_amd.LattePipeline()
    cfg, sd, z = _fixture()
    pipe = latte_amd.LattePipeline(transformer=_model(cfg, sd, "f16"), scheduler=DDIMScheduler())
    assert pipe.vae_scale_factor == 8
    with pytest.raises(ValueError):
        pipe()
    with pytest.raises(latte_amd.LatteError):
        pipe(prompt="a cat")                                  # no tokenizer / text encoder given
    with pytest.raises(latte_amd.LatteError):
        pipe(prompt_embeds=torch.zeros(1, 6, cfg.caption_channels), negative_prompt_embeds=torch.zeros(1, 6, cfg.caption_channels),
             enable_vae_temporal_decoder=True)
    emb = torch.arange(2 * 5 * 3, dtype=torch.float32).reshape(2, 1, 5, 3)
    mask = torch.tensor([[1, 1, 1, 0, 0], [1, 1, 0, 0, 0]], dtype=torch.float32)
    out, keep = pipe.mask_text_embeddings(emb, mask)          # batch > 1: zeroed, not cut (pipeline_latte.py:122-125)
    assert keep == 5 and float(out[1, 0, 2:].abs().sum()) == 0.0
    out, keep = pipe.mask_text_embeddings(emb[:1], mask[:1])  # batch 1: cut to the kept tokens
    assert keep == 3 and out.shape == (1, 1, 3, 3)
    s = DDIMScheduler()
    s.set_timesteps(50)
    assert s.timesteps[:3].tolist() == [980, 960, 940] and s.timesteps[-1].item() == 0


def test_pipeline_text_preprocessing_on_cpu():
    """pipeline_latte.py:182,230-231,359-379: prompt and negative prompt are lower-cased and stripped before the tokenizer
    (T5 is case sensitive); clean_caption=True without bs4 / ftfy warns and does the same, as the reference."""
    import warnings
    from types import SimpleNamespace
    from latte_amd.schedulers import DDIMScheduler
    cfg, sd, z = _fixture()
    seen = []

    def tokenizer(texts, **kw):
        seen.append(list(texts))
        n = kw["max_length"]
        return SimpleNamespace(input_ids=torch.zeros(len(texts), n, dtype=torch.int64),
                               attention_mask=torch.ones(len(texts), n, dtype=torch.int64))

    def text_encoder(ids, attention_mask=None):
        return (torch.zeros(ids.shape[0], ids.shape[1], cfg.caption_channels),)

    pipe = latte_amd.LattePipeline(tokenizer=tokenizer, text_encoder=text_encoder, transformer=_model(cfg, sd, "f16"),
                                   scheduler=DDIMScheduler())
    pe, ne = pipe.encode_prompt(["  A Dog Running On The BEACH \n"], negative_prompt=" Blurry ", device="cpu")
    assert seen == [["a dog running on the beach"], ["blurry"]]
    assert pe.shape[0] == 1 and ne.shape[:2] == pe.shape[:2]
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        assert pipe._text_preprocessing("  MiXed ", clean_caption=True) == ["mixed"]
        assert any("clean_caption" in str(x.message) for x in w)


@pytest.mark.gpu
def test_pipeline_guided_ddim_chain_matches_oracle_loop():
    """The denoising loop of pipeline_latte.py:700-760 (guidance pair [negative, prompt], learned-sigma drop, scheduler step)
    around the engine denoiser vs the same loop around the oracle denoiser; then the per-frame VAE decode hand-off."""
    from oracle import latte_t2v_oracle as to
    from oracle import vae_oracle as vo
    from latte_amd.schedulers import DDIMScheduler
    cfg, sd, z = _fixture()
    g = torch.Generator("cpu").manual_seed(9)
    pe, ne = torch.randn(1, 6, cfg.caption_channels, generator=g), torch.randn(1, 6, cfg.caption_channels, generator=g)
    lat = torch.randn(1, 4, cfg.video_length, cfg.sample_size, cfg.sample_size, generator=g)
    steps, scale = 4, 4.5
    sch = DDIMScheduler()
    sch.set_timesteps(steps)
    want = lat.clone()
    with torch.no_grad():
        for t in sch.timesteps:
            x2 = torch.cat([want] * 2)
            out = to.latte_t2v_forward(sd, cfg, x2, t.reshape(1).expand(2), torch.cat([ne, pe]))
            unc, txt = out.chunk(2)
            eps = (unc + scale * (txt - unc)).chunk(2, dim=1)[0]
            want = sch.step(eps, t, want, return_dict=False)[0]
    pipe = latte_amd.LattePipeline(transformer=_model(cfg, sd, "f16"), scheduler=DDIMScheduler()).to("cuda")
    got = pipe(prompt_embeds=pe, negative_prompt_embeds=ne, num_inference_steps=steps, guidance_scale=scale, latents=lat,
               output_type="latents").video
    print(f"t2v guided chain rel-L2 vs oracle loop: {rel_l2(got, want):.3e}")
    assert rel_l2(got, want) < TOL
    # that was the loop fused into the engine (latte_t2v_guided_ddim_loop); the step-by-step loop around the engine denoiser
    # (any scheduler object) must give the same latents
    pipe.allow_fused_loop = False
    slow = pipe(prompt_embeds=pe, negative_prompt_embeds=ne, num_inference_steps=steps, guidance_scale=scale, latents=lat,
                output_type="latents").video
    assert rel_l2(got, slow) < 1e-5 and rel_l2(slow, want) < TOL
    # a text context installed once serves every later forward without encoder_hidden_states
    m = pipe.transformer
    x2 = torch.cat([lat, lat]).cuda()
    tt = torch.tensor([500, 500]).cuda()
    direct = m(x2, tt, torch.cat([ne, pe]).cuda()).sample
    m.set_text(torch.cat([ne, pe]))
    out = torch.empty_like(direct)
    from latte_amd._lib import check, load_library, ptr, stream_ptr
    check(load_library().latte_t2v_forward(m._h, ptr(x2.contiguous()), ptr(tt), None, None, 2, 6, 1, ptr(out), stream_ptr()))
    torch.cuda.synchronize()
    assert torch.equal(out, direct)
    # decode hand-off (pipeline_latte.py:773-785) on a 16x16 latent (the VAE engine's smallest): uint8 [b, f, h, w, c]
    vsd = vo.init_state_dict(seed=2)
    vae = latte_amd.AutoencoderKL(latent_size=16, max_frames=2, compute_dtype="f16")
    vae.load_state_dict(vsd)
    pipe2 = latte_amd.LattePipeline(vae=vae, transformer=_model(cfg, sd, "f16"), scheduler=DDIMScheduler()).to("cuda")
    zl = torch.randn(1, 4, 2, 16, 16, generator=g)
    video = pipe2.decode_latents(zl.cuda())
    assert video.dtype == torch.uint8 and tuple(video.shape) == (1, 2, 128, 128, 3)
    ref = vo.decode(vsd, (zl / 0.18215).permute(0, 2, 1, 3, 4).reshape(2, 4, 16, 16))
    ref = ((ref / 2.0 + 0.5).clamp(0, 1) * 255).to(torch.uint8).permute(0, 2, 3, 1)
    assert (video[0].int() - ref.int()).abs().float().mean() < 1.0


def test_ddim_scheduler_standin_closed_form():
    """latte_amd.schedulers.DDIMScheduler (memory-derived stand-in, unpinned): a full eta = 0 chain on a model that returns the
    TRUE noise recovers x0 exactly, the last step lands on alpha_bar_prev = 1, and the update is the DDIM closed form."""
    from latte_amd.schedulers import DDIMScheduler
    s = DDIMScheduler()
    s.set_timesteps(10)
    assert s.timesteps.tolist() == [900, 800, 700, 600, 500, 400, 300, 200, 100, 0]
    g = torch.Generator().manual_seed(0)
    x0 = torch.randn(1, 4, 2, 4, 4, generator=g, dtype=torch.float64)
    eps = torch.randn(1, 4, 2, 4, 4, generator=g, dtype=torch.float64)
    a = float(s.alphas_cumprod[900])
    x = a ** 0.5 * x0 + (1 - a) ** 0.5 * eps
    for t in s.timesteps:
        at = float(s.alphas_cumprod[int(t)])
        true_eps = (x - at ** 0.5 * x0) / (1 - at) ** 0.5
        x = s.step(true_eps, t, x, return_dict=False)[0]
    assert float((x - x0).abs().max()) < 1e-9
    # one step against the closed form
    s.set_timesteps(4)
    t = s.timesteps[1]                      # 500 -> 250
    xt = torch.randn(2, 3, generator=g, dtype=torch.float64)
    e = torch.randn(2, 3, generator=g, dtype=torch.float64)
    at, ap = float(s.alphas_cumprod[500]), float(s.alphas_cumprod[250])
    want = ap ** 0.5 * (xt - (1 - at) ** 0.5 * e) / at ** 0.5 + (1 - ap) ** 0.5 * e
    assert float((s.step(e, t, xt, return_dict=False)[0] - want).abs().max()) < 1e-12
