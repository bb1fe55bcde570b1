"""CPU restatement of the reference LatteT2V denoiser (Latte-1 text-to-video), fp32, plain torch ops.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).  SURVEY.md section 8(f) rank 2: groundwork for the NEXT row of the
hot path; no engine code consumes this yet.

Parity status: **partially pinned**.  ``oracle/validate_t2v_oracle.py`` runs the unmodified
``/root/reference/models/latte_t2v.py`` and this file on identical weights / inputs and they agree to fp32 round-off --
but the reference file imports its leaf modules from diffusers==0.24.0, which is absent here, so it runs on
``oracle/diffusers_standin.py``, a memory-derived restatement of those leaves.  What IS pinned: everything the reference
file itself defines -- the frame / token rearranges, ``BasicTransformerBlock_`` (temporal block, latte_t2v.py:126-396),
``FeedForward`` (:69-124), ``AdaLayerNormSingle`` (:398-428), the adaLN-single output head and unpatchify (:904-934).
What is NOT: the numerics of diffusers' ``Attention``, the spatial ``BasicTransformerBlock``, ``PatchEmbed``,
``CaptionProjection`` and ``CombinedTimestepSizeEmbeddings`` as real diffusers computes them.

All ``t2v:N`` citations are ``/root/reference/models/latte_t2v.py``.  The model is a pure function of a
reference-format ``state_dict``.
"""
import math
from dataclasses import dataclass

import numpy as np
import torch
import torch.nn.functional as F


@dataclass
class T2VConfig:
    """The constructor arguments Latte-1 uses (t2v:475-502; values of the released transformer/config.json as far as the
    reference code fixes them: PixArt-alpha XL/2 geometry)."""
    num_attention_heads: int = 16
    attention_head_dim: int = 72
    in_channels: int = 4
    out_channels: int = 8
    num_layers: int = 28
    sample_size: int = 64          # latent side (512 px / 8)
    patch_size: int = 2
    cross_attention_dim: int = 1152
    caption_channels: int = 4096
    video_length: int = 16
    norm_eps: float = 1e-6

    @property
    def inner_dim(self):
        return self.num_attention_heads * self.attention_head_dim


# ----------------------------------------------------------------------------- fixed tables
def sincos_1d(embed_dim, pos):
    omega = np.arange(embed_dim // 2, dtype=np.float64)
    omega /= embed_dim / 2.0
    omega = 1.0 / 10000 ** omega
    out = np.einsum("m,d->md", np.asarray(pos).reshape(-1), omega)
    return np.concatenate([np.sin(out), np.cos(out)], axis=1)


def pos_embed_2d(embed_dim, grid, base_size, interpolation_scale):
    """PatchEmbed's table (diffusers get_2d_sincos_pos_embed): positions scaled by base_size / interpolation_scale."""
    gh = np.arange(grid, dtype=np.float32) / (grid / base_size) / interpolation_scale
    gw = np.arange(grid, dtype=np.float32) / (grid / base_size) / interpolation_scale
    g = np.stack(np.meshgrid(gw, gh), axis=0).reshape([2, 1, grid, grid])
    return np.concatenate([sincos_1d(embed_dim // 2, g[0]), sincos_1d(embed_dim // 2, g[1])], axis=1)


def temp_pos_embed(embed_dim, length):
    """t2v:670-671,943-945: get_1d_sincos_pos_embed_from_grid on arange(length) (an int64 torch column there)."""
    return sincos_1d(embed_dim, np.arange(length, dtype=np.float64))


def timestep_embedding(t, dim=256):
    """diffusers Timesteps(256, flip_sin_to_cos=True, downscale_freq_shift=0): [cos | sin]."""
    half = dim // 2
    freqs = torch.exp(-math.log(10000.0) * torch.arange(0, half, dtype=torch.float32) / half)
    args = t[:, None].float() * freqs[None]
    return torch.cat([torch.cos(args), torch.sin(args)], dim=-1)


# ----------------------------------------------------------------------------- weights
def state_dict_keys(cfg: T2VConfig):
    D, p = cfg.inner_dim, cfg.patch_size
    ks = {"scale_shift_table": (2, D), "pos_embed.proj.weight": (D, cfg.in_channels, p, p), "pos_embed.proj.bias": (D,)}

    def attn(pre, kv):
        for n, i in (("to_q", D), ("to_k", kv), ("to_v", kv), ("to_out.0", D)):
            ks[pre + n + ".weight"] = (D, i)
            ks[pre + n + ".bias"] = (D,)

    for kind, cross in (("transformer_blocks", True), ("temporal_transformer_blocks", False)):
        for i in range(cfg.num_layers):
            b = f"{kind}.{i}."
            ks[b + "scale_shift_table"] = (6, D)
            attn(b + "attn1.", D)
            if cross:
                attn(b + "attn2.", cfg.cross_attention_dim)
            ks[b + "ff.net.0.proj.weight"] = (4 * D, D)
            ks[b + "ff.net.0.proj.bias"] = (4 * D,)
            ks[b + "ff.net.2.weight"] = (D, 4 * D)
            ks[b + "ff.net.2.bias"] = (D,)
    ks["proj_out.weight"] = (p * p * cfg.out_channels, D)
    ks["proj_out.bias"] = (p * p * cfg.out_channels,)
    ks["adaln_single.emb.timestep_embedder.linear_1.weight"] = (D, 256)
    ks["adaln_single.emb.timestep_embedder.linear_1.bias"] = (D,)
    ks["adaln_single.emb.timestep_embedder.linear_2.weight"] = (D, D)
    ks["adaln_single.emb.timestep_embedder.linear_2.bias"] = (D,)
    ks["adaln_single.linear.weight"] = (6 * D, D)
    ks["adaln_single.linear.bias"] = (6 * D,)
    ks["caption_projection.linear_1.weight"] = (D, cfg.caption_channels)
    ks["caption_projection.linear_1.bias"] = (D,)
    ks["caption_projection.linear_2.weight"] = (D, D)
    ks["caption_projection.linear_2.bias"] = (D,)
    ks["caption_projection.y_embedding"] = (120, cfg.caption_channels)   # null-caption buffer of PixArt checkpoints; unused at inference
    return ks


def init_state_dict(cfg: T2VConfig, seed=0):
    g = torch.Generator("cpu").manual_seed(seed)
    sd = {}
    for k, shp in state_dict_keys(cfg).items():
        if k.endswith("scale_shift_table"):
            sd[k] = torch.randn(shp, generator=g) / shp[1] ** 0.5
        elif k.endswith("bias"):
            sd[k] = 0.02 * torch.randn(shp, generator=g)
        else:
            fan_in = math.prod(shp[1:])
            sd[k] = torch.randn(shp, generator=g) / math.sqrt(fan_in)
    return sd


# ----------------------------------------------------------------------------- forward
def _attention(sd, pre, x, heads, context=None, bias=None):
    """diffusers Attention (default processor): softmax(q k^T * hd^-0.5 + additive bias) v, then to_out[0]."""
    B, L, D = x.shape
    ctx = x if context is None else context
    hd = D // heads
    q = F.linear(x, sd[pre + "to_q.weight"], sd[pre + "to_q.bias"]).view(B, L, heads, hd).transpose(1, 2)
    k = F.linear(ctx, sd[pre + "to_k.weight"], sd[pre + "to_k.bias"]).view(B, -1, heads, hd).transpose(1, 2)
    v = F.linear(ctx, sd[pre + "to_v.weight"], sd[pre + "to_v.bias"]).view(B, -1, heads, hd).transpose(1, 2)
    s = (q @ k.transpose(-2, -1)) * (hd ** -0.5)
    if bias is not None:
        s = s + bias[:, None]                                   # [B, 1, Lkv] -> every head, every query
    o = (s.softmax(dim=-1) @ v).transpose(1, 2).reshape(B, L, D)
    return F.linear(o, sd[pre + "to_out.0.weight"], sd[pre + "to_out.0.bias"])


def _ff(sd, pre, x):
    """FeedForward with 'gelu-approximate' (t2v:69-124): Linear -> GELU(tanh) -> Linear."""
    h = F.gelu(F.linear(x, sd[pre + "ff.net.0.proj.weight"], sd[pre + "ff.net.0.proj.bias"]), approximate="tanh")
    return F.linear(h, sd[pre + "ff.net.2.weight"], sd[pre + "ff.net.2.bias"])


def _block(sd, pre, x, t6, heads, eps, context=None, ctx_bias=None):
    """adaLN-single block.  Temporal (t2v:272-396): LN -> modulate -> self-attention, gated; LN (norm3) -> modulate -> FF,
    gated.  Spatial (diffusers BasicTransformerBlock, ada_norm_single): the same plus cross-attention on the
    UN-normalised stream between the two (PixArt: no norm before attn2, no gate after it)."""
    B, _, D = x.shape
    sh1, sc1, g1, sh2, sc2, g2 = (sd[pre + "scale_shift_table"][None] + t6.reshape(B, 6, -1)).chunk(6, dim=1)   # t2v:301-304
    h = F.layer_norm(x, (D,), eps=eps) * (1 + sc1) + sh1
    x = g1 * _attention(sd, pre + "attn1.", h, heads) + x                                                       # t2v:330-336
    if context is not None:
        x = _attention(sd, pre + "attn2.", x, heads, context, ctx_bias) + x
    h = F.layer_norm(x, (D,), eps=eps) * (1 + sc2) + sh2                                                        # t2v:352-354
    return g2 * _ff(sd, pre, h) + x                                                                              # t2v:381-386


def latte_t2v_forward(sd, cfg: T2VConfig, x, t, encoder_hidden_states, encoder_attention_mask=None,
                      enable_temporal_attentions=True):
    """``LatteT2V.forward`` (t2v:677-941) at inference with use_image_num = 0.
    x: [B, C, F, H, W] (channels BEFORE frames, t2v:729), t: int64 [B], encoder_hidden_states: [B, Lk, caption_channels],
    encoder_attention_mask: [B, Lk] 1 = keep | None.  -> [B, out_channels, F, H, W]."""
    B, C, Fr, H, W = x.shape
    D, p, heads, eps = cfg.inner_dim, cfg.patch_size, cfg.num_attention_heads, cfg.norm_eps
    hs = x.permute(0, 2, 1, 3, 4).reshape(B * Fr, C, H, W).float()                                       # t2v:731
    bias = None
    if encoder_attention_mask is not None:                                                             # t2v:746-749
        bias = (1 - encoder_attention_mask.float()) * -10000.0
        bias = bias.unsqueeze(1).repeat_interleave(Fr, dim=0)                                          # 'b 1 l -> (b f) 1 l'
    gh = H // p
    T = gh * (W // p)
    # PatchEmbed: conv + fixed 2-D positions (interpolation_scale = max(sample_size // 64, 1), t2v:571-581)
    tok = F.conv2d(hs, sd["pos_embed.proj.weight"], sd["pos_embed.proj.bias"], stride=p).flatten(2).transpose(1, 2)
    interp = max(cfg.sample_size // 64, 1)
    pe = torch.from_numpy(pos_embed_2d(D, gh, cfg.sample_size // p, interp)).float()
    tok = tok + pe[None]
    # adaLN-single (t2v:398-428,775-779): embedded timestep, and its 6D modulation shared by every block
    temb = timestep_embedding(t, 256)
    temb = F.linear(temb, sd["adaln_single.emb.timestep_embedder.linear_1.weight"], sd["adaln_single.emb.timestep_embedder.linear_1.bias"])
    temb = F.linear(F.silu(temb), sd["adaln_single.emb.timestep_embedder.linear_2.weight"],
                    sd["adaln_single.emb.timestep_embedder.linear_2.bias"])
    t6 = F.linear(F.silu(temb), sd["adaln_single.linear.weight"], sd["adaln_single.linear.bias"])
    # caption projection (t2v:781-793): Linear -> GELU(tanh) -> Linear, then one copy per frame
    ctx = F.linear(encoder_hidden_states.float(), sd["caption_projection.linear_1.weight"], sd["caption_projection.linear_1.bias"])
    ctx = F.linear(F.gelu(ctx, approximate="tanh"), sd["caption_projection.linear_2.weight"], sd["caption_projection.linear_2.bias"])
    ctx_spatial = ctx.repeat_interleave(Fr, dim=0)                                                     # 'b t d -> (b f) t d'
    t6_spatial = t6.repeat_interleave(Fr, dim=0)                                                       # t2v:795
    t6_temp = t6.repeat_interleave(T, dim=0)                                                           # t2v:796
    tpe = torch.from_numpy(temp_pos_embed(D, cfg.video_length)).float()[None]
    h = tok
    for i in range(cfg.num_layers):
        h = _block(sd, f"transformer_blocks.{i}.", h, t6_spatial, heads, eps, ctx_spatial, bias)       # t2v:857-865
        if enable_temporal_attentions:
            h = h.reshape(B, Fr, T, D).permute(0, 2, 1, 3).reshape(B * T, Fr, D)                       # t2v:869
            if i == 0 and Fr > 1:
                h = h + tpe                                                                            # t2v:889-890
            h = _block(sd, f"temporal_transformer_blocks.{i}.", h, t6_temp, heads, eps)                # t2v:892-900
            h = h.reshape(B, T, Fr, D).permute(0, 2, 1, 3).reshape(B * Fr, T, D)                       # t2v:902
    # output head (t2v:913-918): LN, modulate with scale_shift_table + the embedded (not projected) timestep, proj_out
    e = temb.repeat_interleave(Fr, dim=0)
    shift, scale = (sd["scale_shift_table"][None] + e[:, None]).chunk(2, dim=1)
    h = F.layer_norm(h, (D,), eps=1e-6) * (1 + scale) + shift
    h = F.linear(h, sd["proj_out.weight"], sd["proj_out.bias"])
    co = cfg.out_channels
    h = h.reshape(B * Fr, gh, gh, p, p, co).permute(0, 5, 1, 3, 2, 4).reshape(B * Fr, co, gh * p, gh * p)   # t2v:924-931
    return h.reshape(B, Fr, co, H, W).permute(0, 2, 1, 3, 4).contiguous()                             # '(b f) c h w -> b c f h w'
