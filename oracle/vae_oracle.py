"""CPU oracle of the SD-VAE decoder — TEST INFRASTRUCTURE ONLY (imported by tests/, smoke(), bench).

PARITY UNPINNED.  The reference calls ``AutoencoderKL.from_pretrained(...).decode(z / 0.18215).sample``
(/root/reference/sample/sample.py:69,113-115; sample_ddp.py:90,165-168) from diffusers==0.24.0
(/root/reference/environment.yml:12), which is neither vendored under /root/reference nor installable
here, and no weights exist offline.  This file restates the published architecture of
``diffusers.models.autoencoder_kl.AutoencoderKL.decode`` for the ``stabilityai/sd-vae-ft-*`` config
(block_out_channels (128,256,512,512), layers_per_block 2, latent_channels 4, norm_num_groups 32,
act "silu") in plain torch ops; agreement with real diffusers is unverified until it can be imported.

Structure followed (diffusers 0.24.0):
  AutoencoderKL.decode      : z -> post_quant_conv (1x1) -> Decoder
  vae.Decoder.forward       : conv_in -> mid_block -> up_blocks[0..3] -> conv_norm_out -> SiLU -> conv_out
  UNetMidBlock2D            : resnets[0] -> attentions[0] -> resnets[1]
  UpDecoderBlock2D          : 3 x ResnetBlock2D (+ Upsample2D: nearest x2 then conv3x3, blocks 0..2)
  ResnetBlock2D (temb None) : GN(32,eps 1e-6) -> SiLU -> conv3x3 -> GN -> SiLU -> conv3x3, + (conv_shortcut 1x1)(x)
  Attention (1 head, dim 512, residual_connection, group_norm): softmax(q k^T / sqrt(512)) v -> to_out[0]
"""
import math

import torch
import torch.nn.functional as F

BLOCK_OUT = (128, 256, 512, 512)
LAYERS_PER_BLOCK = 2
LATENT_CHANNELS = 4
GROUPS = 32
EPS = 1e-6
SCALING_FACTOR = 0.18215


def decoder_keys(block_out=BLOCK_OUT, layers=LAYERS_PER_BLOCK, latent=LATENT_CHANNELS):
    """state_dict keys (diffusers naming) -> shapes of every tensor `decode` touches."""
    ks = {"post_quant_conv.weight": (latent, latent, 1, 1), "post_quant_conv.bias": (latent,)}
    top = block_out[-1]
    ks["decoder.conv_in.weight"] = (top, latent, 3, 3)
    ks["decoder.conv_in.bias"] = (top,)

    def resnet(prefix, cin, cout):
        ks[prefix + "norm1.weight"] = (cin,)
        ks[prefix + "norm1.bias"] = (cin,)
        ks[prefix + "conv1.weight"] = (cout, cin, 3, 3)
        ks[prefix + "conv1.bias"] = (cout,)
        ks[prefix + "norm2.weight"] = (cout,)
        ks[prefix + "norm2.bias"] = (cout,)
        ks[prefix + "conv2.weight"] = (cout, cout, 3, 3)
        ks[prefix + "conv2.bias"] = (cout,)
        if cin != cout:
            ks[prefix + "conv_shortcut.weight"] = (cout, cin, 1, 1)
            ks[prefix + "conv_shortcut.bias"] = (cout,)

    resnet("decoder.mid_block.resnets.0.", top, top)
    a = "decoder.mid_block.attentions.0."
    ks[a + "group_norm.weight"] = (top,)
    ks[a + "group_norm.bias"] = (top,)
    for n in ("to_q", "to_k", "to_v", "to_out.0"):
        ks[a + n + ".weight"] = (top, top)
        ks[a + n + ".bias"] = (top,)
    resnet("decoder.mid_block.resnets.1.", top, top)
    rev = list(reversed(block_out))
    prev = rev[0]
    for i, cout in enumerate(rev):
        for r in range(layers + 1):
            resnet(f"decoder.up_blocks.{i}.resnets.{r}.", prev if r == 0 else cout, cout)
        prev = cout
        if i != len(rev) - 1:
            ks[f"decoder.up_blocks.{i}.upsamplers.0.conv.weight"] = (cout, cout, 3, 3)
            ks[f"decoder.up_blocks.{i}.upsamplers.0.conv.bias"] = (cout,)
    ks["decoder.conv_norm_out.weight"] = (block_out[0],)
    ks["decoder.conv_norm_out.bias"] = (block_out[0],)
    ks["decoder.conv_out.weight"] = (3, block_out[0], 3, 3)
    ks["decoder.conv_out.bias"] = (3,)
    return ks


def init_state_dict(seed=0, block_out=BLOCK_OUT, layers=LAYERS_PER_BLOCK):
    """Random decoder weights with sane magnitudes (kaiming-like convs, norm weight ~1)."""
    g = torch.Generator("cpu").manual_seed(seed)
    sd = {}
    for k, shp in decoder_keys(block_out, layers).items():
        if "norm" in k and k.endswith("weight"):
            sd[k] = 1.0 + 0.1 * torch.randn(shp, generator=g)
        elif k.endswith("bias"):
            sd[k] = 0.05 * torch.randn(shp, generator=g)
        else:
            fan_in = math.prod(shp[1:])
            sd[k] = torch.randn(shp, generator=g) / math.sqrt(fan_in)
    return sd


def _resnet(sd, p, x):
    h = F.silu(F.group_norm(x, GROUPS, sd[p + "norm1.weight"], sd[p + "norm1.bias"], EPS))
    h = F.conv2d(h, sd[p + "conv1.weight"], sd[p + "conv1.bias"], padding=1)
    h = F.silu(F.group_norm(h, GROUPS, sd[p + "norm2.weight"], sd[p + "norm2.bias"], EPS))
    h = F.conv2d(h, sd[p + "conv2.weight"], sd[p + "conv2.bias"], padding=1)
    if p + "conv_shortcut.weight" in sd:
        x = F.conv2d(x, sd[p + "conv_shortcut.weight"], sd[p + "conv_shortcut.bias"])
    return x + h                                           # output_scale_factor = 1.0


def _attention(sd, p, x):
    n, c, hh, ww = x.shape
    res = x
    h = F.group_norm(x, GROUPS, sd[p + "group_norm.weight"], sd[p + "group_norm.bias"], EPS)
    h = h.view(n, c, hh * ww).transpose(1, 2)              # [n, hw, c]
    q = F.linear(h, sd[p + "to_q.weight"], sd[p + "to_q.bias"])
    k = F.linear(h, sd[p + "to_k.weight"], sd[p + "to_k.bias"])
    v = F.linear(h, sd[p + "to_v.weight"], sd[p + "to_v.bias"])
    a = torch.softmax(q @ k.transpose(1, 2) * (c ** -0.5), dim=-1)     # heads = 1, dim_head = c
    o = F.linear(a @ v, sd[p + "to_out.0.weight"], sd[p + "to_out.0.bias"])
    return o.transpose(1, 2).reshape(n, c, hh, ww) + res   # rescale_output_factor = 1


def decode(sd, z, block_out=BLOCK_OUT, layers=LAYERS_PER_BLOCK, trace=None):
    """z: [N, 4, h, w] fp32 (already divided by the scaling factor) -> [N, 3, 8h, 8w] fp32.
    trace: optional list receiving the activation after conv_in, each mid-block member, every up-block resnet
    and upsampler (the stage order of latte_debug_vae_trace)."""
    def t(x):
        if trace is not None:
            trace.append(x.clone())
        return x
    with torch.no_grad():
        x = F.conv2d(z, sd["post_quant_conv.weight"], sd["post_quant_conv.bias"])
        x = t(F.conv2d(x, sd["decoder.conv_in.weight"], sd["decoder.conv_in.bias"], padding=1))
        x = t(_resnet(sd, "decoder.mid_block.resnets.0.", x))
        x = t(_attention(sd, "decoder.mid_block.attentions.0.", x))
        x = t(_resnet(sd, "decoder.mid_block.resnets.1.", x))
        nb = len(block_out)
        for i in range(nb):
            for r in range(layers + 1):
                x = t(_resnet(sd, f"decoder.up_blocks.{i}.resnets.{r}.", x))
            if i != nb - 1:
                x = F.interpolate(x, scale_factor=2.0, mode="nearest")
                x = t(F.conv2d(x, sd[f"decoder.up_blocks.{i}.upsamplers.0.conv.weight"],
                               sd[f"decoder.up_blocks.{i}.upsamplers.0.conv.bias"], padding=1))
        x = F.silu(F.group_norm(x, GROUPS, sd["decoder.conv_norm_out.weight"], sd["decoder.conv_norm_out.bias"], EPS))
        return F.conv2d(x, sd["decoder.conv_out.weight"], sd["decoder.conv_out.bias"], padding=1)


def to_uint8_video(samples):
    """sample.py:122: ((x*0.5+0.5)*255).add_(0.5).clamp_(0,255).to(uint8), [n,c,h,w] -> [n,h,w,c]."""
    return ((samples * 0.5 + 0.5) * 255).add_(0.5).clamp_(0, 255).to(torch.uint8).permute(0, 2, 3, 1).contiguous()
