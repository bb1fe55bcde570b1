"""CPU oracle of ``AutoencoderKLTemporalDecoder.decode`` — TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED.  The reference builds ``AutoencoderKLTemporalDecoder.from_pretrained(path, subfolder="vae_temporal_decoder")``
(/root/reference/sample/sample_t2x.py:31-32) and calls ``vae.decode(latents[i : i + 14], num_frames=n).sample`` on chunks of 14
frames (/root/reference/sample/pipeline_latte.py:779-798); the class comes from diffusers==0.24.0, which is neither vendored under
/root/reference nor installable here, and no weights exist offline.  This file restates, from memory of the published source
(``diffusers/models/autoencoder_kl_temporal_decoder.py``, ``unet_3d_blocks.py``, ``resnet.py``), the decoder of the
stable-video-diffusion VAE in plain torch ops; agreement with real diffusers is unverified until it can be imported.

Structure restated (diffusers 0.24.0):
  AutoencoderKLTemporalDecoder.decode(z, num_frames) : TemporalDecoder(z, image_only_indicator = zeros[B, num_frames]) -- there is
                                                        NO post_quant_conv in this class
  TemporalDecoder.forward     : conv_in -> mid_block -> up_blocks[0..3] -> conv_norm_out (GN 32, eps 1e-6) -> SiLU -> conv_out
                                -> time_conv_out = Conv3d(3, 3, (3, 1, 1), padding (1, 0, 0)) over the frames of a batch item
  MidBlockTemporalDecoder     : resnets[0] -> attentions[0] (1 head of 512, as the SD VAE) -> resnets[1]
  UpBlockTemporalDecoder      : 3 x SpatioTemporalResBlock (+ Upsample2D: nearest x2 then conv3x3, blocks 0..2)
  SpatioTemporalResBlock      : ResnetBlock2D(eps 1e-6) per frame, then TemporalResnetBlock(eps 1e-5) on [B, C, T, H, W]:
                                GN(32) over (C/32, T, H, W) -> SiLU -> Conv3d (3,1,1) -> GN -> SiLU -> Conv3d (3,1,1), + input;
                                AlphaBlender("learned", switch_spatial_to_temporal_mix=True): a = 1 - sigmoid(mix_factor),
                                out = a * x_spatial + (1 - a) * x_temporal
"""
import math

import torch
import torch.nn.functional as F

from oracle import vae_oracle as vo

BLOCK_OUT = (128, 256, 512, 512)
GROUPS = 32
EPS, TEMPORAL_EPS = 1e-6, 1e-5


def decoder_keys(block_out=BLOCK_OUT, layers=2, latent=4):
    ks = {}
    top = block_out[-1]
    ks["decoder.conv_in.weight"] = (top, latent, 3, 3)
    ks["decoder.conv_in.bias"] = (top,)

    def st_block(prefix, cin, cout):
        s, t = prefix + "spatial_res_block.", prefix + "temporal_res_block."
        for p, ci in ((s, cin), (t, cout)):
            ks[p + "norm1.weight"] = (ci,)
            ks[p + "norm1.bias"] = (ci,)
            ks[p + "norm2.weight"] = (cout,)
            ks[p + "norm2.bias"] = (cout,)
            ks[p + "conv1.bias"] = (cout,)
            ks[p + "conv2.bias"] = (cout,)
        ks[s + "conv1.weight"] = (cout, cin, 3, 3)
        ks[s + "conv2.weight"] = (cout, cout, 3, 3)
        if cin != cout:
            ks[s + "conv_shortcut.weight"] = (cout, cin, 1, 1)
            ks[s + "conv_shortcut.bias"] = (cout,)
        ks[t + "conv1.weight"] = (cout, cout, 3, 1, 1)
        ks[t + "conv2.weight"] = (cout, cout, 3, 1, 1)
        ks[prefix + "time_mixer.mix_factor"] = (1,)

    st_block("decoder.mid_block.resnets.0.", top, top)
    a = "decoder.mid_block.attentions.0."
    ks[a + "group_norm.weight"] = (top,)
    ks[a + "group_norm.bias"] = (top,)
    for n in ("to_q", "to_k", "to_v", "to_out.0"):
        ks[a + n + ".weight"] = (top, top)
        ks[a + n + ".bias"] = (top,)
    st_block("decoder.mid_block.resnets.1.", top, top)
    rev = list(reversed(block_out))
    prev = rev[0]
    for i, cout in enumerate(rev):
        for r in range(layers + 1):
            st_block(f"decoder.up_blocks.{i}.resnets.{r}.", prev if r == 0 else cout, cout)
        prev = cout
        if i != len(rev) - 1:
            ks[f"decoder.up_blocks.{i}.upsamplers.0.conv.weight"] = (cout, cout, 3, 3)
            ks[f"decoder.up_blocks.{i}.upsamplers.0.conv.bias"] = (cout,)
    ks["decoder.conv_norm_out.weight"] = (block_out[0],)
    ks["decoder.conv_norm_out.bias"] = (block_out[0],)
    ks["decoder.conv_out.weight"] = (3, block_out[0], 3, 3)
    ks["decoder.conv_out.bias"] = (3,)
    ks["decoder.time_conv_out.weight"] = (3, 3, 3, 1, 1)
    ks["decoder.time_conv_out.bias"] = (3,)
    return ks


def init_state_dict(seed=0, mix=None):
    """Random decoder weights with sane magnitudes; ``mix``: value for every ``time_mixer.mix_factor`` (default: N(0, 1) draws)."""
    g = torch.Generator("cpu").manual_seed(seed)
    sd = {}
    for k, shp in decoder_keys().items():
        if k.endswith("mix_factor"):
            sd[k] = torch.randn(shp, generator=g) if mix is None else torch.full(shp, float(mix))
        elif "norm" in k and k.endswith("weight"):
            sd[k] = 1.0 + 0.1 * torch.randn(shp, generator=g)
        elif k.endswith("bias"):
            sd[k] = 0.05 * torch.randn(shp, generator=g)
        else:
            sd[k] = torch.randn(shp, generator=g) / math.sqrt(math.prod(shp[1:]))
    return sd


def _temporal_resnet(sd, p, x5):
    """TemporalResnetBlock on [B, C, T, H, W]."""
    h = F.silu(F.group_norm(x5, GROUPS, sd[p + "norm1.weight"], sd[p + "norm1.bias"], TEMPORAL_EPS))
    h = F.conv3d(h, sd[p + "conv1.weight"], sd[p + "conv1.bias"], padding=(1, 0, 0))
    h = F.silu(F.group_norm(h, GROUPS, sd[p + "norm2.weight"], sd[p + "norm2.bias"], TEMPORAL_EPS))
    h = F.conv3d(h, sd[p + "conv2.weight"], sd[p + "conv2.bias"], padding=(1, 0, 0))
    return x5 + h


def _st_block(sd, p, x, num_frames):
    """SpatioTemporalResBlock: x [B*T, C, H, W]."""
    x = vo._resnet(sd, p + "spatial_res_block.", x)
    bt, c, hh, ww = x.shape
    x5 = x.reshape(bt // num_frames, num_frames, c, hh, ww).permute(0, 2, 1, 3, 4)
    xt = _temporal_resnet(sd, p + "temporal_res_block.", x5)
    alpha = 1.0 - torch.sigmoid(sd[p + "time_mixer.mix_factor"])      # switch_spatial_to_temporal_mix
    out = alpha * x5 + (1.0 - alpha) * xt
    return out.permute(0, 2, 1, 3, 4).reshape(bt, c, hh, ww)


def decode(sd, z, num_frames, trace=None):
    """z: [B * num_frames, 4, h, w] fp32 (already divided by the scaling factor) -> [B * num_frames, 3, 8h, 8w] fp32."""
    def t(x):
        if trace is not None:
            trace.append(x.clone())
        return x
    with torch.no_grad():
        x = t(F.conv2d(z, sd["decoder.conv_in.weight"], sd["decoder.conv_in.bias"], padding=1))
        x = t(_st_block(sd, "decoder.mid_block.resnets.0.", x, num_frames))
        x = t(vo._attention(sd, "decoder.mid_block.attentions.0.", x))
        x = t(_st_block(sd, "decoder.mid_block.resnets.1.", x, num_frames))
        for i in range(4):
            for r in range(3):
                x = t(_st_block(sd, f"decoder.up_blocks.{i}.resnets.{r}.", x, num_frames))
            if i != 3:
                x = F.interpolate(x, scale_factor=2.0, mode="nearest")
                x = t(F.conv2d(x, sd[f"decoder.up_blocks.{i}.upsamplers.0.conv.weight"],
                               sd[f"decoder.up_blocks.{i}.upsamplers.0.conv.bias"], padding=1))
        x = F.silu(F.group_norm(x, GROUPS, sd["decoder.conv_norm_out.weight"], sd["decoder.conv_norm_out.bias"], EPS))
        x = F.conv2d(x, sd["decoder.conv_out.weight"], sd["decoder.conv_out.bias"], padding=1)
        bt, c, hh, ww = x.shape
        x5 = x.reshape(bt // num_frames, num_frames, c, hh, ww).permute(0, 2, 1, 3, 4)
        x5 = F.conv3d(x5, sd["decoder.time_conv_out.weight"], sd["decoder.time_conv_out.bias"], padding=(1, 0, 0))
        return x5.permute(0, 2, 1, 3, 4).reshape(bt, c, hh, ww)
