"""Random weights of the right shapes for plumbing runs and benchmarks (there are no checkpoints offline).

Not test infrastructure and not an oracle: these only FILL state dicts with reference-format keys (diffusers
``AutoencoderKL`` decoder keys of the sd-vae-ft architecture, ``LatteT2V`` keys of models/latte_t2v.py) so that
``bench.py`` and ``tools/*`` can time the device path.  ``Latte`` itself initialises like the reference
(latte.py:257-295); ``randomize_zero_init`` re-draws what that leaves at zero so outputs are not trivially 0.
"""
import math

import torch


def randomize_zero_init(model, std=0.02, seed=1):
    g = torch.Generator("cpu").manual_seed(seed)
    with torch.no_grad():
        for _, p in model.named_parameters():
            if p.requires_grad and float(p.detach().abs().max()) == 0.0:
                p.copy_(torch.randn(p.shape, generator=g) * std)
    return model


def _fill(keys, seed):
    g = torch.Generator("cpu").manual_seed(seed)
    sd = {}
    for k, shp in keys.items():
        if k.endswith("scale_shift_table"):
            sd[k] = torch.randn(shp, generator=g) / shp[-1] ** 0.5
        elif "norm" in k and k.endswith("weight"):
            sd[k] = 1.0 + 0.1 * torch.randn(shp, generator=g)
        elif k.endswith("bias"):
            sd[k] = 0.02 * torch.randn(shp, generator=g)
        else:
            sd[k] = torch.randn(shp, generator=g) / math.sqrt(max(math.prod(shp[1:]), 1))
    return sd


def vae_decoder_keys(block_out=(128, 256, 512, 512), layers=2, latent=4):
    """diffusers AutoencoderKL decoder half (post_quant_conv + decoder.*), sd-vae-ft layout."""
    ks = {"post_quant_conv.weight": (latent, latent, 1, 1), "post_quant_conv.bias": (latent,)}
    top = block_out[-1]
    ks["decoder.conv_in.weight"], ks["decoder.conv_in.bias"] = (top, latent, 3, 3), (top,)

    def resnet(p, cin, cout):
        ks[p + "norm1.weight"], ks[p + "norm1.bias"] = (cin,), (cin,)
        ks[p + "conv1.weight"], ks[p + "conv1.bias"] = (cout, cin, 3, 3), (cout,)
        ks[p + "norm2.weight"], ks[p + "norm2.bias"] = (cout,), (cout,)
        ks[p + "conv2.weight"], ks[p + "conv2.bias"] = (cout, cout, 3, 3), (cout,)
        if cin != cout:
            ks[p + "conv_shortcut.weight"], ks[p + "conv_shortcut.bias"] = (cout, cin, 1, 1), (cout,)

    resnet("decoder.mid_block.resnets.0.", top, top)
    a = "decoder.mid_block.attentions.0."
    ks[a + "group_norm.weight"], ks[a + "group_norm.bias"] = (top,), (top,)
    for n in ("to_q", "to_k", "to_v", "to_out.0"):
        ks[a + n + ".weight"], ks[a + n + ".bias"] = (top, top), (top,)
    resnet("decoder.mid_block.resnets.1.", top, top)
    rev, prev = list(reversed(block_out)), block_out[-1]
    for i, cout in enumerate(rev):
        for r in range(layers + 1):
            resnet(f"decoder.up_blocks.{i}.resnets.{r}.", prev if r == 0 else cout, cout)
        prev = cout
        if i != len(rev) - 1:
            ks[f"decoder.up_blocks.{i}.upsamplers.0.conv.weight"] = (cout, cout, 3, 3)
            ks[f"decoder.up_blocks.{i}.upsamplers.0.conv.bias"] = (cout,)
    ks["decoder.conv_norm_out.weight"], ks["decoder.conv_norm_out.bias"] = (block_out[0],), (block_out[0],)
    ks["decoder.conv_out.weight"], ks["decoder.conv_out.bias"] = (3, block_out[0], 3, 3), (3,)
    return ks


def vae_decoder_state_dict(seed=0):
    return _fill(vae_decoder_keys(), seed)


def vae_temporal_decoder_keys():
    """diffusers AutoencoderKLTemporalDecoder decoder keys: the SD-VAE decoder's with every resnet split into
    ``spatial_res_block`` / ``temporal_res_block`` (Conv3d (3,1,1)) / ``time_mixer.mix_factor``, no post_quant_conv, plus
    ``decoder.time_conv_out``."""
    ks = {}
    base = vae_decoder_keys()
    for k, shp in base.items():
        if k.startswith("post_quant_conv"):
            continue
        if ".resnets." in k:
            head, tail = k.rsplit(".", 2)[0], ".".join(k.rsplit(".", 2)[1:])      # "...resnets.N", "conv1.weight"
            ks[f"{head}.spatial_res_block.{tail}"] = shp
            if not tail.startswith("conv_shortcut"):
                c = base[f"{head}.norm2.weight"][0]                               # the temporal block works on out_channels
                ks[f"{head}.temporal_res_block.{tail}"] = (c, c, 3, 1, 1) if tail.endswith("weight") and tail.startswith("conv") else (c,)
            ks[f"{head}.time_mixer.mix_factor"] = (1,)
        else:
            ks[k] = shp
    ks["decoder.time_conv_out.weight"], ks["decoder.time_conv_out.bias"] = (3, 3, 3, 1, 1), (3,)
    return ks


def vae_temporal_decoder_state_dict(seed=0):
    sd = _fill(vae_temporal_decoder_keys(), seed)
    for k in sd:
        if k.endswith("mix_factor"):
            sd[k] = torch.zeros(1)            # AlphaBlender(alpha = 0.0): sigmoid(0) = 0.5
    return sd


def t2v_keys(num_attention_heads=16, attention_head_dim=72, in_channels=4, out_channels=8, num_layers=28, patch_size=2,
             cross_attention_dim=1152, caption_channels=4096, **unused):
    """State-dict keys of LatteT2V (models/latte_t2v.py) in the Latte-1 configuration."""
    D, p = num_attention_heads * attention_head_dim, patch_size
    ks = {"scale_shift_table": (2, D), "pos_embed.proj.weight": (D, in_channels, p, p), "pos_embed.proj.bias": (D,)}

    def attn(pre, kv):
        for n, i in (("to_q", D), ("to_k", kv), ("to_v", kv), ("to_out.0", D)):
            ks[pre + n + ".weight"], ks[pre + n + ".bias"] = (D, i), (D,)

    for kind, cross in (("transformer_blocks", True), ("temporal_transformer_blocks", False)):
        for i in range(num_layers):
            b = f"{kind}.{i}."
            ks[b + "scale_shift_table"] = (6, D)
            attn(b + "attn1.", D)
            if cross:
                attn(b + "attn2.", cross_attention_dim)
            ks[b + "ff.net.0.proj.weight"], ks[b + "ff.net.0.proj.bias"] = (4 * D, D), (4 * D,)
            ks[b + "ff.net.2.weight"], ks[b + "ff.net.2.bias"] = (D, 4 * D), (D,)
    ks["proj_out.weight"], ks["proj_out.bias"] = (p * p * out_channels, D), (p * p * out_channels,)
    e = "adaln_single.emb.timestep_embedder."
    ks[e + "linear_1.weight"], ks[e + "linear_1.bias"] = (D, 256), (D,)
    ks[e + "linear_2.weight"], ks[e + "linear_2.bias"] = (D, D), (D,)
    ks["adaln_single.linear.weight"], ks["adaln_single.linear.bias"] = (6 * D, D), (6 * D,)
    ks["caption_projection.linear_1.weight"], ks["caption_projection.linear_1.bias"] = (D, caption_channels), (D,)
    ks["caption_projection.linear_2.weight"], ks["caption_projection.linear_2.bias"] = (D, D), (D,)
    return ks


def t2v_state_dict(seed=0, **config):
    return _fill(t2v_keys(**config), seed)
