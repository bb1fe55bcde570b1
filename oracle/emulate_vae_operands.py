"""TEST INFRASTRUCTURE ONLY (like everything under oracle/): the SD-VAE decoder oracle (oracle/vae_oracle.py) with the engine's operand
rounding emulated on the CPU -- the inputs and weights of every convolution / linear rounded to f16 (fp32 accumulate, fp32 GroupNorm
and residual stream), q / k / v and the softmax probabilities of the mid-block attention rounded to f16 -- to see WHERE the decoder's
distance to the fp32 oracle comes from and what a hi + lo split of the activation operand of some convolutions would buy.

    python -m oracle.emulate_vae_operands            # table for the smoke's case (latent 16 x 16, weights seed 1, latent seed 5)

Nothing in latte_amd/ imports this."""
import sys

import torch
import torch.nn.functional as F

from . import vae_oracle as vo


def _r(x, on):
    return x.half().float() if on else x


class Rounding:
    """acts / weights: which operands are rounded to f16; exact_stages: names of stages ('mid', 'up0' .. 'up3', 'in', 'out') whose
    ACTIVATION operand is kept exact (what a hi + lo split of that operand gives, to first order)."""

    def __init__(self, acts=True, weights=True, exact_stages=()):
        self.acts, self.weights, self.exact = acts, weights, set(exact_stages)

    def a(self, x, stage):
        return _r(x, self.acts and stage not in self.exact)

    def w(self, x):
        return _r(x, self.weights)


def _conv(rd, stage, sd, p, x, padding):
    return F.conv2d(rd.a(x, stage), rd.w(sd[p + "weight"]), sd[p + "bias"], padding=padding)


def _resnet(rd, stage, sd, p, x):
    h = F.silu(F.group_norm(x, vo.GROUPS, sd[p + "norm1.weight"], sd[p + "norm1.bias"], vo.EPS))
    h = _conv(rd, stage, sd, p + "conv1.", h, 1)
    h = F.silu(F.group_norm(h, vo.GROUPS, sd[p + "norm2.weight"], sd[p + "norm2.bias"], vo.EPS))
    h = _conv(rd, stage, sd, p + "conv2.", h, 1)
    if p + "conv_shortcut.weight" in sd:
        x = _conv(rd, stage, sd, p + "conv_shortcut.", x, 0)
    return x + h


def _attention(rd, stage, sd, p, x):
    n, c, hh, ww = x.shape
    h = F.group_norm(x, vo.GROUPS, sd[p + "group_norm.weight"], sd[p + "group_norm.bias"], vo.EPS)
    h = rd.a(h.view(n, c, hh * ww).transpose(1, 2), stage)
    q, k, v = (rd.a(F.linear(h, rd.w(sd[p + f"to_{n_}.weight"]), sd[p + f"to_{n_}.bias"]), stage) for n_ in "qkv")
    a = rd.a(torch.softmax(q @ k.transpose(1, 2) * (c ** -0.5), dim=-1), stage)
    o = F.linear(rd.a(a @ v, stage), rd.w(sd[p + "to_out.0.weight"]), sd[p + "to_out.0.bias"])
    return o.transpose(1, 2).reshape(n, c, hh, ww) + x


def decode_emulated(sd, z, rd, block_out=vo.BLOCK_OUT, layers=vo.LAYERS_PER_BLOCK):
    with torch.no_grad():
        x = F.conv2d(z, sd["post_quant_conv.weight"], sd["post_quant_conv.bias"])
        x = _conv(rd, "in", sd, "decoder.conv_in.", x, 1)
        x = _resnet(rd, "mid", sd, "decoder.mid_block.resnets.0.", x)
        x = _attention(rd, "mid", sd, "decoder.mid_block.attentions.0.", x)
        x = _resnet(rd, "mid", sd, "decoder.mid_block.resnets.1.", x)
        nb = len(block_out)
        for i in range(nb):
            for r in range(layers + 1):
                x = _resnet(rd, f"up{i}", sd, f"decoder.up_blocks.{i}.resnets.{r}.", x)
            if i != nb - 1:
                x = F.interpolate(x, scale_factor=2.0, mode="nearest")
                x = _conv(rd, f"up{i}", sd, f"decoder.up_blocks.{i}.upsamplers.0.conv.", x, 1)
        x = F.silu(F.group_norm(x, vo.GROUPS, sd["decoder.conv_norm_out.weight"], sd["decoder.conv_norm_out.bias"], vo.EPS))
        return _conv(rd, "out", sd, "decoder.conv_out.", x, 1)


def rel_l2(a, b):
    return float((a - b).norm() / b.norm())


def budget_table(weight_seed=1, latent_seed=5, size=16, frames=1):
    sd = vo.init_state_dict(seed=weight_seed)
    z = torch.randn(frames, 4, size, size, generator=torch.Generator().manual_seed(latent_seed))
    ref = vo.decode(sd, z)
    rows = {}
    rows["activations + weights f16 (the engine)"] = rel_l2(decode_emulated(sd, z, Rounding()), ref)
    rows["activations f16 only"] = rel_l2(decode_emulated(sd, z, Rounding(weights=False)), ref)
    rows["weights f16 only"] = rel_l2(decode_emulated(sd, z, Rounding(acts=False)), ref)
    for st in ("in", "mid", "up0", "up1", "up2", "up3", "out"):
        rows[f"engine, activation operand of '{st}' exact"] = rel_l2(decode_emulated(sd, z, Rounding(exact_stages=(st,))), ref)
    rows["engine, activation operand of mid + up0 exact"] = rel_l2(decode_emulated(sd, z, Rounding(exact_stages=("mid", "up0"))), ref)
    rows["engine, activation operand of up2 + up3 + out exact"] = rel_l2(decode_emulated(sd, z, Rounding(exact_stages=("up2", "up3", "out"))), ref)
    return rows


if __name__ == "__main__":
    kw = {}
    if len(sys.argv) > 1:
        kw["weight_seed"] = int(sys.argv[1])
    if len(sys.argv) > 2:
        kw["latent_seed"] = int(sys.argv[2])
    for k, v in budget_table(**kw).items():
        print(f"{v:.3e}  {k}")
