"""CPU restatement of the reference Latte denoiser (fp32, plain torch ops).

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).  Parity status: PINNED against the
reference itself — ``oracle/validate_oracle.py`` runs ``/root/reference/models/latte.py``
unmodified (timm stand-in) and this file on identical weights/inputs; results are recorded in
``oracle/VALIDATION.md`` and golden vectors produced by the reference live in ``tests/golden/``.

All ``latte.py:N`` citations are ``/root/reference/models/latte.py``.  The model is a pure
function of a reference-format ``state_dict`` (key names of SURVEY.md §8(b)) — no nn.Module.
"""
import math
from dataclasses import dataclass

import numpy as np
import torch
import torch.nn.functional as F


@dataclass
class LatteConfig:
    """Mirror of ``Latte.__init__`` arguments (latte.py:208-223)."""
    input_size: int = 32
    patch_size: int = 2
    in_channels: int = 4
    hidden_size: int = 1152
    depth: int = 28
    num_heads: int = 16
    mlp_ratio: float = 4.0
    num_frames: int = 16
    num_classes: int = 1000
    learn_sigma: bool = True
    extras: int = 1

    @property
    def out_channels(self):
        return self.in_channels * 2 if self.learn_sigma else self.in_channels

    @property
    def num_patches(self):
        return (self.input_size // self.patch_size) ** 2


# latte.py:464-499 — (depth, hidden, heads) per family; patch size from the suffix.
PRESETS = {"XL": (28, 1152, 16), "L": (24, 1024, 16), "B": (12, 768, 12), "S": (12, 384, 6)}


def preset_config(name: str, **kw) -> LatteConfig:
    fam, patch = name.replace("Latte-", "").split("/")
    depth, hidden, heads = PRESETS[fam]
    return LatteConfig(patch_size=int(patch), hidden_size=hidden, depth=depth, num_heads=heads, **kw)


# ----------------------------------------------------------------------------- embeddings
def sincos_1d(embed_dim: int, pos: np.ndarray) -> np.ndarray:
    """latte.py:438-457: omega_i = 10000^(-i/(d/2)), out = [sin(pos*omega) | cos(pos*omega)], fp64."""
    omega = np.arange(embed_dim // 2, dtype=np.float64)
    omega /= embed_dim / 2.0
    omega = 1.0 / 10000 ** omega
    out = np.einsum("m,d->md", pos.reshape(-1), omega)
    return np.concatenate([np.sin(out), np.cos(out)], axis=1)


def pos_embed_2d(embed_dim: int, grid_size: int) -> np.ndarray:
    """latte.py:410-436: meshgrid(grid_w, grid_h) with w first; first half of dims encodes grid[0]."""
    gh = np.arange(grid_size, dtype=np.float32)
    gw = np.arange(grid_size, dtype=np.float32)
    grid = np.stack(np.meshgrid(gw, gh), axis=0).reshape([2, 1, grid_size, grid_size])
    emb_a = sincos_1d(embed_dim // 2, grid[0])
    emb_b = sincos_1d(embed_dim // 2, grid[1])
    return np.concatenate([emb_a, emb_b], axis=1)


def temp_embed_1d(embed_dim: int, length: int) -> np.ndarray:
    """latte.py:406-408 (positions are an int64 torch arange there; values identical)."""
    return sincos_1d(embed_dim, np.arange(length, dtype=np.float64))


def timestep_embedding(t: torch.Tensor, dim: int = 256, max_period: float = 10000.0) -> torch.Tensor:
    """latte.py:97-117: cos first, then sin; freqs computed in fp32."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(0, half, dtype=torch.float32) / half)
    args = t[:, None].float() * freqs[None]
    return torch.cat([torch.cos(args), torch.sin(args)], dim=-1)


# ----------------------------------------------------------------------------- weights
TEXT_TOKENS, TEXT_DIM = 77, 768          # latte.py:241: nn.Linear(77 * 768, hidden_size)


def text_projection_weight(D: int, seed: int = 0, scale: float = 0.01) -> torch.Tensor:
    """[D, 77*768] weight of ``text_embedding_projection`` (latte.py:238-242) from a closed-form integer hash, so that
    golden fixtures need not store 30+ MB: value(i) = (lcg(i + seed * 2^32) >> 40) / 2^23 - 1, times ``scale``.
    uint64 wrap-around arithmetic in numpy is exact and platform independent."""
    n = D * TEXT_TOKENS * TEXT_DIM
    i = np.arange(n, dtype=np.uint64) + np.uint64(seed) * np.uint64(1 << 32)
    h = i * np.uint64(6364136223846793005) + np.uint64(1442695040888963407)
    h ^= h >> np.uint64(29)
    h = h * np.uint64(0xBF58476D1CE4E5B9)
    v = (h >> np.uint64(40)).astype(np.float64) / float(1 << 23) - 1.0
    return torch.from_numpy((v * scale).astype(np.float32)).reshape(D, TEXT_TOKENS * TEXT_DIM)


def init_state_dict(cfg: LatteConfig, seed: int = 0, gate_std: float = 0.02, final_std: float = None,
                    v_scale: float = 1.0) -> dict:
    """Synthetic reference-format weights.  Same *distributions* as ``initialize_weights``
    (latte.py:257-295) but NOT the same RNG stream; every tensor the reference zero-inits
    (adaLN modulation, final layer) is drawn N(0, gate_std) so outputs are non-trivial.

    ``final_std`` (default: gate_std) is the scale of ``final_layer.linear.{weight,bias}`` alone and ``v_scale`` multiplies the
    rows of that layer which produce the learned-range variance channels v (unpatchify's last index c >= in_channels,
    latte.py:297-310, split at gaussian_diffusion.py:291).  A trained checkpoint predicts eps of rms ~ 1 and v inside [-1, 1]
    (frac = (v + 1) / 2 interpolates two log-variances, gaussian_diffusion.py:292-297); gate_std 0.3 on the final layer gives rms ~ 10
    for both, which makes the last DDPM step exponentiate far outside that range (profiles/r5_ddpm_conditioning.log).  Both
    are post-scalings of the same draws, so the RNG stream -- and every fixture generated with the defaults -- is unchanged."""
    g = torch.Generator("cpu").manual_seed(seed)
    D, p, C = cfg.hidden_size, cfg.patch_size, cfg.in_channels
    Hm = int(D * cfg.mlp_ratio)

    def xavier(out_f, in_f):
        a = math.sqrt(6.0 / (in_f + out_f))
        return (torch.rand(out_f, in_f, generator=g) * 2 - 1) * a

    def normal(*shape, std=0.02):
        return torch.randn(*shape, generator=g) * std

    sd = {}
    sd["pos_embed"] = torch.from_numpy(pos_embed_2d(D, cfg.input_size // p)).float().unsqueeze(0)
    sd["temp_embed"] = torch.from_numpy(temp_embed_1d(D, cfg.num_frames)).float().unsqueeze(0)
    sd["x_embedder.proj.weight"] = xavier(D, C * p * p).view(D, C, p, p)
    sd["x_embedder.proj.bias"] = normal(D)
    sd["t_embedder.mlp.0.weight"] = normal(D, 256)
    sd["t_embedder.mlp.0.bias"] = normal(D)
    sd["t_embedder.mlp.2.weight"] = normal(D, D)
    sd["t_embedder.mlp.2.bias"] = normal(D)
    if cfg.extras == 2:
        sd["y_embedder.embedding_table.weight"] = normal(cfg.num_classes + 1, D)
    if cfg.extras == 78:
        sd["text_embedding_projection.1.weight"] = text_projection_weight(D, seed)
        sd["text_embedding_projection.1.bias"] = normal(D)
    for i in range(cfg.depth):
        pre = f"blocks.{i}."
        sd[pre + "attn.qkv.weight"] = xavier(3 * D, D)
        sd[pre + "attn.qkv.bias"] = normal(3 * D)
        sd[pre + "attn.proj.weight"] = xavier(D, D)
        sd[pre + "attn.proj.bias"] = normal(D)
        sd[pre + "mlp.fc1.weight"] = xavier(Hm, D)
        sd[pre + "mlp.fc1.bias"] = normal(Hm)
        sd[pre + "mlp.fc2.weight"] = xavier(D, Hm)
        sd[pre + "mlp.fc2.bias"] = normal(D)
        sd[pre + "adaLN_modulation.1.weight"] = normal(6 * D, D, std=gate_std)
        sd[pre + "adaLN_modulation.1.bias"] = normal(6 * D, std=gate_std)
    fstd = gate_std if final_std is None else final_std
    sd["final_layer.linear.weight"] = normal(p * p * cfg.out_channels, D, std=fstd)
    sd["final_layer.linear.bias"] = normal(p * p * cfg.out_channels, std=fstd)
    if v_scale != 1.0 and cfg.out_channels > C:
        vrows = (torch.arange(p * p * cfg.out_channels) % cfg.out_channels) >= C      # row = (p q) * out_channels + c
        sd["final_layer.linear.weight"][vrows] *= v_scale
        sd["final_layer.linear.bias"][vrows] *= v_scale
    sd["final_layer.adaLN_modulation.1.weight"] = normal(2 * D, D, std=gate_std)
    sd["final_layer.adaLN_modulation.1.bias"] = normal(2 * D, std=gate_std)
    return sd


# ----------------------------------------------------------------------------- forward
def _modulate(x, shift, scale):
    """latte.py:28-29."""
    return x * (1 + scale.unsqueeze(1)) + shift.unsqueeze(1)


def _attention(sd, pre, x, num_heads):
    """latte.py:48-77, attention_mode='math' (the default selected by get_models)."""
    S, L, D = x.shape
    hd = D // num_heads
    qkv = F.linear(x, sd[pre + "attn.qkv.weight"], sd[pre + "attn.qkv.bias"])
    qkv = qkv.reshape(S, L, 3, num_heads, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    attn = (q @ k.transpose(-2, -1)) * (hd ** -0.5)      # scale applied after the product (:67)
    attn = attn.softmax(dim=-1)
    o = (attn @ v).transpose(1, 2).reshape(S, L, D)
    return F.linear(o, sd[pre + "attn.proj.weight"], sd[pre + "attn.proj.bias"])


def _block(sd, i, x, c, num_heads):
    """TransformerBlock.forward, latte.py:177-181 (adaLN-Zero; LN eps 1e-6, no affine :166,:168;
    MLP = fc1 -> GELU(tanh) -> fc2, :170-171)."""
    pre = f"blocks.{i}."
    D = x.shape[-1]
    mod = F.linear(F.silu(c), sd[pre + "adaLN_modulation.1.weight"], sd[pre + "adaLN_modulation.1.bias"])
    sh1, sc1, g1, sh2, sc2, g2 = mod.chunk(6, dim=1)
    h = _modulate(F.layer_norm(x, (D,), eps=1e-6), sh1, sc1)
    x = x + g1.unsqueeze(1) * _attention(sd, pre, h, num_heads)
    h = _modulate(F.layer_norm(x, (D,), eps=1e-6), sh2, sc2)
    h = F.linear(h, sd[pre + "mlp.fc1.weight"], sd[pre + "mlp.fc1.bias"])
    h = F.gelu(h, approximate="tanh")
    h = F.linear(h, sd[pre + "mlp.fc2.weight"], sd[pre + "mlp.fc2.bias"])
    return x + g2.unsqueeze(1) * h


def latte_forward(sd: dict, cfg: LatteConfig, x: torch.Tensor, t: torch.Tensor, y=None, text_embedding=None) -> torch.Tensor:
    """``Latte.forward`` (latte.py:314-377), fp32.  x:[B,F,C,H,W], t:int64[B], y:int64[B]|None,
    text_embedding:[B,77,768]|None (extras == 78)  -> [B,F,out_channels,H,W]."""
    B, Fr, C, H, W = x.shape
    p, D = cfg.patch_size, cfg.hidden_size
    T = (H // p) * (W // p)
    xf = x.reshape(B * Fr, C, H, W).float()
    tok = F.conv2d(xf, sd["x_embedder.proj.weight"], sd["x_embedder.proj.bias"], stride=p)   # :331
    tok = tok.flatten(2).transpose(1, 2) + sd["pos_embed"]
    temb = timestep_embedding(t, 256)                                                       # :332
    temb = F.linear(temb, sd["t_embedder.mlp.0.weight"], sd["t_embedder.mlp.0.bias"])
    temb = F.linear(F.silu(temb), sd["t_embedder.mlp.2.weight"], sd["t_embedder.mlp.2.bias"])
    c = c_final = temb
    if cfg.extras == 2:
        c = c_final = temb + sd["y_embedder.embedding_table.weight"][y]                     # :337,:348,:370-371
    elif cfg.extras == 78:
        # :238-242,:341: Sequential(SiLU, Linear(77*768, D)) on the flattened text embedding; the blocks see
        # t + text (:350,:363), the final layer sees t only (:372-373)
        txt = F.linear(F.silu(text_embedding.reshape(B, -1).float()), sd["text_embedding_projection.1.weight"],
                       sd["text_embedding_projection.1.bias"])
        c = temb + txt
    c_spatial = c.repeat_interleave(Fr, dim=0)        # 'n d -> (n c) d', c=frames   (:333)
    c_temp = c.repeat_interleave(T, dim=0)            # 'n d -> (n c) d', c=tokens   (:334)
    h = tok
    for i in range(0, cfg.depth, 2):
        h = _block(sd, i, h, c_spatial, cfg.num_heads)                                       # :353
        h = h.reshape(B, Fr, T, D).permute(0, 2, 1, 3).reshape(B * T, Fr, D)                 # :355
        if i == 0:
            h = h + sd["temp_embed"]                                                         # :357-358
        h = _block(sd, i + 1, h, c_temp, cfg.num_heads)                                      # :367
        h = h.reshape(B, T, Fr, D).permute(0, 2, 1, 3).reshape(B * Fr, T, D)                 # :368
    # FinalLayer (latte.py:197-201); conditioning is t (+y), never the text embedding (:370-373)
    mod = F.linear(F.silu(c_final.repeat_interleave(Fr, dim=0)), sd["final_layer.adaLN_modulation.1.weight"],
                   sd["final_layer.adaLN_modulation.1.bias"])
    shift, scale = mod.chunk(2, dim=1)
    h = _modulate(F.layer_norm(h, (D,), eps=1e-6), shift, scale)
    h = F.linear(h, sd["final_layer.linear.weight"], sd["final_layer.linear.bias"])
    # unpatchify (latte.py:297-310): (N, T, p*p*c) -> (N, c, H, W), einsum 'nhwpqc->nchpwq'
    co = cfg.out_channels
    gh = H // p
    h = h.reshape(B * Fr, gh, gh, p, p, co).permute(0, 5, 1, 3, 2, 4).reshape(B * Fr, co, gh * p, gh * p)
    return h.reshape(B, Fr, co, H, W)


def latte_forward_with_cfg(sd, cfg, x, t, y, cfg_scale, text_embedding=None):
    """``Latte.forward_with_cfg`` (latte.py:379-398): first half duplicated, guidance on the first
    4 channels only (hard-coded 4 at :394), variance channels of both halves kept."""
    half = x[: len(x) // 2]
    combined = torch.cat([half, half], dim=0)
    out = latte_forward(sd, cfg, combined, t, y, text_embedding)
    eps, rest = out[:, :, :4], out[:, :, 4:]
    cond, uncond = torch.split(eps, len(eps) // 2, dim=0)
    half_eps = uncond + cfg_scale * (cond - uncond)
    return torch.cat([torch.cat([half_eps, half_eps], dim=0), rest], dim=2)
