"""Stand-in for the pieces of diffusers==0.24.0 that /root/reference/models/latte_t2v.py imports.

TEST INFRASTRUCTURE ONLY, and MEMORY-DERIVED: diffusers is pinned by the reference (environment.yml:12) but is neither
vendored under /root/reference nor installable here, so the classes below restate the PUBLISHED 0.24.0 behaviour of the
few modules the Latte-1 text-to-video model touches (PixArt-alpha style blocks).  They exist so that the reference's own
file can be loaded UNMODIFIED (oracle/reference_loader.load_reference_latte_t2v) and its glue -- frame / token
rearranges, temporal blocks (defined in the reference file itself), adaLN-single, the output head -- can pin
oracle/latte_t2v_oracle.py.  Agreement of THESE classes with real diffusers is unverified ("parity unpinned" for every
leaf they restate: Attention, BasicTransformerBlock, PatchEmbed, CaptionProjection, CombinedTimestepSizeEmbeddings,
GELU).  The temporal block `BasicTransformerBlock_` (latte_t2v.py:126-396) is a copy of the same diffusers block and was
used to cross-check the spatial block written here.
"""
import math
import sys
import types
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn


# ----------------------------------------------------------------------------- utils / config plumbing
class BaseOutput(OrderedDict):
    """diffusers.utils.BaseOutput: a dataclass that is also a dict / tuple.  Only attribute access is needed here."""

    def __init__(self, *a, **k):
        super().__init__(*a, **k)

    def __post_init__(self):
        for f in getattr(self, "__dataclass_fields__", {}):
            self[f] = getattr(self, f)


def deprecate(*a, **k):
    return None


def maybe_allow_in_graph(cls):
    return cls


class _Config(dict):
    __getattr__ = dict.__getitem__


def register_to_config(init):
    """configuration_utils.register_to_config: record the constructor arguments in ``self.config`` BEFORE the body runs
    (latte_t2v.py reads ``self.config.sample_size`` inside ``__init__``)."""
    import functools
    import inspect

    @functools.wraps(init)
    def wrapper(self, *args, **kwargs):
        sig = inspect.signature(init)
        bound = sig.bind(self, *args, **kwargs)
        bound.apply_defaults()
        cfg = _Config({k: v for k, v in bound.arguments.items() if k != "self"})
        object.__setattr__(self, "_internal_config", cfg)
        init(self, *args, **kwargs)

    return wrapper


class ConfigMixin:
    @property
    def config(self):
        return self._internal_config


class ModelMixin(nn.Module):
    pass


class LoRACompatibleLinear(nn.Linear):
    def forward(self, x, scale: float = 1.0):
        return super().forward(x)


class LoRACompatibleConv(nn.Conv2d):
    def forward(self, x, scale: float = 1.0):
        return super().forward(x)


# ----------------------------------------------------------------------------- embeddings
def get_1d_sincos_pos_embed_from_grid(embed_dim, pos):
    """embeddings.get_1d_sincos_pos_embed_from_grid: [sin | cos], omega_i = 10000^(-i / (D/2)), fp64."""
    omega = np.arange(embed_dim // 2, dtype=np.float64)
    omega /= embed_dim / 2.0
    omega = 1.0 / 10000 ** omega
    pos = np.asarray(pos).reshape(-1)
    out = np.einsum("m,d->md", pos, omega)
    return np.concatenate([np.sin(out), np.cos(out)], axis=1)


def get_2d_sincos_pos_embed(embed_dim, grid_size, interpolation_scale=1.0, base_size=16):
    if isinstance(grid_size, int):
        grid_size = (grid_size, grid_size)
    grid_h = np.arange(grid_size[0], dtype=np.float32) / (grid_size[0] / base_size) / interpolation_scale
    grid_w = np.arange(grid_size[1], dtype=np.float32) / (grid_size[1] / base_size) / interpolation_scale
    grid = np.stack(np.meshgrid(grid_w, grid_h), axis=0).reshape([2, 1, grid_size[1], grid_size[0]])
    emb_h = get_1d_sincos_pos_embed_from_grid(embed_dim // 2, grid[0])
    emb_w = get_1d_sincos_pos_embed_from_grid(embed_dim // 2, grid[1])
    return np.concatenate([emb_h, emb_w], axis=1)


class PatchEmbed(nn.Module):
    """embeddings.PatchEmbed (0.24.0): Conv2d(k = s = patch) -> flatten -> + fixed 2-D sin-cos positions."""

    def __init__(self, height=224, width=224, patch_size=16, in_channels=3, embed_dim=768, layer_norm=False, flatten=True,
                 bias=True, interpolation_scale=1):
        super().__init__()
        num_patches = (height // patch_size) * (width // patch_size)
        self.flatten, self.layer_norm = flatten, layer_norm
        self.proj = nn.Conv2d(in_channels, embed_dim, kernel_size=(patch_size, patch_size), stride=patch_size, bias=bias)
        self.norm = nn.LayerNorm(embed_dim, elementwise_affine=False, eps=1e-6) if layer_norm else None
        self.patch_size = patch_size
        self.height, self.width = height // patch_size, width // patch_size
        self.base_size = height // patch_size
        self.interpolation_scale = interpolation_scale
        pe = get_2d_sincos_pos_embed(embed_dim, int(num_patches ** 0.5), base_size=self.base_size,
                                     interpolation_scale=self.interpolation_scale)
        self.register_buffer("pos_embed", torch.from_numpy(pe).float().unsqueeze(0), persistent=False)

    def forward(self, latent):
        height, width = latent.shape[-2] // self.patch_size, latent.shape[-1] // self.patch_size
        latent = self.proj(latent)
        if self.flatten:
            latent = latent.flatten(2).transpose(1, 2)
        if self.layer_norm:
            latent = self.norm(latent)
        if self.height != height or self.width != width:
            pe = get_2d_sincos_pos_embed(self.pos_embed.shape[-1], (height, width), base_size=self.base_size,
                                         interpolation_scale=self.interpolation_scale)
            pos_embed = torch.from_numpy(pe).float().unsqueeze(0).to(latent.device)
        else:
            pos_embed = self.pos_embed
        return (latent + pos_embed).to(latent.dtype)


class CaptionProjection(nn.Module):
    """embeddings.CaptionProjection: Linear -> GELU(tanh) -> Linear on the T5 token features."""

    def __init__(self, in_features, hidden_size, num_tokens=120):
        super().__init__()
        self.linear_1 = nn.Linear(in_features, hidden_size, bias=True)
        self.act_1 = nn.GELU(approximate="tanh")
        self.linear_2 = nn.Linear(hidden_size, hidden_size, bias=True)
        self.register_buffer("y_embedding", nn.Parameter(torch.randn(num_tokens, in_features) / in_features ** 0.5))

    def forward(self, caption, force_drop_ids=None):
        return self.linear_2(self.act_1(self.linear_1(caption)))


def get_timestep_embedding(timesteps, embedding_dim, flip_sin_to_cos=False, downscale_freq_shift=1, scale=1, max_period=10000):
    half_dim = embedding_dim // 2
    exponent = -math.log(max_period) * torch.arange(start=0, end=half_dim, dtype=torch.float32, device=timesteps.device)
    exponent = exponent / (half_dim - downscale_freq_shift)
    emb = torch.exp(exponent)
    emb = timesteps[:, None].float() * emb[None, :]
    emb = scale * emb
    emb = torch.cat([torch.sin(emb), torch.cos(emb)], dim=-1)
    if flip_sin_to_cos:
        emb = torch.cat([emb[:, half_dim:], emb[:, :half_dim]], dim=-1)
    return emb


class Timesteps(nn.Module):
    def __init__(self, num_channels, flip_sin_to_cos, downscale_freq_shift):
        super().__init__()
        self.num_channels, self.flip_sin_to_cos, self.downscale_freq_shift = num_channels, flip_sin_to_cos, downscale_freq_shift

    def forward(self, timesteps):
        return get_timestep_embedding(timesteps, self.num_channels, self.flip_sin_to_cos, self.downscale_freq_shift)


class TimestepEmbedding(nn.Module):
    def __init__(self, in_channels, time_embed_dim):
        super().__init__()
        self.linear_1 = nn.Linear(in_channels, time_embed_dim)
        self.act = nn.SiLU()
        self.linear_2 = nn.Linear(time_embed_dim, time_embed_dim)

    def forward(self, sample):
        return self.linear_2(self.act(self.linear_1(sample)))


class CombinedTimestepSizeEmbeddings(nn.Module):
    """embeddings.CombinedTimestepSizeEmbeddings without the additional (resolution / aspect ratio) conditions, which
    PixArt-alpha only enables at sample_size 128 (latte_t2v.py:655)."""

    def __init__(self, embedding_dim, size_emb_dim, use_additional_conditions=False):
        super().__init__()
        if use_additional_conditions:
            raise NotImplementedError("stand-in: additional conditions (sample_size 128) are not restated")
        self.outdim = size_emb_dim
        self.time_proj = Timesteps(num_channels=256, flip_sin_to_cos=True, downscale_freq_shift=0)
        self.timestep_embedder = TimestepEmbedding(in_channels=256, time_embed_dim=embedding_dim)
        self.use_additional_conditions = False

    def forward(self, timestep, resolution, aspect_ratio, batch_size, hidden_dtype):
        return self.timestep_embedder(self.time_proj(timestep).to(dtype=hidden_dtype))


class _Unused(nn.Module):
    def __init__(self, *a, **k):
        raise NotImplementedError("stand-in: this diffusers class is imported by latte_t2v.py but not used by Latte-1")


# ----------------------------------------------------------------------------- activations / attention
class GELU(nn.Module):
    def __init__(self, dim_in, dim_out, approximate="none"):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out)
        self.approximate = approximate

    def forward(self, hidden_states):
        return F.gelu(self.proj(hidden_states), approximate=self.approximate)


class GEGLU(nn.Module):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = LoRACompatibleLinear(dim_in, dim_out * 2)

    def forward(self, hidden_states, scale: float = 1.0):
        hidden_states, gate = self.proj(hidden_states).chunk(2, dim=-1)
        return hidden_states * F.gelu(gate)


class ApproximateGELU(nn.Module):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out)

    def forward(self, x):
        x = self.proj(x)
        return x * torch.sigmoid(1.702 * x)


class Attention(nn.Module):
    """attention_processor.Attention with the default AttnProcessor2_0: q / k / v projections (bias = attention_bias),
    heads of dim_head, softmax(q k^T * dim_head^-0.5 + additive mask) v, to_out[0] with bias."""

    def __init__(self, query_dim, cross_attention_dim=None, heads=8, dim_head=64, dropout=0.0, bias=False,
                 upcast_attention=False, **unused):
        super().__init__()
        self.inner_dim = dim_head * heads
        self.heads = heads
        self.scale = dim_head ** -0.5
        kv_dim = cross_attention_dim if cross_attention_dim is not None else query_dim
        self.to_q = LoRACompatibleLinear(query_dim, self.inner_dim, bias=bias)
        self.to_k = LoRACompatibleLinear(kv_dim, self.inner_dim, bias=bias)
        self.to_v = LoRACompatibleLinear(kv_dim, self.inner_dim, bias=bias)
        self.to_out = nn.ModuleList([LoRACompatibleLinear(self.inner_dim, query_dim, bias=True), nn.Dropout(dropout)])

    def prepare_attention_mask(self, attention_mask, target_length, batch_size):
        if attention_mask is None:
            return None
        if attention_mask.shape[-1] != target_length:
            attention_mask = F.pad(attention_mask, (0, target_length), value=0.0)
        if attention_mask.shape[0] < batch_size * self.heads:
            attention_mask = attention_mask.repeat_interleave(self.heads, dim=0)
        return attention_mask

    def forward(self, hidden_states, encoder_hidden_states=None, attention_mask=None, **unused):
        batch_size, sequence_length, _ = (hidden_states.shape if encoder_hidden_states is None else encoder_hidden_states.shape)
        if attention_mask is not None:
            attention_mask = self.prepare_attention_mask(attention_mask, sequence_length, batch_size)
            attention_mask = attention_mask.view(batch_size, self.heads, -1, attention_mask.shape[-1])
        query = self.to_q(hidden_states)
        enc = hidden_states if encoder_hidden_states is None else encoder_hidden_states
        key, value = self.to_k(enc), self.to_v(enc)
        head_dim = self.inner_dim // self.heads
        query = query.view(batch_size, -1, self.heads, head_dim).transpose(1, 2)
        key = key.view(batch_size, -1, self.heads, head_dim).transpose(1, 2)
        value = value.view(batch_size, -1, self.heads, head_dim).transpose(1, 2)
        out = F.scaled_dot_product_attention(query, key, value, attn_mask=attention_mask, dropout_p=0.0, is_causal=False)
        out = out.transpose(1, 2).reshape(batch_size, -1, self.heads * head_dim).to(query.dtype)
        return self.to_out[1](self.to_out[0](out))


class _FeedForward(nn.Module):
    """attention.FeedForward (same as the copy at latte_t2v.py:69-124)."""

    def __init__(self, dim, dim_out=None, mult=4, dropout=0.0, activation_fn="geglu", final_dropout=False):
        super().__init__()
        inner_dim = int(dim * mult)
        dim_out = dim_out if dim_out is not None else dim
        if activation_fn == "gelu":
            act_fn = GELU(dim, inner_dim)
        if activation_fn == "gelu-approximate":
            act_fn = GELU(dim, inner_dim, approximate="tanh")
        elif activation_fn == "geglu":
            act_fn = GEGLU(dim, inner_dim)
        elif activation_fn == "geglu-approximate":
            act_fn = ApproximateGELU(dim, inner_dim)
        self.net = nn.ModuleList([act_fn, nn.Dropout(dropout), LoRACompatibleLinear(inner_dim, dim_out)])
        if final_dropout:
            self.net.append(nn.Dropout(dropout))

    def forward(self, hidden_states, scale: float = 1.0):
        for module in self.net:
            hidden_states = module(hidden_states, scale) if isinstance(module, (GEGLU, LoRACompatibleLinear)) else module(hidden_states)
        return hidden_states


class BasicTransformerBlock(nn.Module):
    """attention.BasicTransformerBlock (0.24.0), restricted to what the spatial blocks of LatteT2V construct
    (latte_t2v.py:589-606): norm_type 'ada_norm_single' (PixArt-alpha): self-attention gated by the adaLN-single table,
    cross-attention on the un-normalised stream (no norm2 there), norm2 + modulate + feed-forward, gated."""

    def __init__(self, dim, num_attention_heads, attention_head_dim, dropout=0.0, cross_attention_dim=None,
                 activation_fn="geglu", num_embeds_ada_norm=None, attention_bias=False, only_cross_attention=False,
                 double_self_attention=False, upcast_attention=False, norm_elementwise_affine=True, norm_type="layer_norm",
                 norm_eps=1e-5, final_dropout=False, attention_type="default", positional_embeddings=None,
                 num_positional_embeddings=None):
        super().__init__()
        if norm_type != "ada_norm_single" or only_cross_attention or double_self_attention or attention_type != "default" \
                or positional_embeddings is not None:
            raise NotImplementedError("stand-in: only the PixArt-alpha (ada_norm_single) block of Latte-1 is restated")
        self.norm1 = nn.LayerNorm(dim, elementwise_affine=norm_elementwise_affine, eps=norm_eps)
        self.attn1 = Attention(query_dim=dim, heads=num_attention_heads, dim_head=attention_head_dim, dropout=dropout,
                               bias=attention_bias, cross_attention_dim=None, upcast_attention=upcast_attention)
        if cross_attention_dim is not None:
            self.norm2 = nn.LayerNorm(dim, elementwise_affine=norm_elementwise_affine, eps=norm_eps)
            self.attn2 = Attention(query_dim=dim, cross_attention_dim=cross_attention_dim, heads=num_attention_heads,
                                   dim_head=attention_head_dim, dropout=dropout, bias=attention_bias,
                                   upcast_attention=upcast_attention)
        else:
            self.norm2, self.attn2 = None, None
        self.ff = _FeedForward(dim, dropout=dropout, activation_fn=activation_fn, final_dropout=final_dropout)
        self.scale_shift_table = nn.Parameter(torch.randn(6, dim) / dim ** 0.5)

    def forward(self, hidden_states, attention_mask=None, encoder_hidden_states=None, encoder_attention_mask=None,
                timestep=None, cross_attention_kwargs=None, class_labels=None):
        batch_size = hidden_states.shape[0]
        shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp = (
            self.scale_shift_table[None] + timestep.reshape(batch_size, 6, -1)).chunk(6, dim=1)
        norm_hidden_states = self.norm1(hidden_states)
        norm_hidden_states = norm_hidden_states * (1 + scale_msa) + shift_msa
        attn_output = self.attn1(norm_hidden_states, encoder_hidden_states=None, attention_mask=attention_mask)
        hidden_states = gate_msa * attn_output + hidden_states
        if self.attn2 is not None:
            attn_output = self.attn2(hidden_states, encoder_hidden_states=encoder_hidden_states,
                                     attention_mask=encoder_attention_mask)          # PixArt: norm2 is not applied here
            hidden_states = attn_output + hidden_states
        norm_hidden_states = self.norm2(hidden_states)
        norm_hidden_states = norm_hidden_states * (1 + scale_mlp) + shift_mlp
        ff_output = self.ff(norm_hidden_states)
        return gate_mlp * ff_output + hidden_states


# ----------------------------------------------------------------------------- module tree
def install():
    """Register the stand-in package tree under ``diffusers`` (no-op when a real diffusers is importable)."""
    if "diffusers" in sys.modules:
        return
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m
    mod("diffusers", __standin__=True)
    mod("diffusers.models", Transformer2DModel=_Unused)
    mod("diffusers.utils", USE_PEFT_BACKEND=False, BaseOutput=BaseOutput, deprecate=deprecate)
    mod("diffusers.utils.torch_utils", maybe_allow_in_graph=maybe_allow_in_graph)
    mod("diffusers.models.embeddings", get_1d_sincos_pos_embed_from_grid=get_1d_sincos_pos_embed_from_grid,
        ImagePositionalEmbeddings=_Unused, CaptionProjection=CaptionProjection, PatchEmbed=PatchEmbed,
        CombinedTimestepSizeEmbeddings=CombinedTimestepSizeEmbeddings, SinusoidalPositionalEmbedding=_Unused)
    mod("diffusers.configuration_utils", ConfigMixin=ConfigMixin, register_to_config=register_to_config)
    mod("diffusers.models.modeling_utils", ModelMixin=ModelMixin)
    mod("diffusers.models.attention", BasicTransformerBlock=BasicTransformerBlock)
    mod("diffusers.models.lora", LoRACompatibleConv=LoRACompatibleConv, LoRACompatibleLinear=LoRACompatibleLinear)
    mod("diffusers.models.normalization", AdaLayerNorm=_Unused, AdaLayerNormZero=_Unused)
    mod("diffusers.models.attention_processor", Attention=Attention)
    mod("diffusers.models.activations", GEGLU=GEGLU, GELU=GELU, ApproximateGELU=ApproximateGELU)
