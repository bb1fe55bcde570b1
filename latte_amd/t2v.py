"""Host-side mirror of ``models.latte_t2v.LatteT2V`` (Latte-1 text-to-video denoiser) for the call sites of
``sample/pipeline_latte.py`` (:738-746: ``self.transformer(latent_model_input, encoder_hidden_states=prompt_embeds,
timestep=current_timestep, added_cond_kwargs=..., enable_temporal_attentions=..., return_dict=False)[0]``).

SURVEY.md section 8(f) rank 2 -- the row after the class-conditional sampling path.  Only the denoiser forward runs on the
MI355X engine (``latte_t2v_*`` in include/latte_amd.h); the T5 text encoder, the diffusers schedulers and the temporal
VAE decoder of ``LattePipeline`` are not part of it yet.  The configuration is the one Latte-1 ships (PixArt-alpha blocks:
``norm_type='ada_norm_single'``, ``attention_bias=True``, ``activation_fn='gelu-approximate'``).  There is no CPU fallback.
"""
import json
import os
from types import SimpleNamespace

import torch

from . import _lib
from ._lib import LatteError, check, load_library, ptr, stream_ptr


class Transformer3DModelOutput:
    def __init__(self, sample):
        self.sample = sample


class LatteT2V:
    def __init__(self, num_attention_heads=16, attention_head_dim=72, in_channels=4, out_channels=8, num_layers=28,
                 sample_size=64, patch_size=2, cross_attention_dim=1152, attention_bias=True, activation_fn="gelu-approximate",
                 norm_type="ada_norm_single", norm_elementwise_affine=False, norm_eps=1e-6, caption_channels=4096,
                 video_length=16, compute_dtype="f16", max_batch=2, max_text_tokens=120, **unused):
        if norm_type != "ada_norm_single" or not attention_bias or activation_fn != "gelu-approximate" \
                or norm_elementwise_affine or abs(norm_eps - 1e-6) > 1e-12:
            raise LatteError("latte_amd.LatteT2V implements the Latte-1 configuration only (ada_norm_single, attention_bias, "
                             "gelu-approximate, LayerNorm without affine, eps 1e-6)")
        self.config = SimpleNamespace(num_attention_heads=num_attention_heads, attention_head_dim=attention_head_dim,
                                      in_channels=in_channels, out_channels=out_channels or in_channels, num_layers=num_layers,
                                      sample_size=sample_size, patch_size=patch_size, cross_attention_dim=cross_attention_dim,
                                      caption_channels=caption_channels, video_length=video_length, norm_type=norm_type)
        if compute_dtype not in (None, "f16"):
            raise LatteError("latte_amd.LatteT2V runs with f16 MFMA operands only -- the type the reference runs this transformer in "
                             "(sample_t2x.py:29); bf16's 2^-9 operand roundoff, amplified by the guidance pair, misses the 1e-3 parity bar")
        self.compute_dtype, self.max_batch, self.max_text_tokens = "f16", max_batch, max_text_tokens
        self._sd, self._device, self._h, self._key, self._synced = {}, torch.device("cpu"), None, None, False

    # ------------------------------------------------------------------ loading (latte_t2v.py from_pretrained_2d)
    @classmethod
    def from_pretrained_2d(cls, pretrained_model_path, subfolder=None, **kwargs):
        root = pretrained_model_path if subfolder is None else os.path.join(pretrained_model_path, subfolder)
        with open(os.path.join(root, "config.json")) as f:
            cfg = {k: v for k, v in json.load(f).items() if not k.startswith("_")}
        cfg.update(kwargs)
        model = cls(**cfg)
        st = os.path.join(root, "diffusion_pytorch_model.safetensors")
        if os.path.exists(st):
            from safetensors.torch import load_file
            sd = load_file(st)
        else:
            sd = torch.load(os.path.join(root, "diffusion_pytorch_model.bin"), map_location="cpu")
        model.load_state_dict(sd)
        return model

    # models/__init__.py:41 calls the diffusers ModelMixin spelling: same directory layout (config.json + weights)
    from_pretrained = from_pretrained_2d

    def load_state_dict(self, state_dict, strict=True):
        self._sd = {k: v.detach().to(torch.float32) for k, v in state_dict.items()}
        self._synced = False
        return self

    def state_dict(self):
        return dict(self._sd)

    def to(self, *args, **kwargs):
        for a in list(args) + list(kwargs.values()):
            if isinstance(a, (str, torch.device)):
                self._device = torch.device(a)
                if self._device.type == "cuda" and self._device.index is None:
                    self._device = torch.device("cuda", torch.cuda.current_device())
            elif a == torch.bfloat16:
                raise LatteError("latte_amd.LatteT2V runs with f16 MFMA operands only (see the constructor)")
        self._synced = False
        return self

    def cuda(self):
        return self.to("cuda")

    def eval(self):
        return self

    def requires_grad_(self, flag=False):
        return self

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                load_library().latte_t2v_destroy(self._h)
        except Exception:
            pass

    # ------------------------------------------------------------------ engine
    def _engine(self, batch, n_text):
        _lib.require_gpu()
        lib = load_library()
        if self._device.type != "cuda":
            raise LatteError("latte_amd.LatteT2V runs on an MI355X only: call .to('cuda') (there is no CPU fallback)")
        want_b, want_k = max(batch, self.max_batch), max(n_text, self.max_text_tokens)
        key = (self._device.index, want_b, want_k, self.compute_dtype)
        if self._h is None or self._key != key:
            if self._h is not None:
                lib.latte_t2v_destroy(self._h)
                self._h = None
            c = self.config
            cfg = _lib.T2VConfig(c.num_attention_heads, c.attention_head_dim, c.in_channels, c.out_channels, c.num_layers,
                                 c.sample_size, c.patch_size, c.cross_attention_dim, c.caption_channels, c.video_length, want_k,
                                 _lib.DTYPES[self.compute_dtype])
            h = _lib.c_void()
            with torch.cuda.device(self._device):
                check(lib.latte_t2v_create(cfg, want_b, h))
            self._h, self._key, self._synced = h, key, False
            self.max_batch, self.max_text_tokens = want_b, want_k
        if not self._synced:
            with torch.cuda.device(self._device):
                for i in range(lib.latte_t2v_num_keys(self._h)):
                    k = lib.latte_t2v_key(self._h, i).decode()
                    if k not in self._sd:
                        if k == "caption_projection.y_embedding":
                            continue
                        raise LatteError(f'Missing key(s) in state_dict: "{k}"')
                    t = self._sd[k].to(device=self._device, dtype=torch.float32).contiguous()
                    check(lib.latte_t2v_load_tensor(self._h, k.encode(), ptr(t), t.numel(), 1, stream_ptr()))
                check(lib.latte_t2v_check_weights(self._h))
                torch.cuda.current_stream().synchronize()
            self._synced = True
        return self._h

    def set_engine_option(self, name, value):
        """latte_t2v_set_option on the live engine (tuning / test hook: e.g. fuse_qkv_attn); an engine must exist."""
        if self._h is None:
            raise LatteError("set_engine_option: no engine yet (run a forward or set_text first)")
        check(load_library().latte_t2v_set_option(self._h, name.encode(), int(value)))

    # ------------------------------------------------------------------ forward (latte_t2v.py:677-941)
    def forward(self, hidden_states, timestep=None, encoder_hidden_states=None, added_cond_kwargs=None, class_labels=None,
                cross_attention_kwargs=None, attention_mask=None, encoder_attention_mask=None, use_image_num=0,
                enable_temporal_attentions=True, return_dict=True):
        if use_image_num != 0:
            raise LatteError("use_image_num != 0 is the joint image-video TRAINING path (latte_t2v.py:751-757)")
        if attention_mask is not None or class_labels is not None:
            raise LatteError("attention_mask / class_labels are not used by the Latte-1 sampling path")
        if hidden_states.dim() != 5:
            raise LatteError("hidden_states must be [B, C, F, H, W]")
        if timestep is None or encoder_hidden_states is None:
            raise LatteError("LatteT2V.forward needs timestep and encoder_hidden_states")
        c = self.config
        B, C, F, H, W = hidden_states.shape
        if (C, F, H, W) != (c.in_channels, c.video_length, c.sample_size, c.sample_size):
            raise LatteError(f"input shape {tuple(hidden_states.shape)} does not match the model (C={c.in_channels}, "
                             f"F={c.video_length}, H=W={c.sample_size})")
        dev = self._device
        x = hidden_states.to(device=dev, dtype=torch.float32).contiguous()
        t = torch.as_tensor(timestep, device=dev).to(torch.int64).reshape(-1)
        if t.numel() == 1:
            t = t.expand(B)
        t = t.contiguous()
        enc = encoder_hidden_states.to(device=dev, dtype=torch.float32).contiguous()
        if enc.dim() != 3 or enc.shape[0] != B or enc.shape[2] != c.caption_channels:
            raise LatteError(f"encoder_hidden_states must be [B, tokens, {c.caption_channels}]")
        n_text = enc.shape[1]
        mask = None
        if encoder_attention_mask is not None:
            if encoder_attention_mask.dim() != 2:
                raise LatteError("encoder_attention_mask must be [B, tokens] (the 3-d form is the image-joint training path)")
            mask = encoder_attention_mask.to(device=dev, dtype=torch.float32).contiguous()
        eng = self._engine(B, n_text)
        out = torch.empty(B, c.out_channels, F, H, W, device=dev, dtype=torch.float32)
        with torch.cuda.device(dev):
            check(load_library().latte_t2v_forward(eng, ptr(x), ptr(t), ptr(enc), ptr(mask), B, n_text,
                                                   int(bool(enable_temporal_attentions)), ptr(out), stream_ptr()))
        return Transformer3DModelOutput(out) if return_dict else (out,)

    __call__ = forward

    # ------------------------------------------------------------------ chain-level entry points (pipeline_latte.py:700-760)
    def set_text(self, encoder_hidden_states, encoder_attention_mask=None):
        """Install the text context of a sampling chain in the engine: caption projection, every spatial block's
        cross-attention K|V and the mask bias are computed once instead of once per step."""
        dev = self._device
        enc = encoder_hidden_states.to(device=dev, dtype=torch.float32).contiguous()
        if enc.dim() != 3 or enc.shape[2] != self.config.caption_channels:
            raise LatteError(f"encoder_hidden_states must be [B, tokens, {self.config.caption_channels}]")
        mask = None
        if encoder_attention_mask is not None:
            mask = encoder_attention_mask.to(device=dev, dtype=torch.float32).contiguous()
        eng = self._engine(enc.shape[0], enc.shape[1])
        with torch.cuda.device(dev):
            check(load_library().latte_t2v_set_text(eng, ptr(enc), ptr(mask), enc.shape[0], enc.shape[1], stream_ptr()))
            torch.cuda.current_stream().synchronize()          # the inputs may be released by the caller
        return self

    def guided_ddim_loop(self, latents, timesteps, alphas_t, alphas_prev, guidance_scale, enable_temporal_attentions=True):
        """The classifier-free-guidance DDIM loop (eta = 0) inside the engine on ``latents`` [b, C, F, H, W]; ``set_text`` must
        hold the 2 b rows [negative | prompt].  ``timesteps`` / ``alphas_*``: one entry per step (host sequences)."""
        import numpy as np
        dev = self._device
        x = latents.to(device=dev, dtype=torch.float32).contiguous().clone()
        ts = np.ascontiguousarray(np.asarray(timesteps, dtype=np.int64))
        at = np.ascontiguousarray(np.asarray(alphas_t, dtype=np.float64))
        ap = np.ascontiguousarray(np.asarray(alphas_prev, dtype=np.float64))
        if not (len(ts) == len(at) == len(ap)) or len(ts) == 0:
            raise LatteError("guided_ddim_loop: timesteps, alphas_t and alphas_prev must have one entry per step")
        if self._h is None:
            raise LatteError("guided_ddim_loop: call set_text first")
        with torch.cuda.device(dev):
            check(load_library().latte_t2v_guided_ddim_loop(self._h, ptr(x), x.shape[0], len(ts), ts.ctypes.data, at.ctypes.data,
                                                            ap.ctypes.data, float(guidance_scale),
                                                            int(bool(enable_temporal_attentions)), stream_ptr()))
        return x
