"""Host-side mirror of ``diffusers.models.AutoencoderKL`` for the call sites of the reference sampling
scripts (sample/sample.py:69,113-115; sample/sample_ddp.py:90,165-168):

    vae = AutoencoderKL.from_pretrained(path, subfolder="vae").to(device)
    samples = vae.decode(samples / 0.18215).sample

Only the DECODER runs on the MI355X engine (``latte_vae_*`` in include/latte_amd.h); ``encode`` belongs to
the training path (train.py:210) and raises.  Weights keep their diffusers state-dict names.  There is no
CPU fallback.
"""
import json
import os
from types import SimpleNamespace

import torch

from . import _lib
from ._lib import LatteError, check, load_library, ptr, stream_ptr

_LEGACY_ATTN = {"query": "to_q", "key": "to_k", "value": "to_v", "proj_attn": "to_out.0"}


class DecoderOutput:
    def __init__(self, sample):
        self.sample = sample


class AutoencoderKL:
    """SD-VAE (``stabilityai/sd-vae-ft-*`` architecture) decoder on the HIP engine.

    The MFMA operands (GroupNorm+SiLU outputs, conv weights, attention q/k/v/P) are f16 -- the type the reference decodes
    in (``vae.to(dtype=torch.float16)``, sample.py:74; ``torch_dtype=torch.float16``, sample_t2x.py:32-34) -- and the
    residual stream is fp32: 1e-3 relative L2 against the fp32 restatement at the full 32x32 -> 256x256 size.  bf16
    operands (same MFMA rate, 2^-9 instead of 2^-11 unit roundoff in every conv operand) measured 7.3e-3 on the same
    decode and are not offered: ``compute_dtype`` other than f16 and ``.to(torch.bfloat16)`` raise."""
    _TEMPORAL = False

    def __init__(self, latent_size=32, max_frames=16, compute_dtype="f16", scaling_factor=0.18215,
                 block_out_channels=(128, 256, 512, 512), layers_per_block=2, latent_channels=4, norm_num_groups=32):
        if tuple(block_out_channels) != (128, 256, 512, 512) or layers_per_block != 2 or latent_channels != 4 \
                or norm_num_groups != 32:
            raise LatteError("latte_amd.AutoencoderKL implements the sd-vae-ft architecture only "
                             "(block_out_channels (128,256,512,512), layers_per_block 2, latent_channels 4, 32 groups)")
        self.config = SimpleNamespace(scaling_factor=scaling_factor, block_out_channels=list(block_out_channels),
                                      layers_per_block=layers_per_block, latent_channels=latent_channels,
                                      norm_num_groups=norm_num_groups, in_channels=3, out_channels=3)
        if compute_dtype not in ("f16", "fp16", "float16"):
            raise LatteError("latte_amd.AutoencoderKL decodes with f16 MFMA operands only (class docstring): "
                             f"compute_dtype={compute_dtype!r} is not available")
        self.latent_size, self.max_frames, self.compute_dtype = latent_size, max_frames, "f16"
        self._sd = {}
        self._device = torch.device("cpu")
        self._h = None
        self._key = None
        self._synced = False

    # ------------------------------------------------------------------ diffusers-style loading
    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, subfolder=None, **kw):
        root = pretrained_model_name_or_path if subfolder is None else os.path.join(pretrained_model_name_or_path, subfolder)
        cfg = {}
        cfg_path = os.path.join(root, "config.json")
        if os.path.exists(cfg_path):
            with open(cfg_path) as f:
                cfg = json.load(f)
        args = {k: cfg[k] for k in ("scaling_factor", "block_out_channels", "layers_per_block", "latent_channels",
                                    "norm_num_groups") if k in cfg}
        args.update({k: v for k, v in kw.items() if k in ("latent_size", "max_frames", "compute_dtype")})
        vae = cls(**args)
        st = os.path.join(root, "diffusion_pytorch_model.safetensors")
        pt = os.path.join(root, "diffusion_pytorch_model.bin")
        if os.path.exists(st):
            from safetensors.torch import load_file
            sd = load_file(st)
        elif os.path.exists(pt):
            sd = torch.load(pt, map_location="cpu")
        else:
            raise LatteError(f"no diffusion_pytorch_model.safetensors / .bin under {root}")
        vae.load_state_dict(sd)
        return vae

    def load_state_dict(self, state_dict, strict=True):
        sd = {}
        for k, v in state_dict.items():
            if k.startswith("encoder.") or k.startswith("quant_conv."):
                continue                                            # encoder half: not on the sampling path
            parts = k.split(".")
            if "attentions" in parts and parts[-2] in _LEGACY_ATTN:  # pre-0.18 attention names
                parts[-2:-1] = _LEGACY_ATTN[parts[-2]].split(".")
                k = ".".join(parts)
                if v.dim() == 4:
                    v = v.reshape(v.shape[0], v.shape[1])
            sd[k] = v.detach().to(torch.float32)
        self._sd = sd
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
                raise LatteError("latte_amd.AutoencoderKL decodes with f16 MFMA operands only (class docstring)")
            elif a in (torch.float16, torch.float32):                # sample.py:74 vae.to(dtype=torch.float16); fp32 = the default
                pass
        self._synced = False
        return self

    def eval(self):
        return self

    def requires_grad_(self, flag=False):
        return self

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                load_library().latte_vae_destroy(self._h)
        except Exception:
            pass

    # ------------------------------------------------------------------ engine
    def _engine(self, n_frames, latent_size):
        _lib.require_gpu()
        lib = load_library()
        if self._device.type != "cuda":
            raise LatteError("latte_amd.AutoencoderKL runs on an MI355X only: call .to('cuda') (there is no CPU fallback)")
        want = max(self.max_frames, min(n_frames, 64))
        key = (self._device.index, want, latent_size, self.compute_dtype)
        if self._h is None or self._key != key:
            if self._h is not None:
                lib.latte_vae_destroy(self._h)
                self._h = None
            h = _lib.c_void()
            with torch.cuda.device(self._device):
                create = lib.latte_vae_create_temporal if self._TEMPORAL else lib.latte_vae_create
                check(create(latent_size, want, _lib.DTYPES[self.compute_dtype], h))
            self._h, self._key, self._synced = h, key, False
            self.max_frames, self.latent_size = want, latent_size
        if not self._synced:
            with torch.cuda.device(self._device):
                for i in range(lib.latte_vae_num_keys(self._h)):
                    k = lib.latte_vae_key(self._h, i).decode()
                    if k not in self._sd:
                        raise LatteError(f'Missing key(s) in state_dict: "{k}"')
                    t = self._sd[k].to(device=self._device, dtype=torch.float32).contiguous()
                    check(lib.latte_vae_load_tensor(self._h, k.encode(), ptr(t), t.numel(), 1, stream_ptr()))
                check(lib.latte_vae_check_weights(self._h))
                torch.cuda.current_stream().synchronize()
            self._synced = True
        return self._h

    def _run(self, z, z_scale, out_mode):
        if z.dim() != 4 or z.shape[1] != 4 or z.shape[2] != z.shape[3]:
            raise LatteError("z must be [N, 4, h, w] with h == w")
        n, _, h, _ = z.shape
        z32 = z.to(device=self._device, dtype=torch.float32).contiguous()
        eng = self._engine(n, h)
        H = 8 * h
        out = (torch.empty(n, 3, H, H, device=self._device, dtype=torch.float32) if out_mode == 0
               else torch.empty(n, H, H, 3, device=self._device, dtype=torch.uint8))
        lib = load_library()
        with torch.cuda.device(self._device):
            for s in range(0, n, self.max_frames):
                m = min(self.max_frames, n - s)
                check(lib.latte_vae_decode(eng, ptr(z32[s:s + m]), m, float(z_scale), out_mode, ptr(out[s:s + m]), stream_ptr()))
        return out

    KERNEL_CLASSES = ("conv3x3", "groupnorm_stats", "groupnorm_apply", "attention_and_1x1", "small")

    def profile_decode(self, z, z_scale=1.0):
        """Measurement hook (bench.py): one decode of z [N <= max_frames, 4, h, w] with a HIP event behind every launch ->
        {class: (milliseconds, launches)} for KERNEL_CLASSES."""
        import ctypes
        n, _, h, _ = z.shape
        z32 = z.to(device=self._device, dtype=torch.float32).contiguous()
        eng = self._engine(n, h)
        out = torch.empty(n, 3, 8 * h, 8 * h, device=self._device, dtype=torch.float32)
        k = len(self.KERNEL_CLASSES)
        ms, cnt = (ctypes.c_float * k)(), (ctypes.c_int * k)()
        with torch.cuda.device(self._device):
            check(load_library().latte_vae_profile_decode(eng, ptr(z32), n, float(z_scale), 0, ptr(out), ms, cnt, k, stream_ptr()))
        return {c: (float(ms[i]), int(cnt[i])) for i, c in enumerate(self.KERNEL_CLASSES)}

    def decode(self, z, return_dict=True):
        """``AutoencoderKL.decode``: z [N,4,h,w] (already divided by scaling_factor) -> ``.sample`` fp32 [N,3,8h,8w]."""
        out = self._run(z, 1.0, 0)
        return DecoderOutput(out) if return_dict else (out,)

    def decode_video_uint8(self, latents):
        """sample.py:110-122 in one engine call per chunk: latents [B,F,4,h,w] (NOT yet divided by 0.18215) ->
        uint8 video [B,F,8h,8w,3] = ((decode(z/0.18215)*0.5+0.5)*255+0.5).clamp(0,255)."""
        b, f = latents.shape[:2]
        out = self._run(latents.reshape(b * f, *latents.shape[2:]), 1.0 / self.config.scaling_factor, 1)
        return out.view(b, f, *out.shape[1:])

    def encode(self, x):
        raise LatteError("AutoencoderKL.encode is the training path (train.py:210) and is outside the MI355X sampling engine")


class AutoencoderKLTemporalDecoder(AutoencoderKL):
    """``diffusers.AutoencoderKLTemporalDecoder`` (stable-video-diffusion's VAE) for the call sites of the reference's text-to-video
    path (sample_t2x.py:31-32 ``from_pretrained(path, subfolder="vae_temporal_decoder")``; pipeline_latte.py:779-798
    ``vae.decode(latents[i : i + 14], num_frames=n).sample``): the SD-VAE decoder with a temporal resnet (GroupNorm over the
    frames, Conv3d (3,1,1)) blended into every block by a learned factor, and a temporal convolution on the RGB output.
    Decoder only, on the same HIP kernels as ``AutoencoderKL`` (the Conv3d is the implicit-GEMM conv kernel with three taps along
    an "image" whose rows are the frames).  Restated from memory of diffusers 0.24.0: parity unpinned."""
    _TEMPORAL = True

    def load_state_dict(self, state_dict, strict=True):
        super().load_state_dict(state_dict, strict)
        self._sd.pop("post_quant_conv.weight", None)          # this class has none (only the encoder-side quant_conv)
        self._sd.pop("post_quant_conv.bias", None)
        return self

    def decode(self, z, num_frames=1, return_dict=True, image_only_indicator=None):
        """z [B * num_frames, 4, h, w] (already divided by scaling_factor) -> ``.sample`` fp32 [B * num_frames, 3, 8h, 8w];
        every run of ``num_frames`` frames is one video chunk (temporal mixing stays inside it)."""
        n = z.shape[0]
        if num_frames <= 0 or n % num_frames:
            raise LatteError("z.shape[0] must be a multiple of num_frames")
        self.max_frames = max(self.max_frames, num_frames)
        z32 = z.to(device=self._device, dtype=torch.float32).contiguous()
        h = z.shape[2]
        eng = self._engine(num_frames, h)
        out = torch.empty(n, 3, 8 * h, 8 * h, device=self._device, dtype=torch.float32)
        lib = load_library()
        with torch.cuda.device(self._device):
            for s in range(0, n, num_frames):
                check(lib.latte_vae_decode(eng, ptr(z32[s:s + num_frames]), num_frames, 1.0, 0, ptr(out[s:s + num_frames]), stream_ptr()))
        return DecoderOutput(out) if return_dict else (out,)

    def decode_video_uint8(self, latents):
        raise LatteError("use decode(z, num_frames=...) (the text-to-video pipeline converts to uint8 itself, pipeline_latte.py:796)")
