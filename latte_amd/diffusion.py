"""Host-side mirror of the reference sampler interface (``diffusion/__init__.py``, ``respace.py``,
``gaussian_diffusion.py``) on top of the C-ABI.

* schedules: ``latte_schedule_*`` (host fp64, bit-exact timestep maps);
* per-step update: ``latte_sampler_step`` (one fused HIP kernel instead of ~25 elementwise ops and
  8-12 host->device coefficient copies per step, gaussian_diffusion.py:869-881);
* whole loop: when the model callable is a ``latte_amd.Latte`` method the loop runs inside the
  engine (``latte_sample_loop``); any other callable following the model-callable protocol is driven
  step by step with the same update kernel.
"""
import ctypes

import numpy as np
import torch

from . import _lib
from ._lib import LatteError, check, load_library, ptr, stream_ptr
from .models import Latte

_TABLES = ["betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_recip_alphas_cumprod",
           "sqrt_recipm1_alphas_cumprod", "posterior_variance", "posterior_log_variance_clipped",
           "posterior_mean_coef1", "posterior_mean_coef2", "log_betas", "sqrt_alphas_cumprod",
           "sqrt_one_minus_alphas_cumprod"]
_LOSS = {"mse": 0, "rescaled_mse": 1, "kl": 2, "rescaled_kl": 3}
_METHOD = {"ddpm": 0, "ddim": 1}


class SpacedDiffusion:
    """Stand-in for ``respace.SpacedDiffusion`` as built by ``create_diffusion`` (diffusion/__init__.py:32-46):
    EPSILON or START_X mean type, LEARNED_RANGE / FIXED_LARGE / FIXED_SMALL variance."""

    SEGMENT_BYTES = 256 << 20   # per-step noise / trail buffers of one fused-loop segment (see _loop)

    def __init__(self, timestep_respacing, noise_schedule="linear", diffusion_steps=1000, predict_xstart=False,
                 learn_sigma=True, sigma_small=False, loss_type="mse"):
        if loss_type not in _LOSS:
            raise LatteError(f"loss_type must be one of {sorted(_LOSS)}")
        self.loss_type = loss_type
        lib = load_library()
        h = _lib.c_void()
        if isinstance(timestep_respacing, (list, tuple)):
            timestep_respacing = ",".join(str(int(v)) for v in timestep_respacing)
        spec = "" if timestep_respacing is None else str(timestep_respacing)
        check(lib.latte_schedule_create(int(diffusion_steps), spec.encode(), noise_schedule.encode(), h))
        self._h = h
        check(lib.latte_schedule_set_model_types(h, int(bool(predict_xstart)), int(bool(learn_sigma)), int(bool(sigma_small))))
        self.predict_xstart, self.learn_sigma, self.sigma_small = bool(predict_xstart), bool(learn_sigma), bool(sigma_small)
        self.original_num_steps = int(diffusion_steps)
        self.num_timesteps = lib.latte_schedule_num_timesteps(h)
        n = self.num_timesteps
        tm = np.empty(n, dtype=np.int64)
        check(lib.latte_schedule_timestep_map(h, tm.ctypes.data_as(ctypes.c_void_p), n))
        self.timestep_map = tm.tolist()
        self.use_timesteps = set(self.timestep_map)
        for name in _TABLES:
            if name == "posterior_log_variance_clipped" and n < 2:
                setattr(self, name, np.array([]))
                continue
            arr = np.empty(n, dtype=np.float64)
            check(lib.latte_schedule_table(h, name.encode(), arr.ctypes.data_as(ctypes.c_void_p), n))
            setattr(self, name, arr)

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                load_library().latte_schedule_destroy(self._h)
        except Exception:
            pass

    # ------------------------------------------------------------------ helpers
    @staticmethod
    def _engine_model(model):
        """(Latte instance, uses_cfg) when `model` is a bound latte_amd.Latte method, else (None, False)."""
        owner = getattr(model, "__self__", None)
        if isinstance(model, Latte):
            return model, False
        if isinstance(owner, Latte):
            fn = getattr(model, "__func__", None)
            if fn is Latte.forward:
                return owner, False
            if fn is Latte.forward_with_cfg:
                return owner, True
        return None, False

    def _device(self, model, device):
        if device is not None:
            return torch.device(device)
        owner = getattr(model, "__self__", model)
        return next(owner.parameters()).device                                     # gd:487-488

    def _step(self, method, model_output, x, index, noise, eta, clip_denoised, denoised_fn=None, cond_fn=None,
              model_kwargs=None):
        """p_mean_variance + p_sample / ddim_sample of one step on the engine (gd:254-336, :380-421, :517-564), with the
        two caller hooks: ``denoised_fn`` on the x_start prediction before the clamp (gd:316-321) and ``cond_fn(x, t)``
        with ORIGINAL timesteps (rs:100-104; condition_mean for DDPM, condition_score for DDIM)."""
        _lib.require_gpu()
        B, F, C = x.shape[:3]
        want = (B, F, C * 2 if self.learn_sigma else C, *x.shape[3:])              # gd:290 / :338
        if model_output.shape != want:
            raise AssertionError(f"model output shape {tuple(model_output.shape)} != {want}")
        x32 = x.float().contiguous()
        mo = model_output.float().contiguous()
        nz = None if noise is None else noise.float().contiguous()
        sample = torch.empty_like(x32)
        x0 = torch.empty_like(x32)
        hw = int(np.prod(x.shape[3:]))
        lib = load_library()
        args = (self._h, _METHOD[method], int(index), float(eta), int(bool(clip_denoised)), ptr(x32), ptr(mo), ptr(nz))
        x0_in = grad = None
        with torch.cuda.device(x32.device):
            if denoised_fn is not None:
                check(lib.latte_sampler_step_ex(*args, None, None, 1, B, F, C, hw, None, ptr(x0), stream_ptr()))
                x0_in = denoised_fn(x0).float().contiguous()
                if x0_in.shape != x32.shape:
                    raise AssertionError("denoised_fn must keep the shape of its argument")
            if cond_fn is not None:
                t = torch.full((B,), self.timestep_map[index], device=x32.device, dtype=torch.int64)
                grad = cond_fn(x, t, **(model_kwargs or {})).float().contiguous()
                if grad.shape != x32.shape:
                    raise AssertionError("cond_fn must return a gradient of x's shape")
            check(lib.latte_sampler_step_ex(*args, ptr(x0_in), ptr(grad), 0, B, F, C, hw, ptr(sample), ptr(x0), stream_ptr()))
        return {"sample": sample, "pred_xstart": x0}

    def _call_model(self, model, x, index, model_kwargs):
        # respace.py:125-130: the model sees ORIGINAL timesteps
        t = torch.full((x.shape[0],), self.timestep_map[index], device=x.device, dtype=torch.int64)
        out = model(x, t, **model_kwargs)
        if isinstance(out, tuple):                                                  # gd:284-287
            out = out[0]
        return out

    # ------------------------------------------------------------------ single steps (gd:380-421, :517-564)
    def p_sample(self, model, x, t, clip_denoised=True, denoised_fn=None, cond_fn=None, model_kwargs=None):
        index = self._uniform_index(t)
        out = self._call_model(model, x, index, model_kwargs or {})
        return self._step("ddpm", out, x, index, torch.randn_like(x.float()), 0.0, clip_denoised, denoised_fn, cond_fn,
                          model_kwargs)

    def ddim_sample(self, model, x, t, clip_denoised=True, denoised_fn=None, cond_fn=None, model_kwargs=None, eta=0.0):
        index = self._uniform_index(t)
        out = self._call_model(model, x, index, model_kwargs or {})
        return self._step("ddim", out, x, index, torch.randn_like(x.float()), eta, clip_denoised, denoised_fn, cond_fn,
                          model_kwargs)

    @staticmethod
    def _uniform_index(t):
        tv = t.tolist()
        if any(v != tv[0] for v in tv):
            raise LatteError("the fused sampler step takes one timestep index for the whole batch "
                             "(the sampling loops always do, gaussian_diffusion.py:503,671)")
        return int(tv[0])

    # ------------------------------------------------------------------ loops (gd:423-515, :604-684)
    def _loop(self, method, model, shape, noise, clip_denoised, denoised_fn, cond_fn, model_kwargs, device, progress,
              eta, progressive):
        _lib.require_gpu()
        model_kwargs = dict(model_kwargs or {})
        device = self._device(model, device)
        assert isinstance(shape, (tuple, list))
        img = noise if noise is not None else torch.randn(*shape, device=device)
        img = img.to(device=device, dtype=torch.float32).contiguous().clone()
        n = self.num_timesteps
        needs_noise = method == "ddpm" or eta != 0.0
        owner, uses_cfg = self._engine_model(model)
        known = {"y", "use_fp16", "cfg_scale", "text_embedding"}
        if owner is not None and set(model_kwargs) <= known and denoised_fn is None and cond_fn is None:
            # ---- whole chain inside the engine
            B = img.shape[0]
            y = model_kwargs.get("y")
            x32, _, y64 = owner._prep(img, torch.zeros(B, dtype=torch.int64), y)
            cfg_scale = float(model_kwargs.get("cfg_scale", 7.0)) if uses_cfg else 1.0
            eng = owner.engine(B, guided=uses_cfg)
            owner._set_text(model_kwargs.get("text_embedding"), B, guided=uses_cfg)
            # The chain runs in segments of consecutive steps so that the per-step noise (and the progressive trails) of
            # only ONE segment exist at a time: 65 MB per 250-step sample-chain otherwise, 4.2 GB at B = 64.  The draws
            # are the reference's -- one torch.randn_like per step, in step order (gd:413,555) -- whatever the split.
            numel_bytes = x32.numel() * 4
            seg = max(1, min(n, self.SEGMENT_BYTES // max(numel_bytes, 1)))
            lib = load_library()
            i = n - 1
            while i >= 0:
                lo = max(0, i - seg + 1)
                cnt = i - lo + 1
                nz = torch.stack([torch.randn_like(x32) for _ in range(cnt)]) if needs_noise else None
                trail_s = trail_0 = None
                if progressive:
                    trail_s = torch.empty((cnt,) + tuple(x32.shape), device=device, dtype=torch.float32)
                    trail_0 = torch.empty_like(trail_s)
                with torch.cuda.device(device):
                    check(lib.latte_sample_loop_ex(eng, self._h, _METHOD[method], float(eta), int(bool(clip_denoised)),
                                                   int(uses_cfg), cfg_scale, ptr(x32), ptr(y64), B, i, lo, ptr(nz),
                                                   ptr(trail_s), ptr(trail_0), stream_ptr()))
                    if nz is not None or progressive:
                        torch.cuda.current_stream().synchronize()      # the segment's buffers are released / handed out next
                if progressive:
                    for k in range(cnt):
                        yield {"sample": trail_s[k], "pred_xstart": trail_0[k]}
                i = lo - 1
            if not progressive:
                yield {"sample": x32, "pred_xstart": None}
            return
        # ---- generic model callable, step by step
        indices = list(range(n))[::-1]
        if progress:
            from tqdm.auto import tqdm
            indices = tqdm(indices)
        for i in indices:
            out = self._call_model(model, img, i, model_kwargs)
            nz = torch.randn_like(img) if needs_noise else None
            r = self._step(method, out, img, i, nz, eta, clip_denoised, denoised_fn, cond_fn, model_kwargs)
            yield r
            img = r["sample"]

    def p_sample_loop_progressive(self, model, shape, noise=None, clip_denoised=True, denoised_fn=None, cond_fn=None,
                                  model_kwargs=None, device=None, progress=False):
        yield from self._loop("ddpm", model, shape, noise, clip_denoised, denoised_fn, cond_fn, model_kwargs, device,
                              progress, 0.0, True)

    def ddim_sample_loop_progressive(self, model, shape, noise=None, clip_denoised=True, denoised_fn=None,
                                     cond_fn=None, model_kwargs=None, device=None, progress=False, eta=0.0):
        yield from self._loop("ddim", model, shape, noise, clip_denoised, denoised_fn, cond_fn, model_kwargs, device,
                              progress, eta, True)

    def p_sample_loop(self, model, shape, noise=None, clip_denoised=True, denoised_fn=None, cond_fn=None,
                      model_kwargs=None, device=None, progress=False):
        final = None
        for final in self._loop("ddpm", model, shape, noise, clip_denoised, denoised_fn, cond_fn, model_kwargs, device,
                                progress, 0.0, False):
            pass
        return final["sample"]

    def ddim_sample_loop(self, model, shape, noise=None, clip_denoised=True, denoised_fn=None, cond_fn=None,
                         model_kwargs=None, device=None, progress=False, eta=0.0):
        final = None
        for final in self._loop("ddim", model, shape, noise, clip_denoised, denoised_fn, cond_fn, model_kwargs, device,
                                progress, eta, False):
            pass
        return final["sample"]

    # ------------------------------------------------------------------ training path, forward evaluation
    def q_sample(self, x_start, t, noise=None):
        """``GaussianDiffusion.q_sample`` (gd:216-229) with one respaced timestep per sample, on the engine."""
        _lib.require_gpu()
        x0 = x_start.float().contiguous()
        if noise is None:
            noise = torch.randn_like(x0)
        if noise.shape != x0.shape:
            raise AssertionError("noise.shape == x_start.shape")                      # gd:225
        nz = noise.to(device=x0.device, dtype=torch.float32).contiguous()
        t64 = t.to(device=x0.device, dtype=torch.int64).contiguous()
        if t64.shape != (x0.shape[0],) or int(t64.min()) < 0 or int(t64.max()) >= self.num_timesteps:
            raise IndexError("t must hold one timestep index in [0, num_timesteps) per sample")
        out = torch.empty_like(x0)
        with torch.cuda.device(x0.device):
            check(load_library().latte_q_sample(self._h, ptr(x0), ptr(nz), ptr(t64), x0.shape[0], x0[0].numel(), ptr(out),
                                                stream_ptr()))
        return out

    def training_losses(self, model, x_start, t, model_kwargs=None, noise=None):
        """``SpacedDiffusion.training_losses`` (rs:95-98 -> gd:719-795) as called by train.py:224-226: the VALUES of the
        loss terms -- {"loss", "mse", "vb"} of shape [N] for the MSE loss types, {"loss"} for the KL ones -- computed by the
        engine from ``model(x_t, timestep_map[t], **model_kwargs)``.  Forward evaluation only: nothing here builds an
        autograd graph, the backward kernels are the next slice of SURVEY.md section 8(f) rank 3."""
        _lib.require_gpu()
        model_kwargs = model_kwargs or {}
        x0 = x_start.float().contiguous()
        if x0.dim() != 5:
            raise LatteError("x_start must be [N, F, C, H, W]")
        if noise is None:
            noise = torch.randn_like(x0)                                              # gd:732-733
        nz = noise.to(device=x0.device, dtype=torch.float32).contiguous()
        t64 = t.to(device=x0.device, dtype=torch.int64).contiguous()
        x_t = self.q_sample(x0, t64, nz)
        tmap = torch.tensor(self.timestep_map, device=x0.device, dtype=torch.int64)
        out = model(x_t, tmap[t64], **model_kwargs)                                  # rs:125-130: ORIGINAL timesteps
        if isinstance(out, tuple):
            out = out[0]
        B, F, C = x0.shape[:3]
        want = (B, F, C * 2 if self.learn_sigma else C, *x0.shape[3:])               # gd:762 / :784
        if tuple(out.shape) != want:
            raise AssertionError(f"model output shape {tuple(out.shape)} != {want}")
        mo = out.detach().float().contiguous()
        hw = int(np.prod(x0.shape[3:]))
        lib = load_library()
        nws = int(lib.latte_training_workspace_floats(B, x0[0].numel()))
        ws = torch.empty(nws, device=x0.device, dtype=torch.float32)
        mse, vb, loss = (torch.empty(B, device=x0.device, dtype=torch.float32) for _ in range(3))
        with torch.cuda.device(x0.device):
            check(lib.latte_training_losses(self._h, _LOSS[self.loss_type], ptr(x0), ptr(x_t), ptr(nz), ptr(mo), ptr(t64), B, F, C,
                                            hw, ptr(ws), nws, ptr(mse), ptr(vb), ptr(loss), stream_ptr()))
        if self.loss_type in ("kl", "rescaled_kl"):
            return {"loss": loss}
        terms = {"mse": mse, "loss": loss}
        if self.learn_sigma:
            terms["vb"] = vb
        return terms


def create_diffusion(timestep_respacing, noise_schedule="linear", use_kl=False, sigma_small=False,
                     predict_xstart=False, learn_sigma=True, rescale_learned_sigmas=False, diffusion_steps=1000):
    """``diffusion.create_diffusion`` (diffusion/__init__.py:10-47).  ``use_kl`` / ``rescale_learned_sigmas`` select the
    training loss (LossType RESCALED_KL / RESCALED_MSE / MSE, :22-27) and do not touch sampling."""
    if timestep_respacing is None or timestep_respacing == "":
        timestep_respacing = [diffusion_steps]
    loss_type = "rescaled_kl" if use_kl else ("rescaled_mse" if rescale_learned_sigmas else "mse")
    return SpacedDiffusion(timestep_respacing, noise_schedule=noise_schedule, diffusion_steps=diffusion_steps,
                           predict_xstart=predict_xstart, learn_sigma=learn_sigma, sigma_small=sigma_small, loss_type=loss_type)
