"""A minimal DDIM scheduler with the interface ``LattePipeline`` uses (``set_timesteps``, ``timesteps``,
``init_noise_sigma``, ``scale_model_input``, ``step(...)[0]``, ``order``).

``sample/sample_t2x.py:43-114`` builds the scheduler from diffusers (``DDIMScheduler.from_pretrained(..., beta_start,
beta_end, beta_schedule, variance_type, clip_sample=False)``); diffusers is not available offline, so any object with this
interface can be passed to the pipeline and THIS class exists for self-contained runs and tests.  It restates the published
DDIM update (Song et al. 2020, eq. 12) with diffusers' "leading" timestep spacing; it is memory-derived and NOT pinned
against diffusers.  Host-side fp64 tables, a handful of elementwise torch ops per step on the latents: plumbing, not the
hot path (the denoiser call is).
"""
import numpy as np
import torch


class DDIMScheduler:
    order = 1
    init_noise_sigma = 1.0

    def __init__(self, num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="linear", clip_sample=False,
                 set_alpha_to_one=True, steps_offset=0, **unused):
        if beta_schedule == "linear":
            betas = np.linspace(beta_start, beta_end, num_train_timesteps, dtype=np.float64)
        elif beta_schedule == "scaled_linear":
            betas = np.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=np.float64) ** 2
        else:
            raise ValueError(f"unsupported beta_schedule {beta_schedule!r}")
        self.alphas_cumprod = np.cumprod(1.0 - betas)
        self.final_alpha_cumprod = 1.0 if set_alpha_to_one else float(self.alphas_cumprod[0])
        self.num_train_timesteps, self.clip_sample, self.steps_offset = num_train_timesteps, clip_sample, steps_offset
        self.num_inference_steps = None
        self.timesteps = torch.arange(num_train_timesteps - 1, -1, -1)

    def set_timesteps(self, num_inference_steps, device=None):
        step_ratio = self.num_train_timesteps // num_inference_steps                       # "leading" spacing
        ts = (np.arange(0, num_inference_steps) * step_ratio).round()[::-1].copy().astype(np.int64) + self.steps_offset
        self.num_inference_steps = num_inference_steps
        self.timesteps = torch.from_numpy(ts).to(device)

    def scale_model_input(self, sample, timestep=None):
        return sample

    def step(self, model_output, timestep, sample, eta=0.0, generator=None, return_dict=True):
        t = int(timestep)
        prev_t = t - self.num_train_timesteps // self.num_inference_steps
        a_t = float(self.alphas_cumprod[t])
        a_prev = float(self.alphas_cumprod[prev_t]) if prev_t >= 0 else self.final_alpha_cumprod
        x0 = (sample - (1.0 - a_t) ** 0.5 * model_output) / a_t ** 0.5
        if self.clip_sample:
            x0 = x0.clamp(-1.0, 1.0)
        sigma = eta * ((1.0 - a_prev) / (1.0 - a_t)) ** 0.5 * (1.0 - a_t / a_prev) ** 0.5
        eps = (sample - a_t ** 0.5 * x0) / (1.0 - a_t) ** 0.5 if self.clip_sample else model_output
        prev = a_prev ** 0.5 * x0 + (1.0 - a_prev - sigma ** 2) ** 0.5 * eps
        if eta > 0:
            noise = torch.randn(model_output.shape, generator=generator, device=model_output.device, dtype=model_output.dtype)
            prev = prev + sigma * noise
        return (prev,) if not return_dict else type("DDIMSchedulerOutput", (), {"prev_sample": prev, "pred_original_sample": x0})()
