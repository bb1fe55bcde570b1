"""CPU restatement of the reference sampler (schedules in numpy fp64, updates in torch fp32).

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).  Parity status: PINNED — checked against
``/root/reference/diffusion`` run in the build container (``oracle/validate_oracle.py``,
``oracle/VALIDATION.md``) and against the known answers / golden tables in ``tests/golden/``.

Citations: ``gd:N`` = /root/reference/diffusion/gaussian_diffusion.py, ``rs:N`` = .../respace.py,
``init:N`` = .../diffusion/__init__.py.
"""
import numpy as np
import torch


# ----------------------------------------------------------------------------- schedules
def space_timesteps(num_timesteps: int, section_counts) -> list:
    """rs:12-62.  Returns the SORTED retained indices.  ``round`` is Python's (banker's)."""
    if isinstance(section_counts, str):
        if section_counts.startswith("ddim"):
            want = int(section_counts[4:])
            for stride in range(1, num_timesteps):
                if len(range(0, num_timesteps, stride)) == want:
                    return list(range(0, num_timesteps, stride))
            raise ValueError(f"cannot create exactly {num_timesteps} steps with an integer stride")
        section_counts = [int(s) for s in section_counts.split(",")]
    size_per, extra = divmod(num_timesteps, len(section_counts))
    start, kept = 0, []
    for i, count in enumerate(section_counts):
        size = size_per + (1 if i < extra else 0)
        if size < count:
            raise ValueError(f"cannot divide section of {size} steps into {count}")
        frac = 1 if count <= 1 else (size - 1) / (count - 1)
        cur = 0.0
        for _ in range(count):
            kept.append(start + round(cur))
            cur += frac
        start += size
    return sorted(set(kept))


def named_betas(name: str, n: int) -> np.ndarray:
    """gd:98-122 (``linear`` = Ho et al. scaled by 1000/n; ``squaredcos_cap_v2`` via gd:125-141)."""
    if name == "linear":
        scale = 1000 / n
        return np.linspace(scale * 0.0001, scale * 0.02, n, dtype=np.float64)
    if name == "squaredcos_cap_v2":
        import math
        f = lambda t: math.cos((t + 0.008) / 1.008 * math.pi / 2) ** 2
        return np.array([min(1 - f((i + 1) / n) / f(i / n), 0.999) for i in range(n)])
    raise NotImplementedError(name)


class Schedule:
    """Tables of ``GaussianDiffusion.__init__`` (gd:153-201) for the respaced betas of
    ``SpacedDiffusion.__init__`` (rs:73-87)."""

    def __init__(self, timestep_respacing="", noise_schedule="linear", diffusion_steps=1000, predict_xstart=False,
                 learn_sigma=True, sigma_small=False):
        # init:32-45: what the model predicts
        self.predict_xstart = bool(predict_xstart)
        self.var_type = "learned_range" if learn_sigma else ("fixed_small" if sigma_small else "fixed_large")
        base_betas = named_betas(noise_schedule, diffusion_steps)
        if timestep_respacing is None or timestep_respacing == "":
            timestep_respacing = [diffusion_steps]                              # init:29-30
        use = set(space_timesteps(diffusion_steps, timestep_respacing))
        base_ac = np.cumprod(1.0 - base_betas, axis=0)
        last, new_betas, self.timestep_map = 1.0, [], []
        for i, ac in enumerate(base_ac):                                         # rs:80-85
            if i in use:
                new_betas.append(1 - ac / last)
                last = ac
                self.timestep_map.append(i)
        betas = np.array(new_betas, dtype=np.float64)
        self.betas = betas
        self.num_timesteps = int(betas.shape[0])
        alphas = 1.0 - betas
        self.alphas_cumprod = np.cumprod(alphas, axis=0)
        self.alphas_cumprod_prev = np.append(1.0, self.alphas_cumprod[:-1])
        self.sqrt_recip_alphas_cumprod = np.sqrt(1.0 / self.alphas_cumprod)
        self.sqrt_recipm1_alphas_cumprod = np.sqrt(1.0 / self.alphas_cumprod - 1)
        self.posterior_variance = betas * (1.0 - self.alphas_cumprod_prev) / (1.0 - self.alphas_cumprod)
        self.posterior_log_variance_clipped = (
            np.log(np.append(self.posterior_variance[1], self.posterior_variance[1:]))
            if self.num_timesteps > 1 else np.array([]))
        self.posterior_mean_coef1 = betas * np.sqrt(self.alphas_cumprod_prev) / (1.0 - self.alphas_cumprod)
        self.posterior_mean_coef2 = (1.0 - self.alphas_cumprod_prev) * np.sqrt(alphas) / (1.0 - self.alphas_cumprod)
        self.log_betas = np.log(betas)                                           # gd:293
        self.sqrt_alphas_cumprod = np.sqrt(self.alphas_cumprod)                  # gd:176 (q_sample, training)
        self.sqrt_one_minus_alphas_cumprod = np.sqrt(1.0 - self.alphas_cumprod)  # gd:177


def _coef(arr: np.ndarray, i: int) -> torch.Tensor:
    """gd:869-881: fp64 table entry cast to fp32 (then broadcast by the caller's arithmetic)."""
    return torch.tensor(arr[i], dtype=torch.float64).float()


# ----------------------------------------------------------------------------- per-step math
def p_mean_variance(s: Schedule, model_out: torch.Tensor, x: torch.Tensor, i: int, clip_denoised=False, denoised_fn=None):
    """gd:254-336, EPSILON mean + LEARNED_RANGE variance (what create_diffusion builds, init:32-46).
    ``model_out`` is [B,F,2C,H,W]; ``i`` is the respaced index.  ``denoised_fn`` is applied to the x_start
    prediction BEFORE the clamp (process_xstart, gd:316-321)."""
    C = x.shape[2]
    var_type = getattr(s, "var_type", "learned_range")
    if var_type == "learned_range":
        eps, v = torch.split(model_out, C, dim=2)                                # gd:291
        min_log = _coef(s.posterior_log_variance_clipped, i)
        max_log = _coef(s.log_betas, i)
        frac = (v + 1) / 2
        log_var = frac * max_log + (1 - frac) * min_log                          # gd:296
    else:
        # gd:298-313: fixed_large = log(append(posterior_variance[1], betas[1:])), fixed_small = clipped posterior
        eps = model_out
        table = (np.log(np.append(s.posterior_variance[1], s.betas[1:])) if var_type == "fixed_large"
                 else s.posterior_log_variance_clipped)
        log_var = _coef(table, i) + torch.zeros_like(x)                          # _extract_into_tensor broadcast
    x0 = _coef(s.sqrt_recip_alphas_cumprod, i) * x - _coef(s.sqrt_recipm1_alphas_cumprod, i) * eps
    if getattr(s, "predict_xstart", False):
        x0 = eps                                                                 # START_X, gd:323-324
    if denoised_fn is not None:
        x0 = denoised_fn(x0)
    if clip_denoised:
        x0 = x0.clamp(-1, 1)
    mean = _coef(s.posterior_mean_coef1, i) * x0 + _coef(s.posterior_mean_coef2, i) * x
    return {"mean": mean, "log_variance": log_var, "variance": torch.exp(log_var), "pred_xstart": x0}


def p_sample(s, model_out, x, i, noise, clip_denoised=False, denoised_fn=None, cond_grad=None):
    """gd:380-421 (DDPM ancestral step); no noise at i == 0.  ``cond_grad`` = cond_fn(x, t) of condition_mean
    (gd:345-356): mean += variance * gradient; pred_xstart stays the unconditioned one."""
    out = p_mean_variance(s, model_out, x, i, clip_denoised, denoised_fn)
    if cond_grad is not None:
        out["mean"] = out["mean"].float() + out["variance"] * cond_grad.float()
    mask = 0.0 if i == 0 else 1.0
    sample = out["mean"] + mask * torch.exp(0.5 * out["log_variance"]) * noise
    return {"sample": sample, "pred_xstart": out["pred_xstart"]}


def ddim_sample(s, model_out, x, i, noise=None, eta=0.0, clip_denoised=False, denoised_fn=None, cond_grad=None):
    """gd:517-564.  ``cond_grad`` = cond_fn(x, t) of condition_score (gd:358-375): eps -= sqrt(1 - alpha_bar) * gradient,
    pred_xstart re-derived from that eps (not clamped again)."""
    out = p_mean_variance(s, model_out, x, i, clip_denoised, denoised_fn)
    x0 = out["pred_xstart"]
    if cond_grad is not None:
        e = (_coef(s.sqrt_recip_alphas_cumprod, i) * x - x0) / _coef(s.sqrt_recipm1_alphas_cumprod, i)
        e = e - (1 - _coef(s.alphas_cumprod, i)).sqrt() * cond_grad
        x0 = _coef(s.sqrt_recip_alphas_cumprod, i) * x - _coef(s.sqrt_recipm1_alphas_cumprod, i) * e
    eps = (_coef(s.sqrt_recip_alphas_cumprod, i) * x - x0) / _coef(s.sqrt_recipm1_alphas_cumprod, i)
    ab = _coef(s.alphas_cumprod, i)
    ab_prev = _coef(s.alphas_cumprod_prev, i)
    sigma = eta * torch.sqrt((1 - ab_prev) / (1 - ab)) * torch.sqrt(1 - ab / ab_prev)
    mean_pred = x0 * torch.sqrt(ab_prev) + torch.sqrt(1 - ab_prev - sigma ** 2) * eps
    mask = 0.0 if i == 0 else 1.0
    if noise is None:
        noise = torch.zeros_like(x)
    return {"sample": mean_pred + mask * sigma * noise, "pred_xstart": x0}


def sample_loop(s: Schedule, model_fn, x: torch.Tensor, method="ddim", eta=0.0, noises=None,
                clip_denoised=False, progressive=False, denoised_fn=None, cond_fn=None):
    """gd:423-515 / gd:604-684.  ``model_fn(x, t_original:int64[B]) -> [B,F,2C,H,W]``; the loop
    index ``i`` is mapped through ``timestep_map`` exactly as ``_WrappedModel`` does (rs:125-130).
    ``noises[k]`` is the noise used at the k-th executed step (k=0 is i=T-1) so both sides of a
    parity run consume identical draws."""
    trail = []
    B = x.shape[0]
    for k, i in enumerate(range(s.num_timesteps - 1, -1, -1)):
        t = torch.full((B,), s.timestep_map[i], dtype=torch.int64)
        out = model_fn(x, t)
        nz = None if noises is None else noises[k]
        grad = None if cond_fn is None else cond_fn(x, t)          # the wrapped cond_fn sees ORIGINAL timesteps (rs:100-104)
        if method == "ddim":
            r = ddim_sample(s, out, x, i, nz, eta, clip_denoised, denoised_fn, grad)
        else:
            r = p_sample(s, out, x, i, nz if nz is not None else torch.zeros_like(x), clip_denoised, denoised_fn, grad)
        x = r["sample"]
        if progressive:
            trail.append(r)
    return (x, trail) if progressive else x


# ----------------------------------------------------------------------------- training losses
def _coef_t(arr: np.ndarray, t: torch.Tensor, ndim: int) -> torch.Tensor:
    """gd:869-881 for a BATCH of timesteps: ``from_numpy(arr)[t].float()`` broadcast over the trailing dims."""
    res = torch.from_numpy(np.asarray(arr))[t].float()
    while res.dim() < ndim:
        res = res[..., None]
    return res


def _mean_flat(x):
    return x.mean(dim=list(range(1, x.dim())))                                   # gd:20-24


def normal_kl(mean1, logvar1, mean2, logvar2):
    """diffusion_utils.py:10-36."""
    return 0.5 * (-1.0 + logvar2 - logvar1 + torch.exp(logvar1 - logvar2) + ((mean1 - mean2) ** 2) * torch.exp(-logvar2))


def _approx_cdf(x):
    return 0.5 * (1.0 + torch.tanh(np.sqrt(2.0 / np.pi) * (x + 0.044715 * torch.pow(x, 3))))   # diffusion_utils.py:39-44


def discretized_gaussian_log_likelihood(x, means, log_scales):
    """diffusion_utils.py:62-88."""
    centered = x - means
    inv_stdv = torch.exp(-log_scales)
    cdf_plus = _approx_cdf(inv_stdv * (centered + 1.0 / 255.0))
    cdf_min = _approx_cdf(inv_stdv * (centered - 1.0 / 255.0))
    log_cdf_plus = torch.log(cdf_plus.clamp(min=1e-12))
    log_one_minus_cdf_min = torch.log((1.0 - cdf_min).clamp(min=1e-12))
    cdf_delta = cdf_plus - cdf_min
    return torch.where(x < -0.999, log_cdf_plus,
                       torch.where(x > 0.999, log_one_minus_cdf_min, torch.log(cdf_delta.clamp(min=1e-12))))


def q_sample(s: Schedule, x_start, t, noise):
    """gd:216-229."""
    return (_coef_t(s.sqrt_alphas_cumprod, t, x_start.dim()) * x_start
            + _coef_t(s.sqrt_one_minus_alphas_cumprod, t, x_start.dim()) * noise)


def _vb_terms_bpd(s: Schedule, model_out, x_start, x_t, t):
    """gd:686-717 with clip_denoised=False (the only way training_losses calls it): per-sample KL / decoder NLL in bits."""
    nd = x_start.dim()
    true_mean = _coef_t(s.posterior_mean_coef1, t, nd) * x_start + _coef_t(s.posterior_mean_coef2, t, nd) * x_t   # gd:232-241
    true_logvar = _coef_t(s.posterior_log_variance_clipped, t, nd)
    C = x_t.shape[2]
    if s.var_type == "learned_range":
        eps, v = torch.split(model_out, C, dim=2)
        min_log, max_log = _coef_t(s.posterior_log_variance_clipped, t, nd), _coef_t(s.log_betas, t, nd)
        frac = (v + 1) / 2
        log_var = frac * max_log + (1 - frac) * min_log
    else:
        eps = model_out
        table = (np.log(np.append(s.posterior_variance[1], s.betas[1:])) if s.var_type == "fixed_large"
                 else s.posterior_log_variance_clipped)
        log_var = _coef_t(table, t, nd) + torch.zeros_like(x_t)
    if s.predict_xstart:
        x0 = eps
    else:
        x0 = _coef_t(s.sqrt_recip_alphas_cumprod, t, nd) * x_t - _coef_t(s.sqrt_recipm1_alphas_cumprod, t, nd) * eps
    mean = _coef_t(s.posterior_mean_coef1, t, nd) * x0 + _coef_t(s.posterior_mean_coef2, t, nd) * x_t
    kl = _mean_flat(normal_kl(true_mean, true_logvar, mean, log_var)) / np.log(2.0)
    nll = _mean_flat(-discretized_gaussian_log_likelihood(x_start, mean, 0.5 * log_var)) / np.log(2.0)
    return torch.where(t == 0, nll, kl)


def training_losses(s: Schedule, model_fn, x_start, t, noise, loss_type="mse"):
    """gd:719-795 through rs:95-98 (the model sees ORIGINAL timesteps).  ``t`` = respaced indices int64[N];
    ``loss_type`` in {"mse", "rescaled_mse", "kl", "rescaled_kl"} (init:22-27: use_kl -> rescaled_kl,
    rescale_learned_sigmas -> rescaled_mse, else mse).  -> dict of [N] tensors."""
    x_t = q_sample(s, x_start, t, noise)
    t_orig = torch.tensor(s.timestep_map, dtype=torch.int64)[t]
    model_out = model_fn(x_t, t_orig)
    terms = {}
    if loss_type in ("kl", "rescaled_kl"):
        terms["loss"] = _vb_terms_bpd(s, model_out, x_start, x_t, t)
        if loss_type == "rescaled_kl":
            terms["loss"] = terms["loss"] * s.num_timesteps
        return terms
    C = x_t.shape[2]
    pred = model_out
    if s.var_type == "learned_range":
        pred = model_out[:, :, :C]
        terms["vb"] = _vb_terms_bpd(s, model_out, x_start, x_t, t)           # mean prediction detached: forward value equal
        if loss_type == "rescaled_mse":
            terms["vb"] = terms["vb"] * (s.num_timesteps / 1000.0)
    target = x_start if s.predict_xstart else noise
    terms["mse"] = _mean_flat((target - pred) ** 2)
    terms["loss"] = terms["mse"] + terms["vb"] if "vb" in terms else terms["mse"]
    return terms


# ----------------------------------------------------------------------------- hook fixtures
def example_denoised_fn(x0):
    """A deterministic ``denoised_fn`` for parity fixtures (the reference never ships one; sample.py passes none)."""
    return 0.9 * x0 + 0.05


def example_cond_fn(x, t, **kwargs):
    """A deterministic ``cond_fn``: stands in for grad log p(y | x); depends on x AND on the ORIGINAL timestep, so a
    wrong timestep mapping (rs:100-104) shows."""
    return 0.05 * torch.tanh(x) * (t.float().view(-1, 1, 1, 1, 1) / 1000.0 + 0.5)


def synthetic_model(x, t, out_channels):
    """A cheap deterministic stand-in for a denoiser (the sampler arithmetic is under test, not the network):
    [B,F,C,H,W], int64[B] -> [B,F,out_channels,H,W]."""
    s = torch.sin(x * 1.7 + 0.3) * 0.8 + 0.1 * x
    s = s * (0.5 + t.float().view(-1, 1, 1, 1, 1) / 2000.0)
    if out_channels == x.shape[2]:
        return s
    return torch.cat([s, torch.cos(x * 0.9 - 0.2) * 0.7], dim=2)
