"""CPU fp32 restatement of ONE optimisation step of the reference's training loop (train.py:197-236).

TEST INFRASTRUCTURE ONLY (tests/, the golden generator and oracle validation import it; the product path never does).

What the step is, with the reference lines it follows:
  * ``x_t = q_sample(x_start, t, noise)``; ``model_output = model(x_t, timestep_map[t], y)`` with the label dropout of
    ``LabelEmbedder.token_drop`` in train mode (latte.py:138-153: dropped labels become ``num_classes``) -- the drop mask is an
    explicit argument here so that the step is a pure function;
  * ``training_losses`` (gaussian_diffusion.py:719-795): ``mse = mean_flat((noise - eps)^2)`` and, for learned sigma, the
    variational bound evaluated on ``cat([eps.detach(), v])`` -- the variance head is trained by the bound, the mean head by
    the MSE only (:753-757);
  * ``loss = terms["loss"].mean()``; ``loss.backward()`` (train.py:224-226);
  * ``clip_grad_norm_`` (utils.py:72-117): total 2-norm over all gradients, ``g *= clamp(max_norm / (norm + 1e-6), max=1)``
    only when ``clip_grad`` (train.py:228-231: from ``start_clip_iter`` on);
  * ``torch.optim.AdamW(lr=1e-4, weight_decay=0)`` (train.py:127), restated from its documented update rule;
  * ``update_ema(ema, model, decay=0.9999)`` over ALL named parameters incl. the frozen sin-cos tables (utils.py:191-200).

Pinned by ``oracle/validate_oracle.py`` against the unmodified reference objects (``Latte(...).train()``, ``create_diffusion``,
``torch.optim.AdamW``, ``utils.clip_grad_norm_`` / ``update_ema`` restated there because utils.py imports tensorboard).
"""
import math

import torch

from oracle import diffusion_oracle as do
from oracle import latte_oracle as lo

FROZEN = ("pos_embed", "temp_embed")   # nn.Parameter(requires_grad=False), latte.py:246-247


def trainable_keys(sd):
    return [k for k in sd if k not in FROZEN]


def loss_and_grads(sd, cfg, sched, x_start, t, noise, y=None, drop_mask=None, loss_type="mse"):
    """-> (terms dict of [N] tensors, model_output, {key: grad}) for ``terms['loss'].mean()``."""
    params = {k: (v.detach().clone().requires_grad_(k not in FROZEN)) for k, v in sd.items()}
    yy = y
    if y is not None and drop_mask is not None:
        yy = torch.where(drop_mask, torch.full_like(y, cfg.num_classes), y)       # latte.py:146-148
    x_t = do.q_sample(sched, x_start, t, noise)
    t_orig = torch.tensor(sched.timestep_map, dtype=torch.int64)[t]
    out = lo.latte_forward(params, cfg, x_t, t_orig, yy)
    C = x_t.shape[2]
    terms = {}
    if sched.var_type == "learned_range":
        eps, v = out[:, :, :C], out[:, :, C:]
        frozen = torch.cat([eps.detach(), v], dim=2)                                # gd:753-757
        terms["vb"] = do._vb_terms_bpd(sched, frozen, x_start, x_t, t)
        if loss_type == "rescaled_mse":
            terms["vb"] = terms["vb"] * (sched.num_timesteps / 1000.0)
    else:
        eps = out
    target = x_start if sched.predict_xstart else noise
    terms["mse"] = do._mean_flat((target - eps) ** 2)
    terms["loss"] = terms["mse"] + terms["vb"] if "vb" in terms else terms["mse"]
    loss = terms["loss"].mean()
    keys = trainable_keys(params)
    grads = torch.autograd.grad(loss, [params[k] for k in keys])
    return {k: v.detach() for k, v in terms.items()}, out.detach(), dict(zip(keys, grads))


def grad_norm(grads):
    """utils.py:103: norm of the per-tensor norms."""
    return torch.norm(torch.stack([torch.norm(g, 2.0) for g in grads.values()]), 2.0)


def clip_grads(grads, max_norm, clip=True):
    total = grad_norm(grads)
    if clip:
        coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
        grads = {k: g * coef for k, g in grads.items()}
    return total, grads


def adamw_step(sd, grads, state, step, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
    """torch.optim.AdamW, single-tensor form.  ``state`` = {key: (exp_avg, exp_avg_sq)} (zeros at step 1); ``step`` counts
    from 1.  -> (new sd, new state)."""
    b1, b2 = betas
    new_sd, new_state = dict(sd), {}
    bc1, bc2 = 1 - b1 ** step, 1 - b2 ** step
    for k, g in grads.items():
        m, v = state.get(k, (torch.zeros_like(g), torch.zeros_like(g)))
        p = sd[k] * (1 - lr * weight_decay)
        m = m * b1 + g * (1 - b1)                       # exp_avg.lerp_(grad, 1 - beta1)
        v = v * b2 + (g * g) * (1 - b2)
        denom = v.sqrt() / math.sqrt(bc2) + eps
        new_sd[k] = p - (lr / bc1) * (m / denom)
        new_state[k] = (m, v)
    return new_sd, new_state


def update_ema(ema_sd, sd, decay=0.9999):
    return {k: ema_sd[k] * decay + sd[k] * (1 - decay) for k in sd}


def train_step(sd, ema_sd, state, step, cfg, sched, x_start, t, noise, y=None, drop_mask=None, lr=1e-4, clip_max_norm=0.1,
               clip=False, ema_decay=0.9999, loss_type="mse", world_grads=None):
    """One iteration of train.py:197-236 with gradient_accumulation_steps = 1.  ``world_grads``: gradients of the OTHER ranks'
    micro-batches (list of dicts) -- DDP averages them (train.py:125)."""
    terms, out, grads = loss_and_grads(sd, cfg, sched, x_start, t, noise, y, drop_mask, loss_type)
    if world_grads:
        n = 1 + len(world_grads)
        grads = {k: (g + sum(w[k] for w in world_grads)) / n for k, g in grads.items()}
    total, grads = clip_grads(grads, clip_max_norm, clip)
    sd, state = adamw_step(sd, grads, state, step, lr=lr)
    ema_sd = update_ema(ema_sd, sd, ema_decay)
    return dict(terms=terms, model_out=out, grads=grads, grad_norm=total, sd=sd, state=state, ema=ema_sd)
