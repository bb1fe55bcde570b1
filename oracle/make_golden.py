"""Generate ``tests/golden/*.npz`` by running the REAL reference (build container only).

TEST INFRASTRUCTURE ONLY.  Usage:  python -m oracle.make_golden
The fixtures are outputs of /root/reference code (models/latte.py via the timm stand-in,
diffusion/* unmodified); the oracle restatement and the HIP engine are both tested against them.
"""
import hashlib
import json
import os

import numpy as np
import torch

from oracle import diffusion_oracle as dor
from oracle import latte_oracle as lo
from oracle.reference_loader import (load_reference_diffusion, load_reference_latte,
                                     randomize_zero_init)

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")

TINY = dict(depth=2, hidden_size=128, patch_size=2, num_heads=2, input_size=8, num_frames=4,
            num_classes=5, extras=2, learn_sigma=True)
TINY4 = dict(depth=2, hidden_size=128, patch_size=2, num_heads=2, input_size=8, num_frames=16,
             num_classes=1000, extras=1, learn_sigma=True)

TINY78 = dict(depth=2, hidden_size=128, patch_size=2, num_heads=2, input_size=8, num_frames=4,
              num_classes=1000, extras=78, learn_sigma=True)

TABLES = ["betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_recip_alphas_cumprod",
          "sqrt_recipm1_alphas_cumprod", "posterior_variance", "posterior_log_variance_clipped",
          "posterior_mean_coef1", "posterior_mean_coef2"]


def schedules(rd):
    kat = {}
    arrays = {}
    for spec in ["", "1000", "250", "100", "50", "25", "10", "5", "1", "ddim250", "ddim100", "ddim50",
                 "ddim10", "10,15,20", "1,1,1", "333"]:
        d = rd.create_diffusion(spec)
        tm = np.asarray(d.timestep_map, dtype=np.int64)
        kat[spec] = {"n": int(d.num_timesteps), "sha256": hashlib.sha256(tm.tobytes()).hexdigest(),
                     "head": tm[:6].tolist(), "tail": tm[-4:].tolist()}
        arrays[f"map::{spec}"] = tm
        if spec in ("250", "10", "50", "ddim250", ""):
            for name in TABLES:
                arrays[f"{name}::{spec}"] = np.asarray(getattr(d, name), dtype=np.float64)
            arrays[f"log_betas::{spec}"] = np.log(d.betas)
    # other diffusion_steps / schedule names through the same factory (init:10-20)
    d = rd.create_diffusion("20", noise_schedule="squaredcos_cap_v2", diffusion_steps=400)
    arrays["map::cos400/20"] = np.asarray(d.timestep_map, dtype=np.int64)
    arrays["betas::cos400/20"] = d.betas
    arrays["alphas_cumprod::cos400/20"] = d.alphas_cumprod
    np.savez_compressed(os.path.join(OUT, "schedules.npz"), **arrays)
    with open(os.path.join(OUT, "schedules_kat.json"), "w") as f:
        json.dump(kat, f, indent=1, sort_keys=True)


def tiny_model(rl, rd, name, kw, use_cfg, seed):
    torch.manual_seed(seed)
    model = rl.Latte(**kw).eval()
    randomize_zero_init(model, seed=seed + 1)
    # the reference zero-inits every Linear bias; give them signal too so bias paths are tested
    g = torch.Generator("cpu").manual_seed(seed + 2)
    with torch.no_grad():
        for n_, p_ in model.named_parameters():
            if n_.endswith(".bias") and float(p_.abs().max()) == 0.0:
                p_.copy_(torch.randn(p_.shape, generator=g) * 0.02)
    if kw["extras"] == 78:
        # the [D, 59136] text projection comes from a closed-form hash (oracle.latte_oracle.text_projection_weight)
        # and is NOT stored in the fixture; tests rebuild it from text_w_seed
        with torch.no_grad():
            model.text_embedding_projection[1].weight.copy_(lo.text_projection_weight(kw["hidden_size"], seed))
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    g = torch.Generator("cpu").manual_seed(seed + 3)
    B, Fr, C, S = 2, kw["num_frames"], 4, kw["input_size"]
    x = torch.randn(B, Fr, C, S, S, generator=g)
    t = torch.tensor([999, 37], dtype=torch.int64)
    out = {"cfg_json": np.frombuffer(json.dumps(kw).encode(), dtype=np.uint8)}
    for k, v in sd.items():
        if k != "text_embedding_projection.1.weight":
            out["sd::" + k] = v.numpy()
    if kw["extras"] == 78:
        out["text_w_seed"] = np.int64(seed)
    out["x"] = x.numpy()
    out["t"] = t.numpy()
    with torch.no_grad():
        if kw["extras"] == 2:
            y = torch.tensor([3, 1], dtype=torch.int64)
            out["y"] = y.numpy()
            out["forward"] = model(x, t, y=y).numpy()
            # classifier-free guidance batch as sample.py:88-94 builds it
            xc = torch.cat([x[:1], x[:1]], 0)
            yc = torch.tensor([3, kw["num_classes"]], dtype=torch.int64)
            out["x_cfg"], out["y_cfg"] = xc.numpy(), yc.numpy()
            out["cfg_scale"] = np.float32(7.0)
            out["forward_with_cfg"] = model.forward_with_cfg(xc, t, y=yc, cfg_scale=7.0).numpy()
        elif kw["extras"] == 78:
            # text-conditioned variant (latte.py:238-242,340-363): [B,77,768] embeddings, guidance batch = [text, null text]
            te = torch.randn(B, 77, 768, generator=g)
            out["text_embedding"] = te.numpy()
            out["forward"] = model(x, t, text_embedding=te).numpy()
            xc = torch.cat([x[:1], x[:1]], 0)
            out["x_cfg"] = xc.numpy()
            out["cfg_scale"] = np.float32(7.0)
            out["forward_with_cfg"] = model.forward_with_cfg(xc, t, cfg_scale=7.0, text_embedding=te).numpy()
        else:
            out["forward"] = model(x, t).numpy()

        # sampling loops (sample.py:100-107): clip_denoised=False, explicit noise=z
        steps = 5
        diff = rd.create_diffusion(str(steps))
        if use_cfg:
            z = out["x_cfg"]
            z = torch.from_numpy(z)
            fn = model.forward_with_cfg
            mk = (dict(y=torch.from_numpy(out["y_cfg"]), cfg_scale=7.0) if kw["extras"] == 2
                  else dict(text_embedding=te, cfg_scale=7.0))
        else:
            z = x
            fn = model.forward
            mk = (dict(y=torch.from_numpy(out["y"])) if kw["extras"] == 2
                  else dict(text_embedding=te) if kw["extras"] == 78 else dict(y=None))
        out["loop_steps"] = np.int64(steps)
        for method, gen in (("ddim", diff.ddim_sample_loop_progressive), ("ddpm", diff.p_sample_loop_progressive)):
            torch.manual_seed(seed + 10)
            noises = [torch.randn_like(z) for _ in range(steps)]
            torch.manual_seed(seed + 10)
            samples, x0s = [], []
            for r in gen(fn, z.shape, z, clip_denoised=False, model_kwargs=mk, device="cpu"):
                samples.append(r["sample"].numpy())
                x0s.append(r["pred_xstart"].numpy())
            out[f"{method}_noises"] = np.stack([n_.numpy() for n_ in noises])
            out[f"{method}_samples"] = np.stack(samples)
            out[f"{method}_pred_xstart"] = np.stack(x0s)
        # DDIM with eta > 0 exercises the sigma / noise branch (gd:549-563)
        torch.manual_seed(seed + 10)
        out["ddim_eta05_final"] = diff.ddim_sample_loop(fn, z.shape, z, clip_denoised=False,
                                                        model_kwargs=mk, device="cpu", eta=0.5).numpy()
        if name == "tiny_uncond":
            # denoised_fn / cond_fn hooks (gd:316-321, :345-375; rs:100-104), clip_denoised=True, DDIM with eta > 0:
            # same seed -> same randn_like draws as the *_noises arrays above
            hooks = dict(denoised_fn=dor.example_denoised_fn, cond_fn=dor.example_cond_fn)
            torch.manual_seed(seed + 10)
            out["ddpm_hooks_final"] = diff.p_sample_loop(fn, z.shape, z, clip_denoised=True, model_kwargs=mk, device="cpu",
                                                         **hooks).numpy()
            torch.manual_seed(seed + 10)
            out["ddim_hooks_final"] = diff.ddim_sample_loop(fn, z.shape, z, clip_denoised=True, model_kwargs=mk,
                                                            device="cpu", eta=0.3, **hooks).numpy()
            torch.manual_seed(seed + 10)
            out["ddim_denoised_only_final"] = diff.ddim_sample_loop(fn, z.shape, z, clip_denoised=False, model_kwargs=mk,
                                                                    device="cpu", denoised_fn=dor.example_denoised_fn).numpy()
    np.savez_compressed(os.path.join(OUT, f"{name}.npz"), **out)


def sampler_types(rd):
    """create_diffusion's other model types (init:32-45): fixed variances (learn_sigma=False [, sigma_small]) and
    x_start prediction, DDPM and DDIM, run by the reference on the synthetic model."""
    g = torch.Generator("cpu").manual_seed(77)
    z = torch.randn(2, 4, 4, 8, 8, generator=g)
    out = {"z": z.numpy()}
    steps = 6
    for tag, kw in (("fixed_large", dict(learn_sigma=False)), ("fixed_small", dict(learn_sigma=False, sigma_small=True)),
                    ("xstart_learned", dict(predict_xstart=True)), ("xstart_fixed_small", dict(predict_xstart=True, learn_sigma=False, sigma_small=True))):
        d = rd.create_diffusion(str(steps), **kw)
        oc = 8 if kw.get("learn_sigma", True) else 4
        fn = lambda x, t, **k: dor.synthetic_model(x, t, oc)
        for method in ("ddpm", "ddim"):
            torch.manual_seed(5)
            noises = [torch.randn_like(z) for _ in range(steps)]
            torch.manual_seed(5)
            if method == "ddpm":
                fin = d.p_sample_loop(fn, z.shape, z, clip_denoised=True, device="cpu")
            else:
                fin = d.ddim_sample_loop(fn, z.shape, z, clip_denoised=False, device="cpu", eta=0.4)
            out[f"{tag}::{method}"] = fin.numpy()
            out["noises"] = np.stack([n_.numpy() for n_ in noises])
    out["steps"] = np.int64(steps)
    np.savez_compressed(os.path.join(OUT, "sampler_types.npz"), **out)


TRAINING_CASES = [   # (tag, create_diffusion kwargs, respacing): the loss types / model types of init:10-46
    ("mse_learned", dict(), ""),                                           # train.py:100: create_diffusion(timestep_respacing="")
    ("rescaled_mse_learned_100", dict(rescale_learned_sigmas=True), "100"),
    ("rescaled_kl_learned", dict(use_kl=True), ""),
    ("mse_fixed_large", dict(learn_sigma=False), ""),
    ("mse_xstart_learned", dict(predict_xstart=True), "250"),
]


def training_inputs(n_steps):
    """Inputs of the training-loss fixtures: data partly outside +-0.999 (both tail branches of the discretized likelihood),
    t with the decoder-NLL case t == 0, the last step and interior steps."""
    g = torch.Generator("cpu").manual_seed(4242)
    x0 = (torch.randn(5, 4, 4, 8, 8, generator=g) * 0.6).clamp(-1.0, 1.0)
    noise = torch.randn(5, 4, 4, 8, 8, generator=g)
    t = torch.tensor([0, n_steps - 1, n_steps // 2, 1, (2 * n_steps) // 3], dtype=torch.int64)
    return x0, noise, t


def training(rd):
    """GaussianDiffusion.training_losses (gd:719-795) run by the reference on the synthetic model."""
    out = {}
    for tag, kw, spec in TRAINING_CASES:
        d = rd.create_diffusion(spec, **kw)
        x0, noise, t = training_inputs(d.num_timesteps)
        oc = 8 if kw.get("learn_sigma", True) else 4
        terms = d.training_losses(lambda x, tt, **k: dor.synthetic_model(x, tt, oc), x0, t, model_kwargs={}, noise=noise)
        for k, v in terms.items():
            out[f"{tag}::{k}"] = v.numpy()
        out[f"{tag}::x_t"] = d.q_sample(x0, t, noise=noise).numpy()
    np.savez_compressed(os.path.join(OUT, "training_losses.npz"), **out)


TRAIN_STEP = dict(depth=2, hidden_size=128, patch_size=2, num_heads=2, input_size=8, num_frames=4, num_classes=5, extras=2,
                  learn_sigma=True)


def train_step_inputs():
    """Inputs of the training-step fixture (tests/golden/train_step.npz): weights = oracle.latte_oracle.init_state_dict(seed 11)
    (so the fixture only has to store what the REFERENCE computed), three samples with t = 0 (decoder NLL), an interior step and
    the last one, one dropped label."""
    from oracle import latte_oracle as lo
    cfg = lo.LatteConfig(**TRAIN_STEP)
    sd = lo.init_state_dict(cfg, seed=11)
    g = torch.Generator("cpu").manual_seed(31)
    x0 = (torch.randn(3, 4, 4, 8, 8, generator=g) * 0.6).clamp(-1.0, 1.0)
    noise = torch.randn(3, 4, 4, 8, 8, generator=g)
    t = torch.tensor([0, 500, 999], dtype=torch.int64)
    y = torch.tensor([1, 4, 2], dtype=torch.int64)
    drop = torch.tensor([False, False, True])
    return cfg, sd, x0, noise, t, y, drop


def train_step(rl, rd):
    """One iteration of train.py:197-236 run by the reference objects: Latte(...).train() with the fixture's weights, the label
    dropout forced to the fixture's mask through the RNG-free path (labels replaced by num_classes before the call with
    class_dropout_prob = 0 -- LabelEmbedder.token_drop does exactly that replacement, latte.py:146-148),
    create_diffusion("").training_losses, loss.mean().backward(), the gradient norm of utils.clip_grad_norm_ (:103),
    torch.optim.AdamW(lr=1e-4, weight_decay=0).step(), update_ema(decay=0.9999)."""
    import copy
    cfg, sd, x0, noise, t, y, drop = train_step_inputs()
    # (class_dropout_prob = 0 in the constructor would remove the null-class row from the table, latte.py:130-131: build with
    #  the row, then switch the RNG path off)
    model = rl.Latte(**TRAIN_STEP)
    model.y_embedder.dropout_prob = 0.0
    model.load_state_dict(sd)
    model.train()
    ema = copy.deepcopy(model)
    yy = torch.where(drop, torch.full_like(y, TRAIN_STEP["num_classes"]), y)
    d = rd.create_diffusion("")
    opt = torch.optim.AdamW(model.parameters(), lr=1e-4, weight_decay=0)
    terms = d.training_losses(model, x0, t, dict(y=yy), noise=noise)
    terms["loss"].mean().backward()
    out = {f"terms::{k}": v.detach().numpy() for k, v in terms.items()}
    grads = {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}
    out["grad_norm"] = torch.norm(torch.stack([torch.norm(g_, 2.0) for g_ in grads.values()]), 2.0).numpy()
    for k, g_ in grads.items():
        out[f"grad::{k}"] = g_.numpy()
    opt.step()
    with torch.no_grad():
        ep = dict(ema.named_parameters())
        for n_, p_ in model.named_parameters():
            ep[n_].mul_(0.9999).add_(p_.data, alpha=1 - 0.9999)
    # the updated parameters / EMA of two tensors as spot checks (the rest follows from the gradients through the oracle)
    for k in ("blocks.1.attn.qkv.weight", "final_layer.linear.bias", "y_embedder.embedding_table.weight"):
        out[f"param::{k}"] = dict(model.named_parameters())[k].detach().numpy()
        out[f"ema::{k}"] = ep[k].detach().numpy()
    np.savez_compressed(os.path.join(OUT, "train_step.npz"), **out)


def main():
    os.makedirs(OUT, exist_ok=True)
    rl, rd = load_reference_latte(), load_reference_diffusion()
    schedules(rd)
    tiny_model(rl, rd, "tiny_classcond", TINY, use_cfg=True, seed=100)
    tiny_model(rl, rd, "tiny_uncond", TINY4, use_cfg=False, seed=200)
    tiny_model(rl, rd, "tiny_textcond", TINY78, use_cfg=True, seed=300)
    sampler_types(rd)
    training(rd)
    train_step(rl, rd)
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
