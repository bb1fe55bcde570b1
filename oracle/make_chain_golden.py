"""Generate ``tests/golden/chain250.npz``: 250-step sampling chains run by the REAL reference (build container only).

TEST INFRASTRUCTURE ONLY.  Usage:  python -m oracle.make_chain_golden [--skip-xl]

The benchmarked chain is DDIM on the "250" respacing (sample/sample.py:67 ``create_diffusion("250")`` driving
gaussian_diffusion.py:604-684, DDPM: :423-515).  Every case here runs the reference's own loop -- its ``Latte`` module
(models/latte.py, unmodified, timm stand-in) under its ``SpacedDiffusion.{ddim,p}_sample_loop`` -- for all 250 steps and
stores the sample after every 50th executed step plus the final latents.  The noise the loops draw
(``th.randn_like(x)`` at gaussian_diffusion.py:413,555) comes from ``torch.manual_seed(noise_seed)``; the k-th draw equals the
k-th ``torch.randn(x.shape)`` under the same seed (asserted below), which is how the parity test regenerates it.  Weights are
``oracle.latte_oracle.init_state_dict(cfg, seed)`` (regenerable from the seed, nothing large is stored).  The oracle's
``sample_loop`` is run beside the reference on the small cases and must agree to 1e-6 (it is bit-identical on the short
chains of oracle/validate_oracle.py; over 250 steps fp32 non-associativity between the two forward implementations may
show in the last bits).

Cases (name -> model, latent, conditioning, steps):
  s2_uncond   Latte-S/2  4 x 8x8    unconditional                  250 (ddim, ddpm)
  s2_guided   Latte-S/2  4 x 8x8    class-cond, CFG 7.0 (2 rows)    250 (ddim, ddpm)
  b2_uncond   Latte-B/2  16 x 16x16 unconditional                  250 (ddim, ddpm)
  b2_guided   Latte-B/2  16 x 16x16 class-cond, CFG 7.0            250 (ddim, ddpm)
  xl_segment  Latte-XL/2 16 x 32x32 unconditional, B = 2            the first 24 steps of the "250" respacing (ddim)
Round 4 -- weights at TRAINED-CHECKPOINT gate magnitudes (every tensor the reference zero-initialises drawn N(0, gate_std)
with gate_std = 0.3 instead of 0.02: gate_msa / gate_mlp of latte.py:178-180 are then O(0.1-1) as in a trained checkpoint and
the block branches reach the latents at full weight), and the benchmarked chain at its full length and size:
  s2_uncond_g03, b2_uncond_g03, b2_guided_g03     as above with gate_std 0.3            250 (ddim, ddpm)
  xl_full       Latte-XL/2 16 x 32x32 unconditional, B = 1, gate_std 0.02                250 (ddim)   -> chain250_xl.npz
  xl_full_g03   the same with gate_std 0.3                                               250 (ddim)   -> chain250_xl.npz
Round 5 -- BASELINE config 3's own model and call (sample/sample_ddp.py:140-160: UCF101 class-conditional Latte-XL/2 through
``forward_with_cfg``, latte.py:379-398, cfg_scale 7.0) at trained-scale gates, full length:
  xl_guided_g03 Latte-XL/2 16 x 32x32 class-cond (label 23 + null class), CFG 7.0, B = 1 (2 rows), gate_std 0.3   250 (ddim, ddpm: the
                YAMLs' default sample_method)   -> chain250_xl.npz
Round 6 -- the YAML-default sampler (DDPM, configs/ffs/ffs_sample.yaml:23-24) on WELL-CONDITIONED trained-scale weights.  The ``*_g03`` cases
above also draw ``final_layer.linear`` at 0.3, which makes eps and the learned-range v come out at rms ~ 10: the chains run at latent rms
1e3 - 1e6 and the reference's own last DDPM step exponentiates v far outside [-1, 1] (gaussian_diffusion.py:292-297), amplifying ANY relative
difference 3.2 x (profiles/r5_ddpm_conditioning.log).  They stay as stress rows.  The ``*_g03b`` / ``xl_uncond_g03`` cases keep every gate at
0.3 but draw the final projection as a trained checkpoint has it -- ``init_state_dict(final_std=0.03, v_scale=0.15)``: eps of rms ~ 1 (x CFG),
|v| < 0.6 on every step so ``frac`` stays inside [0, 1] -- and the latents stay at rms 1 ... 6e2 (the growth a random eps-model gives any chain):
  b2_guided_g03b  Latte-B/2  16 x 16x16 class-cond, CFG 7.0                                    250 (ddim, ddpm)
  xl_uncond_g03   Latte-XL/2 16 x 32x32 unconditional, B = 1: BASELINE config 2's own chain    250 (ddpm)         -> chain250_xl.npz
  xl_guided_g03b  Latte-XL/2 16 x 32x32 class-cond (label 23 + null), CFG 7.0, B = 1 (2 rows)  250 (ddpm)         -> chain250_xl.npz
``--only a,b`` regenerates the named cases and merges them into the existing file (the XL chains take ~30 min each, the guided one
~75 min per sampler); ``--methods ddpm`` restricts the regenerated cases to the named samplers (the others are kept from the file).
"""
import os
import sys
import time

import numpy as np
import torch

from oracle import diffusion_oracle as do
from oracle import latte_oracle as lo
from oracle.reference_loader import load_reference_diffusion, load_reference_latte

GOLDEN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
OUT = os.path.join(GOLDEN, "chain250.npz")
OUT_XL = os.path.join(GOLDEN, "chain250_xl.npz")      # the full-length XL/2 chains (kept apart: generated separately)

CFG_SCALE = 7.0
EVERY = 50
# name -> (preset, kwargs, weight seed, latent seed, noise seed, batch, guided label rows or None, steps, methods)
CASES = {
    "s2_uncond": ("Latte-S/2", dict(input_size=8, num_frames=4, extras=1), 11, 12, 13, 1, None, 250, ("ddim", "ddpm")),
    "s2_guided": ("Latte-S/2", dict(input_size=8, num_frames=4, num_classes=101, extras=2), 21, 22, 23, 1, [17], 250,
                  ("ddim", "ddpm")),
    "b2_uncond": ("Latte-B/2", dict(input_size=16, num_frames=16, extras=1), 31, 32, 33, 1, None, 250, ("ddim", "ddpm")),
    "b2_guided": ("Latte-B/2", dict(input_size=16, num_frames=16, num_classes=101, extras=2), 41, 42, 43, 1, [5], 250,
                  ("ddim", "ddpm")),
    "xl_segment": ("Latte-XL/2", dict(input_size=32, num_frames=16, extras=1), 51, 52, 53, 2, None, 24, ("ddim",)),
    "s2_uncond_g03": ("Latte-S/2", dict(input_size=8, num_frames=4, extras=1), 61, 62, 63, 1, None, 250, ("ddim", "ddpm")),
    "b2_uncond_g03": ("Latte-B/2", dict(input_size=16, num_frames=16, extras=1), 71, 72, 73, 1, None, 250, ("ddim", "ddpm")),
    "b2_guided_g03": ("Latte-B/2", dict(input_size=16, num_frames=16, num_classes=101, extras=2), 81, 82, 83, 1, [5], 250,
                      ("ddim", "ddpm")),
    "xl_full": ("Latte-XL/2", dict(input_size=32, num_frames=16, extras=1), 91, 92, 93, 1, None, 250, ("ddim",)),
    "xl_full_g03": ("Latte-XL/2", dict(input_size=32, num_frames=16, extras=1), 101, 102, 103, 1, None, 250, ("ddim",)),
    "xl_guided_g03": ("Latte-XL/2", dict(input_size=32, num_frames=16, num_classes=101, extras=2), 111, 112, 113, 1, [23], 250,
                      ("ddim", "ddpm")),
    "b2_guided_g03b": ("Latte-B/2", dict(input_size=16, num_frames=16, num_classes=101, extras=2), 121, 122, 123, 1, [5], 250,
                       ("ddim", "ddpm")),
    "xl_uncond_g03": ("Latte-XL/2", dict(input_size=32, num_frames=16, extras=1), 131, 132, 133, 1, None, 250, ("ddpm",)),
    "xl_guided_g03b": ("Latte-XL/2", dict(input_size=32, num_frames=16, num_classes=101, extras=2), 141, 142, 143, 1, [23], 250,
                       ("ddpm",)),
}
GATE_STD = {"s2_uncond_g03": 0.3, "b2_uncond_g03": 0.3, "b2_guided_g03": 0.3, "xl_full_g03": 0.3,
            "xl_guided_g03": 0.3, "b2_guided_g03b": 0.3, "xl_uncond_g03": 0.3, "xl_guided_g03b": 0.3}   # default 0.02
# name -> (final_std, v_scale) of init_state_dict: the final projection at trained-checkpoint output scale (default: gate_std, 1.0)
FINAL = {"b2_guided_g03b": (0.03 * (1152 / 768) ** 0.5, 0.15), "xl_uncond_g03": (0.03, 0.15), "xl_guided_g03b": (0.03, 0.15)}
XL_FILE_CASES = ("xl_full", "xl_full_g03", "xl_guided_g03", "xl_uncond_g03", "xl_guided_g03b")


def case_file(name):
    return OUT_XL if name in XL_FILE_CASES else OUT


def case_inputs(name):
    """-> (preset, kw, state_dict, x0 [rows, F, 4, S, S], y or None, steps, methods, noise_seed).  Guided cases: rows = 2 x batch,
    x0 = cat([z, z]), y = [labels..., null class...] (sample/sample.py:92-99)."""
    preset, kw, wseed, xseed, nseed, B, labels, steps, methods = CASES[name]
    cfg = lo.preset_config(preset, **kw)
    fstd, vsc = FINAL.get(name, (None, 1.0))
    sd = lo.init_state_dict(cfg, seed=wseed, gate_std=GATE_STD.get(name, 0.02), final_std=fstd, v_scale=vsc)
    g = torch.Generator("cpu").manual_seed(xseed)
    z = torch.randn(B, kw["num_frames"], 4, kw["input_size"], kw["input_size"], generator=g)
    if labels is None:
        return preset, kw, cfg, sd, z, None, steps, methods, nseed
    y = torch.tensor(list(labels) + [kw["num_classes"]] * B, dtype=torch.int64)
    return preset, kw, cfg, sd, torch.cat([z, z]), y, steps, methods, nseed


def chain_noises(seed, shape, n):
    """The n noise tensors a reference loop draws after torch.manual_seed(seed) (one th.randn_like(x) per step)."""
    g = torch.Generator("cpu").manual_seed(seed)
    return [torch.randn(shape, generator=g) for _ in range(n)]


def main():
    rl, rd = load_reference_latte(), load_reference_diffusion()
    # the k-th randn_like under the global seed == the k-th generator draw of the same shape (what the test regenerates)
    torch.manual_seed(99)
    a = [torch.randn_like(torch.empty(2, 3, 5)) for _ in range(3)]
    b = chain_noises(99, (2, 3, 5), 3)
    assert all(torch.equal(p, q) for p, q in zip(a, b))
    only = None
    only_methods = None
    for i, a in enumerate(sys.argv):
        if a == "--only":
            only = set(sys.argv[i + 1].split(","))
        if a == "--methods":
            only_methods = set(sys.argv[i + 1].split(","))
    arrays = {OUT: {}, OUT_XL: {}}
    for path in arrays:
        if only is not None and os.path.exists(path):      # merge into what is there
            with np.load(path) as z:
                arrays[path] = {k: z[k] for k in z.files}
    for name in CASES:
        if only is not None and name not in only:
            continue
        if name.startswith("xl") and "--skip-xl" in sys.argv:
            continue
        if only is None and name in XL_FILE_CASES:       # the long chains only on request
            continue
        preset, kw, cfg, sd, x0, y, steps, methods, nseed = case_inputs(name)
        model = rl.Latte_models[preset](**kw).eval()
        model.load_state_dict(sd, strict=True)
        d = rd.create_diffusion("250")
        s = do.Schedule("250")
        for method in methods:
            if only_methods is not None and method not in only_methods:
                continue
            t0 = time.time()
            loop = d.ddim_sample_loop_progressive if method == "ddim" else d.p_sample_loop_progressive
            if y is None:
                fn, mk = model.forward, dict(y=None)
            else:
                fn, mk = model.forward_with_cfg, dict(y=y, cfg_scale=CFG_SCALE)
            keep, rms = [], []
            torch.manual_seed(nseed)
            with torch.no_grad():
                for k, r in enumerate(loop(fn, x0.shape, noise=x0.clone(), clip_denoised=False, model_kwargs=mk, device="cpu")):
                    rms.append(float(r["sample"].pow(2).mean().sqrt()))
                    if (k + 1) % EVERY == 0 or k + 1 == steps:
                        keep.append((k, r["sample"].clone()))
                    if k + 1 == steps:
                        break
            arrays[case_file(name)][f"{name}::{method}::steps"] = np.asarray([k for k, _ in keep], dtype=np.int64)
            arrays[case_file(name)][f"{name}::{method}::samples"] = torch.stack([v for _, v in keep]).numpy()
            arrays[case_file(name)][f"{name}::{method}::rms"] = np.asarray(rms, dtype=np.float32)      # latent rms after every step
            msg = f"{name} {method}: {steps} reference steps in {time.time() - t0:.1f}s, |x_final| rms {float(keep[-1][1].pow(2).mean().sqrt()):.3f}"
            if not name.startswith("xl"):   # the oracle beside the reference (pin over the full chain length)
                nz = chain_noises(nseed, x0.shape, steps)
                if y is None:
                    mfn = lambda xx, tt: lo.latte_forward(sd, cfg, xx, tt)
                else:
                    mfn = lambda xx, tt: lo.latte_forward_with_cfg(sd, cfg, xx, tt, y, CFG_SCALE)
                with torch.no_grad():
                    want = do.sample_loop(s, mfn, x0.clone(), method=method, noises=nz)
                err = float((want - keep[-1][1]).norm() / keep[-1][1].norm())
                msg += f"; oracle loop vs reference loop rel-L2 {err:.2e}"
                assert err < 1e-6, err
            print(msg, flush=True)
    for path, arr in arrays.items():
        if arr:
            np.savez_compressed(path, **arr)
            print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
