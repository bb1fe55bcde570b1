"""One-command pin of the third-party arithmetic the oracles restate from memory (run wherever ``diffusers==0.24.0`` imports).

TEST INFRASTRUCTURE ONLY.  Usage:

    python -m oracle.pin_third_party            # checks every restatement against diffusers, writes the fixture
    python -m oracle.pin_third_party --check    # the same comparisons, nothing written

The reference leans on diffusers 0.24.0 (requirements / environment.yml) for four pieces that are NOT vendored under
/root/reference and cannot be installed in the build container (no network), so four oracles are "parity unpinned":

  piece (reference call site)                                             restatement in this repo
  ----------------------------------------------------------------------  ----------------------------------------------------
  AutoencoderKL.decode            (sample/sample.py:69,113-115)           oracle/vae_oracle.py          decode()
  AutoencoderKLTemporalDecoder.decode (sample/pipeline_latte.py:779-798)  oracle/vae_temporal_oracle.py decode()
  DDIMScheduler.set_timesteps / .step (sample/pipeline_latte.py:747-758,  latte_amd/schedulers.py       DDIMScheduler
                                   sample/sample_t2x.py:43-50)
  the diffusers leaves imported by models/latte_t2v.py:9-20               oracle/diffusers_standin.py (under the reference's own
   (PatchEmbed, CaptionProjection, CombinedTimestepSizeEmbeddings,        latte_t2v.py) == oracle/latte_t2v_oracle.py
    Attention, FeedForward / GELU, AdaLayerNormSingle)

This script builds each diffusers object with RANDOM weights of the real architecture (no checkpoint is needed: the
restatements take the same state-dict keys), runs both sides on seeded inputs in fp32 on the CPU, asserts agreement
(1e-5 relative: fp32 re-association only) and stores inputs + diffusers outputs in ``tests/golden/third_party_pin.npz``.
From then on ``tests/test_third_party_pin.py`` checks the restatements against that fixture everywhere (no diffusers
needed), and rows a22 / f2 of SURVEY.md section 8 are pinned.  Until it has been run the header of each oracle keeps saying
"parity unpinned" and so does DESIGN.md.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden", "third_party_pin.npz")
TOL = 1e-5
WANT_VERSION = "0.24.0"


def diffusers_available():
    try:
        import diffusers
        if getattr(diffusers, "__standin__", False):      # oracle/diffusers_standin.py registered itself in this process
            return False
        from diffusers.models import AutoencoderKL  # noqa: F401
        return True
    except Exception:
        return False


def rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


# ---------------------------------------------------------------------------------------------- SD-VAE decoder
def vae_inputs():
    g = torch.Generator("cpu").manual_seed(101)
    return torch.randn(2, 4, 16, 16, generator=g)


def vae_restated(z, seed=7):
    from oracle import vae_oracle as vo
    return vo.decode(vo.init_state_dict(seed=seed), z)


def pin_vae(arrays):
    """AutoencoderKL(sd-vae-ft config).decode(z).sample vs oracle/vae_oracle.py on the same random decoder weights."""
    from diffusers.models import AutoencoderKL
    from oracle import vae_oracle as vo
    vae = AutoencoderKL(in_channels=3, out_channels=3, down_block_types=("DownEncoderBlock2D",) * 4,
                        up_block_types=("UpDecoderBlock2D",) * 4, block_out_channels=(128, 256, 512, 512), layers_per_block=2,
                        act_fn="silu", latent_channels=4, norm_num_groups=32, sample_size=256).eval()
    sd = vo.init_state_dict(seed=7)
    own = vae.state_dict()
    missing = [k for k in sd if k not in own]
    assert not missing, f"decoder keys the real AutoencoderKL does not have: {missing[:5]}"
    dec = [k for k in own if k.startswith(("decoder.", "post_quant_conv."))]
    assert sorted(dec) == sorted(sd), sorted(set(dec) ^ set(sd))[:8]
    vae.load_state_dict({**own, **sd})
    z = vae_inputs()
    with torch.no_grad():
        want = vae.decode(z).sample
    err = rel(vae_restated(z), want)
    arrays["vae::z"], arrays["vae::out"] = z.numpy(), want.numpy()
    return "AutoencoderKL.decode (2 frames, 16x16 latent)", err


# ---------------------------------------------------------------------------------------------- temporal decoder
def vae_t_inputs():
    g = torch.Generator("cpu").manual_seed(102)
    return torch.randn(3, 4, 16, 16, generator=g)


def vae_t_restated(z, seed=9):
    from oracle import vae_temporal_oracle as vt
    return vt.decode(vt.init_state_dict(seed=seed), z, num_frames=z.shape[0])


def pin_vae_temporal(arrays):
    """AutoencoderKLTemporalDecoder.decode(z, num_frames).sample vs oracle/vae_temporal_oracle.py."""
    from diffusers.models import AutoencoderKLTemporalDecoder
    from oracle import vae_temporal_oracle as vt
    vae = AutoencoderKLTemporalDecoder(in_channels=3, out_channels=3, down_block_types=("DownEncoderBlock2D",) * 4,
                                       block_out_channels=(128, 256, 512, 512), layers_per_block=2, latent_channels=4,
                                       sample_size=256).eval()
    sd = vt.init_state_dict(seed=9)
    own = vae.state_dict()
    dec = [k for k in own if k.startswith("decoder.")]
    assert sorted(dec) == sorted(k for k in sd if k.startswith("decoder.")), sorted(set(dec) ^ set(sd))[:8]
    vae.load_state_dict({**own, **{k: v for k, v in sd.items() if k in own}})
    z = vae_t_inputs()
    with torch.no_grad():
        want = vae.decode(z, num_frames=z.shape[0]).sample
    err = rel(vae_t_restated(z), want)
    arrays["vae_t::z"], arrays["vae_t::out"] = z.numpy(), want.numpy()
    return "AutoencoderKLTemporalDecoder.decode (one 3-frame chunk, 16x16 latent)", err


# ---------------------------------------------------------------------------------------------- DDIM scheduler
SCHED_KW = dict(num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="linear", clip_sample=False)


def sched_inputs():
    g = torch.Generator("cpu").manual_seed(103)
    return torch.randn(2, 4, 4, 8, 8, generator=g), torch.randn(2, 4, 4, 8, 8, generator=g)


def sched_restated(n_steps, sample, eps):
    from latte_amd.schedulers import DDIMScheduler
    s = DDIMScheduler(**SCHED_KW)
    s.set_timesteps(n_steps)
    ts = s.timesteps.clone()
    outs = []
    x = sample
    for t in ts:
        x = s.step(eps, t, x, eta=0.0, return_dict=False)[0]
        outs.append(x)
    return ts, torch.stack(outs)


def pin_scheduler(arrays):
    """diffusers.DDIMScheduler (the arguments of sample_t2x.py:43-50: linear betas 1e-4..2e-2, clip_sample=False) driven as
    pipeline_latte.py:747-758 does: set_timesteps(n), then step(noise_pred, t, latents) over all timesteps, eta = 0."""
    from diffusers import DDIMScheduler
    sample, eps = sched_inputs()
    worst = 0.0
    for n in (50, 20):
        s = DDIMScheduler(**SCHED_KW)
        s.set_timesteps(n)
        ts, mine = sched_restated(n, sample, eps)
        assert torch.equal(torch.as_tensor(s.timesteps).long(), ts.long()), (s.timesteps[:5], ts[:5])
        x, outs = sample, []
        for t in s.timesteps:
            x = s.step(eps, t, x, eta=0.0, return_dict=False)[0]
            outs.append(x)
        want = torch.stack(outs)
        worst = max(worst, max(rel(mine[i], want[i]) for i in range(n)))
        arrays[f"sched::{n}::timesteps"] = np.asarray(s.timesteps).astype(np.int64)
        arrays[f"sched::{n}::trajectory"] = want.numpy()
    arrays["sched::sample"], arrays["sched::eps"] = sample.numpy(), eps.numpy()
    return "DDIMScheduler.set_timesteps / .step (50 and 20 steps, eta 0)", worst


# ---------------------------------------------------------------------------------------------- latte_t2v.py leaves
def t2v_case():
    from oracle import latte_t2v_oracle as to
    cfg = to.T2VConfig(num_attention_heads=2, attention_head_dim=72, num_layers=2, sample_size=16, cross_attention_dim=144,
                       caption_channels=48, video_length=4)
    g = torch.Generator("cpu").manual_seed(104)
    x = torch.randn(2, cfg.in_channels, cfg.video_length, cfg.sample_size, cfg.sample_size, generator=g)
    t = torch.tensor([999, 37])
    enc = torch.randn(2, 10, cfg.caption_channels, generator=g)
    mask = torch.ones(2, 10)
    mask[1, 7:] = 0
    return cfg, x, t, enc, mask


def pin_t2v_leaves(arrays):
    """The reference's own models/latte_t2v.py, imported on the REAL diffusers (not on oracle/diffusers_standin.py), against
    oracle/latte_t2v_oracle.py -- which is pinned to the same file on the stand-in, so agreement here pins the stand-in's leaves.
    Needs /root/reference (build container) as well as diffusers."""
    import importlib.util
    from oracle import latte_t2v_oracle as to
    from oracle.reference_loader import REFERENCE_ROOT, reference_available
    from oracle.validate_t2v_oracle import build
    if not reference_available():
        return "latte_t2v.py on real diffusers", None
    assert not getattr(sys.modules.get("diffusers"), "__standin__", False), \
        "run the pin in a fresh process: the diffusers stand-in is installed in this one"
    spec = importlib.util.spec_from_file_location("_reference_latte_t2v_real", os.path.join(REFERENCE_ROOT, "models", "latte_t2v.py"))
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    cfg, x, t, enc, mask = t2v_case()
    net = build(ref, cfg, seed=11)
    sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
    assert set(sd) == set(to.state_dict_keys(cfg)), sorted(set(sd) ^ set(to.state_dict_keys(cfg)))[:8]
    with torch.no_grad():
        want = net(x, timestep=t, encoder_hidden_states=enc, encoder_attention_mask=mask,
                   added_cond_kwargs={"resolution": None, "aspect_ratio": None}, enable_temporal_attentions=True,
                   return_dict=False)[0]
        got = to.latte_t2v_forward(sd, cfg, x, t, enc, mask)
    for k, v in sd.items():
        arrays["t2v::sd::" + k] = v.numpy()
    arrays["t2v::out"] = want.numpy()
    return "models/latte_t2v.py on real diffusers leaves vs oracle/latte_t2v_oracle.py", rel(got, want)


def main():
    if not diffusers_available():
        print("diffusers is not importable here: nothing pinned (the oracles stay 'parity unpinned').  Run this script in an "
              f"environment with diffusers=={WANT_VERSION} and torch, from the repo root.")
        return 2
    import diffusers
    if diffusers.__version__ != WANT_VERSION:
        print(f"warning: diffusers {diffusers.__version__}, the reference pins {WANT_VERSION}")
    arrays, rows, bad = {}, [], 0
    for fn in (pin_vae, pin_vae_temporal, pin_scheduler, pin_t2v_leaves):
        what, err = fn(arrays)
        if err is None:
            rows.append((what, "skipped (needs /root/reference)"))
            continue
        ok = err < TOL
        bad += not ok
        rows.append((what, f"rel-L2 {err:.2e} {'OK' if ok else 'MISMATCH (restatement differs from diffusers)'}"))
    for what, res in rows:
        print(f"| {what} | {res} |")
    if bad:
        print("NOT writing the fixture: fix the restatement(s) above first")
        return 1
    if "--check" not in sys.argv:
        arrays["diffusers_version"] = np.frombuffer(diffusers.__version__.encode(), dtype=np.uint8)
        np.savez_compressed(OUT, **arrays)
        with open(os.path.join(ROOT, "oracle", "VALIDATION.md"), "a") as f:
            f.write(f"\n## Third-party pin (python -m oracle.pin_third_party, diffusers {diffusers.__version__})\n\n| check | result |\n|---|---|\n")
            for what, res in rows:
                f.write(f"| {what} | {res} |\n")
        print("wrote", OUT)
    return 0


if __name__ == "__main__":
    sys.exit(main())
