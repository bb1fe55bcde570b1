"""Checkpoint ingestion through the reference's own entry points:
``find_model`` on a ``torch.save``d training checkpoint (utils.py:274-287, sample.py:62-64), ``get_models`` for every model
family it dispatches (models/__init__.py:31-51), and the diffusers directory layout (``config.json`` +
``diffusion_pytorch_model.safetensors``) for ``AutoencoderKL.from_pretrained`` / ``LatteT2V.from_pretrained[_2d]``
(sample.py:69, sample_t2x.py:29-34).  CPU tests check what is loaded; the ``gpu`` tests check that the loaded model computes
what the reference computed."""
import json
import os

import numpy as np
import pytest
import torch

import latte_amd
from _util import GOLDEN, load_golden_model, rel_l2

TOL = 1e-3


def _save_ckpt(tmp_path, sd, with_ema=True):
    other = {k: torch.zeros_like(v) for k, v in sd.items()}
    ck = {"model": other if with_ema else sd, "opt": {"step": 3}, "args": None}
    if with_ema:
        ck["ema"] = sd
    path = os.path.join(tmp_path, "0003000.pt")
    torch.save(ck, path)                                    # train.py:258-266 layout
    return path


def _t2v_fixture():
    from oracle import latte_t2v_oracle as to
    z = np.load(os.path.join(GOLDEN, "tiny_t2v.npz"))
    cfg = json.loads(bytes(z["cfg_json"]).decode())
    sd = {k[4:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd::")}
    return to.T2VConfig(**cfg), cfg, sd, z


def _write_diffusers_dir(root, cfg, sd, safetensors=True):
    os.makedirs(root, exist_ok=True)
    with open(os.path.join(root, "config.json"), "w") as f:
        json.dump({"_class_name": "X", "_diffusers_version": "0.24.0", **cfg}, f)
    if safetensors:
        from safetensors.torch import save_file
        save_file({k: v.contiguous() for k, v in sd.items()}, os.path.join(root, "diffusion_pytorch_model.safetensors"))
    else:
        torch.save(sd, os.path.join(root, "diffusion_pytorch_model.bin"))


# ------------------------------------------------------------------------------------------------ CPU
def test_find_model_prefers_ema_and_loads_strictly(tmp_path, capsys):
    kw, sd, _ = load_golden_model("tiny_classcond")
    path = _save_ckpt(str(tmp_path), sd, with_ema=True)
    got = latte_amd.find_model(path)
    assert "Using Ema!" in capsys.readouterr().out
    assert set(got) == set(sd) and all(torch.equal(got[k], sd[k]) for k in sd)
    m = latte_amd.Latte(**kw)
    m.load_state_dict(got)                                  # strict (sample.py:64)
    assert all(torch.equal(v, sd[k]) for k, v in m.state_dict().items())
    path = _save_ckpt(str(tmp_path), sd, with_ema=False)
    got = latte_amd.find_model(path)
    assert "Using model!" in capsys.readouterr().out and torch.equal(got["pos_embed"], sd["pos_embed"])
    with pytest.raises(AssertionError):
        latte_amd.find_model(os.path.join(str(tmp_path), "missing.pt"))


def test_get_models_dispatch(tmp_path):
    args = latte_amd.Config(model="Latte-S/2", latent_size=8, num_classes=5, num_frames=4, learn_sigma=True, extras=2)
    m = latte_amd.get_models(args)
    assert isinstance(m, latte_amd.Latte) and m.hidden_size == 384 and m.y_embedder.embedding_table.weight.shape[0] == 6
    tcfg, cfg, sd, _ = _t2v_fixture()
    _write_diffusers_dir(os.path.join(str(tmp_path), "transformer"), cfg, sd)
    t2v = latte_amd.get_models(latte_amd.Config(model="LatteT2V", pretrained_model_path=str(tmp_path),
                                                video_length=cfg["video_length"]))
    assert isinstance(t2v, latte_amd.LatteT2V) and t2v.config.video_length == cfg["video_length"]
    assert set(t2v.state_dict()) == set(sd) and all(torch.equal(t2v.state_dict()[k], sd[k]) for k in sd)
    with pytest.raises(latte_amd.LatteError):
        latte_amd.get_models(latte_amd.Config(model="LatteIMG-XL/2", latent_size=32, num_classes=0, num_frames=16,
                                              learn_sigma=True, extras=1))
    with pytest.raises(latte_amd.LatteError):
        latte_amd.get_models(latte_amd.Config(model="DiT-XL/2"))


@pytest.mark.parametrize("safetensors", [True, False])
def test_vae_from_pretrained_directory(tmp_path, safetensors):
    from oracle import vae_oracle as vo
    sd = vo.init_state_dict(seed=8)
    full = {**sd, "encoder.conv_in.weight": torch.zeros(128, 3, 3, 3), "quant_conv.weight": torch.zeros(8, 8, 1, 1)}
    cfg = {"scaling_factor": 0.18215, "block_out_channels": [128, 256, 512, 512], "layers_per_block": 2,
           "latent_channels": 4, "norm_num_groups": 32, "act_fn": "silu", "sample_size": 256}
    _write_diffusers_dir(os.path.join(str(tmp_path), "vae"), cfg, full, safetensors)
    vae = latte_amd.AutoencoderKL.from_pretrained(str(tmp_path), subfolder="vae")
    assert vae.config.scaling_factor == 0.18215 and vae.config.block_out_channels == [128, 256, 512, 512]
    got = vae.state_dict()
    assert set(got) == set(sd) and all(torch.equal(got[k], sd[k]) for k in sd)      # decoder half only
    with pytest.raises(latte_amd.LatteError):
        latte_amd.AutoencoderKL.from_pretrained(str(tmp_path), subfolder="nothing_here")


# ------------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
def test_checkpoint_file_to_gpu_parity(tmp_path):
    """torch.save({'ema': sd}) -> find_model -> load_state_dict -> .to(device) -> the reference's own forward output."""
    kw, sd, r = load_golden_model("tiny_classcond")
    path = _save_ckpt(str(tmp_path), sd)
    args = latte_amd.Config(model=None, latent_size=kw["input_size"], num_classes=kw["num_classes"],
                            num_frames=kw["num_frames"], learn_sigma=kw["learn_sigma"], extras=kw["extras"])
    m = latte_amd.Latte(**kw)
    m.load_state_dict(latte_amd.find_model(path))
    m = m.to("cuda").eval()
    x, t, y = (torch.from_numpy(r[k]).cuda() for k in ("x", "t", "y"))
    assert rel_l2(m(x, t, y=y), torch.from_numpy(r["forward"])) < TOL
    out = m.forward_with_cfg(torch.from_numpy(r["x_cfg"]).cuda(), t, y=torch.from_numpy(r["y_cfg"]).cuda(),
                             cfg_scale=float(r["cfg_scale"]))
    assert rel_l2(out, torch.from_numpy(r["forward_with_cfg"])) < TOL
    del args


@pytest.mark.gpu
def test_t2v_directory_to_gpu_parity(tmp_path):
    tcfg, cfg, sd, z = _t2v_fixture()
    _write_diffusers_dir(os.path.join(str(tmp_path), "transformer"), cfg, sd)
    m = latte_amd.get_models(latte_amd.Config(model="LatteT2V", pretrained_model_path=str(tmp_path),
                                              video_length=cfg["video_length"])).to("cuda", dtype=torch.float16)  # sample_t2x.py:29
    x, t = torch.from_numpy(z["x"]).cuda(), torch.from_numpy(z["t"]).cuda()
    enc, mask = torch.from_numpy(z["encoder_hidden_states"]).cuda(), torch.from_numpy(z["encoder_attention_mask"]).cuda()
    out = m(x, timestep=t, encoder_hidden_states=enc, encoder_attention_mask=mask, return_dict=False)[0]
    assert rel_l2(out, torch.from_numpy(z["forward"])) < TOL


@pytest.mark.gpu
def test_vae_directory_to_gpu_decode(tmp_path):
    from oracle import vae_oracle as vo
    sd = vo.init_state_dict(seed=8)
    cfg = {"scaling_factor": 0.18215, "block_out_channels": [128, 256, 512, 512], "layers_per_block": 2,
           "latent_channels": 4, "norm_num_groups": 32}
    _write_diffusers_dir(os.path.join(str(tmp_path), "vae"), cfg, sd)
    vae = latte_amd.AutoencoderKL.from_pretrained(str(tmp_path), subfolder="vae", latent_size=16, max_frames=1).to("cuda")
    vae.to(dtype=torch.float16)                                                             # sample.py:74
    zl = torch.randn(1, 4, 16, 16, generator=torch.Generator().manual_seed(3))
    got = vae.decode(zl.cuda()).sample
    assert rel_l2(got, vo.decode(sd, zl)) < TOL
