"""Parity of the HIP engine (through the C-ABI + host shim) with the reference.

* golden fixtures produced by the REAL reference (tests/golden, oracle/make_golden.py);
* the oracle restatement on seeded inputs at S/2, B/2, L/2 and full XL/2 size;
* size-independent properties of the sampler at the full XL/2 latent size.
Tolerance: north_star's 1e-3 relative (rel-L2) on denoised latents / model outputs, written below.
"""
import numpy as np
import pytest
import torch

import latte_amd
from _util import engine_model, load_golden_model, rel_l2
from latte_amd._lib import check, load_library, ptr, stream_ptr

pytestmark = pytest.mark.gpu
TOL = 1e-3   # north_star: "within 1e-3 rel fp for denoised latents"
DTYPES = ["bf16", "f16"]


@pytest.mark.parametrize("cd", DTYPES)
@pytest.mark.parametrize("name", ["tiny_classcond", "tiny_uncond"])
def test_forward_matches_reference_golden(name, cd):
    kw, sd, r = load_golden_model(name)
    m = engine_model(kw, sd, cd)
    x, t = torch.from_numpy(r["x"]).cuda(), torch.from_numpy(r["t"]).cuda()
    y = torch.from_numpy(r["y"]).cuda() if "y" in r else None
    assert rel_l2(m(x, t, y=y), torch.from_numpy(r["forward"])) < TOL
    if "forward_with_cfg" in r:
        out = m.forward_with_cfg(torch.from_numpy(r["x_cfg"]).cuda(), t, y=torch.from_numpy(r["y_cfg"]).cuda(),
                                 cfg_scale=float(r["cfg_scale"]))
        assert rel_l2(out, torch.from_numpy(r["forward_with_cfg"])) < TOL


@pytest.mark.parametrize("cd", DTYPES)
@pytest.mark.parametrize("name", ["tiny_classcond", "tiny_uncond"])
@pytest.mark.parametrize("method", ["ddim", "ddpm"])
def test_fused_loop_matches_reference_trajectory(name, cd, method):
    """latte_sample_loop fed the reference's own noise draws: every step's sample and pred_xstart."""
    kw, sd, r = load_golden_model(name)
    m = engine_model(kw, sd, cd)
    steps = int(r["loop_steps"])
    d = latte_amd.create_diffusion(str(steps))
    if "x_cfg" in r:
        z, y, scale = torch.from_numpy(r["x_cfg"]).cuda(), torch.from_numpy(r["y_cfg"]).cuda(), float(r["cfg_scale"])
    else:
        z, scale = torch.from_numpy(r["x"]).cuda(), 1.0
        y = torch.from_numpy(r["y"]).cuda() if "y" in r else None
    xx = z.clone().contiguous()
    nz = torch.from_numpy(r[f"{method}_noises"]).cuda().contiguous()
    ts = torch.empty((steps,) + tuple(xx.shape), device="cuda")
    t0 = torch.empty_like(ts)
    check(load_library().latte_sample_loop(m.engine(xx.shape[0]), d._h, 1 if method == "ddim" else 0, 0.0, 0, scale,
                                           ptr(xx), ptr(y), xx.shape[0], steps - 1, 0, ptr(nz), ptr(ts), ptr(t0),
                                           stream_ptr()))
    torch.cuda.synchronize()
    for k in range(steps):
        assert rel_l2(ts[k], torch.from_numpy(r[f"{method}_samples"][k])) < TOL, k
        assert rel_l2(t0[k], torch.from_numpy(r[f"{method}_pred_xstart"][k])) < TOL, k
    assert torch.equal(xx, ts[-1])


@pytest.mark.parametrize("cd", DTYPES)
def test_reference_style_driver_ddim(cd):
    """The body of sample/sample.py:88-107 written against latte_amd, vs the reference's final latents."""
    kw, sd, r = load_golden_model("tiny_classcond")
    model = engine_model(kw, sd, cd)
    diffusion = latte_amd.create_diffusion(str(int(r["loop_steps"])))
    z = torch.from_numpy(r["x_cfg"]).cuda()
    model_kwargs = dict(y=torch.from_numpy(r["y_cfg"]).cuda(), cfg_scale=7.0, use_fp16=False)
    samples = diffusion.ddim_sample_loop(model.forward_with_cfg, z.shape, z, clip_denoised=False,
                                         model_kwargs=model_kwargs, progress=False, device="cuda")
    assert rel_l2(samples, torch.from_numpy(r["ddim_samples"][-1])) < TOL
    # generic-callable path (any function following the model-callable protocol) gives the same chain
    fn = lambda x, t, **kw_: model.forward_with_cfg(x, t, **kw_)
    samples2 = diffusion.ddim_sample_loop(fn, z.shape, z, clip_denoised=False, model_kwargs=model_kwargs, device="cuda")
    assert rel_l2(samples2, samples) < 1e-6
    # progressive generator yields every step
    trail = list(diffusion.ddim_sample_loop_progressive(model.forward_with_cfg, z.shape, z, clip_denoised=False,
                                                        model_kwargs=model_kwargs, device="cuda"))
    assert len(trail) == int(r["loop_steps"])
    assert rel_l2(trail[0]["pred_xstart"], torch.from_numpy(r["ddim_pred_xstart"][0])) < TOL


def test_ddim_eta_noise_path():
    """eta > 0 (sigma/noise branch, gd:549-563) through the step API with the reference's draws."""
    kw, sd, r = load_golden_model("tiny_classcond")
    m = engine_model(kw, sd, "f16")
    steps = int(r["loop_steps"])
    d = latte_amd.create_diffusion(str(steps))
    x = torch.from_numpy(r["x_cfg"]).cuda()
    y = torch.from_numpy(r["y_cfg"]).cuda()
    nz = torch.from_numpy(r["ddim_noises"]).cuda()
    for k, i in enumerate(range(steps - 1, -1, -1)):
        t = torch.full((x.shape[0],), d.timestep_map[i], device="cuda", dtype=torch.int64)
        out = m.forward_with_cfg(x, t, y=y, cfg_scale=7.0)
        x = d._step("ddim", out, x, i, nz[k], 0.5, False)["sample"]
    assert rel_l2(x, torch.from_numpy(r["ddim_eta05_final"])) < TOL


ORACLE_CASES = [
    ("Latte-S/2", dict(input_size=8, num_frames=4, num_classes=101, extras=2), 2),      # plumbing config (8x8)
    ("Latte-S/2", dict(input_size=64, num_frames=4, extras=1), 1),                       # plumbing config (64x64)
    ("Latte-B/2", dict(input_size=16, num_frames=16, extras=1), 1),
    ("Latte-L/2", dict(input_size=16, num_frames=8, num_classes=10, extras=2), 1),
    ("Latte-S/4", dict(input_size=32, num_frames=4, extras=1), 1),
    ("Latte-S/8", dict(input_size=32, num_frames=2, extras=1), 2),
    ("Latte-XL/2", dict(input_size=32, num_frames=16, num_classes=101, extras=2), 1),    # headline size
    ("Latte-S/2", dict(input_size=32, num_frames=16, extras=1), 2),     # 256 tokens x 16 frames at hd = 64: the fused qkv + attention kernel
]


_ORACLE_FORWARD = {}     # case index -> (sd, x, t, y, ref): the fp32 oracle forward is the same for both operand types (one run per case)


def _oracle_forward(case):
    from oracle import latte_oracle as lo
    key = ORACLE_CASES.index(case)
    if key not in _ORACLE_FORWARD:
        name, kw, B = case
        cfg = lo.preset_config(name, **kw)
        sd = lo.init_state_dict(cfg, seed=0)
        g = torch.Generator("cpu").manual_seed(1)
        x = torch.randn(B, kw["num_frames"], 4, kw["input_size"], kw["input_size"], generator=g)
        t = torch.tensor([999, 12][:B])
        y = torch.tensor([7, kw.get("num_classes", 0)][:B]) if kw["extras"] == 2 else None
        with torch.no_grad():
            ref = lo.latte_forward(sd, cfg, x, t, y)
        _ORACLE_FORWARD.clear()          # keep one case alive (XL/2 weights are 2.7 GB in fp32): the dtypes of a case run back to back
        _ORACLE_FORWARD[key] = (sd, x, t, y, ref)
    return _ORACLE_FORWARD[key]


@pytest.mark.parametrize("cd", DTYPES)
@pytest.mark.parametrize("case", ORACLE_CASES, ids=lambda c: f"{c[0]}-{c[1]['input_size']}x{c[1]['num_frames']}")
def test_forward_matches_oracle(case, cd):
    name, kw, B = case
    sd, x, t, y, ref = _oracle_forward(case)
    m = latte_amd.Latte_models[name](compute_dtype=cd, max_batch=B, **kw)
    m.load_state_dict(sd)
    m = m.cuda()
    out = m(x.cuda(), t.cuda(), y=None if y is None else y.cuda())
    assert rel_l2(out, ref) < TOL
    assert rel_l2(out[:, :, :4], ref[:, :, :4]) < TOL       # the epsilon channels on their own


# Weights that look like a TRAINED checkpoint (round 4).  The reference zero-initialises the adaLN modulation and the final
# layer (latte.py:284-295); the synthetic recipe re-draws them N(0, 0.02), which leaves every gate_msa / gate_mlp
# (latte.py:178-180) at ~0.03 and damps the block branches ~30x before they reach the output.  A trained checkpoint
# (sample/sample.py:62-64) has gates of O(0.1 - 1): here those tensors are drawn N(0, gate_std) with gate_std up to 1.0, so the
# half-precision rounding of every block linear reaches the model output at full weight.
GATE_CASES = [
    ("Latte-S/2", dict(input_size=16, num_frames=8, extras=1), 1),
    ("Latte-B/2", dict(input_size=16, num_frames=16, extras=1), 1),
    ("Latte-S/2", dict(input_size=32, num_frames=16, num_classes=101, extras=2), 2),    # fused qkv + attention kernel, hd = 64
    ("Latte-XL/2", dict(input_size=32, num_frames=16, num_classes=101, extras=2), 1),    # headline size
]
BF16_TOL_AT_TRAINED_GATES = 1e-2   # bf16 is NOT a 1e-3 type at these gates (measured ~3e-3, recorded); sanity bound only


def _record_gate(key, val):
    import json
    import os
    from _util import ROOT
    path = os.path.join(ROOT, "gpurun_out", "gate_parity.json")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    tab = {}
    if os.path.exists(path):
        with open(path) as f:
            tab = json.load(f)
    tab[key] = val
    with open(path, "w") as f:
        json.dump(tab, f, indent=1, sort_keys=True)


@pytest.mark.parametrize("gate_std", [0.1, 0.3, 1.0])
@pytest.mark.parametrize("case", GATE_CASES, ids=lambda c: f"{c[0]}-{c[1]['input_size']}x{c[1]['num_frames']}")
def test_forward_at_trained_scale_gates(case, gate_std):
    """f16 operands (the default type) hold north_star's 1e-3 on the model output AND on its epsilon channels at every gate
    magnitude; bf16 is measured beside it and written to gpurun_out/gate_parity.json (-> profiles/r4_gate_parity.json)."""
    from oracle import latte_oracle as lo
    name, kw, B = case
    cfg = lo.preset_config(name, **kw)
    sd = lo.init_state_dict(cfg, seed=0, gate_std=gate_std)
    g = torch.Generator("cpu").manual_seed(1)
    x = torch.randn(B, kw["num_frames"], 4, kw["input_size"], kw["input_size"], generator=g)
    t = torch.tensor([999, 12][:B])
    y = torch.tensor([7, kw.get("num_classes", 0)][:B]) if kw["extras"] == 2 else None
    with torch.no_grad():
        ref = lo.latte_forward(sd, cfg, x, t, y)
    err = {}
    for cd in ("f16", "bf16"):
        m = latte_amd.Latte_models[name](compute_dtype=cd, max_batch=B, **kw)
        m.load_state_dict(sd)
        m = m.cuda()
        out = m(x.cuda(), t.cuda(), y=None if y is None else y.cuda())
        assert torch.isfinite(out).all()
        err[cd] = (rel_l2(out, ref), rel_l2(out[:, :, :4], ref[:, :, :4]))
        del m
    _record_gate(f"forward::{name}::{kw['input_size']}x{kw['num_frames']}::gate_std={gate_std}",
                 {"f16": err["f16"][0], "f16_eps": err["f16"][1], "bf16": err["bf16"][0], "bf16_eps": err["bf16"][1]})
    print(name, gate_std, err)
    assert err["f16"][0] < TOL and err["f16"][1] < TOL, err
    assert err["bf16"][0] < BF16_TOL_AT_TRAINED_GATES, err


# forward_with_cfg at trained-scale gates, BASELINE config 3's own model and call (sample/sample_ddp.py:140-160: UCF101 class-conditional
# Latte-XL/2 through forward_with_cfg, cfg_scale 7.0).  Round 4 asserted one draw (gate_std 0.3, t = 500, one seed: 8.7e-4); round 5
# asserts the grid gate_std {0.3, 1.0} x t {999, 500, 50} x two (latent, label) seeds.  The weights of a gate_std and their engine are
# built once and shared by that gate_std's six cases (XL/2: 2.7 GB of fp32 weights on the host).
_GUIDED = {}
GUIDED_GRID = [(gs, t, seed) for gs in (0.3, 1.0) for t in (999, 500, 50) for seed in (1, 2)]


def _guided_model(gate_std):
    from oracle import latte_oracle as lo
    if _GUIDED.get("gate_std") != gate_std:
        _GUIDED.clear()
        name, kw = "Latte-XL/2", dict(input_size=32, num_frames=16, num_classes=101, extras=2)
        cfg = lo.preset_config(name, **kw)
        sd = lo.init_state_dict(cfg, seed=0, gate_std=gate_std)
        m = latte_amd.Latte_models[name](max_batch=2, **kw)       # default operand type
        m.load_state_dict(sd)
        _GUIDED.update(gate_std=gate_std, cfg=cfg, sd=sd, m=m.cuda())
    return _GUIDED["cfg"], _GUIDED["sd"], _GUIDED["m"]


@pytest.mark.parametrize("gate_std,t_val,seed", GUIDED_GRID, ids=[f"g{gs}-t{t}-s{sd_}" for gs, t, sd_ in GUIDED_GRID])
def test_guided_forward_at_trained_scale_gates(gate_std, t_val, seed):
    """forward_with_cfg (latte.py:379-398) at CFG 7.0 with trained-scale gates: the guidance combination amplifies the two
    halves' operand rounding; f16 (default type) stays under 1e-3 on the guided output AND on its epsilon channels, XL/2 at the
    headline latent size, at the start, the middle and the end of the chain's timestep range."""
    from oracle import latte_oracle as lo
    cfg, sd, m = _guided_model(gate_std)
    g = torch.Generator("cpu").manual_seed(seed)
    z = torch.randn(1, 16, 4, 32, 32, generator=g)
    x = torch.cat([z, z])
    t = torch.tensor([t_val, t_val])
    y = torch.tensor([int(torch.randint(0, 101, (1,), generator=g)), 101])
    with torch.no_grad():
        ref = lo.latte_forward_with_cfg(sd, cfg, x, t, y, 7.0)
    assert m.operand_dtype(guided=True) == "f16"
    assert m.get_engine_option("guided_split", 2, guided=True) == 20       # the default: attention output + fp8 remainder, fc1's operand + fp4 remainder
    out = m.forward_with_cfg(x.cuda(), t.cuda(), y=y.cuda(), cfg_scale=7.0)
    assert m.get_engine_option("guided_split_active", 2, guided=True) == 20
    e = (rel_l2(out, ref), rel_l2(out[:, :, :4], ref[:, :, :4]))
    rec = {"f16": e[0], "f16_eps": e[1]}
    # measured beside it, not asserted: the f16-pair form of round 5 (3), one operand only (4: attention output, 8: fc1's), plain f16 (0)
    for gs in (3, 4, 8, 16, 12, 1, 0):       # 12: both remainders as fp8 (the first form of round 6); 16: fc1's operand with the fp4 remainder only
        m.set_engine_option("guided_split", gs, 2, guided=True)
        o = m.forward_with_cfg(x.cuda(), t.cuda(), y=y.cuda(), cfg_scale=7.0)
        rec[f"guided_split_{gs}"] = rel_l2(o, ref)
    m.set_engine_option("guided_split", 20, 2, guided=True)
    _record_gate(f"guided_forward::Latte-XL/2::32x16::gate_std={gate_std}::t={t_val}::seed={seed}", rec)
    print(gate_std, t_val, seed, rec)
    assert e[0] < TOL and e[1] < TOL, e


def test_guided_forward_latte_l2_width():
    """The split-pair operands of guided calls on a width whose gated GEMMs do NOT run on the 192-wide 12-wave kernel (Latte-L/2:
    D = 1024 = 5.33 x 192, 16 heads of 64): the K-concatenated out-projection / fc1 go through whatever tile the shape rule picks."""
    from oracle import latte_oracle as lo
    kw = dict(input_size=32, num_frames=16, num_classes=101, extras=2)
    cfg = lo.preset_config("Latte-L/2", **kw)
    sd = lo.init_state_dict(cfg, seed=7, gate_std=0.3)
    g = torch.Generator("cpu").manual_seed(8)
    z = torch.randn(1, 16, 4, 32, 32, generator=g)
    x, t, y = torch.cat([z, z]), torch.tensor([500, 500]), torch.tensor([33, 101])
    with torch.no_grad():
        ref = lo.latte_forward_with_cfg(sd, cfg, x, t, y, 7.0)
    m = latte_amd.Latte_models["Latte-L/2"](max_batch=2, **kw)
    m.load_state_dict(sd)
    m = m.cuda()
    e = {}
    for gs in (3, 0):
        m.set_engine_option("guided_split", gs, 2, guided=True)
        out = m.forward_with_cfg(x.cuda(), t.cuda(), y=y.cuda(), cfg_scale=7.0)
        e[gs] = rel_l2(out[:, :, :4], ref[:, :, :4])
    print(e)
    assert e[3] < TOL and e[3] < e[0]


def test_guided_split_contract():
    """Engine option guided_split (round 5; round 6: bits 2 / 3 = the fp8-remainder form, bit 4 = fc1's operand with an fp4 remainder; default 20): (a) unguided calls do not
    depend on it; (b) guided_split = 0 is the plain f16 path: the guided output is the guidance combination of the SAME engine's
    unguided outputs of the doubled batch, bit for bit; (c) the split operands move the guided output by no more than an operand
    rounding (they remove one) and closer to the fp32 oracle, and the fp8 remainder does what the f16 remainder does; (d) the derived
    weight copies ([W | W], W8) follow a reloaded weight; (e) the engine reports what it really ran."""
    from oracle import latte_oracle as lo
    kw = dict(input_size=32, num_frames=16, num_classes=101, extras=2)      # 256 tokens x 16 frames: fused kernels, pair output
    cfg = lo.preset_config("Latte-S/2", **kw)
    sd = lo.init_state_dict(cfg, seed=5, gate_std=0.3)
    g = torch.Generator("cpu").manual_seed(6)
    z = torch.randn(1, 16, 4, 32, 32, generator=g)
    x, t, y = torch.cat([z, z]).cuda(), torch.tensor([300, 300]).cuda(), torch.tensor([9, 101]).cuda()
    with torch.no_grad():
        ref = lo.latte_forward_with_cfg(sd, cfg, x.cpu(), t.cpu(), y.cpu(), 7.0)
    m = latte_amd.Latte_models["Latte-S/2"](max_batch=2, **kw)
    m.load_state_dict(sd)
    m = m.cuda()
    plain = m(x, t, y=y)
    outs = {}
    assert m.get_engine_option("guided_split", 2, guided=True) == 20            # the default of an f16 engine
    for gs in (0, 1, 2, 3, 4, 8, 12, 15, 16, 20, 31):
        m.set_engine_option("guided_split", gs, 2, guided=True)
        outs[gs] = m.forward_with_cfg(x, t, y=y, cfg_scale=7.0)
        assert torch.equal(m(x, t, y=y), plain), gs                               # (a)
        assert m.get_engine_option("guided_split_active", 2, guided=True) == {15: 12, 31: 20}.get(gs, gs)   # (e): fp8 wins over the pair, fp4 over fp8
    assert m.get_engine_option("guided_split_failed", 2, guided=True) == 0
    assert torch.equal(outs[15], outs[12]) and torch.equal(outs[31], outs[20])
    cond, uncond = plain[:1, :, :4], plain[1:, :, :4]
    eps = uncond + 7.0 * (cond - uncond)
    want0 = torch.cat([torch.cat([eps, eps]), plain[:, :, 4:]], dim=2)
    assert rel_l2(outs[0], want0) < 1e-6                                            # (b) same forward, fp32 combine
    e = {gs: rel_l2(o[:, :, :4], ref[:, :, :4]) for gs, o in outs.items()}
    print(e)
    assert e[3] < e[0] and e[1] < e[0] and e[2] < e[0] and e[3] < TOL               # (c)
    assert e[12] < e[0] and e[4] < e[0] and e[8] < e[0] and e[12] < TOL
    assert e[20] < e[0] and e[16] < e[0] and e[20] < TOL and abs(e[20] - e[12]) < 0.1 * e[12] and abs(e[16] - e[8]) < 0.1 * e[8]   # the fp4 remainder does what the fp8 one does
    assert all(rel_l2(outs[gs], outs[0]) < 3e-3 for gs in (1, 2, 3, 4, 8, 12, 16, 20))
    # the fp8 remainder keeps 4 bits of a term that is 2^-12 of the product: against the oracle the two forms are the same (their
    # outputs still differ by a few 1e-4 from each other: any change upstream re-draws the roundings of every operand downstream)
    assert abs(e[12] - e[3]) < 0.05 * e[3] and abs(e[4] - e[1]) < 0.05 * e[1] and abs(e[8] - e[2]) < 0.05 * e[2]
    # (d) reload a block weight: both paths must follow it
    sd2 = dict(sd)
    sd2["blocks.3.mlp.fc1.weight"] = sd["blocks.3.mlp.fc1.weight"] * 1.5
    sd2["blocks.4.attn.proj.weight"] = sd["blocks.4.attn.proj.weight"] * 0.5
    m.load_state_dict(sd2)
    m.mark_weights_dirty() if hasattr(m, "mark_weights_dirty") else None
    with torch.no_grad():
        ref2 = lo.latte_forward_with_cfg(sd2, cfg, x.cpu(), t.cpu(), y.cpu(), 7.0)
    for gs in (3, 12, 20):
        m.set_engine_option("guided_split", gs, 2, guided=True)
        got2 = m.forward_with_cfg(x, t, y=y, cfg_scale=7.0)
        assert rel_l2(got2[:, :, :4], ref2[:, :, :4]) < TOL and rel_l2(got2, outs[gs]) > 1e-2


@pytest.mark.parametrize("cd", DTYPES)
def test_ddim10_plumbing_config_matches_oracle(cd):
    """BASELINE.json configs[0]: Latte-S/2, 4 frames, DDIM 10 steps, batch 1 — denoised latents."""
    from oracle import diffusion_oracle as do
    from oracle import latte_oracle as lo
    kw = dict(input_size=16, num_frames=4, extras=1)
    cfg = lo.preset_config("Latte-S/2", **kw)
    sd = lo.init_state_dict(cfg, seed=3)
    x = torch.randn(1, 4, 4, 16, 16, generator=torch.Generator("cpu").manual_seed(0))
    with torch.no_grad():
        want = do.sample_loop(do.Schedule("10"), lambda xx, tt: lo.latte_forward(sd, cfg, xx, tt), x, method="ddim")
    m = latte_amd.Latte_models["Latte-S/2"](compute_dtype=cd, **kw)
    m.load_state_dict(sd)
    m = m.cuda()
    d = latte_amd.create_diffusion("10")
    got = d.ddim_sample_loop(m.forward, x.shape, x.cuda(), clip_denoised=False, model_kwargs=dict(y=None))
    assert rel_l2(got, want) < TOL


def test_sampler_step_properties_at_full_size():
    """Size-independent properties of the update kernel at the XL/2 latent size (B=8):
    DDIM(eta=0) is deterministic and ignores noise; at index 0 both samplers return their mean;
    pred_xstart follows its closed form; the DDPM noise scale is monotone in the variance channel."""
    d = latte_amd.create_diffusion("250")
    g = torch.Generator("cpu").manual_seed(0)
    B, F, C, H = 8, 16, 4, 32
    x = torch.randn(B, F, C, H, H, generator=g).cuda()
    mo = torch.randn(B, F, 2 * C, H, H, generator=g).cuda()
    nz = torch.randn(B, F, C, H, H, generator=g).cuda()
    a = d._step("ddim", mo, x, 100, nz, 0.0, False)
    b = d._step("ddim", mo, x, 100, None, 0.0, False)
    assert torch.equal(a["sample"], b["sample"])
    p0 = d._step("ddpm", mo, x, 0, nz, 0.0, False)
    p0b = d._step("ddpm", mo, x, 0, torch.zeros_like(nz), 0.0, False)
    assert torch.equal(p0["sample"], p0b["sample"])
    i = 100
    x0 = float(np.float32(d.sqrt_recip_alphas_cumprod[i])) * x.double() - \
        float(np.float32(d.sqrt_recipm1_alphas_cumprod[i])) * mo[:, :, :C].double()
    assert rel_l2(a["pred_xstart"], x0) < 1e-6
    lo_v = d._step("ddpm", torch.cat([mo[:, :, :C], -torch.ones_like(mo[:, :, C:])], 2), x, 200, nz, 0.0, False)
    hi_v = d._step("ddpm", torch.cat([mo[:, :, :C], torch.ones_like(mo[:, :, C:])], 2), x, 200, nz, 0.0, False)
    mean = d._step("ddpm", mo, x, 200, torch.zeros_like(nz), 0.0, False)["sample"]
    assert ((hi_v["sample"] - mean).abs() >= (lo_v["sample"] - mean).abs() - 1e-6).all()


def test_engine_errors_are_loud():
    m = latte_amd.Latte_models["Latte-S/2"](input_size=8, num_frames=4, num_classes=5, extras=2).cuda()
    x = torch.zeros(1, 4, 4, 8, 8, device="cuda")
    t = torch.zeros(1, dtype=torch.int64, device="cuda")
    with pytest.raises(latte_amd.LatteError):
        m(x, t)                                   # class-conditional model without labels
    with pytest.raises(latte_amd.LatteError):
        m(torch.zeros(1, 4, 4, 16, 16, device="cuda"), t, y=t)   # wrong latent size
    with pytest.raises(latte_amd.LatteError):
        m.forward_with_cfg(x, t, y=t)             # odd batch
    sd = m.state_dict()
    sd.pop("blocks.3.mlp.fc2.bias")
    with pytest.raises(RuntimeError):
        m.load_state_dict(sd)                     # strict key check (sample.py:64)


def test_temb_table_and_chain_conditioning(lib):
    """latte_engine_temb_table == the per-step t_embedder path, and a loop run from an INSTALLED table (the RCCL
    broadcast payload) is bit-identical to a loop that computes its own."""
    import latte_amd
    from latte_amd._lib import check, ptr, stream_ptr
    kw, sd, r = load_golden_model("tiny_classcond")
    x, y = torch.from_numpy(r["x"]).cuda(), torch.from_numpy(r["y"]).cuda()
    m = engine_model(kw, sd, "bf16")
    d = latte_amd.create_diffusion("10")
    eng = m.engine(x.shape[0])
    table = torch.zeros(d.num_timesteps, m.hidden_size, device="cuda")
    check(lib.latte_engine_temb_table(eng, d._h, ptr(table), stream_ptr()))
    torch.cuda.synchronize()
    from oracle import latte_oracle as lo
    import torch.nn.functional as F
    tf = lo.timestep_embedding(torch.tensor(d.timestep_map), 256)                       # latte.py:97-117
    want = F.linear(F.silu(F.linear(tf, sd["t_embedder.mlp.0.weight"], sd["t_embedder.mlp.0.bias"])),
                    sd["t_embedder.mlp.2.weight"], sd["t_embedder.mlp.2.bias"])          # latte.py:90-94
    assert rel_l2(table, want) < 1e-5
    a = d.ddim_sample_loop(m.forward, x.shape, x.clone(), clip_denoised=False, model_kwargs=dict(y=y))
    check(lib.latte_engine_set_temb_table(eng, d._h, ptr(table), stream_ptr()))
    b = d.ddim_sample_loop(m.forward, x.shape, x.clone(), clip_denoised=False, model_kwargs=dict(y=y))
    check(lib.latte_engine_set_temb_table(eng, None, None, stream_ptr()))
    assert torch.equal(a, b)
    # and the fused loop (precomputed conditioning) equals stepping the model callable by hand
    c = d.ddim_sample_loop(lambda xx, tt, **k: m.forward(xx, tt, **k), x.shape, x.clone(), clip_denoised=False,
                           model_kwargs=dict(y=y), device="cuda")
    assert rel_l2(a, c) < 1e-6
    # a table installed for ANOTHER timestep_map with the same number of steps must not be used ("10" vs "ddim10"),
    # and re-loading the t_embedder weights uninstalls it
    d2 = latte_amd.create_diffusion("ddim10")
    assert d2.num_timesteps == d.num_timesteps and list(d2.timestep_map) != list(d.timestep_map)
    own = d2.ddim_sample_loop(m.forward, x.shape, x.clone(), clip_denoised=False, model_kwargs=dict(y=y))
    check(lib.latte_engine_set_temb_table(eng, d._h, ptr(table), stream_ptr()))
    other = d2.ddim_sample_loop(m.forward, x.shape, x.clone(), clip_denoised=False, model_kwargs=dict(y=y))
    assert torch.equal(own, other)
    sd2 = dict(sd)
    sd2["t_embedder.mlp.2.bias"] = sd["t_embedder.mlp.2.bias"] + 0.25
    m.load_state_dict(sd2)
    eng = m.engine(x.shape[0])
    fresh = d.ddim_sample_loop(m.forward, x.shape, x.clone(), clip_denoised=False, model_kwargs=dict(y=y))
    assert not torch.equal(fresh, a)                      # the stale table (old weights) would reproduce `a`
    m.load_state_dict(sd)


def test_chain_conditioning_is_chunked_consistently(lib):
    """latte_sample_loop produces the adaLN rows of the chain in chunks of <= 256 rows: a 300-step class-conditional
    chain (2 samples -> 3 chunks) equals the same chain stepped through the model callable (conditioning per step)."""
    import latte_amd
    kw, sd, r = load_golden_model("tiny_classcond")
    x, y = torch.from_numpy(r["x"]).cuda(), torch.from_numpy(r["y"]).cuda()
    m = engine_model(kw, sd, "f16")
    d = latte_amd.create_diffusion("300")
    a = d.ddim_sample_loop(m.forward, x.shape, x.clone(), clip_denoised=False, model_kwargs=dict(y=y))
    c = d.ddim_sample_loop(lambda xx, tt, **k: m.forward(xx, tt, **k), x.shape, x.clone(), clip_denoised=False,
                           model_kwargs=dict(y=y), device="cuda")
    assert torch.isfinite(a).all() and rel_l2(a, c) < 1e-5


def test_xl_guided_ddim_chain_matches_oracle():
    """Headline size, class-conditional with classifier-free guidance (BASELINE config 3's model): two DDIM steps of the
    '250' respacing from the oracle's loop vs the fused engine loop on the doubled batch, identical noise."""
    from oracle import diffusion_oracle as do
    from oracle import latte_oracle as lo
    kw = dict(input_size=32, num_frames=16, num_classes=101, extras=2)
    cfg = lo.preset_config("Latte-XL/2", **kw)
    sd = lo.init_state_dict(cfg, seed=3)
    g = torch.Generator("cpu").manual_seed(4)
    z = torch.randn(1, 16, 4, 32, 32, generator=g)
    x = torch.cat([z, z])
    y = torch.tensor([17, 101])
    scale = 4.0
    s = do.Schedule("250")
    want = x.clone()
    with torch.no_grad():
        for i in (249, 248):                                   # the loop body of gd:637-684 for two steps
            t = torch.full((2,), s.timestep_map[i], dtype=torch.int64)
            out = lo.latte_forward_with_cfg(sd, cfg, want, t, y, scale)
            want = do.ddim_sample(s, out, want, i, None, 0.0, False)["sample"]
    m = latte_amd.Latte_models["Latte-XL/2"](compute_dtype="bf16", max_batch=2, **kw)
    m.load_state_dict(sd)
    m = m.cuda()
    d = latte_amd.create_diffusion("250")
    xx = x.cuda().contiguous()
    check(load_library().latte_sample_loop(m.engine(2), d._h, 1, 0.0, 0, scale, ptr(xx), ptr(y.cuda()), 2, 249, 248, None, None,
                                           None, stream_ptr()))
    torch.cuda.synchronize()
    assert rel_l2(xx, want) < TOL


def test_results_do_not_depend_on_batch_composition():
    """Sharding claim: a sample's chain is the same whether it runs alone or inside a batch —
    different M means different GEMM tile shapes and workgroup schedules, the K order of every dot product does not."""
    kw, sd, r = load_golden_model("tiny_uncond")
    g = torch.Generator("cpu").manual_seed(11)
    xs = torch.randn(5, *r["x"].shape[1:], generator=g).cuda()
    d = latte_amd.create_diffusion("6")
    m5 = engine_model(kw, sd, "bf16", max_batch=5)
    all5 = d.ddim_sample_loop(m5.forward, xs.shape, xs.clone(), clip_denoised=False, model_kwargs=dict(y=None))
    m1 = engine_model(kw, sd, "bf16", max_batch=1)
    for b in (0, 3):
        one = d.ddim_sample_loop(m1.forward, xs[b:b + 1].shape, xs[b:b + 1].clone(), clip_denoised=False,
                                 model_kwargs=dict(y=None))
        assert torch.equal(one[0], all5[b])


def test_xl_chain_is_bitwise_reproducible():
    """Race screen at the headline size: the persistent GEMM's DMA / epilogue ordering rests on in-order vmcnt arguments
     and a violated hand-off shows up as rare, run-dependent garbage.  Two runs of the same 40-step
    DDPM chain (engine noise stream re-seeded) on B = 8 -- every GEMM walks 3 to 9 tiles per workgroup -- must be
    bit-identical and finite, and must differ from a run with another seed (the noise really is used)."""
    from oracle import latte_oracle as lo
    kw = dict(input_size=32, num_frames=16, extras=1)
    cfg = lo.preset_config("Latte-XL/2", **kw)
    sd = lo.init_state_dict(cfg, seed=5)
    m = latte_amd.Latte_models["Latte-XL/2"](compute_dtype="bf16", max_batch=8, **kw)
    m.load_state_dict(sd)
    m = m.cuda()
    d = latte_amd.create_diffusion("250")
    lib = load_library()
    eng = m.engine(8)
    x0 = torch.randn(8, 16, 4, 32, 32, generator=torch.Generator().manual_seed(3)).cuda()

    def run(seed):
        m.set_engine_option("seed", seed, 8)
        x = x0.clone()
        check(lib.latte_sample_loop(eng, d._h, 0, 0.0, 0, 1.0, ptr(x), None, 8, 249, 210, None, None, None, stream_ptr()))
        torch.cuda.synchronize()
        return x

    a, b, c = run(11), run(11), run(12)
    assert torch.isfinite(a).all()
    assert torch.equal(a, b)
    assert not torch.equal(a, c)


@pytest.mark.parametrize("B", [1, 2])
def test_small_batch_gated_gemms(B):
    """B = 1 at XL/2 (M = 4096): the gated GEMMs (out-projection, fc2; N = 1152) have 96 tiles of 256 x 192 for 256 CUs.  The rule
    (gemm.hip: gemm_small_tile_ok) gives them the 128 x 144 tile -- 256 tiles, one per CU -- which sums the contraction in the
    same order as the 256 x 192 kernels: bit-identical to the forward with the 12-wave kernel forced.  A forced split of the
    contraction (engine.cpp: gated_gemm, 2 partial products + a reduction into the residual stream) agrees up to the
    re-association, every path is bit-identical run to run, and at B = 2 (512 small tiles: not taken; 192 large ones: no split)
    the options must change nothing."""
    from oracle import latte_oracle as lo
    kw = dict(input_size=32, num_frames=16, extras=1)
    cfg = lo.preset_config("Latte-XL/2", **kw)
    sd = lo.init_state_dict(cfg, seed=7)
    m = latte_amd.Latte_models["Latte-XL/2"](compute_dtype="bf16", max_batch=B, **kw)
    m.load_state_dict(sd)
    m = m.cuda()
    g = torch.Generator().manual_seed(9)
    x = torch.randn(B, 16, 4, 32, 32, generator=g).cuda()
    t = torch.randint(0, 1000, (B,), generator=g).cuda()
    outs = {}
    for mode in (0, 1, 2, 0):
        m.set_engine_option("gated_split_k", mode, B)
        o = m.forward(x, t)
        torch.cuda.synchronize()
        assert torch.isfinite(o).all()
        if mode in outs:
            assert torch.equal(o, outs[mode]), "not deterministic"
        outs[mode] = o.clone()
    m.set_engine_option("gated_split_k", 0, B)
    assert torch.equal(outs[0], outs[1])            # the rule never splits: small tile at B = 1, enough large tiles at B = 2
    if B == 1:
        assert not torch.equal(outs[2], outs[1]), "the forced split did not split"
    assert rel_l2(outs[2], outs[1]) < 6e-4          # fp32 re-association, amplified by the half-precision operand rounding downstream
    for gname in ("proj", "fc2"):
        m.set_engine_option("gemm_variant_" + gname, 11, B)
    big = m.forward(x, t)
    torch.cuda.synchronize()
    for gname in ("proj", "fc2"):
        m.set_engine_option("gemm_variant_" + gname, 0, B)
    assert torch.equal(big, outs[0]), "128 x 144 tile and 256 x 192 tile disagree"


@pytest.mark.parametrize("cd", [None, "f16"])
def test_text_conditioned_variant_matches_reference_golden(cd):
    """extras == 78 (latte.py:238-242,340-363): text_embedding_projection inside the engine, blocks conditioned on
    t + text, final layer on t alone; forward, forward_with_cfg and both guided loops (progressive trajectories through
    the host shim, fused chain and generic-callable path) against the reference's own outputs.

    Operand types: ``None`` = the shim's rule (latte_amd.Latte docstring: bf16 operands for the plain forward, f16 for the
    guided callable -- this fixture's text conditioning is strong, |text projection| ~ |t_emb|, so the two halves' rounding
    errors are uncorrelated and eps_u + 7 (eps_c - eps_u) amplifies them by sqrt(36 + 49) ~ 9) and pinned f16.  Everything
    is held to north_star's 1e-3."""
    kw, sd, r = load_golden_model("tiny_textcond")
    m = engine_model(kw, sd, cd)
    x, t = torch.from_numpy(r["x"]).cuda(), torch.from_numpy(r["t"]).cuda()
    te = torch.from_numpy(r["text_embedding"]).cuda()
    assert rel_l2(m(x, t, text_embedding=te), torch.from_numpy(r["forward"])) < TOL
    gtol = TOL
    xc = torch.from_numpy(r["x_cfg"]).cuda()
    out = m.forward_with_cfg(xc, t, cfg_scale=7.0, text_embedding=te)
    assert rel_l2(out, torch.from_numpy(r["forward_with_cfg"])) < gtol
    with pytest.raises(latte_amd.LatteError, match="text_embedding"):
        m(x, t)
    steps = int(r["loop_steps"])
    d = latte_amd.create_diffusion(str(steps))
    mk = dict(text_embedding=te, cfg_scale=7.0)
    trail = list(d.ddim_sample_loop_progressive(m.forward_with_cfg, xc.shape, xc, clip_denoised=False, model_kwargs=mk,
                                                device="cuda"))
    for k in range(steps):
        assert rel_l2(trail[k]["sample"], torch.from_numpy(r["ddim_samples"][k])) < gtol, k
        assert rel_l2(trail[k]["pred_xstart"], torch.from_numpy(r["ddim_pred_xstart"][k])) < gtol, k
    fn = lambda xx, tt, **kw_: m.forward_with_cfg(xx, tt, **kw_)          # generic callable, step by step
    s2 = d.ddim_sample_loop(fn, xc.shape, xc, clip_denoised=False, model_kwargs=mk, device="cuda")
    assert rel_l2(s2, torch.from_numpy(r["ddim_samples"][-1])) < gtol
    # DDPM with the reference's noise draws
    xx = xc.clone().contiguous()
    nz = torch.from_numpy(r["ddpm_noises"]).cuda().contiguous()
    ts = torch.empty((steps,) + tuple(xx.shape), device="cuda")
    m._set_text(te, xx.shape[0], guided=True)
    check(load_library().latte_sample_loop(m.engine(xx.shape[0], guided=True), d._h, 0, 0.0, 0, 7.0, ptr(xx), None, xx.shape[0],
                                           steps - 1, 0, ptr(nz), ptr(ts), None, stream_ptr()))
    torch.cuda.synchronize()
    for k in range(steps):
        assert rel_l2(ts[k], torch.from_numpy(r["ddpm_samples"][k])) < gtol, k


@pytest.mark.parametrize("cd", DTYPES)
def test_sampling_hooks_match_reference_golden(cd):
    """denoised_fn / cond_fn of p_sample / ddim_sample (gd:316-321, :345-375; hook timesteps are the ORIGINAL ones,
    rs:100-104) through latte_sampler_step_ex, against loops run by the reference with the same hooks, clip_denoised=True
    and (DDIM) eta = 0.3, fed the reference's own noise draws."""
    from oracle import diffusion_oracle as do
    kw, sd, r = load_golden_model("tiny_uncond")
    m = engine_model(kw, sd, cd)
    steps = int(r["loop_steps"])
    d = latte_amd.create_diffusion(str(steps))
    z = torch.from_numpy(r["x"]).cuda()
    hooks = dict(denoised_fn=do.example_denoised_fn, cond_fn=do.example_cond_fn)

    def chain(method, eta, clip, noises, **hk):
        x = z.clone()
        for k, i in enumerate(range(steps - 1, -1, -1)):
            out = d._call_model(m.forward, x, i, {})
            x = d._step(method, out, x, i, noises[k], eta, clip, hk.get("denoised_fn"), hk.get("cond_fn"), {})["sample"]
        return x

    nz = torch.from_numpy(r["ddpm_noises"]).cuda()
    assert rel_l2(chain("ddpm", 0.0, True, nz, **hooks), torch.from_numpy(r["ddpm_hooks_final"])) < TOL
    nz = torch.from_numpy(r["ddim_noises"]).cuda()
    assert rel_l2(chain("ddim", 0.3, True, nz, **hooks), torch.from_numpy(r["ddim_hooks_final"])) < TOL
    only = chain("ddim", 0.0, False, nz, denoised_fn=do.example_denoised_fn)
    assert rel_l2(only, torch.from_numpy(r["ddim_denoised_only_final"])) < TOL
    # the public loops take the hooks too (own noise draws: finite, and different from the unhooked chain)
    a = d.ddim_sample_loop(m.forward, z.shape, z, clip_denoised=True, device="cuda", **hooks)
    b = d.ddim_sample_loop(m.forward, z.shape, z, clip_denoised=True, device="cuda")
    assert torch.isfinite(a).all() and rel_l2(a, b) > 1e-3
    c = d.p_sample_loop(m.forward, z.shape, z, clip_denoised=True, device="cuda", **hooks)
    assert torch.isfinite(c).all()


SAMPLER_TYPES = {"fixed_large": dict(learn_sigma=False), "fixed_small": dict(learn_sigma=False, sigma_small=True),
                 "xstart_learned": dict(predict_xstart=True),
                 "xstart_fixed_small": dict(predict_xstart=True, learn_sigma=False, sigma_small=True)}


@pytest.mark.parametrize("tag", sorted(SAMPLER_TYPES))
def test_other_model_types_match_reference_golden(tag):
    """create_diffusion(learn_sigma=False[, sigma_small=True]) / predict_xstart=True (init:32-45; gd:289-313,323-328)
    on the update kernel, against loops the reference ran on a synthetic model callable (same noise draws)."""
    import os
    from _util import GOLDEN
    from oracle import diffusion_oracle as do
    r = np.load(os.path.join(GOLDEN, "sampler_types.npz"))
    kw = SAMPLER_TYPES[tag]
    steps = int(r["steps"])
    d = latte_amd.create_diffusion(str(steps), **kw)
    oc = 8 if kw.get("learn_sigma", True) else 4
    z = torch.from_numpy(r["z"]).cuda()
    nz = torch.from_numpy(r["noises"]).cuda()
    fn = lambda xx, tt, **k: do.synthetic_model(xx, tt, oc)
    for method, eta, clip in (("ddpm", 0.0, True), ("ddim", 0.4, False)):
        x = z.clone()
        for k, i in enumerate(range(steps - 1, -1, -1)):
            x = d._step(method, d._call_model(fn, x, i, {}), x, i, nz[k], eta, clip)["sample"]
        assert rel_l2(x, torch.from_numpy(r[f"{tag}::{method}"])) < 1e-4, method      # fp32 update kernel, torch-GPU model fn
    out = d.p_sample_loop(fn, z.shape, z, clip_denoised=True, device="cuda")           # public loop, own noise
    assert torch.isfinite(out).all()
    with pytest.raises(AssertionError):                                               # wrong channel count (gd:290 / :338)
        d._step("ddpm", torch.zeros(2, 4, 12 - oc, 8, 8, device="cuda"), z, 0, None, 0.0, False)


def test_fused_loop_with_fixed_variance_model():
    """A learn_sigma=False Latte (C output channels) with create_diffusion(learn_sigma=False): whole chain inside the
    engine vs the oracle; mixing a learn_sigma=True model with it fails loudly."""
    from oracle import diffusion_oracle as do
    from oracle import latte_oracle as lo
    kw = dict(input_size=8, num_frames=4, extras=1, learn_sigma=False)
    cfg = lo.preset_config("Latte-S/2", **kw)
    sd = lo.init_state_dict(cfg, seed=9)
    m = latte_amd.Latte_models["Latte-S/2"](compute_dtype="f16", **kw)
    m.load_state_dict(sd)
    m = m.cuda()
    steps = 5
    d = latte_amd.create_diffusion(str(steps), learn_sigma=False)
    s = do.Schedule(str(steps), learn_sigma=False)
    g = torch.Generator().manual_seed(1)
    z = torch.randn(2, 4, 4, 8, 8, generator=g)
    noises = [torch.randn(2, 4, 4, 8, 8, generator=g) for _ in range(steps)]
    want = do.sample_loop(s, lambda xx, tt: lo.latte_forward(sd, cfg, xx, tt, None), z, method="ddpm", noises=noises)
    xx = z.clone().cuda().contiguous()
    nz = torch.stack(noises).cuda().contiguous()
    check(load_library().latte_sample_loop(m.engine(2), d._h, 0, 0.0, 0, 1.0, ptr(xx), None, 2, steps - 1, 0, ptr(nz), None,
                                           None, stream_ptr()))
    torch.cuda.synchronize()
    assert rel_l2(xx, want) < TOL
    d2 = latte_amd.create_diffusion(str(steps))
    with pytest.raises(latte_amd.LatteError, match="learn_sigma"):
        check(load_library().latte_sample_loop(m.engine(2), d2._h, 0, 0.0, 0, 1.0, ptr(xx), None, 2, steps - 1, 0, ptr(nz),
                                               None, None, stream_ptr()))


@pytest.mark.parametrize("name", ["Latte-S/2", "Latte-XL/2"])
def test_fused_qkv_attention_path_is_bit_identical_to_the_unfused(name):
    """16 frames of 256 tokens: every block's QKV projection + attention core runs as ONE kernel (csrc/qkv_attn.hip, engine option
    fuse_qkv_attn, default on).  The forward must be bit-identical with the option off (separate qkv GEMM + attention kernels), the
    launch classes must show which path ran, and partial fusion (spatial only / temporal only) must agree as well."""
    from oracle import latte_oracle as lo
    kw = dict(input_size=32, num_frames=16, extras=1)
    cfg = lo.preset_config(name, **kw)
    sd = lo.init_state_dict(cfg, seed=8)
    B = 2
    m = latte_amd.Latte_models[name](compute_dtype="bf16", max_batch=B, **kw)
    m.load_state_dict(sd)
    m = m.cuda()
    x = torch.randn(B, 16, 4, 32, 32, generator=torch.Generator("cpu").manual_seed(2)).cuda()
    t = torch.tensor([900, 41]).cuda()
    outs = {}
    for opt in (3, 0, 1, 2, 3 + 4, 3 + 8, 3 + 12):     # + 4 / + 8: schedule variants of the fused kernel (same bits)
        m.set_engine_option("fuse_qkv_attn", opt, B)
        outs[opt] = m(x, t).clone()
        prof = m.profile_forward(x, t)
        half = cfg.depth // 2
        assert prof["qkv_attn_spatial"][1] == (half if opt & 1 else 0) and prof["attn_spatial"][1] == (0 if opt & 1 else half)
        assert prof["qkv_attn_temporal"][1] == (half if opt & 2 else 0) and prof["attn_temporal"][1] == (0 if opt & 2 else half)
        assert prof["gemm_qkv"][1] == (0 if opt & 1 else half) + (0 if opt & 2 else half)
    assert torch.isfinite(outs[3]).all()
    for opt in (0, 1, 2, 7, 11, 15):
        assert torch.equal(outs[opt], outs[3]), opt
