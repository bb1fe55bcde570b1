"""CPU emulation of the ENGINE's operand rounding inside the oracle forward (fp32 torch ops, nothing from the product).

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).  It is not the oracle — ``latte_oracle.latte_forward`` is — but a
model of where the HIP engine rounds to half precision, used to PREDICT, on CPU, what a numerics design costs before a GPU
minute is spent on it (tests/test_operand_budget.py), and to tell a kernel bug from a rounding budget when a GPU parity
case moves.  Where the engine rounds:

* the MFMA operands of the four block linears (latte.py:43-45,171): activation and weight rounded to bf16 / f16, fp32
  accumulate, fp32 bias / gate / residual;
* q, k, v rounded to half (they live in LDS as half), the softmax probabilities rounded to half before P·V, the attention
  output and the GELU output rounded to half (they are the next GEMM's A operand);
* everything else (patch embed, conditioning, adaLN linears, LayerNorm statistics, final layer) is fp32.

(Round 4 also modelled a LayerNorm fusion here -- the engine path it predicted was measured without gain and removed in round 5.)
"""
import torch
import torch.nn.functional as F

from . import latte_oracle as lo

_DT = {"bf16": torch.bfloat16, "f16": torch.float16, "fp32": None}


def _rnd(x, dt):
    return x if dt is None else x.to(dt).float()


# Rounding points of a block (round 5: per-point control, to attribute a parity budget on CPU).  ``exact`` = a set of point names that
# are NOT rounded (equivalently: carried as a split hi + lo operand pair, whose residual error is 2^-22):
#   a_qkv  LN-modulate output feeding the qkv linear        w_qkv  its weight
#   qkv    q / k / v as they sit in LDS                     p      the softmax probabilities fed to P V
#   a_proj attention output feeding the out-projection      w_proj
#   a_fc1  LN-modulate output feeding fc1                   w_fc1
#   a_fc2  GELU output feeding fc2                          w_fc2
POINTS = ("a_qkv", "w_qkv", "qkv", "p", "a_proj", "w_proj", "a_fc1", "w_fc1", "a_fc2", "w_fc2")
_EXACT = frozenset()          # module state of one emulated forward (set by latte_forward_emulated)
_PAIRED = frozenset()         # points whose second batch half is carried as (rounded first half) + rounded DIFFERENCE (guided pairs)
_EXACT_BLOCKS = None          # None = every block; else the set of block indices in which `_EXACT` applies
_CUR_BLOCK = -1


def _rp(x, dt, point):
    """round at a named point unless that point is exempt in the current block"""
    if point in _EXACT and (_EXACT_BLOCKS is None or _CUR_BLOCK in _EXACT_BLOCKS):
        return x
    if point in _PAIRED and dt is not None:
        # rows [0, n/2) and [n/2, n) are the two halves of a guidance pair (same latent, different conditioning): the second half's
        # operand is (first half's rounded value) + (rounded difference) -- its rounding error is the first half's, to ~2^-11 |difference|
        n = x.shape[0] // 2
        a = _rnd(x[:n], dt)
        return torch.cat([a, a + _rnd(x[n:] - a, dt)], dim=0)
    return _rnd(x, dt)


def _attention_core(q, k, v, hd, dt):
    attn = (q @ k.transpose(-2, -1)) * (hd ** -0.5)
    attn = attn.softmax(dim=-1)
    # the engine feeds exp2(s - max) (un-normalised, <= 1) to the PV MFMA as half and divides by the fp32 row sum after
    m = attn.max(dim=-1, keepdim=True).values
    p = _rp(attn / m, dt, "p")
    return (p @ v) * m


def _block(sd, i, x, c_rows, num_heads, dt):
    global _CUR_BLOCK
    _CUR_BLOCK = i
    pre = f"blocks.{i}."
    S, L, D = x.shape
    hd = D // num_heads
    mod = F.linear(F.silu(c_rows), sd[pre + "adaLN_modulation.1.weight"], sd[pre + "adaLN_modulation.1.bias"])
    sh1, sc1, g1, sh2, sc2, g2 = (m.unsqueeze(1) for m in mod.chunk(6, dim=1))

    def modulated_linear(x, sh, sc, w, b, name):
        """LN(x)·(1+sc)+sh -> half -> · W^T + b (the engine's separate LayerNorm-modulate pass in front of the linear)."""
        wq = _rp(w, dt, "w_" + name)
        a = _rp(F.layer_norm(x, (D,), eps=1e-6) * (1 + sc) + sh, dt, "a_" + name)
        return a @ wq.t() + b

    qkv = modulated_linear(x, sh1, sc1, sd[pre + "attn.qkv.weight"], sd[pre + "attn.qkv.bias"], "qkv")
    qkv = _rp(qkv, dt, "qkv").reshape(S, L, 3, num_heads, hd).permute(2, 0, 3, 1, 4)
    o = _attention_core(qkv[0], qkv[1], qkv[2], hd, dt).transpose(1, 2).reshape(S, L, D)
    o = _rp(o, dt, "a_proj") @ _rp(sd[pre + "attn.proj.weight"], dt, "w_proj").t() + sd[pre + "attn.proj.bias"]
    x = x + g1 * o
    h = modulated_linear(x, sh2, sc2, sd[pre + "mlp.fc1.weight"], sd[pre + "mlp.fc1.bias"], "fc1")
    h = _rp(F.gelu(h, approximate="tanh"), dt, "a_fc2")
    h = h @ _rp(sd[pre + "mlp.fc2.weight"], dt, "w_fc2").t() + sd[pre + "mlp.fc2.bias"]
    return x + g2 * h


def latte_forward_emulated(sd, cfg, x, t, y=None, operand="bf16", exact=(), exact_blocks=None, paired=()):
    """``latte_oracle.latte_forward`` with the engine's half-precision roundings applied (class-cond / uncond only).
    ``exact``: names of ``POINTS`` that are not rounded (in ``exact_blocks`` only, when given) -- what a split hi + lo operand buys."""
    global _EXACT, _EXACT_BLOCKS, _PAIRED
    assert all(e in POINTS for e in tuple(exact) + tuple(paired)), (exact, paired)
    _PAIRED = frozenset(paired)
    _EXACT, _EXACT_BLOCKS = frozenset(exact), (None if exact_blocks is None else frozenset(exact_blocks))
    dt = _DT[operand]
    B, Fr, C, H, W = x.shape
    p, D = cfg.patch_size, cfg.hidden_size
    T = (H // p) * (W // p)
    xf = x.reshape(B * Fr, C, H, W).float()
    tok = F.conv2d(xf, sd["x_embedder.proj.weight"], sd["x_embedder.proj.bias"], stride=p)
    tok = tok.flatten(2).transpose(1, 2) + sd["pos_embed"]
    temb = lo.timestep_embedding(t, 256)
    temb = F.linear(temb, sd["t_embedder.mlp.0.weight"], sd["t_embedder.mlp.0.bias"])
    temb = F.linear(F.silu(temb), sd["t_embedder.mlp.2.weight"], sd["t_embedder.mlp.2.bias"])
    c = temb
    if cfg.extras == 2:
        c = temb + sd["y_embedder.embedding_table.weight"][y]
    c_spatial = c.repeat_interleave(Fr, dim=0)
    c_temp = c.repeat_interleave(T, dim=0)
    h = tok
    for i in range(0, cfg.depth, 2):
        h = _block(sd, i, h, c_spatial, cfg.num_heads, dt)
        h = h.reshape(B, Fr, T, D).permute(0, 2, 1, 3).reshape(B * T, Fr, D)
        if i == 0:
            h = h + sd["temp_embed"]
        h = _block(sd, i + 1, h, c_temp, cfg.num_heads, dt)
        h = h.reshape(B, T, Fr, D).permute(0, 2, 1, 3).reshape(B * Fr, T, D)
    mod = F.linear(F.silu(c.repeat_interleave(Fr, dim=0)), sd["final_layer.adaLN_modulation.1.weight"],
                   sd["final_layer.adaLN_modulation.1.bias"])
    shift, scale = mod.chunk(2, dim=1)
    h = F.layer_norm(h, (D,), eps=1e-6) * (1 + scale.unsqueeze(1)) + shift.unsqueeze(1)
    h = F.linear(h, sd["final_layer.linear.weight"], sd["final_layer.linear.bias"])
    co = cfg.out_channels
    gh = H // p
    h = h.reshape(B * Fr, gh, gh, p, p, co).permute(0, 5, 1, 3, 2, 4).reshape(B * Fr, co, gh * p, gh * p)
    return h.reshape(B, Fr, co, H, W)


def latte_forward_with_cfg_emulated(sd, cfg, x, t, y, cfg_scale, **emu):
    """``latte_oracle.latte_forward_with_cfg`` (latte.py:379-398) on the emulated forward."""
    half = x[: len(x) // 2]
    out = latte_forward_emulated(sd, cfg, torch.cat([half, half], dim=0), t, y, **emu)
    eps, rest = out[:, :, :4], out[:, :, 4:]
    cond, uncond = torch.split(eps, len(eps) // 2, dim=0)
    half_eps = uncond + cfg_scale * (cond - uncond)
    return torch.cat([torch.cat([half_eps, half_eps], dim=0), rest], dim=2)


def budget_table(cases, gate_stds=(0.02, 0.1, 0.3, 1.0), operands=("bf16", "f16"), seed=0):
    """rel-L2 of the emulated forward against the fp32 oracle per (case, gate_std, operand type)."""
    rows = []
    for name, kw, B in cases:
        cfg = lo.preset_config(name, **kw)
        g = torch.Generator("cpu").manual_seed(1)
        x = torch.randn(B, kw["num_frames"], 4, kw["input_size"], kw["input_size"], generator=g)
        t = torch.tensor([999, 12][:B])
        y = torch.tensor([7, kw.get("num_classes", 0)][:B]) if kw.get("extras", 1) == 2 else None
        for gs in gate_stds:
            sd = lo.init_state_dict(cfg, seed=seed, gate_std=gs)
            with torch.no_grad():
                ref = lo.latte_forward(sd, cfg, x, t, y)
                for op in operands:
                    out = latte_forward_emulated(sd, cfg, x, t, y, operand=op)
                    e = float((out - ref).double().norm() / ref.double().norm())
                    rows.append(dict(model=name, latent=kw["input_size"], frames=kw["num_frames"], gate_std=gs, operand=op, rel_l2=e))
                    print(rows[-1], flush=True)
    return rows


if __name__ == "__main__":
    import json
    import sys
    cases = [("Latte-S/2", dict(input_size=16, num_frames=8, extras=1), 1),
             ("Latte-B/2", dict(input_size=16, num_frames=16, extras=1), 1)]
    if "--xl" in sys.argv:
        cases.append(("Latte-XL/2", dict(input_size=32, num_frames=16, num_classes=101, extras=2), 1))
    rows = budget_table(cases)
    if "--json" in sys.argv:
        print(json.dumps(rows))
