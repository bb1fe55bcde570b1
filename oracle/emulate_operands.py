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

``ln_fused=True`` models the round-4 LayerNorm fusion: the gated GEMM's epilogue emits the half operand
``x·(1+scale)`` and per-row Σx / Σx², the consuming GEMM applies ``r·(acc − μ·u[n]) + v[n]`` with ``u = (1+scale)·Wᵀ``,
``v = shift·Wᵀ + b`` (fp32 on the rounded weights).
"""
import torch
import torch.nn.functional as F

from . import latte_oracle as lo

_DT = {"bf16": torch.bfloat16, "f16": torch.float16, "fp32": None}


def _rnd(x, dt):
    return x if dt is None else x.to(dt).float()


def _attention_core(q, k, v, hd, dt):
    attn = (q @ k.transpose(-2, -1)) * (hd ** -0.5)
    attn = attn.softmax(dim=-1)
    # the engine feeds exp2(s - max) (un-normalised, <= 1) to the PV MFMA as half and divides by the fp32 row sum after
    m = attn.max(dim=-1, keepdim=True).values
    p = _rnd(attn / m, dt)
    return (p @ v) * m


def _block(sd, i, x, c_rows, num_heads, dt, ln_fused, prescale):
    pre = f"blocks.{i}."
    S, L, D = x.shape
    hd = D // num_heads
    mod = F.linear(F.silu(c_rows), sd[pre + "adaLN_modulation.1.weight"], sd[pre + "adaLN_modulation.1.bias"])
    sh1, sc1, g1, sh2, sc2, g2 = (m.unsqueeze(1) for m in mod.chunk(6, dim=1))

    def modulated_linear(x, sh, sc, w, b):
        """LN(x)·(1+sc)+sh -> half -> · W^T + b, either as the engine's separate LN pass or as the fused algebra."""
        wq = _rnd(w, dt)
        if not ln_fused:
            a = _rnd(F.layer_norm(x, (D,), eps=1e-6) * (1 + sc) + sh, dt)
            return a @ wq.t() + b
        mu = x.mean(dim=-1, keepdim=True)
        var = (x * x).mean(dim=-1, keepdim=True) - mu * mu          # Σx² / D − μ², as the epilogue partials give it
        r = torch.rsqrt(var.clamp_min(0) + 1e-6)
        a = _rnd(x * (1 + sc) * prescale, dt)
        acc = (a @ wq.t()) / prescale
        u = (1 + sc) @ wq.t()                                       # [S,1,N] per sample
        v = sh @ wq.t() + b
        return r * (acc - mu * u) + v

    qkv = modulated_linear(x, sh1, sc1, sd[pre + "attn.qkv.weight"], sd[pre + "attn.qkv.bias"])
    qkv = _rnd(qkv, dt).reshape(S, L, 3, num_heads, hd).permute(2, 0, 3, 1, 4)
    o = _attention_core(qkv[0], qkv[1], qkv[2], hd, dt).transpose(1, 2).reshape(S, L, D)
    o = _rnd(o, dt) @ _rnd(sd[pre + "attn.proj.weight"], dt).t() + sd[pre + "attn.proj.bias"]
    x = x + g1 * o
    h = modulated_linear(x, sh2, sc2, sd[pre + "mlp.fc1.weight"], sd[pre + "mlp.fc1.bias"])
    h = _rnd(F.gelu(h, approximate="tanh"), dt)
    h = h @ _rnd(sd[pre + "mlp.fc2.weight"], dt).t() + sd[pre + "mlp.fc2.bias"]
    return x + g2 * h


def latte_forward_emulated(sd, cfg, x, t, y=None, operand="bf16", ln_fused=False, prescale=1.0):
    """``latte_oracle.latte_forward`` with the engine's half-precision roundings applied (class-cond / uncond only)."""
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
        h = _block(sd, i, h, c_spatial, cfg.num_heads, dt, ln_fused, prescale)
        h = h.reshape(B, Fr, T, D).permute(0, 2, 1, 3).reshape(B * T, Fr, D)
        if i == 0:
            h = h + sd["temp_embed"]
        h = _block(sd, i + 1, h, c_temp, cfg.num_heads, dt, ln_fused, prescale)
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


def budget_table(cases, gate_stds=(0.02, 0.1, 0.3, 1.0), operands=("bf16", "f16"), ln_fused=(False, True), seed=0):
    """rel-L2 of the emulated forward against the fp32 oracle per (case, gate_std, operand type, LN fusion)."""
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
                    for lf in ln_fused:
                        out = latte_forward_emulated(sd, cfg, x, t, y, operand=op, ln_fused=lf)
                        e = float((out - ref).double().norm() / ref.double().norm())
                        rows.append(dict(model=name, latent=kw["input_size"], frames=kw["num_frames"], gate_std=gs, operand=op,
                                         ln_fused=lf, rel_l2=e))
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
