"""Pin oracle/latte_t2v_oracle.py against the reference's own models/latte_t2v.py (build container only).

TEST INFRASTRUCTURE ONLY.  The reference file runs UNMODIFIED on oracle/diffusers_standin.py (diffusers 0.24.0 is absent),
so this pins the reference's own glue and temporal blocks; the diffusers leaves stay memory-derived (see the oracle's
header).  Usage: python -m oracle.validate_t2v_oracle  -> appends to oracle/VALIDATION.md, writes tests/golden/tiny_t2v.npz
"""
import json
import os

import numpy as np
import torch

from oracle import latte_t2v_oracle as to
from oracle.reference_loader import load_reference_latte_t2v

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def build(ref, cfg, seed):
    torch.manual_seed(seed)
    net = ref.LatteT2V(num_attention_heads=cfg.num_attention_heads, attention_head_dim=cfg.attention_head_dim,
                       in_channels=cfg.in_channels, out_channels=cfg.out_channels, num_layers=cfg.num_layers,
                       sample_size=cfg.sample_size, patch_size=cfg.patch_size, cross_attention_dim=cfg.cross_attention_dim,
                       attention_bias=True, activation_fn="gelu-approximate", norm_type="ada_norm_single",
                       norm_elementwise_affine=False, norm_eps=cfg.norm_eps, caption_channels=cfg.caption_channels,
                       video_length=cfg.video_length).eval()
    g = torch.Generator("cpu").manual_seed(seed + 1)
    with torch.no_grad():                      # give every bias signal (nn.Linear inits are fine, zero tensors are not)
        for _, p_ in net.named_parameters():
            if float(p_.abs().max()) == 0.0:
                p_.copy_(torch.randn(p_.shape, generator=g) * 0.02)
    return net


def main():
    ref = load_reference_latte_t2v()
    rows = []
    cases = [("tiny (D=128, 2+2 layers, 4 frames, 8x8 latent, 6 text tokens)",
              to.T2VConfig(num_attention_heads=2, attention_head_dim=64, num_layers=2, sample_size=8, cross_attention_dim=128,
                           caption_channels=64, video_length=4), 2, 6),
             ("hd=72 geometry (D=144, 3+3 layers, 16 frames, 16x16 latent, 20 text tokens)",
              to.T2VConfig(num_attention_heads=2, attention_head_dim=72, num_layers=3, sample_size=16, cross_attention_dim=144,
                           caption_channels=48, video_length=16), 1, 20)]
    golden = None
    for title, cfg, B, Lk in cases:
        net = build(ref, cfg, seed=11)
        sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
        assert set(sd) == set(to.state_dict_keys(cfg)), set(sd) ^ set(to.state_dict_keys(cfg))
        g = torch.Generator("cpu").manual_seed(3)
        x = torch.randn(B, cfg.in_channels, cfg.video_length, cfg.sample_size, cfg.sample_size, generator=g)
        t = torch.tensor([999, 37][:B])
        enc = torch.randn(B, Lk, cfg.caption_channels, generator=g)
        mask = torch.ones(B, Lk)
        mask[-1, Lk - 2:] = 0
        with torch.no_grad():
            want = net(x, timestep=t, encoder_hidden_states=enc, encoder_attention_mask=mask,
                       added_cond_kwargs={"resolution": None, "aspect_ratio": None}, enable_temporal_attentions=True,
                       return_dict=False)[0]
            got = to.latte_t2v_forward(sd, cfg, x, t, enc, mask)
            want_s = net(x, timestep=t, encoder_hidden_states=enc, encoder_attention_mask=None,
                         added_cond_kwargs={"resolution": None, "aspect_ratio": None}, enable_temporal_attentions=False,
                         return_dict=False)[0]
            got_s = to.latte_t2v_forward(sd, cfg, x, t, enc, None, enable_temporal_attentions=False)
        rel = float((got - want).norm() / want.norm())
        rel_s = float((got_s - want_s).norm() / want_s.norm())
        rows.append((f"LatteT2V {title}: forward (text mask, temporal blocks on)", f"rel-L2 {rel:.2e}, max|d| {float((got - want).abs().max()):.2e}"))
        rows.append((f"LatteT2V {title}: forward (no mask, temporal blocks off)", f"rel-L2 {rel_s:.2e}"))
        assert rel < 2e-6 and rel_s < 2e-6, (rel, rel_s)
        if golden is None:
            golden = {"cfg_json": np.frombuffer(json.dumps(cfg.__dict__).encode(), dtype=np.uint8), "x": x.numpy(), "t": t.numpy(),
                      "encoder_hidden_states": enc.numpy(), "encoder_attention_mask": mask.numpy(), "forward": want.numpy(),
                      "forward_spatial_only": want_s.numpy()}
            for k, v in sd.items():
                golden["sd::" + k] = v.numpy()
    np.savez_compressed(os.path.join(OUT, "tiny_t2v.npz"), **golden)
    with open("oracle/VALIDATION.md", "a") as f:
        f.write("\n## LatteT2V (SURVEY §8(f) rank 2 groundwork)\n\n`python -m oracle.validate_t2v_oracle`: the unmodified "
                "`/root/reference/models/latte_t2v.py` on the diffusers stand-in (`oracle/diffusers_standin.py`, memory-derived) "
                "vs `oracle/latte_t2v_oracle.py`.  SDPA in the stand-in vs explicit softmax in the oracle: fp32 round-off only.\n\n"
                "| check | result |\n|---|---|\n")
        for a, b in rows:
            f.write(f"| {a} | {b} |\n")
    for a, b in rows:
        print(a, "->", b)


if __name__ == "__main__":
    main()
