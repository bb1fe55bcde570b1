"""Where the LayerNorm fusion's time goes (measurement build: LATTE_AMD_LIB=latte_amd/lib/liblatte_amd_dbg.so).

  LATTE_AMD_LIB=latte_amd/lib/liblatte_amd_dbg.so python tools/ln_fusion_probe.py [B]

XL/2 forward at B (default 8), f16, per-class kernel time by HIP events (latte_profile_forward, best of 6 interleaved rounds) for:
the separate LayerNorm kernel (fuse_ln = 0), the fusion as shipped, and its ablations (LnFuse::dbg bits, results garbage):
1 = no row-sum slot stores, 2 = no operand stores, 16 = operand stores as direct 8-byte stores instead of through the LDS patch.
(profiles/r4_ln_fusion_ablation_v1_atomics.log is this probe on the first version, whose row sums were 64-bit atomics.)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import latte_amd

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
torch.manual_seed(0)
m = latte_amd.Latte_models["Latte-XL/2"](input_size=32, num_frames=16, extras=1, compute_dtype="f16", max_batch=B)
g = torch.Generator("cpu").manual_seed(1)
with torch.no_grad():
    for _, p in m.named_parameters():
        if p.requires_grad and float(p.detach().abs().max()) == 0.0:
            p.copy_(torch.randn(p.shape, generator=g) * 0.02)
m = m.to("cuda").eval()
x = torch.randn(B, 16, 4, 32, 32, generator=g).cuda()
t = torch.full((B,), 500, device="cuda", dtype=torch.int64)
SETTINGS = [("separate LN kernel", 0, 0), ("fused", 1, 0), ("fused, no slot stores", 1, 1), ("fused, no operand stores", 1, 2),
            ("fused, neither", 1, 3), ("fused, direct 8-byte operand stores", 1, 16)]
has_dbg = True
best = {}
for rnd in range(6):
    for name, fuse, dbg in SETTINGS:
        m.set_engine_option("fuse_ln", fuse, B)
        if dbg or has_dbg:
            try:
                m.set_engine_option("ln_dbg", dbg, B)
            except Exception:
                has_dbg = False
                if dbg:
                    continue
        m.profile_forward(x, t)
        prof = m.profile_forward(x, t)
        tot = sum(v[0] for v in prof.values())
        if name not in best or tot < best[name][0]:
            best[name] = (tot, {k: round(v[0], 3) for k, v in prof.items() if v[0] > 0.0})
for name, _, _ in SETTINGS:
    if name in best:
        tot, d = best[name]
        print(f"{name:40s} forward {tot:7.3f} ms  " + "  ".join(f"{k}={v}" for k, v in d.items()), flush=True)
