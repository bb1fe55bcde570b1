#!/usr/bin/env python
"""Interleaved in-model A/B of a per-launch environment switch of the GEMM kernels (read with getenv at every launch):
  python tools/pw_env_probe.py ENV_NAME v0,v1,...      XL/2 forward at B = LATTE_FL_B (default 8), bf16, best of 4 rounds."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from latte_amd.models import Latte_models  # noqa: E402

name, values = sys.argv[1], sys.argv[2].split(",")
dev = torch.device("cuda")
B = int(os.environ.get("LATTE_FL_B", "8"))
m = Latte_models["Latte-XL/2"](compute_dtype="bf16", max_batch=B, input_size=32, num_frames=16, extras=1)
with torch.no_grad():
    for n_, p_ in m.named_parameters():
        if float(p_.abs().max()) == 0.0:
            p_.normal_(0, 0.02)
m = m.to(dev)
x = torch.randn(B, 16, 4, 32, 32, device=dev)
t = torch.full((B,), 500, device=dev, dtype=torch.int64)
best = {v: {} for v in values}
outs = {}
for rnd in range(4):
    for v in values:
        os.environ[name] = v
        prof = m.profile_forward(x, t)
        for k in ("gemm_proj", "gemm_fc2", "gemm_fc1", "ln_modulate"):
            best[v][k] = min(best[v].get(k, 1e9), prof[k][0] / prof[k][1] * 1e3)
        best[v]["total_ms"] = min(best[v].get("total_ms", 1e9), sum(q[0] for q in prof.values()))
        outs[v] = m(x, t).clone()
for v in values:
    b = best[v]
    same = bool(torch.equal(outs[v], outs[values[0]]))
    print(f"{name}={v}: proj {b['gemm_proj']:6.1f} us  fc2 {b['gemm_fc2']:6.1f} us  fc1 {b['gemm_fc1']:6.1f} us  ln {b['ln_modulate']:5.1f} us  "
          f"forward {b['total_ms']:6.2f} ms  output equal to the first setting: {same}")
