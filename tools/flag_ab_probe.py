"""A/B of a compiler-flag build of the library (latte_amd/build.py: LATTE_BUILD_TAG / LATTE_EXTRA_HIPFLAGS) on the XL/2 forward at B = 8:
average forward time, per-class table, and a digest of the output (the two builds must give the same bits).  Run once per library:
  LATTE_AMD_LIB=latte_amd/lib/liblatte_amd_<tag>.so python tools/flag_ab_probe.py"""
import hashlib
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import latte_amd  # noqa: E402

B = 8
torch.manual_seed(0)      # the constructor's default initialisers draw from the global generator
m = latte_amd.Latte_models["Latte-XL/2"](input_size=32, num_frames=16, extras=1, max_batch=B)
gc = torch.Generator("cpu").manual_seed(1)
with torch.no_grad():
    for _, p in m.named_parameters():
        if p.requires_grad and float(p.detach().abs().max()) == 0.0:
            p.copy_(torch.randn(p.shape, generator=gc) * 0.02)
m = m.to("cuda").eval()
x = torch.randn(B, 16, 4, 32, 32, generator=torch.Generator("cpu").manual_seed(2)).cuda()
t = torch.full((B,), 500, device="cuda", dtype=torch.int64)
for _ in range(3):
    out = m(x, t)
torch.cuda.synchronize()
rounds = []
for _ in range(4):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(15):
        out = m(x, t)
    e1.record()
    torch.cuda.synchronize()
    rounds.append(round(e0.elapsed_time(e1) / 15, 3))
m.profile_forward(x, t)
prof = m.profile_forward(x, t)
digest = hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest()[:16]
print(json.dumps({"lib": os.environ.get("LATTE_AMD_LIB", "default"), "forward_ms_rounds": rounds, "output_sha256_16": digest,
                  "per_class_ms": {k: round(v[0], 3) for k, v in prof.items() if v[1]}}))
