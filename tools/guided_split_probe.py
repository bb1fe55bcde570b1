"""Cost of the split-operand linears of guided calls (engine option guided_split): BASELINE config 3's per-GPU
share -- Latte-XL/2 class-conditional, CFG 7.0, 8 samples = 16 sequences -- one forward_with_cfg per setting, interleaved in one
process, HIP events.  Usage: python tools/guided_split_probe.py [rows=16]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import latte_amd  # noqa: E402

rows = int(dict(a.split("=") for a in sys.argv[1:] if "=" in a).get("rows", 16))
kw = dict(input_size=32, num_frames=16, num_classes=101, extras=2)
m = latte_amd.Latte_models["Latte-XL/2"](max_batch=rows, **kw)
with torch.no_grad():
    g = torch.Generator().manual_seed(0)
    for _, p in m.named_parameters():
        if float(p.abs().max()) == 0.0:
            p.copy_(torch.randn(p.shape, generator=g) * 0.3)
m = m.cuda()
z = torch.randn(rows // 2, 16, 4, 32, 32, device="cuda")
x = torch.cat([z, z])
t = torch.full((rows,), 500, device="cuda", dtype=torch.int64)
y = torch.cat([torch.randint(0, 101, (rows // 2,), device="cuda"), torch.full((rows // 2,), 101, device="cuda")])
res, outs = {}, {}
for rep in range(4):
    for gs in (0, 1, 2, 3, 4, 8, 12, 16, 20):
        m.set_engine_option("guided_split", gs, rows, guided=True)
        o = m.forward_with_cfg(x, t, y=y, cfg_scale=7.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            o = m.forward_with_cfg(x, t, y=y, cfg_scale=7.0)
        e1.record()
        torch.cuda.synchronize()
        res.setdefault(gs, []).append(e0.elapsed_time(e1) / 3)
        outs[gs] = o
base = min(res[0])
for gs, v in res.items():
    d = float((outs[gs] - outs[0]).norm() / outs[0].norm())
    print(f"guided_split={gs}: {rows} sequences, forward_with_cfg min {min(v):.2f} ms  med {sorted(v)[len(v) // 2]:.2f} ms  ({min(v) / base - 1:+.1%} vs 0)"
          f"   |out - out(split 0)| / |out(0)| = {d:.2e}", flush=True)
