"""Per-tensor gradient errors of the engine's training step against the reference-generated fixture (GPU box; measurement /
debugging aid for tests/test_training_step.py)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from tests.test_training_step import _trainer, fixture, rel

for dtype in (sys.argv[1:] or ["f16", "bf16"]):
    z, grads_ref, terms_ref = fixture()
    tr, model, (cfg, sd, x0, noise, t, y, drop) = _trainer(dtype)
    out = tr.forward_backward(x0, t, noise, y, drop, return_model_out=True)
    torch.cuda.synchronize()
    print(dtype, {k: (out[k].cpu().tolist(), terms_ref[k].tolist()) for k in ("loss", "mse", "vb")})
    got = {k: v.cpu() for k, v in tr.grad_dict().items()}
    for k in grads_ref:
        print(f"  {k:45s} rel {rel(got[k], grads_ref[k]):.3e}  |ref| {float(grads_ref[k].norm()):.3e} |got| {float(got[k].norm()):.3e}")
