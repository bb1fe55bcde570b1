"""Shared helpers for the test-suite (golden fixture loading, error metrics, engine construction)."""
import json
import os

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def rel_l2(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def load_golden_model(name):
    """-> (kwargs of the reference Latte ctor, state_dict of torch tensors, dict of the other arrays)."""
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    kw = json.loads(bytes(z["cfg_json"]).decode())
    sd = {k[4:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd::")}
    rest = {k: z[k] for k in z.files if not k.startswith("sd::") and k != "cfg_json"}
    if "text_w_seed" in rest:   # extras == 78: the [D, 77*768] projection is a closed-form hash, not stored (oracle/make_golden.py)
        from oracle.latte_oracle import text_projection_weight
        sd["text_embedding_projection.1.weight"] = text_projection_weight(kw["hidden_size"], int(rest["text_w_seed"]))
    return kw, sd, rest


def oracle_config(kw):
    from oracle.latte_oracle import LatteConfig
    return LatteConfig(input_size=kw["input_size"], patch_size=kw["patch_size"], hidden_size=kw["hidden_size"],
                       depth=kw["depth"], num_heads=kw["num_heads"], num_frames=kw["num_frames"],
                       num_classes=kw.get("num_classes", 1000), learn_sigma=kw["learn_sigma"], extras=kw["extras"])


def engine_model(kw, sd, compute_dtype="bf16", max_batch=2, device="cuda"):
    """latte_amd.Latte with the reference-format weights `sd` on the GPU."""
    from latte_amd.models import Latte
    m = Latte(compute_dtype=compute_dtype, max_batch=max_batch, **kw)
    m.load_state_dict(sd, strict=True)
    return m.to(device).eval()
