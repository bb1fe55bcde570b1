"""Parity at the chain length the benchmark runs: 250-step DDIM / DDPM chains of the fused engine loop against chains run by
the REAL reference (tests/golden/chain250.npz, generator oracle/make_chain_golden.py: reference Latte module under the
reference's SpacedDiffusion loops, gaussian_diffusion.py:423-515 / :604-684 driven as sample/sample.py:67,100-107 does).

Operand types are the shim's default rule (latte_amd.Latte docstring): unguided chains bf16 (the benchmarked path), guided
chains (CFG 7.0 through forward_with_cfg) f16.  Tolerance: north_star's 1e-3 relative on the denoised latents, asserted
after every 50th step and on the final latents; the measured drift per checkpoint is written to
gpurun_out/chain250_drift.json (copied to profiles/ by the round script).
"""
import json
import os

import numpy as np
import pytest
import torch

import latte_amd
from _util import GOLDEN, ROOT, rel_l2
from latte_amd._lib import check, load_library, ptr, stream_ptr

pytestmark = pytest.mark.gpu
TOL = 1e-3

CHAIN = [(n, m) for n in ("s2_uncond", "s2_guided", "b2_uncond", "b2_guided") for m in ("ddim", "ddpm")] + [("xl_segment", "ddim")]


def _record(key, drift):
    path = os.path.join(ROOT, "gpurun_out", "chain250_drift.json")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    tab = {}
    if os.path.exists(path):
        with open(path) as f:
            tab = json.load(f)
    tab[key] = drift
    with open(path, "w") as f:
        json.dump(tab, f, indent=1, sort_keys=True)


@pytest.mark.parametrize("name,method", CHAIN, ids=[f"{n}-{m}" for n, m in CHAIN])
def test_chain_matches_reference_chain(name, method):
    from oracle.make_chain_golden import CFG_SCALE, case_inputs, chain_noises
    z = np.load(os.path.join(GOLDEN, "chain250.npz"))
    if f"{name}::{method}::samples" not in z.files:
        pytest.skip("fixture case not generated")
    preset, kw, cfg, sd, x0, y, steps, methods, nseed = case_inputs(name)
    guided = y is not None
    rows = x0.shape[0]
    m = latte_amd.Latte_models[preset](max_batch=rows, **kw)          # compute_dtype=None: bf16 unguided, f16 guided
    m.load_state_dict(sd)
    m = m.cuda()
    assert m.operand_dtype(guided) == ("f16" if guided else "bf16")
    d = latte_amd.create_diffusion("250")
    xx = x0.cuda().contiguous()
    nz = torch.stack(chain_noises(nseed, x0.shape, steps)).cuda().contiguous()
    ts = torch.empty((steps,) + tuple(xx.shape), device="cuda")
    yy = y.cuda() if guided else None
    check(load_library().latte_sample_loop(m.engine(rows, guided=guided), d._h, 1 if method == "ddim" else 0, 0.0, 0,
                                           CFG_SCALE if guided else 1.0, ptr(xx), ptr(yy), rows, 249, 250 - steps, ptr(nz), ptr(ts),
                                           None, stream_ptr()))
    torch.cuda.synchronize()
    assert torch.isfinite(xx).all()
    ks = z[f"{name}::{method}::steps"]
    want = torch.from_numpy(z[f"{name}::{method}::samples"])
    drift = {int(k) + 1: rel_l2(ts[int(k)], want[i]) for i, k in enumerate(ks)}
    _record(f"{name}::{method}::{m.operand_dtype(guided)}", drift)
    print(name, method, drift)
    assert int(ks[-1]) == steps - 1 and torch.equal(xx, ts[-1])
    for k, e in drift.items():
        assert e < TOL, (k, e)
