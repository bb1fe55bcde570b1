"""CPU prediction of the half-precision rounding budget (oracle/emulate_operands.py: the oracle forward with the engine's
rounding points applied) -- the numerics decisions of round 4 without a GPU:

* at trained-checkpoint gate magnitudes (gates of latte.py:178-180 at O(0.1 - 1)) bf16 operands put the model output at
  ~3e-3 of the fp32 reference, f16 operands at ~4e-4: f16 is the default operand type (latte_amd.Latte docstring);
* (round 5) the guided combination of latte.py:394-398 amplifies the part of the rounding that differs between the two halves: which
  rounding points carry it (test_guided_budget_by_rounding_point).
The GPU measurements of the same cases: tests/test_gpu_parity.py::test_forward_at_trained_scale_gates.
"""
import pytest
import torch

from oracle import latte_oracle as lo
from oracle.emulate_operands import latte_forward_emulated

TOL = 1e-3
KW = dict(input_size=16, num_frames=8, extras=1)


_CASE = {}


def _case(gate_std):
    if gate_std not in _CASE:
        cfg = lo.preset_config("Latte-S/2", **KW)
        sd = lo.init_state_dict(cfg, seed=0, gate_std=gate_std)
        x = torch.randn(1, 8, 4, 16, 16, generator=torch.Generator("cpu").manual_seed(1))
        t = torch.tensor([999])
        with torch.no_grad():
            _CASE[gate_std] = (cfg, sd, x, t, lo.latte_forward(sd, cfg, x, t))
    return _CASE[gate_std]


def _errs(gate_std, **emu):
    cfg, sd, x, t, ref = _case(gate_std)
    with torch.no_grad():
        out = latte_forward_emulated(sd, cfg, x, t, **emu)
    return float((out - ref).double().norm() / ref.double().norm())


def test_fp32_emulation_is_the_oracle():
    assert _errs(0.3, operand="fp32") < 1e-6


@pytest.mark.parametrize("gate_std", [0.3, 1.0])
def test_f16_holds_the_bar_at_trained_scale_gates_and_bf16_does_not(gate_std):
    e16, eb = _errs(gate_std, operand="f16"), _errs(gate_std, operand="bf16")
    assert e16 < 0.6 * TOL, e16
    assert eb > 2 * TOL, eb          # why bf16 is not the default: 2^-9 roundoff at full branch weight
    assert 4 < eb / e16 < 16         # the 2^-9 / 2^-11 ratio, not a bug in one path


def test_bf16_only_passes_at_near_zero_gates():
    assert _errs(0.02, operand="bf16") < TOL
