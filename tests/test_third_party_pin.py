"""Pins for the third-party (diffusers 0.24.0) arithmetic the oracles restate from memory -- SURVEY.md section 8 rows a22 / f2.

Two layers, both skip-if-absent (diffusers cannot be installed in the build container, and no fixture exists until someone runs
``python -m oracle.pin_third_party`` where it can be):
  * with diffusers importable: the live comparison (oracle/pin_third_party.py, 1e-5);
  * with ``tests/golden/third_party_pin.npz`` present (written by that script): the restatements against the stored diffusers
    outputs, no diffusers needed -- from then on the pin travels with the repo.
"""
import os

import numpy as np
import pytest
import torch

from _util import GOLDEN
from oracle import pin_third_party as pin

FIXTURE = os.path.join(GOLDEN, "third_party_pin.npz")
have_fixture = pytest.mark.skipif(not os.path.exists(FIXTURE), reason="tests/golden/third_party_pin.npz not generated yet "
                                  "(python -m oracle.pin_third_party where diffusers==0.24.0 imports)")
have_diffusers = pytest.mark.skipif(not pin.diffusers_available(), reason="diffusers is not importable here")


@have_diffusers
@pytest.mark.parametrize("fn", [pin.pin_vae, pin.pin_vae_temporal, pin.pin_scheduler, pin.pin_t2v_leaves], ids=lambda f: f.__name__)
def test_restatement_matches_live_diffusers(fn):
    what, err = fn({})
    if err is None:
        pytest.skip(what + ": needs /root/reference")
    assert err < pin.TOL, (what, err)


@have_fixture
def test_vae_oracle_matches_stored_diffusers_decode():
    z = np.load(FIXTURE)
    got = pin.vae_restated(torch.from_numpy(z["vae::z"]))
    assert pin.rel(got, torch.from_numpy(z["vae::out"])) < pin.TOL


@have_fixture
def test_temporal_decoder_oracle_matches_stored_diffusers_decode():
    z = np.load(FIXTURE)
    got = pin.vae_t_restated(torch.from_numpy(z["vae_t::z"]))
    assert pin.rel(got, torch.from_numpy(z["vae_t::out"])) < pin.TOL


@have_fixture
@pytest.mark.parametrize("n", [50, 20])
def test_ddim_scheduler_matches_stored_diffusers_trajectory(n):
    z = np.load(FIXTURE)
    ts, traj = pin.sched_restated(n, torch.from_numpy(z["sched::sample"]), torch.from_numpy(z["sched::eps"]))
    assert np.array_equal(ts.numpy().astype(np.int64), z[f"sched::{n}::timesteps"])
    want = torch.from_numpy(z[f"sched::{n}::trajectory"])
    assert max(pin.rel(traj[i], want[i]) for i in range(n)) < pin.TOL


@have_fixture
def test_t2v_oracle_matches_stored_reference_on_real_diffusers():
    z = np.load(FIXTURE)
    if "t2v::out" not in z.files:
        pytest.skip("the fixture was written without /root/reference")
    from oracle import latte_t2v_oracle as to
    cfg, x, t, enc, mask = pin.t2v_case()
    sd = {k[len("t2v::sd::"):]: torch.from_numpy(z[k]) for k in z.files if k.startswith("t2v::sd::")}
    with torch.no_grad():
        got = to.latte_t2v_forward(sd, cfg, x, t, enc, mask)
    assert pin.rel(got, torch.from_numpy(z["t2v::out"])) < pin.TOL


def test_pin_script_reports_unpinned_without_diffusers():
    """The one-command recipe exists and says so plainly when it cannot run (exit code 2, nothing written)."""
    if pin.diffusers_available():
        pytest.skip("diffusers is importable: the live tests above run instead")
    assert pin.main() == 2 and not os.path.exists(FIXTURE)
