"""The C-ABI library: it builds (hipcc cross-compiles without a GPU), loads, and exports every
symbol include/*.h declares.  Host-only entry points (schedules, argument validation) are
exercised; no kernel is launched here."""
import ctypes
import hashlib
import json
import os
import re

import numpy as np
import pytest

from _util import GOLDEN, ROOT


def declared_symbols():
    names = set()
    inc = os.path.join(ROOT, "include")
    for f in os.listdir(inc):
        if f.endswith(".h"):
            src = open(os.path.join(inc, f)).read()
            src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
            names |= set(re.findall(r"\b(latte_[a-z0-9_]+)\s*\(", src))
    return sorted(names)


def test_exports_every_declared_symbol(lib):
    syms = declared_symbols()
    assert len(syms) >= 25
    missing = [s for s in syms if not hasattr(lib, s)]
    assert not missing, f"declared in include/*.h but not exported: {missing}"


def test_prototypes_cover_headers(lib):
    from latte_amd._lib import PROTOTYPES
    assert set(declared_symbols()) <= set(PROTOTYPES), sorted(set(declared_symbols()) - set(PROTOTYPES))


def test_version_and_error_string(lib):
    assert b"gfx950" in lib.latte_version()
    h = ctypes.c_void_p()
    rc = lib.latte_schedule_create(1000, b"ddim600", b"linear", ctypes.byref(h))
    assert rc != 0 and b"integer stride" in lib.latte_last_error()
    rc = lib.latte_schedule_create(10, b"20", b"linear", ctypes.byref(h))
    assert rc != 0 and b"cannot divide section" in lib.latte_last_error()
    rc = lib.latte_schedule_create(1000, b"250", b"nope", ctypes.byref(h))
    assert rc != 0 and b"unknown beta schedule" in lib.latte_last_error()


def _schedule(lib, steps, spec, name=b"linear"):
    h = ctypes.c_void_p()
    assert lib.latte_schedule_create(steps, spec.encode(), name, ctypes.byref(h)) == 0, lib.latte_last_error()
    n = lib.latte_schedule_num_timesteps(h)
    tm = np.empty(n, dtype=np.int64)
    assert lib.latte_schedule_timestep_map(h, tm.ctypes.data_as(ctypes.c_void_p), n) == 0
    return h, n, tm


def test_timestep_maps_bit_exact(lib):
    """Integer schedules must be bit-exact with the reference (north_star); KATs from SURVEY.md §8(c)."""
    kat = json.load(open(os.path.join(GOLDEN, "schedules_kat.json")))
    z = np.load(os.path.join(GOLDEN, "schedules.npz"))
    for spec, info in kat.items():
        h, n, tm = _schedule(lib, 1000, spec)
        assert n == info["n"], spec
        assert hashlib.sha256(tm.tobytes()).hexdigest() == info["sha256"], spec
        assert np.array_equal(tm, z[f"map::{spec}"]), spec
        lib.latte_schedule_destroy(h)


@pytest.mark.parametrize("spec", ["250", "10", "50", "ddim250", ""])
def test_fp64_tables(lib, spec):
    z = np.load(os.path.join(GOLDEN, "schedules.npz"))
    h, n, _ = _schedule(lib, 1000, spec)
    for name in ["betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_recip_alphas_cumprod",
                 "sqrt_recipm1_alphas_cumprod", "posterior_variance", "posterior_mean_coef1", "posterior_mean_coef2",
                 "posterior_log_variance_clipped", "log_betas"]:
        arr = np.empty(n, dtype=np.float64)
        assert lib.latte_schedule_table(h, name.encode(), arr.ctypes.data_as(ctypes.c_void_p), n) == 0
        ref = z[f"{name}::{spec}"]
        if "log" in name:
            # numpy's SIMD log and glibc's log may differ in the last place; what the device consumes
            # is the fp32 cast (gaussian_diffusion.py:878), which must be identical
            assert np.max(np.abs(arr - ref) / np.abs(ref)) < 4e-16, (spec, name)
            assert np.array_equal(arr.astype(np.float32), ref.astype(np.float32)), (spec, name)
        else:
            assert np.array_equal(arr, ref), (spec, name)
    lib.latte_schedule_destroy(h)


def test_cosine_schedule_through_abi(lib):
    z = np.load(os.path.join(GOLDEN, "schedules.npz"))
    h, n, tm = _schedule(lib, 400, "20", b"squaredcos_cap_v2")
    assert np.array_equal(tm, z["map::cos400/20"])
    arr = np.empty(n, dtype=np.float64)
    assert lib.latte_schedule_table(h, b"betas", arr.ctypes.data_as(ctypes.c_void_p), n) == 0
    assert np.allclose(arr, z["betas::cos400/20"], rtol=1e-14, atol=0)
    lib.latte_schedule_destroy(h)


def test_engine_argument_validation_without_gpu(lib):
    from latte_amd._lib import ModelConfig
    cfg = ModelConfig(32, 2, 4, 1152, 28, 16, 4608, 16, 101, 1, 2, 0)
    h = ctypes.c_void_p()
    bad = ModelConfig(32, 2, 4, 1100, 28, 16, 4400, 16, 101, 1, 2, 0)
    assert lib.latte_engine_create(ctypes.byref(bad), 1, ctypes.byref(h)) != 0
    assert b"multiple of 128" in lib.latte_last_error()
    bad = ModelConfig(32, 2, 4, 1152, 27, 16, 4608, 16, 101, 1, 2, 0)
    assert lib.latte_engine_create(ctypes.byref(bad), 1, ctypes.byref(h)) != 0
    assert b"even" in lib.latte_last_error()
    bad = ModelConfig(32, 2, 4, 1152, 28, 12, 4608, 16, 101, 1, 2, 0)
    assert lib.latte_engine_create(ctypes.byref(bad), 1, ctypes.byref(h)) != 0
    assert b"head_dim" in lib.latte_last_error()
    assert cfg.hidden_size == 1152
