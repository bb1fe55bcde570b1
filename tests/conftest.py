import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The fp32 oracle (CPU torch) is the slow part of the GPU suite.  On the 256-core host of a GPU box torch's default thread count
    # oversubscribes it: an XL/2 oracle forward takes 3.5-4.3 s on 32 threads against 4.9-7.2 s on 64 and 9.4-14 s on 128
    # (bench.py's cpu_baseline sweep).  Tolerance-based comparisons do not depend on the thread count; the
    # bit-exact ones compare GPU results with each other or with committed fixtures.
    import torch
    torch.set_num_threads(max(1, min(32, os.cpu_count() or 1)))


@pytest.fixture(scope="session")
def lib():
    """The built C-ABI library (built on demand: hipcc cross-compiles without a GPU)."""
    from latte_amd import build as _build
    from latte_amd._lib import load_library
    _build.build()
    return load_library()


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
