"""Run one GEMM shape/variant in a loop (for rocprofv3 --pmc / --kernel-trace)."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from latte_amd import _lib
from latte_amd._lib import check, load_library, stream_ptr

lib = load_library()
M, N, K, epi, variant, iters = (int(v) for v in sys.argv[1:7])
ms = _lib.c_f32()
torch.cuda.init()
check(lib.latte_bench_gemm(M, N, K, epi, 0, variant, iters, ctypes.byref(ms), stream_ptr()))
print(f"M={M} N={N} K={K} epi={epi} variant={variant}: {ms.value*1e3:.1f} us {2.0*M*N*K/(ms.value*1e-3)/1e12:.0f} TF/s")
