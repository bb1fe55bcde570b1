"""Shader clock / power while a GEMM variant runs (rocm-smi sampled from a thread).  Measurement only."""
import ctypes
import os
import subprocess
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from latte_amd import _lib  # noqa: E402
from latte_amd._lib import check, load_library, stream_ptr  # noqa: E402

lib = load_library()
samples = []
stop = False


def sampler():
    while not stop:
        try:
            r = subprocess.run(["rocm-smi", "-c", "-P", "--json"], capture_output=True, text=True, timeout=5)
            samples.append((time.time(), r.stdout.strip()))
        except Exception as ex:  # noqa: BLE001
            samples.append((time.time(), "ERR " + str(ex)))
        time.sleep(0.15)


def main():
    global stop
    torch.zeros(1, device="cuda")
    th = threading.Thread(target=sampler, daemon=True)
    th.start()
    time.sleep(1.0)
    ms = _lib.c_f32()
    marks = []
    for name, cfg in [("idle", None), ("fc1 full epi1", (32768, 4608, 1152, 1, 9)), ("fc1 nostore", (32768, 4608, 1152, 4, 9)),
                                    ("fc1 noDMA", (32768, 4608, 1152, 6, 9)), ("fc1 noLDSread", (32768, 4608, 1152, 7, 9)),
                                    ("fc1 noMFMA", (32768, 4608, 1152, 8, 9)), ("fc2 full epi2", (32768, 1152, 4608, 2, 8))]:
        t0 = time.time()
        if cfg is None:
            time.sleep(1.0)
            marks.append((name, t0, time.time(), 0.0))
            continue
        M, N, K, epi, v = cfg
        check(lib.latte_bench_gemm(M, N, K, epi, 0, v, 6000, ctypes.byref(ms), stream_ptr()))
        marks.append((name, t0, time.time(), ms.value * 1e3))
    stop = True
    th.join(timeout=10)
    import json
    for name, a, b, us in marks:
        rows = []
        for t, s in samples:
            if a + 0.4 <= t <= b:
                try:
                    j = json.loads(s)
                    c = j[next(iter(j))]
                    rows.append({k: v for k, v in c.items() if "sclk" in k.lower() or "power" in k.lower() or "mclk" in k.lower()})
                except Exception:  # noqa: BLE001
                    rows.append(s[:200])
        print(f"== {name}: {us:.1f} us/launch, {len(rows)} samples")
        for r in rows[:: max(1, len(rows) // 4)]:
            print("   ", r)


if __name__ == "__main__":
    main()
