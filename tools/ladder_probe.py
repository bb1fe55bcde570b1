"""Round 6: the probe-to-kernel LADDER.  Round 5 measured what this socket gives a pure f16 MFMA stream under its 1400 W cap (0.72-0.75 of
the 2.5 PF peak on random operands, 0.61 with the 256 x 192 tile's operand DMA beside it) and the production fc2 GEMM at 0.445 -- and
nobody had attributed the distance.  Here the PRODUCTION kernel (gemm_pwr_kernel, the rolling 12-wave producer / consumer GEMM of
csrc/gemm_pw.hip) is taken apart top-down, one feature per rung, in the measurement build (template parameter ABL; results of the ablated
launches are garbage, their instruction streams are the production kernel's minus the feature):

  rung 7  the production launch (gated fp32 read-modify-write epilogue, real operands)
  rung 6  - the epilogue            (accumulators dropped at a tile boundary: no residual read-modify-write burst)
  rung 5  - the tile boundaries     (one K loop over the whole walk: no drain / refill per output tile)
  rung 4  - real operand addresses  (every DMA reads tile 0, K tiles 0-3: an L2-resident pool instead of HBM / Infinity Cache streams)
  rung 3  - waits and barriers      (no vmcnt / lgkmcnt waits, no workgroup barrier inside the K walk)
  rung 2  - LDS fragment re-reads   (fragments read once, reused)
  rung 1  - operand DMA             (= the MFMA stream alone inside this kernel's skeleton)

Every rung on random (f16 ~ N(0,1) x N(0,1)/sqrt(K)) AND all-zero operands, settled under the power cap, with rocm-smi socket power and
shader clock sampled beside it:  us per launch, TFLOP/s of the launch's algorithmic FLOPs, fraction of 2.5 PF, W, MHz, pJ per FLOP.
Usage (measurement build):  LATTE_DEBUG_BUILD=1 python -m latte_amd.build;
  LATTE_AMD_LIB=latte_amd/lib/liblatte_amd_dbg.so python tools/ladder_probe.py [seconds per case] [fc2|proj|both]
"""
import json
import os
import subprocess
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from latte_amd._lib import check, load_library, ptr, stream_ptr  # noqa: E402

lib = load_library()
SECONDS = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
WHICH = sys.argv[2] if len(sys.argv) > 2 else "both"
ONLY = [int(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else None     # ablation masks to run (default: all)
samples, stop = [], False


def sampler():
    while not stop:
        try:
            r = subprocess.run(["rocm-smi", "-c", "-P", "--json"], capture_output=True, text=True, timeout=5)
            samples.append((time.time(), r.stdout.strip()))
        except Exception as ex:  # noqa: BLE001
            samples.append((time.time(), "ERR " + str(ex)))
        time.sleep(0.1)


def smi_between(a, b):
    power, sclk = [], []
    for t, s in samples:
        if not (a + 0.4 <= t <= b):
            continue
        try:
            j = json.loads(s)
            c = j[next(iter(j))]
        except Exception:  # noqa: BLE001
            continue
        for k, v in c.items():
            kl = k.lower()
            try:
                if "power" in kl:
                    power.append(float(str(v).split()[0]))
                elif "sclk" in kl and "clock speed" in kl:
                    sclk.append(float(str(v).strip("()MmHhZz ")))
            except ValueError:
                pass
    mean = lambda v: sum(v) / len(v) if v else float("nan")  # noqa: E731
    return mean(power), mean(sclk), len(power)


RUNGS = [(7, 0, "production launch"), (6, 1, "- epilogue"), (5, 3, "- tile boundaries"), (4, 7, "- real operand addresses (L2-resident pool)"),
         (3, 15, "- waits and barriers"), (2, 31, "- LDS fragment re-reads"), (1, 63, "- operand DMA (MFMA stream alone)")]
# single features taken away from the production launch on their own (not cumulative): where one feature's cost depends on the rest
SINGLES = [(4, "production - real operand addresses only"), (32, "production - operand DMA only"), (16, "production - LDS fragment re-reads only"),
           (5, "production - epilogue - real addresses"), (35, "production - epilogue - tile boundaries - DMA"),
           (64, "production, read-modify-write as fire-and-forget fp32 atomic adds"),
           (128, "production, 6 residual fragments in flight per wave instead of 2"), (256, "production, 12 residual fragments in flight")]


def main():
    global stop
    torch.zeros(1, device="cuda")
    th = threading.Thread(target=sampler, daemon=True)
    th.start()
    time.sleep(0.5)
    g = torch.Generator(device="cuda").manual_seed(0)
    shapes = [("fc2", 32768, 1152, 4608), ("proj", 32768, 1152, 1152)]
    if WHICH != "both":
        shapes = [s for s in shapes if s[0] == WHICH]
    for name, M, N, K in shapes:
        flops = 2.0 * M * N * K
        bias = torch.zeros(N, device="cuda")
        gate = torch.full((N,), 1e-3, device="cuda")
        for pattern in ("random", "zeros"):
            if pattern == "random":
                A = torch.randn(M, K, generator=g, device="cuda").to(torch.float16)
                W = (torch.randn(N, K, generator=g, device="cuda") / K ** 0.5).to(torch.float16)
            else:
                A = torch.zeros(M, K, device="cuda", dtype=torch.float16)
                W = torch.zeros(N, K, device="cuda", dtype=torch.float16)
            out = torch.zeros(M, N, device="cuda")
            for rung, mask, what in RUNGS + [(0, m_, w_) for m_, w_ in SINGLES]:
                if ONLY is not None and mask not in ONLY:
                    continue
                if mask:
                    os.environ["LATTE_PWR_ABL"] = str(mask)
                else:
                    os.environ.pop("LATTE_PWR_ABL", None)

                def launch(n):
                    for _ in range(n):
                        check(lib.latte_debug_gemm(ptr(A), ptr(W), ptr(bias), ptr(out), ptr(gate), M, N, K, 0, M, 2, 1, 1011, stream_ptr()))
                launch(20)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                t0 = time.time()
                n_total = 0
                e0.record()
                while time.time() - t0 < SECONDS:
                    launch(100)
                    n_total += 100
                    torch.cuda.synchronize()
                e1.record()
                torch.cuda.synchronize()
                t1 = time.time()
                us = e0.elapsed_time(e1) * 1e3 / n_total
                tf = flops / us * 1e-6
                p, clk, ns = smi_between(t0, t1)
                print(json.dumps(dict(gemm=name, operands=pattern, rung=rung, abl_mask=mask, what=what, us_per_launch=round(us, 1), tflops=round(tf, 1),
                                      frac_of_2500=round(tf / 2500.0, 3), power_w=round(p, 0), sclk_mhz=round(clk, 0),
                                      pj_per_flop=round(p * us * 1e-6 / flops * 1e12, 3), smi_samples=ns)), flush=True)
            del A, W, out
    os.environ.pop("LATTE_PWR_ABL", None)
    stop = True
    th.join(timeout=10)


def in_model(masks):
    """The same ablation masks INSIDE the XL/2 forward at B = 8 (the residual HBM-cold, A just written by the kernel in front): per-launch
    HIP-event times of the two gated GEMM classes, interleaved in one process.  Only masks with CORRECT results make sense here."""
    import latte_amd
    B = 8
    m = latte_amd.Latte_models["Latte-XL/2"](max_batch=B, input_size=32, num_frames=16, extras=1)
    with torch.no_grad():
        gg = torch.Generator().manual_seed(0)
        for _, p in m.named_parameters():
            if float(p.abs().max()) == 0.0:
                p.copy_(torch.randn(p.shape, generator=gg) * 0.02)
    m = m.cuda()
    x = torch.randn(B, 16, 4, 32, 32, device="cuda")
    t = torch.full((B,), 500, device="cuda", dtype=torch.int64)
    for rep in range(3):
        for mask in masks:
            if mask:
                os.environ["LATTE_PWR_ABL"] = str(mask)
            else:
                os.environ.pop("LATTE_PWR_ABL", None)
            m.profile_forward(x, t)
            pr = [m.profile_forward(x, t) for _ in range(3)]
            row = {k: round(min(p[k][0] for p in pr) / max(pr[0][k][1], 1) * 1e3, 1) for k in ("gemm_proj", "gemm_fc2", "gemm_fc1", "ln_modulate")}
            print(json.dumps(dict(in_model=True, abl_mask=mask, rep=rep, us_per_launch=row, forward_ms=round(min(sum(v[0] for v in p.values()) for p in pr), 3))), flush=True)
    os.environ.pop("LATTE_PWR_ABL", None)


if __name__ == "__main__":
    if WHICH == "inmodel":
        in_model(ONLY or [0])
    else:
        main()
