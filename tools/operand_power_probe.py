"""Is the GEMM rate bound by the kernel's schedule or by the chip's power cap?  The same launches of the model's fc1 / fc2 / proj
shapes are timed on operands that toggle the MFMA data paths differently (random values, half of them zero, constants, zeros) in
both operand types, with rocm-smi power / shader clock sampled from a thread.  A schedule-bound kernel runs every pattern at the
same rate; a power-bound one speeds up as the data gets quieter.  Measurement only (profiles/r4_operand_power_probe.log)."""
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
samples = []
stop = False
SECONDS = float(sys.argv[1]) if len(sys.argv) > 1 else 1.2
PART = sys.argv[2] if len(sys.argv) > 2 else "gemm"      # gemm | fused (fused QKV + attention kernel, whole forward) | side (small batches, VAE)


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
        if not (a + 0.3 <= t <= b):
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


def operands(pattern, M, N, K, tdt, g):
    Mp = (M + 255) // 256 * 256
    if pattern == "random":
        A = torch.randn(Mp, K, generator=g, device="cuda")
        W = torch.randn(N, K, generator=g, device="cuda") / K ** 0.5
    elif pattern == "half_zero":          # what a GELU output looks like: half of the activations (almost) zero
        A = torch.randn(Mp, K, generator=g, device="cuda").clamp_min(0.0)
        W = torch.randn(N, K, generator=g, device="cuda") / K ** 0.5
    elif pattern == "random_x_zero_w":    # operand A toggles, every product is zero
        A = torch.randn(Mp, K, generator=g, device="cuda")
        W = torch.zeros(N, K, device="cuda")
    elif pattern == "constant":           # non-zero, no toggling between consecutive operands
        A = torch.ones(Mp, K, device="cuda")
        W = torch.full((N, K), 1.0 / K, device="cuda")
    elif pattern == "zeros":
        A = torch.zeros(Mp, K, device="cuda")
        W = torch.zeros(N, K, device="cuda")
    else:
        raise ValueError(pattern)
    return A.to(tdt), W.to(tdt)


def timed(launch, per_round, flops, tag):
    launch(10)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.time()
    n_total = 0
    e0.record()
    while time.time() - t0 < SECONDS:
        launch(per_round)
        n_total += per_round
        torch.cuda.synchronize()
    e1.record()
    torch.cuda.synchronize()
    t1 = time.time()
    us = e0.elapsed_time(e1) * 1e3 / n_total
    tf = flops / us * 1e-6
    p, clk, ns = smi_between(t0, t1)
    row = dict(tag, us_per_launch=round(us, 1), tflops=round(tf, 1), frac_of_2500=round(tf / 2500.0, 3), power_w=round(p, 0),
               sclk_mhz=round(clk, 0), smi_samples=ns)
    print(json.dumps(row), flush=True)


def fused_part():
    """The fused QKV + attention kernel on quiet / random operands, and the XL/2 forward at B = 8 with the power sampled."""
    import latte_amd
    g = torch.Generator(device="cuda").manual_seed(0)
    B, F, T, D, H = 8, 16, 256, 1152, 16
    M = B * F * T
    for dt_name, dt, tdt in (("f16", 1, torch.float16), ("bf16", 0, torch.bfloat16)):
        for pattern in ("random", "zeros"):
            if pattern == "random":
                xn = torch.randn(M, D, generator=g, device="cuda").to(tdt)
                W = (torch.randn(3 * D, D, generator=g, device="cuda") / D ** 0.5).to(tdt)
            else:
                xn = torch.zeros(M, D, device="cuda", dtype=tdt)
                W = torch.zeros(3 * D, D, device="cuda", dtype=tdt)
            bias = torch.zeros(3 * D, device="cuda")
            out = torch.zeros(M, D, device="cuda", dtype=tdt)
            for mode in (0, 1):
                L = T if mode == 0 else F
                flops = 2.0 * M * D * 3 * D + 4.0 * M * L * D

                def launch(n):
                    for _ in range(n):
                        check(lib.latte_debug_qkv_attention(ptr(xn), ptr(W), ptr(bias), ptr(out), None, B, F, T, D, H, mode, 7, dt,
                                                            stream_ptr()))
                timed(launch, 200, flops, dict(dtype=dt_name, kernel="qkv_attn_" + ("spatial" if mode == 0 else "temporal"),
                                               pattern=pattern))
    for dt_name in ("f16", "bf16"):
        m = latte_amd.Latte_models["Latte-XL/2"](input_size=32, num_frames=16, extras=1, compute_dtype=dt_name, max_batch=B)
        gc = torch.Generator("cpu").manual_seed(1)
        with torch.no_grad():
            for _, p in m.named_parameters():
                if p.requires_grad and float(p.detach().abs().max()) == 0.0:
                    p.copy_(torch.randn(p.shape, generator=gc) * 0.02)
        m = m.to("cuda").eval()
        x = torch.randn(B, 16, 4, 32, 32, device="cuda")
        t = torch.full((B,), 500, device="cuda", dtype=torch.int64)

        def fwd(n):
            for _ in range(n):
                m(x, t)
        timed(fwd, 10, B * 3.726e12, dict(dtype=dt_name, kernel="XL/2 forward B=8 (whole model, random latents and weights)"))
        del m


def side_part():
    """Which of the side paths run into the power cap?  XL/2 forward at B = 1 / 2 / 4 (with the per-class table of the B = 1 forward),
    the SD-VAE decode of a 16-frame video and the convolution class alone."""
    import latte_amd
    from latte_amd.random_init import vae_decoder_state_dict
    for B in (1, 2, 4):
        m = latte_amd.Latte_models["Latte-XL/2"](input_size=32, num_frames=16, extras=1, max_batch=B)
        gc = torch.Generator("cpu").manual_seed(1)
        with torch.no_grad():
            for _, p in m.named_parameters():
                if p.requires_grad and float(p.detach().abs().max()) == 0.0:
                    p.copy_(torch.randn(p.shape, generator=gc) * 0.02)
        m = m.to("cuda").eval()
        x = torch.randn(B, 16, 4, 32, 32, device="cuda")
        t = torch.full((B,), 500, device="cuda", dtype=torch.int64)

        def fwd(n):
            for _ in range(n):
                m(x, t)
        timed(fwd, 20, B * 3.726e12, dict(dtype=m.operand_dtype(), kernel=f"XL/2 forward B={B}"))
        if B == 1:
            m.profile_forward(x, t)
            prof = m.profile_forward(x, t)
            print(json.dumps({"B=1 per class: ms per forward, launches": {k: (round(v[0], 4), v[1]) for k, v in prof.items() if v[1]}}), flush=True)
        del m
    vae = latte_amd.AutoencoderKL(latent_size=32, max_frames=16)
    vae.load_state_dict(vae_decoder_state_dict(0))
    vae.to("cuda")
    lat = torch.randn(1, 16, 4, 32, 32, device="cuda") * 0.18215

    def dec(n):
        for _ in range(n):
            vae.decode_video_uint8(lat)
    timed(dec, 5, 9.741e12, dict(dtype="f16", kernel="SD-VAE decode, 16 frames 256x256 (9.74 TFLOP of convolutions per video)"))


def main():
    global stop
    torch.zeros(1, device="cuda")
    th = threading.Thread(target=sampler, daemon=True)
    th.start()
    time.sleep(0.5)
    if PART in ("fused", "side"):
        fused_part() if PART == "fused" else side_part()
        stop = True
        th.join(timeout=10)
        return
    g = torch.Generator(device="cuda").manual_seed(0)
    shapes = [("fc2", 32768, 1152, 4608, 2), ("fc1", 32768, 4608, 1152, 1), ("proj", 32768, 1152, 1152, 2)]
    rows = []
    for dt_name, dt, tdt in (("f16", 1, torch.float16), ("bf16", 0, torch.bfloat16)):
        for name, M, N, K, epi in shapes:
            Mp = (M + 255) // 256 * 256
            bias = torch.zeros(N, device="cuda")
            gate = torch.full((N,), 1e-3, device="cuda")
            for pattern in ("random", "half_zero", "random_x_zero_w", "constant", "zeros"):
                A, W = operands(pattern, M, N, K, tdt, g)
                out = torch.zeros(Mp, N, device="cuda", dtype=torch.float32 if epi >= 2 else tdt)

                def launch(n):
                    for _ in range(n):
                        check(lib.latte_debug_gemm(ptr(A), ptr(W), ptr(bias), ptr(out), ptr(gate), M, N, K, 0, M, epi, dt, 0,
                                                   stream_ptr()))
                launch(20)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                t0 = time.time()
                n_total = 0
                e0.record()
                while time.time() - t0 < SECONDS:
                    launch(200)
                    n_total += 200
                    torch.cuda.synchronize()
                e1.record()
                torch.cuda.synchronize()
                t1 = time.time()
                us = e0.elapsed_time(e1) * 1e3 / n_total
                tf = 2.0 * M * N * K / us * 1e-6
                p, clk, ns = smi_between(t0, t1)
                rows.append(dict(dtype=dt_name, gemm=name, pattern=pattern, us_per_launch=round(us, 1), tflops=round(tf, 1),
                                 frac_of_2500=round(tf / 2500.0, 3), power_w=round(p, 0), sclk_mhz=round(clk, 0), smi_samples=ns))
                print(json.dumps(rows[-1]), flush=True)
                del A, W, out
    stop = True
    th.join(timeout=10)
    if samples:
        print("last rocm-smi sample:", samples[-1][1][:400])


if __name__ == "__main__":
    main()
