"""bench.py — denoising steps/sec of the Latte-XL/2 16x256x256 sampling loop on MI355X.

  python bench.py --gpus N --steps K --warmup W

N > 1: one process per GPU over RCCL.  Under torch.distributed.run (WORLD_SIZE / RANK / LOCAL_RANK in the environment, the
driver's launch line) the process is one rank and asserts WORLD_SIZE == --gpus; started plainly (`python bench.py --gpus N`)
it re-executes itself as `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
--master-port <free port> bench.py <same flags>` -- the reference's sample_ddp.py is launched the same way
(sample/sample_ddp.py:60-66: one rank per device, torchrun environment).

One "step" = one pass of the hot path (denoiser forward + sampler update) over the rank's batch of
latents, inputs resident in HBM.  Workload (every N) = BASELINE.json configs[1]: Latte-XL/2, FaceForensics
(unconditional) config, 16 frames of 32x32 latents, per-GPU batch 8 (= BASELINE config 3's per-GPU share
of its batch 64; `--batch 2` is the reference YAML's per_proc_batch_size), DDIM eta=0 over the "250"
respacing, random-init weights (adaLN/final layers re-drawn N(0, 0.02) so the network is not the
identity), synthetic N(0,1) latents.  Weak scaling: every rank runs its own samples, no data-path
collective (the reference's sample_ddp.py has none either); the only payload collective is the one-off
RCCL broadcast of the timestep-embedding table before the timed region.
`value` = aggregate denoising sample-steps/s = n_gpus * batch * K / max-over-ranks time.

Beside the headline, in the same JSON line:
  * `roofline`: the DOMINANT kernel of the step (largest share of the forward by HIP events on the launch
    stream) and `roofline_table`: every kernel class with its bound, algorithmic work per launch, average
    launch time and fraction of the MI355X peak (MFMA 2.5 PFLOP/s dense bf16/f16, HBM 8 TB/s);
  * `dtype` = f16 (round 4): the MFMA operand type that holds the 1e-3 parity bar on weights with trained-checkpoint gate
    magnitudes (profiles/r4_gate_parity.json; latte_amd.Latte docstring) and the reference's own half mode (sample.py:72-75);
    `bf16`: the same timed loop with bf16 operands (BASELINE config 2's word; 1e-3 only at near-zero gates), reported beside it;
  * `config3`: BASELINE config 3's per-GPU share -- UCF101 class-conditional Latte-XL/2, CFG 7.0, 8 samples =
    16 sequences per GPU through forward_with_cfg (aggregate guided sample-steps/s over all ranks);
  * `config5`: BASELINE config 5's per-GPU share -- one optimisation step of train.py on Latte-B/2 16x256x256 synthetic
    latents, local batch 5 (configs/ffs/ffs_train.yaml), forward + loss + backward + gradient all-reduce (N > 1: one RCCL
    all-reduce of the 130 M-parameter fp32 gradient buffer) + clip + AdamW + EMA; samples/s over all ranks and the
    algorithmic TFLOP/s (3 x forward FLOPs);
  * `config4` (N = 1): BASELINE config 4 -- Latte-1 text-to-video 512x512x16: one guided DDIM step of LatteT2V inside the
    engine and the 16-frame AutoencoderKLTemporalDecoder decode, random weights of the real shapes;
  * `power_check` (N = 1): socket power and shader clock (rocm-smi) during ~1.5 s of the headline forward, and the dominant GEMM
    stand-alone on random and on all-zero operands -- on random operands the 1400 W socket cap throttles the MFMA kernels to
    ~1.7-1.9 GHz (of 2.4); on quiet operands the same launches keep the full clock (profiles/r4_operand_power_probe_*.log);
  * `energy` (N = 1, round 5): joules per step / per sample-step / per 250-step video and pJ per algorithmic FLOP = the forward's mean
    socket power x this run's step time; `roofline_table` rows carry `joules_per_launch` (the same power x the class's launch time);
  * `cpu_baseline` (N = 1): the oracle's sampling-loop body (forward + ddim_sample) on the host cores, standing in for the reference's
    loop (the reference itself is not on the GPU box; the oracle is bit-identical to it and runs ~6 % faster
    than it because it skips the reference's repeated adaLN rows: oracle/VALIDATION.md).
"""
import argparse
import ctypes
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOPS_PER_SAMPLE_STEP = {"Latte-XL/2": 3.726e12}  # SURVEY.md §8(d), algorithmic, 16x32x32 latents
MFMA_PEAK_TFLOPS = 2500.0                           # MI355X_MICROARCH.md: bf16/f16 dense
HBM_PEAK_GBS = 8000.0                               # MI355X_MICROARCH.md: HBM3E spec peak


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=250)
    p.add_argument("--warmup", type=int, default=5)
    p.add_argument("--batch", type=int, default=8, help="samples per GPU (8 = config 3's share; 2 = ffs_sample.yaml)")
    p.add_argument("--dtype", default="f16", choices=["bf16", "f16"])
    p.add_argument("--method", default="ddim", choices=["ddim", "ddpm"])
    p.add_argument("--gemm-variant", type=int, default=0)
    p.add_argument("--engine-option", action="append", default=[], metavar="NAME=VALUE",
                   help="latte_engine_set_option on the headline engine before the timed region (A/B measurements, e.g. fuse_qkv_attn=0)")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-vae", action="store_true", help="skip the (untimed-region) VAE decode rate report")
    p.add_argument("--no-side", action="store_true", help="skip the f16 / config-3 side measurements")
    p.add_argument("--launch-check", action="store_true",
                   help="launch N ranks, rendezvous, one all-reduce, print {launch_check, n_gpus, collective_ranks} and exit: "
                        "exercises the --gpus contract without measuring anything (runs without a GPU on the gloo backend)")
    p.add_argument("--cpu-steps", type=int, default=3, help="timed DDIM steps of the CPU baseline (about 5 s each)")
    return p.parse_args()


def build_model(device, dtype, batch, extras=1, num_classes=1000):
    import latte_amd
    torch.manual_seed(0)
    m = latte_amd.Latte_models["Latte-XL/2"](input_size=32, num_frames=16, extras=extras, num_classes=num_classes,
                                             learn_sigma=True, compute_dtype=dtype, max_batch=batch)
    g = torch.Generator("cpu").manual_seed(1)
    with torch.no_grad():
        for _, prm in m.named_parameters():
            if prm.requires_grad and float(prm.detach().abs().max()) == 0.0:
                prm.copy_(torch.randn(prm.shape, generator=g) * 0.02)
    return m.to(device).eval()


def run_steps(lib, model, diffusion, x, n_steps, method, batch, y=None, cfg_scale=1.0, guided=False):
    """Exactly n_steps denoising steps: chains of up to num_timesteps steps through the fused loop."""
    from latte_amd._lib import check, ptr, stream_ptr
    eng = model.engine(batch, guided=guided)
    T = diffusion.num_timesteps
    left = n_steps
    mi = 1 if method == "ddim" else 0
    while left > 0:
        seg = min(left, T)
        check(lib.latte_sample_loop_ex(eng, diffusion._h, mi, 0.0, 0, int(guided), cfg_scale, ptr(x), ptr(y), batch, T - 1,
                                       T - seg, None, None, None, stream_ptr()))
        left -= seg


def timed_steps(lib, model, diffusion, x, steps, method, batch, **kw):
    """Side measurement (not the headline): 2 warm-up steps, then `steps` timed ones; seconds per step."""
    run_steps(lib, model, diffusion, x, 2, method, batch, **kw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run_steps(lib, model, diffusion, x, steps, method, batch, **kw)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def cpu_baseline(n_steps):
    """The oracle (a CPU port of the reference, bit-identical to it: oracle/VALIDATION.md) running the reference's sampling
    loop body -- denoiser forward + `ddim_sample` (gaussian_diffusion.py:604-684) on the "250" respacing -- on this box's
    host cores; reported beside the GPU number, never the thing measured.  Thread count: one untimed-for-the-result forward
    at 32, 64 and 128 threads (those the box has), the fastest count runs the timed steps (MKL / OpenMP GEMMs of this size
    stop scaling, and can collapse, well below the 256 hardware threads of the GPU box)."""
    from oracle import diffusion_oracle as do
    from oracle import latte_oracle as lo
    cfg = lo.preset_config("Latte-XL/2", input_size=32, num_frames=16, extras=1)
    sd = lo.init_state_dict(cfg, seed=0)
    x = torch.randn(1, 16, 4, 32, 32)
    s = do.Schedule("250")
    visible = len(os.sched_getaffinity(0))
    sweep = {}
    with torch.no_grad():
        torch.set_num_threads(min(visible, 32))
        lo.latte_forward(sd, cfg, x, torch.tensor([500]))                       # warm-up (page-in, MKL init)
        for n in sorted({min(visible, c) for c in (32, 64, 128)}):
            torch.set_num_threads(n)
            t0 = time.time()
            lo.latte_forward(sd, cfg, x, torch.tensor([500]))
            sweep[n] = round(time.time() - t0, 3)
        cores = min(sweep, key=sweep.get)
        torch.set_num_threads(cores)
        t0 = time.time()
        for k in range(n_steps):                                                 # the loop body of gd:637-684
            i = s.num_timesteps - 1 - k
            out = lo.latte_forward(sd, cfg, x, torch.full((1,), s.timestep_map[i], dtype=torch.int64))
            x = do.ddim_sample(s, out, x, i, None, 0.0, False)["sample"]
        dt = (time.time() - t0) / n_steps
    return {"value": round(1.0 / dt, 4), "unit": "denoising sample-steps/s", "cores": cores,
            "host_cores_visible": visible, "kind": "port", "seconds_per_forward_by_threads": sweep,
            "sample": f"{n_steps} timed DDIM steps (oracle forward + ddim_sample, fp32, B=1, Latte-XL/2 16x32x32 latents, '250' "
                      "respacing) after 1 warm-up forward and a 32/64/128-thread sweep; the oracle is bit-identical to the "
                      "reference loop and ~6 % faster than it (it skips the repeated adaLN rows; oracle/VALIDATION.md: "
                      "6.61 s vs 7.06 s per forward)"}


def pmc_table():
    """HBM-side bytes per launch from the rocprofv3 PMC passes committed under profiles/ (tools/pmc_collect.py:
    FETCH_SIZE doubled per the guide's gfx950 correction + WRITE_SIZE) -- a committed constant per kernel class and shape,
    NOT a per-run measurement: rocprofv3 cannot run inside this process.  Files are merged oldest round first, so every
    (class, shape) key carries the NEWEST pass that measured it; -> ({key: record}, {key: source file})."""
    import glob
    import re

    def order(path):
        m = re.match(r"r(\d+)([a-z]*)_", os.path.basename(path))
        return (int(m.group(1)), m.group(2)) if m else (-1, "")
    tab, src = {}, {}
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc.json")), key=order):
        with open(path) as f:
            try:
                rec = json.load(f)
            except ValueError:
                continue
        for k, v in rec.items():
            if isinstance(v, dict) and "hbm_bytes_per_launch" in v:
                tab[k] = v
                src[k] = "profiles/" + os.path.basename(path)
    return tab, src


def roofline_table(prof, B, dtype):
    """Per kernel class: algorithmic work per launch (2 M N K for the linears, 4 S L^2 D for the
    attention products; bytes = the operands a launch must read / write once), average launch time by HIP events, and
    the fraction of the roofline that bounds it."""
    D, Hm, F, T = 1152, 4608, 16, 256
    M = B * F * T
    mfma = {"gemm_qkv": (2.0 * M * 3 * D * D, "gemm_pps_kernel<256, EPI_BIAS_H16>: qkv projection M=%d N=3456 K=1152" % M),
            "gemm_proj": (2.0 * M * D * D, "gemm_pwr_kernel<EPI_GATE_RES_F32, tag 0> (12-wave producer/consumer, 256x192): attention out-projection M=%d N=1152 K=1152, gated fp32 residual RMW" % M),
            "gemm_fc1": (2.0 * M * Hm * D, "gemm_pps_kernel<256, EPI_BIAS_GELU_H16>: fc1 M=%d N=4608 K=1152, bias+GELU" % M),
            "gemm_fc2": (2.0 * M * D * Hm, "gemm_pwr_kernel<EPI_GATE_RES_F32, tag 1> (12-wave producer/consumer, 256x192): fc2 M=%d N=1152 K=4608, gated fp32 residual RMW" % M),
            # the fused QKV projection + attention kernel (csrc/qkv_attn.hip): algorithmic FLOPs = the projection (2 M 3D D) + the
            # two attention products (4 S L^2 D); q / k / v never reach HBM
            "qkv_attn_spatial": (2.0 * M * 3 * D * D + 4.0 * B * F * T * T * D,
                                 "qkv_attn_kernel<72, MODE 0>: QKV projection + spatial attention fused, %d sequences x 16 heads x 256 tokens, q/k/v in LDS" % (B * F)),
            "qkv_attn_temporal": (2.0 * M * 3 * D * D + 4.0 * B * T * F * F * D,
                                  "qkv_attn_kernel<72, MODE 1>: QKV projection + temporal attention fused, %d sequences x 16 heads x 16 frames, q/k/v in LDS" % (B * T))}
    qkv_bytes = M * 3 * D * 2 + M * D * 2      # reads q, k, v once, writes the head outputs
    hbm = {"attn_spatial": (qkv_bytes, 4.0 * B * F * T * T * D, "attn_full_kernel<72>: spatial attention, %d sequences x 16 heads x 256 tokens" % (B * F)),
           "attn_temporal": (qkv_bytes, 4.0 * B * T * F * F * D, "attn_small_kernel<72>: temporal attention, %d sequences x 16 heads x 16 frames" % (B * T)),
           "ln_modulate": (M * D * (4 + 2), 0.0, "ln_modulate_kernel: LayerNorm + adaLN modulate, fp32 in, half out")}
    pmc, pmc_src = pmc_table()
    total_ms = sum(v[0] for v in prof.values())
    rows = []
    for k, (ms, n) in prof.items():
        if n == 0 or (k not in mfma and k not in hbm):
            continue
        avg = ms / n
        row = {"class": k, "launches_per_forward": n, "avg_launch_ms": round(avg, 4), "share_of_forward": round(ms / total_ms, 4)}
        if k in mfma:
            fl, name = mfma[k]
            ach = fl / (avg * 1e-3) / 1e12
            row.update({"kernel": name, "bound": "mfma", "achieved": round(ach, 1), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": round(ach / MFMA_PEAK_TFLOPS, 4), "algorithmic_flops_per_launch": fl})
        else:
            by, fl, name = hbm[k]
            ach = by / (avg * 1e-3) / 1e9
            row.update({"kernel": name, "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(ach / HBM_PEAK_GBS, 4), "algorithmic_bytes_per_launch": by})
            if fl:
                row["mfma_frac_of_peak"] = round(fl / (avg * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS, 4)
        rec = pmc.get(f"{k}:M={M}")
        row["traffic"] = rec["hbm_bytes_per_launch"] if rec else None
        row["traffic_source"] = (pmc_src[f"{k}:M={M}"] + " (committed rocprofv3 PMC pass, not measured in this run)") if rec else None
        rows.append(row)
    rows.sort(key=lambda r: -r["share_of_forward"])
    return rows, total_ms


def vae_decoder_flops(h):
    """Algorithmic FLOPs of one frame of AutoencoderKL.decode at an h x h latent, from the layer shapes (2 * pixels * Cin *
    Cout * taps per convolution; the one attention as four 512 x 512 linears + the two L x L products): 0.622 TFLOP at
    h = 32."""
    def conv(H, cin, cout, k=3):
        return 2 * H * H * cin * cout * k * k

    def resnet(H, cin, cout):
        return conv(H, cin, cout) + conv(H, cout, cout) + (conv(H, cin, cout, 1) if cin != cout else 0)

    f = conv(h, 4, 4, 1) + conv(h, 4, 512) + 2 * resnet(h, 512, 512)
    L = h * h
    f += 4 * 2 * L * 512 * 512 + 2 * 2 * L * L * 512
    H, prev = h, 512
    for i, c in enumerate((512, 512, 256, 128)):
        for r in range(3):
            f += resnet(H, prev if r == 0 else c, c)
        prev = c
        if i < 3:
            H *= 2
            f += conv(H, c, c)
    return f + conv(H, 128, 3)


def vae_temporal_decoder_flops(h):
    """Algorithmic FLOPs PER FRAME of AutoencoderKLTemporalDecoder.decode at an h x h latent (stable-video-diffusion's VAE decoder as
    oracle/vae_temporal_oracle.py restates it): the SD-VAE decoder without post_quant_conv, plus, behind every one of the 14 resnets, a temporal
    resnet of two Conv3d (3, 1, 1) c -> c at that resnet's output width / resolution (2 * pixels * c * c * 3 each), plus the 3-tap
    time_conv_out on RGB.  -> (total, FLOPs of the convolutions that run on the MFMA implicit-GEMM kernels: 3x3 spatial + (3,1,1) temporal).
    Counted ONCE: the engine runs the spatial convolutions with a second pass on the activation remainder and the temporal ones as three
    split-operand passes for parity -- that is cost, not algorithmic work."""
    conv3 = lambda H, cin, cout: 2 * H * H * cin * cout * 9
    tconv = lambda H, c: 2 * H * H * c * c * 3
    mfma = 0
    for _ in range(2):                       # mid block
        mfma += 2 * conv3(h, 512, 512) + 2 * tconv(h, 512)
    H, prev = h, 512
    for i, c in enumerate((512, 512, 256, 128)):
        for r in range(3):
            cin = prev if r == 0 else c
            mfma += conv3(H, cin, c) + conv3(H, c, c) + 2 * tconv(H, c)
        prev = c
        if i < 3:
            H *= 2
            mfma += conv3(H, c, c)
    total = vae_decoder_flops(h) - 2 * h * h * 4 * 4 + (mfma - vae_decoder_work(h)[0]) + 2 * H * H * 3 * 3 * 3
    return total, mfma


def vae_decoder_work(h):
    """Per frame of AutoencoderKL.decode at an h x h latent: (FLOPs of the 3x3 convolutions that run on the MFMA implicit-GEMM kernel,
    algorithmic bytes of the GroupNorm statistics passes = every GroupNorm input read once (fp32 stream), algorithmic bytes of the
    GroupNorm apply passes = that input read once more + the half operand written).  30 GroupNorms: 2 per resnet (14 resnets:
    2 mid + 12 up), mid attention, conv_norm_out."""
    conv3 = lambda H, cin, cout: 2 * H * H * cin * cout * 9
    fl, gn_in = 0, 0
    gn = lambda H, c: H * H * c * 4
    fl += 0   # conv_in (4 -> 512) is a small direct kernel, not the MFMA kernel
    for _ in range(2):                       # mid resnets
        fl += 2 * conv3(h, 512, 512)
        gn_in += 2 * gn(h, 512)
    gn_in += gn(h, 512)                      # mid attention's norm
    H, prev = h, 512
    for i, c in enumerate((512, 512, 256, 128)):
        for r in range(3):
            cin = prev if r == 0 else c
            fl += conv3(H, cin, c) + conv3(H, c, c)
            gn_in += gn(H, cin) + gn(H, c)
        prev = c
        if i < 3:
            H *= 2
            fl += conv3(H, c, c)
    gn_in += gn(H, 128)                      # conv_norm_out
    return fl, gn_in, gn_in + gn_in // 2


def vae_conv_bytes(h, frames=16):
    """Algorithmic bytes of the 3x3 convolutions of `frames` frames of AutoencoderKL.decode at an h x h latent: every convolution reads
    its half operand once (the upsampler convolutions at the pre-upsample size), writes its fp32 output once, reads the fp32 residual
    once where it adds one (the second convolution of every resnet), and the weights are read once per decode.  14.4 GB at h = 32,
    16 frames -- the 24.4 GB the PMC pass counts through the fabric (profiles/r4_pmc.json) are 1.7 x that: a 256-pixel tile re-reads
    the rows above / below it for its 9 taps out of the other XCDs' reach, and every tile streams its weight slice again.  At 11 ms
    even the fabric figure is 2.2 TB/s: the convolutions are nowhere near a memory bound."""
    def conv(H, cin, cout, res=False, ups=False):
        hin = H // 2 if ups else H
        return hin * hin * cin * 2 + H * H * cout * 4 + (H * H * cout * 4 if res else 0)
    act = wts = 0
    for _ in range(2):
        act += conv(h, 512, 512) + conv(h, 512, 512, res=True)
        wts += 2 * 9 * 512 * 512 * 2
    H, prev = h, 512
    for i, c in enumerate((512, 512, 256, 128)):
        for r in range(3):
            cin = prev if r == 0 else c
            act += conv(H, cin, c) + conv(H, c, c, res=True)
            wts += 9 * cin * c * 2 + 9 * c * c * 2
        prev = c
        if i < 3:
            H *= 2
            act += conv(H, c, c, ups=True)
            wts += 9 * c * c * 2
    return frames * act + wts


def parse_rocm_smi(txt):
    """(socket power in W, shader clock in MHz) of the first card in `rocm-smi -c -P --json` output; None for what is not there."""
    try:
        card = next(iter(json.loads(txt).values()))
    except Exception:  # noqa: BLE001
        return None, None
    power = clock = None
    if not isinstance(card, dict):
        return None, None
    for k, v in card.items():
        kl = k.lower()
        try:
            if "power" in kl and power is None:
                power = float(str(v).split()[0])
            elif "sclk" in kl and "speed" in kl:
                clock = float(str(v).strip("()MmHhZz "))
        except (ValueError, IndexError):
            pass
    return power, clock


def mean_power_clock(samples, a, b):
    """Mean socket power / shader clock over the rocm-smi samples [(time, json text)] taken in [a + 0.3 s, b] (the first 0.3 s of a
    leg still show the previous leg's power), and the number of power samples."""
    pw, ck = [], []
    for ts, txt in samples:
        if a + 0.3 <= ts <= b:
            p_w, c_mhz = parse_rocm_smi(txt)
            if p_w is not None:
                pw.append(p_w)
            if c_mhz is not None:
                ck.append(c_mhz)
    return (round(sum(pw) / len(pw), 0) if pw else None), (round(sum(ck) / len(ck), 0) if ck else None), len(pw)


def power_check(device, model, x, dtype):
    """Is the step bound by the kernels' schedule or by the socket's power cap?  (tools/operand_power_probe.py is the full table,
    profiles/r4_operand_power_probe_*.log.)  Two short legs outside the timed region: (1) ~1.5 s of the headline forward with
    rocm-smi's socket power and shader clock sampled from a thread; (2) the dominant GEMM (fc2: 32768 x 1152 x 4608, gated
    read-modify-write epilogue) stand-alone on random operands and on all-zero operands -- the same instruction stream and the same
    bytes, but nothing toggles in the MFMA data paths, so the chip keeps its full clock: that rate is what the kernel's schedule
    reaches when the 1400 W cap does not throttle it."""
    import subprocess
    import threading
    from latte_amd._lib import check, load_library, ptr, stream_ptr
    lib = load_library()
    samples, stop = [], [False]

    def sampler():
        while not stop[0]:
            try:
                r = subprocess.run(["rocm-smi", "-c", "-P", "--json"], capture_output=True, text=True, timeout=5)
                samples.append((time.time(), r.stdout))
            except Exception:  # noqa: BLE001
                pass
            time.sleep(0.1)

    def smi_mean(a, b):
        return mean_power_clock(samples, a, b)

    def spin(launch, seconds, per_round):
        launch(per_round)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0, n = time.time(), 0
        e0.record()
        while time.time() - t0 < seconds:
            launch(per_round)
            n += per_round
            torch.cuda.synchronize()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n, t0, time.time()

    th = threading.Thread(target=sampler, daemon=True)
    th.start()
    out = {"socket_power_cap_w": 1400, "shader_clock_max_mhz": 2400}
    try:
        t = torch.full((x.shape[0],), 500, device=device, dtype=torch.int64)
        ms, a, b = spin(lambda n: [model(x, t) for _ in range(n)], 1.5, 5)
        pw, ck, ns = smi_mean(a, b)
        out["forward"] = {"ms": round(ms, 3), "socket_power_w": pw, "shader_clock_mhz": ck, "rocm_smi_samples": ns}
        M, N, K = 32768, 1152, 4608
        tdt = torch.float16 if dtype == "f16" else torch.bfloat16
        bias = torch.zeros(N, device=device)
        gate = torch.full((N,), 1e-3, device=device)
        res = torch.zeros(M, N, device=device)
        for pattern in ("random", "zeros"):
            if pattern == "random":
                A = torch.randn(M, K, device=device).to(tdt)
                W = (torch.randn(N, K, device=device) / K ** 0.5).to(tdt)
            else:
                A = torch.zeros(M, K, device=device, dtype=tdt)
                W = torch.zeros(N, K, device=device, dtype=tdt)

            def launch(n):
                for _ in range(n):
                    check(lib.latte_debug_gemm(ptr(A), ptr(W), ptr(bias), ptr(res), ptr(gate), M, N, K, 0, M, 2, 1 if dtype == "f16" else 0,
                                               0, stream_ptr()))
            ms, a, b = spin(launch, 1.0, 200)
            pw, ck, ns = smi_mean(a, b)
            tf = 2.0 * M * N * K / (ms * 1e-3) / 1e12
            out["fc2_standalone_" + pattern + "_operands"] = {"avg_launch_ms": round(ms, 4), "achieved": round(tf, 1), "unit": "TFLOP/s",
                                                               "frac": round(tf / MFMA_PEAK_TFLOPS, 4), "socket_power_w": pw,
                                                               "shader_clock_mhz": ck}
            del A, W
    finally:
        stop[0] = True
        th.join(timeout=10)
    return out


def energy_columns(res, world):
    """Energy beside time (round 5).  The MFMA kernels of the step run on the socket's power cap (power_check), where a launch is as
    fast as its ENERGY allows: joules = mean socket power of the forward (rocm-smi, random operands, sampled during ~1.5 s of
    back-to-back forwards) x time.  Per kernel class the forward-average power is applied to the class's average launch time -- an
    estimate (the classes draw 1300-1400 W each when run alone: tools/operand_power_probe.py), good to a few per cent; the step figure
    is the product of two quantities measured in this run.  pJ per algorithmic FLOP = joules per sample-step / 3.726 TFLOP."""
    pc = res.get("power_check") or {}
    fw = pc.get("forward") or {}
    pw = fw.get("socket_power_w")
    if not pw:
        return None
    B = res["config"]["per_gpu_batch"]
    for row in res.get("roofline_table", []):
        row["joules_per_launch"] = round(pw * row["avg_launch_ms"] * 1e-3, 4)
    j_step = pw * res["ms_per_step"] * 1e-3
    out = {"socket_power_w_forward": pw, "shader_clock_mhz_forward": fw.get("shader_clock_mhz"),
           "joules_per_step": round(j_step, 2), "joules_per_sample_step": round(j_step / B, 3),
           "picojoules_per_algorithmic_flop": round(j_step / B / FLOPS_PER_SAMPLE_STEP["Latte-XL/2"] * 1e12, 3),
           "joules_per_250_step_video": round(j_step / B * 250, 1),
           "source": "rocm-smi socket power during the power_check forward leg x this run's ms_per_step; per-class rows: the same power x avg_launch_ms"}
    for leg in ("fc2_standalone_random_operands", "fc2_standalone_zeros_operands"):
        if leg in pc and pc[leg].get("socket_power_w"):
            pc[leg]["joules_per_launch"] = round(pc[leg]["socket_power_w"] * pc[leg]["avg_launch_ms"] * 1e-3, 4)
    return out


def vae_decode_rate(device):
    """VAE decode of one 16-frame video (latents 16x4x32x32 -> 16x256x256x3 uint8), random sd-vae-ft-shaped weights;
    reported beside the headline, outside its timed region (decode is ~1.5 % of a 250-step chain)."""
    import latte_amd
    from latte_amd.random_init import vae_decoder_state_dict
    vae = latte_amd.AutoencoderKL(latent_size=32, max_frames=16, compute_dtype="f16")
    vae.load_state_dict(vae_decoder_state_dict(0))
    vae.to(device)
    lat = torch.randn(1, 16, 4, 32, 32, device=device) * 0.18215
    vae.decode_video_uint8(lat)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 3
    for _ in range(n):
        out = vae.decode_video_uint8(lat)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    # per kernel class, HIP events behind every launch of one decode (latte_vae_profile_decode): the convolutions against the MFMA
    # peak with the FLOPs of the layer shapes, the two GroupNorm passes against the HBM peak with their algorithmic bytes
    z = (lat[0] / 0.18215).contiguous()
    vae.profile_decode(z)
    prof = vae.profile_decode(z)
    conv_fl, gn_stats_b, gn_apply_b = (16 * v for v in vae_decoder_work(32))
    total = sum(v[0] for v in prof.values())
    table = []
    for k, (ms, n) in prof.items():
        row = {"class": k, "launches": n, "ms_per_video": round(ms, 3), "share_of_decode": round(ms / total, 4)}
        if k == "conv3x3":
            ach = conv_fl / (ms * 1e-3) / 1e12
            row.update({"bound": "mfma", "achieved": round(ach, 1), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": round(ach / MFMA_PEAK_TFLOPS, 4), "algorithmic_flops": conv_fl,
                        "algorithmic_bytes": vae_conv_bytes(32), "algorithmic_gbs": round(vae_conv_bytes(32) / (ms * 1e-3) / 1e9, 1),
                        "kernel": "conv3x3_kernel: implicit-GEMM 3x3 convolution (+ nearest-2x upsample in the gather), f16 operands"})
        elif k in ("groupnorm_stats", "groupnorm_apply"):
            by = gn_stats_b if k == "groupnorm_stats" else gn_apply_b
            ach = by / (ms * 1e-3) / 1e9
            row.update({"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                        "algorithmic_bytes": by,
                        "kernel": "gn_partial_kernel + gn_finalize_kernel: GroupNorm(32) statistics over the fp32 stream" if k == "groupnorm_stats"
                                  else "gn_apply_kernel: normalise + affine + SiLU, fp32 in, half operand out"})
        table.append(row)
    table.sort(key=lambda r: -r["share_of_decode"])
    return {"ms_per_video": round(dt * 1e3, 2), "frames_per_sec": round(16 / dt, 1), "dtype": "f16",
            "algorithmic_tflops_per_s": round(16 * vae_decoder_flops(32) / 1e12 / dt, 1),
            "algorithmic_tflop_per_frame": round(vae_decoder_flops(32) / 1e12, 4),
            "finite_and_nonconstant": bool(out.float().std() > 0),
            "roofline_table": table, "decode_ms_eager_events": round(total, 3)}


def config4_rate(device, n_steps=4):
    """BASELINE config 4 (configs/t2x/t2v_sample.yaml:24-26, sample/sample_t2x.py): Latte-1 text-to-video, 16 frames of 64x64
    latents (512 px), the classifier-free-guidance pair through LatteT2V inside the engine's guided DDIM loop
    (pipeline_latte.py:735-758), then the 16-frame AutoencoderKLTemporalDecoder decode in chunks of 14 + 2 frames
    (pipeline_latte.py:779-798).  Random weights of the real shapes, 120 T5 tokens (40 unmasked); f16 operands (the reference
    runs this model in fp16: sample_t2x.py:29).  Side measurement outside the timed region."""
    import latte_amd
    from latte_amd.random_init import t2v_state_dict, vae_temporal_decoder_state_dict
    from latte_amd.schedulers import DDIMScheduler
    m = latte_amd.LatteT2V(num_layers=28, compute_dtype="f16", max_batch=2).load_state_dict(t2v_state_dict(0, num_layers=28)).to(device)
    g = torch.Generator("cpu").manual_seed(4000)
    enc = torch.randn(2, 120, 4096, generator=g).to(device)
    mask = torch.ones(2, 120, device=device)
    mask[:, 40:] = 0
    sch = DDIMScheduler()
    sch.set_timesteps(50)
    ts = [int(v) for v in sch.timesteps[:n_steps]]
    ratio = sch.num_train_timesteps // sch.num_inference_steps
    at = [float(sch.alphas_cumprod[v]) for v in ts]
    ap = [float(sch.alphas_cumprod[v - ratio]) if v >= ratio else 1.0 for v in ts]
    lat = torch.randn(1, 4, 16, 64, 64, generator=g).to(device)
    m.set_text(enc, mask)
    m.guided_ddim_loop(lat, ts[:2], at[:2], ap[:2], 7.5)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = m.guided_ddim_loop(lat, ts, at, ap, 7.5)
    torch.cuda.synchronize()
    dstep = (time.perf_counter() - t0) / n_steps
    D, T, F, L = 1152, 1024, 16, 28
    M = 2 * F * T
    lin = 2.0 * M * D * (3 * D + D + 4 * D + 4 * D) * 2 * L + 2.0 * M * D * (D + D) * L      # self blocks (x2) + cross q / out
    attn = L * (4.0 * 2 * F * T * T * D + 4.0 * 2 * T * F * F * D + 4.0 * 2 * F * T * 120 * D)
    res = {"workload": "Latte-1 T2V 512x512x16 (16 x 4 x 64x64 latents, 28 + 28 blocks, guidance pair, 120 T5 tokens), guided DDIM "
                       "step inside the engine + AutoencoderKLTemporalDecoder decode (14 + 2 frame chunks), random weights, f16 operands",
           "guided_step_ms": round(dstep * 1e3, 2), "steps": n_steps, "value": round(1.0 / dstep, 3), "unit": "guided steps/s",
           "seconds_for_50_steps": round(50 * dstep, 2),
           "algorithmic_tflops_per_s": round((lin + attn) / dstep / 1e12, 1),
           "model_mfma_frac": round((lin + attn) / dstep / 1e12 / MFMA_PEAK_TFLOPS, 4),
           "finite": bool(torch.isfinite(out).all())}
    del m
    torch.cuda.empty_cache()
    vae = latte_amd.AutoencoderKLTemporalDecoder(latent_size=64, max_frames=14)
    vae.load_state_dict(vae_temporal_decoder_state_dict(0))
    vae.to(device)
    z = torch.randn(16, 4, 64, 64, generator=g).to(device)     # latents / scaling_factor (pipeline_latte.py:786)

    def decode():
        return [vae.decode(z[i:i + 14].contiguous(), num_frames=min(14, 16 - i)).sample for i in range(0, 16, 14)]
    decode()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    vid = decode()
    torch.cuda.synchronize()
    dd = time.perf_counter() - t0
    tot_fl, mfma_fl = (16 * v for v in vae_temporal_decoder_flops(64))
    res["temporal_decoder"] = {"ms_per_16_frame_video": round(dd * 1e3, 2), "frames_per_sec": round(16 / dd, 1),
                               "finite": bool(all(torch.isfinite(v).all() for v in vid)),
                               "algorithmic_tflop_per_video": round(tot_fl / 1e12, 2),
                               "algorithmic_tflops_per_s": round(tot_fl / 1e12 / dd, 1),
                               "model_mfma_frac": round(tot_fl / 1e12 / dd / MFMA_PEAK_TFLOPS, 4)}
    # per kernel class (HIP events behind every launch, latte_vae_profile_decode) for the 14-frame chunk, scaled to what it is: the convolutions
    # against the MFMA peak with the layer shapes' FLOPs counted ONCE (the engine's split-operand passes for parity are cost, not work)
    z14 = z[:14].contiguous()
    vae2 = latte_amd.AutoencoderKLTemporalDecoder(latent_size=64, max_frames=14)
    vae2.load_state_dict(vae_temporal_decoder_state_dict(0))
    vae2.to(device)
    vae2.decode(z14, num_frames=14)
    vae2.profile_decode(z14)
    prof = vae2.profile_decode(z14)
    tot_ms = sum(v[0] for v in prof.values())
    table = []
    for k, (ms, n) in prof.items():
        row = {"class": k, "launches": n, "ms_per_14_frame_chunk": round(ms, 3), "share_of_decode": round(ms / max(tot_ms, 1e-9), 4)}
        if k == "conv3x3" and ms > 0:
            ach = 14 * vae_temporal_decoder_flops(64)[1] / (ms * 1e-3) / 1e12
            row.update({"bound": "mfma", "achieved": round(ach, 1), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / MFMA_PEAK_TFLOPS, 4),
                        "algorithmic_flops": 14 * vae_temporal_decoder_flops(64)[1],
                        "kernel": "conv3x3_pp_kernel / conv3x3_kernel: implicit-GEMM 3x3 and (3,1,1) convolutions; the split-operand passes of the default "
                                  "mask 0x319c03 (csrc/vae_engine.cpp: vae_split_mask -- activation and weight f16 rounding residuals on the mid block, "
                                  "up block 0 and its upsampler) are cost, not counted work"})
        table.append(row)
    table.sort(key=lambda r: -r["share_of_decode"])
    res["temporal_decoder"]["roofline_table"] = table
    res["temporal_decoder"]["chunk14_ms_eager_events"] = round(tot_ms, 3)
    del vae2
    res["seconds_per_video_50_steps_plus_decode"] = round(50 * dstep + dd, 2)
    del vae
    torch.cuda.empty_cache()
    return res


def note(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks (one per GPU) under torch.distributed.run."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    note(f"--gpus {n} without a launcher: re-executing as {' '.join(cmd[1:9])} ...")
    return subprocess.call(cmd)


def launch_check(args, world, rank, backend):
    """--launch-check: the rendezvous + collective of the N-rank launch and nothing else (no `value` is printed)."""
    ranks, total = 1, rank
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        cuda = backend == "nccl"
        if cuda:
            assert torch.cuda.is_available() and world <= torch.cuda.device_count(), "RCCL: one rank per GPU"
            torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        dist.init_process_group(backend)
        v = torch.tensor([float(rank)], device="cuda" if cuda else "cpu")
        dist.all_reduce(v)
        dist.barrier()
        ranks, total = dist.get_world_size(), int(v.item())
        dist.destroy_process_group()
    assert total == world * (world - 1) // 2
    if rank == 0:
        print(json.dumps({"launch_check": True, "n_gpus": world, "collective_ranks": ranks,
                          "collective_backend": "rccl" if backend == "nccl" else backend}))


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, f"--gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks"
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("LATTE_BENCH_BACKEND", "nccl")
    if args.launch_check:
        return launch_check(args, world, rank, backend)
    assert torch.cuda.is_available(), "bench.py needs the MI355X"
    # "nccl" is RCCL on ROCm: one rank per device.  LATTE_BENCH_BACKEND=gloo exists only to exercise the N > 1 code path on a
    # box with fewer GPUs than ranks (RCCL refuses two ranks on one device); it is never used for reported numbers.
    assert backend != "nccl" or world <= torch.cuda.device_count(), \
        f"--gpus {world} needs {world} devices (RCCL: one rank per GPU), {torch.cuda.device_count()} visible"
    local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)

    import latte_amd
    from latte_amd._lib import load_library
    lib = load_library()
    note('building model')
    model = build_model(device, args.dtype, args.batch)
    note('model on device')
    if args.gemm_variant:
        model.set_engine_option("gemm_variant", args.gemm_variant, args.batch)
    for opt in args.engine_option:
        k, v = opt.split("=")
        model.set_engine_option(k, int(v), args.batch)
    diffusion = latte_amd.create_diffusion("250")
    B = args.batch
    if dist is not None:
        # rank 0's timestep-embedding table [250, 1152] fp32 to every rank over RCCL/xGMI (outside the timed region)
        from latte_amd import parallel
        parallel.broadcast_temb_table(model, diffusion, batch=B)
    g = torch.Generator("cpu").manual_seed(1000 + rank)
    x = torch.randn(B, 16, 4, 32, 32, generator=g).to(device)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    run_steps(lib, model, diffusion, x.clone(), max(args.warmup, 1), args.method, B)
    note('warm-up done')
    xx = x.clone()
    barrier()
    t0 = time.perf_counter()
    run_steps(lib, model, diffusion, xx, args.steps, args.method, B)
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    finite = bool(torch.isfinite(xx).all())
    note(f'timed region done: {elapsed:.3f}s')

    other = None
    if world == 1:
        # BASELINE.json configs[1] words the same FaceForensics job as DDPM-250 (the YAML default `sample_method: 'ddpm'`),
        # the metric as DDIM-250: same model cost, DDPM adds one engine-side noise fill per step.  Short side measurement.
        om = "ddpm" if args.method == "ddim" else "ddim"
        n_other = min(args.steps, 60)
        xo = x.clone()
        run_steps(lib, model, diffusion, xo, 2, om, B)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        run_steps(lib, model, diffusion, xo, n_other, om, B)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t1
        other = {"method": om, "value": round(B * n_other / dt, 3), "unit": "sample-steps/s", "steps": n_other,
                 "finite": bool(torch.isfinite(xo).all())}
    side = {}
    if not args.no_side:
        # the other operand type: same model, same loop, short run
        n_side = min(args.steps, 20)
        other_dt = "f16" if args.dtype == "bf16" else "bf16"
        model.to(dtype=torch.float16 if other_dt == "f16" else torch.bfloat16)
        dt16 = timed_steps(lib, model, diffusion, x.clone(), n_side, args.method, B)
        model.to(dtype=torch.float16 if args.dtype == "f16" else torch.bfloat16)
        side[other_dt] = {"ms_per_step": dt16 * 1e3, "steps": n_side}
        # small batches on the same engine (B = 1 is the reference's sample.py case): fewer tiles than CUs in the N = 1152 GEMMs
        for bs in (2, 1):
            if bs < B:
                side[f"batch{bs}"] = {"ms_per_step": timed_steps(lib, model, diffusion, x[:bs].clone(), n_side, args.method, bs) * 1e3,
                                      "steps": n_side, "batch": bs}
        # BASELINE config 3's per-GPU share: class-conditional (UCF101: 101 classes), CFG 7.0, 8 samples = 16 sequences
        m3 = build_model(device, None, 16, extras=2, num_classes=101)     # operand rule: f16
        gq = torch.Generator("cpu").manual_seed(2000 + rank)
        z3 = torch.randn(8, 16, 4, 32, 32, generator=gq)
        y3 = torch.cat([torch.randint(0, 101, (8,), generator=gq), torch.full((8,), 101)]).to(device)
        x3 = torch.cat([z3, z3]).to(device).contiguous()
        n3 = min(args.steps, 10)
        dt3 = timed_steps(lib, m3, diffusion, x3, n3, args.method, 16, y=y3, cfg_scale=7.0, guided=True)
        side["config3"] = {"ms_per_step": dt3 * 1e3, "steps": n3}
        # the guided call's split-operand linears (engine option guided_split, default 20 = fp8 / fp4 remainders: DESIGN.md section 2) cost
        # time; beside the default: the plain f16 operands of rounds 1-4 (0: AT 1e-3 of the fp32 reference at trained-scale gates, not
        # under it), the attention output's remainder only (4) and round 5's f16 pairs (3)
        for gs in (0, 4, 12, 3):
            m3.set_engine_option("guided_split", gs, 16, guided=True)
            side["config3"][f"ms_per_step_guided_split_{gs}"] = timed_steps(lib, m3, diffusion, x3, n3, args.method, 16, y=y3, cfg_scale=7.0,
                                                                            guided=True) * 1e3
        m3.set_engine_option("guided_split", 20, 16, guided=True)
        # the per-kernel roofline table of the GUIDED call at config 3's M = 65 536 rows (16 sequences), split operands included: the
        # algorithmic FLOPs of a class are those of the plain product, so the correction pass shows as time, not as work
        t3 = torch.full((16,), 500, device=device, dtype=torch.int64)
        m3.profile_forward(x3, t3, y=y3, guided=True)
        tab3, tot3 = roofline_table(m3.profile_forward(x3, t3, y=y3, guided=True), 16, "f16")
        side["config3"]["roofline_table"] = [{k: r[k] for k in ("class", "launches_per_forward", "avg_launch_ms", "share_of_forward", "bound",
                                                               "achieved", "unit", "frac") if k in r} for r in tab3]
        side["config3"]["forward_ms_eager_events"] = round(tot3, 3)
        del m3
        torch.cuda.empty_cache()
        # BASELINE config 5's per-GPU share: one optimisation step (train.py:197-236) of Latte-B/2, local batch 5
        tb = 5
        m5 = latte_amd.Latte_models["Latte-B/2"](input_size=32, num_frames=16, extras=1, max_batch=tb).to(device)
        with torch.no_grad():
            for p_ in m5.parameters():
                if p_.requires_grad and float(p_.abs().max()) == 0.0:
                    p_.normal_(0, 0.02)
        trn = latte_amd.LatteTrainer(m5, latte_amd.create_diffusion(""), max_batch=tb)
        x5 = torch.randn(tb, 16, 4, 32, 32, generator=torch.Generator("cpu").manual_seed(3000 + rank)).to(device)
        for _ in range(2):
            trn.train_step(x5)
        torch.cuda.synchronize()
        n5 = min(args.steps, 10)
        t5 = time.perf_counter()
        for _ in range(n5):
            o5 = trn.train_step(x5)
        torch.cuda.synchronize()
        side["config5"] = {"ms_per_step": (time.perf_counter() - t5) / n5 * 1e3, "steps": n5,
                           "loss_finite": bool(torch.isfinite(o5["loss"]).all())}
        # the same step with the small-kernel consolidation of round 6b off (trainer option fuse_small = 0: the separate finalize /
        # column-sum / adaLN / gate-backward launches of rounds 2 - 6), interleaved on this box
        trn.set_option("fuse_small", 0)
        for _ in range(2):
            trn.train_step(x5)
        torch.cuda.synchronize()
        t5 = time.perf_counter()
        for _ in range(n5):
            trn.train_step(x5)
        torch.cuda.synchronize()
        side["config5"]["ms_per_step_separate_launches"] = (time.perf_counter() - t5) / n5 * 1e3
        del trn, m5
        torch.cuda.empty_cache()
        if dist is not None:   # max over ranks of every side timing
            keys = sorted(side)
            tt = torch.tensor([side[k]["ms_per_step"] for k in keys], device=device, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            for k, v in zip(keys, tt.tolist()):
                side[k]["ms_per_step"] = v
    if rank == 0:
        value = world * B * args.steps / elapsed
        flops = FLOPS_PER_SAMPLE_STEP["Latte-XL/2"]
        # rooflines, timed live with HIP events on the launch stream
        t = torch.full((B,), 500, device=device, dtype=torch.int64)
        model.profile_forward(x, t)
        prof = model.profile_forward(x, t)
        table, total_ms = roofline_table(prof, B, args.dtype)
        dom = table[0]                                   # the dominant kernel = largest share of the forward
        res = {
            "metric": "denoising steps/sec", "value": round(value, 3), "unit": "sample-steps/s",
            "n_gpus": world, "collective_ranks": dist.get_world_size() if dist is not None else 1,
            "collective_backend": (("rccl" if backend == "nccl" else backend) if dist is not None else None), "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "Latte-XL/2 FaceForensics (uncond) 16x256x256 -> latents 16x4x32x32, "
                                   f"{args.method.upper()} on the '250' respacing, random-init weights",
                       "per_gpu_batch": B, "global_batch": B * world, "parallelism": f"dp{world} (independent samples)"},
            "other_sampler": other,
            "latent_frames_per_sec": round(world * B * 16 / (elapsed / args.steps * 250), 3),
            "model_mfma_frac": round(value / world * flops / (MFMA_PEAK_TFLOPS * 1e12), 4),
            "finite": finite,
            "roofline": {"bound": dom["bound"], "kernel": dom["kernel"], "achieved": dom["achieved"], "peak": dom["peak"],
                         "unit": dom["unit"], "frac": dom["frac"], "traffic": dom["traffic"], "traffic_source": dom["traffic_source"],
                         "avg_launch_ms": dom["avg_launch_ms"], "share_of_forward": dom["share_of_forward"]},
            "roofline_table": table,
            "kernel_ms_per_forward": {k: round(v[0], 4) for k, v in prof.items()},
            "forward_ms_eager_events": round(total_ms, 4),
        }
        for k, v in side.items():
            if k == "config3":
                sps = world * 8 / (v["ms_per_step"] * 1e-3)
                res["config3"] = {"workload": "Latte-XL/2 UCF101 class-conditional (101 classes + null), CFG 7.0 through "
                                              "forward_with_cfg, 8 samples = 16 sequences per GPU, f16 operands, attention output and fc1 "
                                              "operand carried as f16 + a low-precision remainder (fp8 / fp4 with a per-row scale) whose product runs on the "
                                              "block-scaled MFMA inside the same GEMM launch (guided_split 20, the default of guided calls)",
                                  "value": round(sps, 3), "unit": "guided sample-steps/s", "ms_per_step": round(v["ms_per_step"], 3),
                                  "other_guided_split_settings": {
                                      f"guided_split_{gs}": {"value": round(world * 8 / (v[f"ms_per_step_guided_split_{gs}"] * 1e-3), 3),
                                                             "ms_per_step": round(v[f"ms_per_step_guided_split_{gs}"], 3),
                                                             "note": {0: "plain f16 operands (rounds 1-4): the guided XL/2 output at trained-scale gates is AT "
                                                                         "1e-3 of the fp32 reference, 0.61-1.07e-3 over 12 draws",
                                                                      4: "fp8 remainder of the attention output only: 0.51-0.87e-3 over the 12 draws",
                                                                      12: "both remainders as fp8 (the first form of round 6): 0.43-0.72e-3",
                                                                      3: "round 5's default: both operands as f16 pairs [hi | lo] . [W | W]: 0.43-0.72e-3, "
                                                                         "the parity of the low-precision forms at two to three times their cost"}[gs]}
                                      for gs in (0, 4, 12, 3) if f"ms_per_step_guided_split_{gs}" in v},
                                  "parity_of_the_default": "0.44-0.73e-3 (eps channels 0.44-0.79e-3) over gate_std {0.3, 1.0} x t {999, 500, 50} x 2 seeds "
                                                           "against the fp32 oracle (profiles/r6_gate_parity_guided.json)",
                                  "steps": v["steps"], "global_batch": 8 * world,
                                  "model_mfma_frac": round(2 * sps / world * flops / (MFMA_PEAK_TFLOPS * 1e12), 4),
                                  "roofline_table": v.get("roofline_table"), "forward_ms_eager_events": v.get("forward_ms_eager_events")}
            elif k == "config5":
                tb, D5, dep5, M5 = 5, 768, 12, 5 * 16 * 256
                fwd5 = dep5 * 2.0 * M5 * 12 * D5 * D5 + (dep5 // 2) * (4.0 * tb * 16 * 256 * 256 * D5 + 4.0 * tb * 256 * 16 * 16 * D5)
                res["config5"] = {"workload": "train.py step, Latte-B/2 16x256x256 synthetic latents, local batch 5, f16 operands (loss-scaled backward: the "
                                              "mantissa of the reference's TF32 matmuls) / fp32 masters: forward + MSE/VB loss + backward + grad all-reduce + clip + AdamW + EMA",
                                  "value": round(world * tb / (v["ms_per_step"] * 1e-3), 3), "unit": "training samples/s",
                                  "ms_per_step": round(v["ms_per_step"], 3), "steps": v["steps"], "global_batch": tb * world,
                                  "algorithmic_tflops_per_gpu": round(3 * fwd5 / (v["ms_per_step"] * 1e-3) / 1e12, 1),
                                  "loss_finite": v.get("loss_finite"),
                                  "fuse_small_0": {"ms_per_step": round(v.get("ms_per_step_separate_launches", 0.0), 3),
                                                   "note": "trainer option fuse_small = 0 (the separate finalize / column-sum / adaLN / gate-backward "
                                                           "launches of rounds 2 - 6) on the same box; the default folds them (latte_amd/csrc/train_fin.hip)"}}
            else:
                res[k] = {"value": round(world * v.get("batch", B) / (v["ms_per_step"] * 1e-3), 3), "unit": "sample-steps/s",
                          "ms_per_step": round(v["ms_per_step"], 4), "steps": v["steps"]}
                if "batch" in v:
                    res[k]["per_gpu_batch"] = v["batch"]
        if world == 1 and not args.no_side:
            note('power check (socket power / shader clock in the forward; dominant GEMM on quiet operands)')
            try:
                res["power_check"] = power_check(device, model, x, args.dtype)
            except Exception as ex:  # noqa: BLE001  (rocm-smi absent or unreadable: the line must still print)
                res["power_check"] = {"error": repr(ex)[:200]}
            res["energy"] = energy_columns(res, world)
        if world == 1 and not args.no_vae:
            res["vae_decode"] = vae_decode_rate(device)
        if world == 1 and not args.no_side:
            note('config 4 (Latte-1 T2V guided step + temporal decoder)')
            res["config4"] = config4_rate(device)
        if world == 1 and not args.no_cpu_baseline:
            note('cpu baseline (oracle on host cores)')
            res["cpu_baseline"] = cpu_baseline(args.cpu_steps)
            note('cpu baseline done')
        print(json.dumps(res))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
