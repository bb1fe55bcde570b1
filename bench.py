"""bench.py — denoising steps/sec of the Latte-XL/2 16x256x256 sampling loop on MI355X.

  python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run)

One "step" = one pass of the hot path (denoiser forward + sampler update) over the rank's batch of
latents, inputs resident in HBM.  Workload = BASELINE.json configs[1]: Latte-XL/2, FaceForensics
(unconditional) config, 16 frames of 32x32 latents, per-GPU batch 8 (= BASELINE config 3's per-GPU share
of its batch 64; `--batch 2` is the reference YAML's per_proc_batch_size), DDIM eta=0 over the "250"
respacing, random-init weights (adaLN/final layers re-drawn N(0, 0.02) so the network is not the
identity), synthetic N(0,1) latents.  Weak scaling: every rank runs its own samples, no data-path
collective (the reference's sample_ddp.py has none either); the only payload collective is the one-off
RCCL broadcast of the timestep-embedding table before the timed region.
`value` = aggregate denoising sample-steps/s = n_gpus * batch * K / max-over-ranks time.
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


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=250)
    p.add_argument("--warmup", type=int, default=5)
    p.add_argument("--batch", type=int, default=8, help="samples per GPU (8 = config 3's share; 2 = ffs_sample.yaml)")
    p.add_argument("--dtype", default="bf16", choices=["bf16", "f16"])
    p.add_argument("--method", default="ddim", choices=["ddim", "ddpm"])
    p.add_argument("--gemm-variant", type=int, default=0)
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-vae", action="store_true", help="skip the (untimed-region) VAE decode rate report")
    p.add_argument("--cpu-forwards", type=int, default=2)
    return p.parse_args()


def build_model(args, device):
    import latte_amd
    torch.manual_seed(0)
    m = latte_amd.Latte_models["Latte-XL/2"](input_size=32, num_frames=16, extras=1, learn_sigma=True,
                                             compute_dtype=args.dtype, max_batch=args.batch)
    g = torch.Generator("cpu").manual_seed(1)
    with torch.no_grad():
        for _, prm in m.named_parameters():
            if prm.requires_grad and float(prm.detach().abs().max()) == 0.0:
                prm.copy_(torch.randn(prm.shape, generator=g) * 0.02)
    return m.to(device).eval()


def run_steps(lib, model, diffusion, x, n_steps, method, batch):
    """Exactly n_steps denoising steps: chains of up to num_timesteps steps through the fused loop."""
    from latte_amd._lib import check, ptr, stream_ptr
    eng = model.engine(batch)
    T = diffusion.num_timesteps
    left = n_steps
    mi = 1 if method == "ddim" else 0
    while left > 0:
        seg = min(left, T)
        check(lib.latte_sample_loop(eng, diffusion._h, mi, 0.0, 0, 1.0, ptr(x), None, batch, T - 1, T - seg, None,
                                    None, None, stream_ptr()))
        left -= seg


def cpu_baseline(n_forwards):
    """The oracle (a CPU port of the reference forward, bit-identical to it: oracle/VALIDATION.md) timed
    on this box's host cores — reported beside the GPU number, never the thing measured."""
    from oracle import latte_oracle as lo
    cfg = lo.preset_config("Latte-XL/2", input_size=32, num_frames=16, extras=1)
    sd = lo.init_state_dict(cfg, seed=0)
    x = torch.randn(1, 16, 4, 32, 32)
    t = torch.tensor([500])
    # MKL/OpenMP GEMMs stop scaling (and can collapse) far below the 256 hardware threads of the GPU box
    cores = min(len(os.sched_getaffinity(0)), 32)
    torch.set_num_threads(cores)
    with torch.no_grad():
        lo.latte_forward(sd, cfg, x, t)                       # warm-up
        t0 = time.time()
        for _ in range(n_forwards):
            lo.latte_forward(sd, cfg, x, t)
        dt = (time.time() - t0) / n_forwards
    return {"value": round(1.0 / dt, 4), "unit": "denoising sample-steps/s", "cores": torch.get_num_threads(),
            "kind": "port",
            "sample": f"{n_forwards} timed fp32 forwards of Latte-XL/2 (B=1, 16x32x32 latents) after 1 warm-up; "
                      "the sampler update is negligible on CPU (<0.1%)"}


def pmc_traffic(kernel_class, M):
    """HBM bytes per launch of the dominant GEMM from the rocprofv3 PMC passes committed under profiles/
    (FETCH_SIZE doubled per the guide's gfx950 correction + WRITE_SIZE); None when that shape was not profiled."""
    path = os.path.join(ROOT, "profiles", "r1_gemm_pmc.json")
    if not os.path.exists(path):
        return None
    with open(path) as f:
        tab = json.load(f)
    rec = tab.get(f"{kernel_class}:M={M}")
    return rec["hbm_bytes_per_launch"] if rec else None


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
    return {"ms_per_video": round(dt * 1e3, 2), "frames_per_sec": round(16 / dt, 1), "dtype": "f16",
            "algorithmic_tflops_per_s": round(16 * 0.62 / dt, 1), "finite_and_nonconstant": bool(out.float().std() > 0)}


def note(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs the MI355X"
    local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # "nccl" is RCCL on ROCm.  LATTE_BENCH_BACKEND=gloo exists only to exercise the N > 1 code path on a box
        # with fewer GPUs than ranks (RCCL refuses two ranks on one device); it is never used for reported numbers.
        backend = os.environ.get("LATTE_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)

    import latte_amd
    from latte_amd._lib import load_library
    lib = load_library()
    note('building model')
    model = build_model(args, device)
    note('model on device')
    if args.gemm_variant:
        model.set_engine_option("gemm_variant", args.gemm_variant, args.batch)
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
    if rank == 0:
        value = world * B * args.steps / elapsed
        flops = FLOPS_PER_SAMPLE_STEP["Latte-XL/2"]
        # roofline of the dominant kernel, timed live with HIP events on the launch stream
        t = torch.full((B,), 500, device=device, dtype=torch.int64)
        model.profile_forward(x, t)
        prof = model.profile_forward(x, t)
        total_ms = sum(v[0] for v in prof.values())
        D, Hm, M = 1152, 4608, B * 16 * 256
        gemm_flops = {"gemm_qkv": 2.0 * M * 3 * D * D, "gemm_proj": 2.0 * M * D * D, "gemm_fc1": 2.0 * M * Hm * D,
                      "gemm_fc2": 2.0 * M * D * Hm}
        # dominant kernel for the roofline object: the fc1 GEMM (largest single-shape kernel; fc2 and proj share one
        # template instantiation, so their rocprof averages are mixed -- DESIGN.md section 5)
        dom = "gemm_fc1"
        avg_ms = prof[dom][0] / max(prof[dom][1], 1)
        achieved = gemm_flops[dom] / (avg_ms * 1e-3) / 1e12
        res = {
            "metric": "denoising steps/sec", "value": round(value, 3), "unit": "sample-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "Latte-XL/2 FaceForensics (uncond) 16x256x256 -> latents 16x4x32x32, "
                                   f"{args.method.upper()} on the '250' respacing, random-init weights",
                       "per_gpu_batch": B, "global_batch": B * world, "parallelism": f"dp{world} (independent samples)"},
            "other_sampler": other,
            "latent_frames_per_sec": round(world * B * 16 / (elapsed / args.steps * 250), 3),
            "model_mfma_frac": round(value / world * flops / (MFMA_PEAK_TFLOPS * 1e12), 4),
            "finite": finite,
            "roofline": {"bound": "mfma", "kernel": f"gemm_pps_kernel ({dom}: M={M} N={Hm} K={D}, bias+GELU epilogue)", "achieved": round(achieved, 1),
                         "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / MFMA_PEAK_TFLOPS, 4),
                         "traffic": pmc_traffic(dom, M), "avg_launch_ms": round(avg_ms, 4)},
            "kernel_ms_per_forward": {k: round(v[0], 4) for k, v in prof.items()},
            "forward_ms_eager_events": round(total_ms, 4),
        }
        if world == 1 and not args.no_vae:
            res["vae_decode"] = vae_decode_rate(device)
        if world == 1 and not args.no_cpu_baseline:
            note('cpu baseline (oracle on host cores)')
            res["cpu_baseline"] = cpu_baseline(args.cpu_forwards)
            note('cpu baseline done')
        print(json.dumps(res))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
