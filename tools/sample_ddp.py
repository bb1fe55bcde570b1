"""sample/sample_ddp.py of the reference, re-hosted on the MI355X engine.

  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 tools/sample_ddp.py \
      --config /path/to/configs/ucf101/ucf101_sample.yaml [--ckpt model.pt] [--num-samples 64] [--out DIR]

One process per GPU, weights replicated, samples sharded by global index (latte_amd.parallel.plan_shards =
sample_ddp.py:116-176); collectives: start / end barrier and ONE RCCL broadcast of the timestep-embedding table.
Without --ckpt / a VAE directory the weights are random (there are no checkpoints offline): the script then
measures and checks plumbing, not picture quality.  Videos are written as {index:04d}.mp4 like the reference's
(sample_ddp.py:174-176; Motion-JPEG samples, latte_amd.video_io -- imageio / H.264 do not exist offline), as lossless
uncompressed .avi, or as uint8 .npy [F, H, W, 3].
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import latte_amd  # noqa: E402
from latte_amd import parallel  # noqa: E402


def randomise_zero_init(model, seed=1):
    g = torch.Generator("cpu").manual_seed(seed)
    with torch.no_grad():
        for _, p in model.named_parameters():
            if p.requires_grad and float(p.detach().abs().max()) == 0.0:
                p.copy_(torch.randn(p.shape, generator=g) * 0.02)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", required=True)
    ap.add_argument("--ckpt", default=None)
    ap.add_argument("--vae", default=None, help="directory with the sd-vae weights (diffusers layout); random if absent")
    ap.add_argument("--num-samples", type=int, default=None)
    ap.add_argument("--steps", type=int, default=None, help="override num_sampling_steps")
    ap.add_argument("--out", default=None)
    ap.add_argument("--no-decode", action="store_true")
    ap.add_argument("--format", choices=["mp4", "avi", "npy"], default="mp4")   # mp4: the reference's {index:04d}.mp4 (:174-176)
    a = ap.parse_args()
    args = latte_amd.load_config(a.config)
    torch.set_grad_enabled(False)
    rank, world, local = parallel.setup_distributed()
    assert torch.cuda.is_available(), "sample_ddp needs MI355X GPUs"
    device = torch.device("cuda", local)
    torch.cuda.set_device(device)
    seed = int(args.get("seed") or 0)

    args.latent_size = args.image_size // 8
    n = int(args.per_proc_batch_size)
    using_cfg = float(args.cfg_scale) > 1.0
    args.max_batch = 2 * n if using_cfg else n
    # The reference seeds torch with seed * world + rank before building anything (sample_ddp.py:63-65), which ties the
    # draws to the world size; here every rank seeds identically (random-weight plumbing runs then hold the SAME replica on
    # every GPU) and the per-sample noise / labels come from the global sample index (parallel.sample_noise).
    torch.manual_seed(seed)
    model = latte_amd.get_models(args)
    if a.ckpt or args.get("ckpt"):
        model.load_state_dict(latte_amd.find_model(a.ckpt or args.ckpt))
    else:
        randomise_zero_init(model)
    model = model.to(device).eval()
    if args.get("use_fp16"):
        model.to(dtype=torch.float16)                              # sample.py:72-75 / sample_ddp.py use_fp16
    diffusion = latte_amd.create_diffusion(str(a.steps or args.num_sampling_steps))
    vae = None
    if not a.no_decode:
        if a.vae:
            vae = latte_amd.AutoencoderKL.from_pretrained(a.vae, latent_size=args.latent_size, max_frames=args.num_frames)
        else:
            from latte_amd.random_init import vae_decoder_state_dict   # random decoder weights (no checkpoint offline)
            vae = latte_amd.AutoencoderKL(latent_size=args.latent_size, max_frames=args.num_frames)
            vae.load_state_dict(vae_decoder_state_dict(0))
        vae.to(device)

    out_dir = a.out or args.get("save_video_path") or "./sample_videos"
    if rank == 0:
        os.makedirs(out_dir, exist_ok=True)
    parallel.barrier()                                             # sample_ddp.py:113
    parallel.broadcast_temb_table(model, diffusion, batch=args.max_batch, guided=using_cfg)   # the one payload collective

    num = a.num_samples or int(args.get("num_fvd_samples") or n * world)
    total, iterations, lists = parallel.plan_shards(num, n, rank, world)
    shape = (args.num_frames, 4, args.latent_size, args.latent_size)
    t0 = time.time()
    for it in range(iterations):
        idx = lists[it]
        z = torch.stack([parallel.sample_noise(i, shape, seed, device) for i in idx])
        if using_cfg:
            z = torch.cat([z, z], 0)
            y = torch.tensor([parallel.sample_label(i, args.num_classes, seed) for i in idx] + [args.num_classes] * n,
                             device=device)
            kw = dict(y=y, cfg_scale=float(args.cfg_scale))
            fn = model.forward_with_cfg
        else:
            kw = dict(y=None)
            fn = model.forward
        loop = diffusion.ddim_sample_loop if args.sample_method == "ddim" else diffusion.p_sample_loop
        samples = loop(fn, z.shape, z, clip_denoised=False, model_kwargs=kw, progress=False, device=device)
        if using_cfg:
            samples, _ = samples.chunk(2, dim=0)                   # sample_ddp.py:159-160
        if vae is not None:
            video = vae.decode_video_uint8(samples)                # decode(z / 0.18215) + uint8, sample_ddp.py:165-172
            for j, i in enumerate(idx):
                if a.format == "mp4":
                    latte_amd.write_mp4(os.path.join(out_dir, f"{i:04d}.mp4"), video[j], fps=8)   # sample_ddp.py:174-176
                elif a.format == "avi":
                    latte_amd.write_avi(os.path.join(out_dir, f"{i:04d}.avi"), video[j], fps=8)   # lossless alternative
                else:
                    np.save(os.path.join(out_dir, f"{i:04d}.npy"), video[j].cpu().numpy())
        else:
            for j, i in enumerate(idx):
                np.save(os.path.join(out_dir, f"{i:04d}_latent.npy"), samples[j].cpu().numpy())
    torch.cuda.synchronize()
    parallel.barrier()                                             # sample_ddp.py:180
    if rank == 0:
        dt = time.time() - t0
        print(f"{total} samples on {world} GPU(s) in {dt:.1f} s -> {total * diffusion.num_timesteps / dt:.1f} sample-steps/s")
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
