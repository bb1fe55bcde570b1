"""sample/sample.py of the reference on the MI355X engine: single process, one video.

  python tools/sample.py --config configs/ffs_sample.yaml [--ckpt model.pt] [--vae DIR] [--save_video_path DIR]

The flow of /root/reference/sample/sample.py:39-126 statement by statement, with its three import lines swapped for
latte_amd: config -> get_models -> find_model / load_state_dict -> create_diffusion ->
AutoencoderKL -> z (doubled with the null class under guidance, :86-98) -> p_sample_loop / ddim_sample_loop (:100-107) ->
vae.decode(z / 0.18215) (:113-115) -> uint8 video (:122) -> file.  Offline there are no checkpoints: without --ckpt the
zero-initialised adaLN / final layers are re-drawn so the run exercises the whole path (plumbing, not picture quality), and
the video is written as sample.mp4 like the reference's (:124-126; Motion-JPEG samples instead of imageio's H.264, see
latte_amd/video_io.py)."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import latte_amd  # noqa: E402
from latte_amd import create_diffusion, find_model, get_models  # noqa: E402


def main(args, cli):
    torch.manual_seed(int(args.get("seed") or 0))                          # sample.py:41
    torch.set_grad_enabled(False)                                          # :42
    device = "cuda"
    if not torch.cuda.is_available():
        raise SystemExit("tools/sample.py needs an MI355X (latte_amd has no CPU fallback)")
    using_cfg = float(args.cfg_scale) > 1.0                                # :51
    args.latent_size = args.image_size // 8                                # :55
    args.max_batch = 2 if using_cfg else 1
    model = get_models(args).to(device)                                    # :56
    if args.get("ckpt"):
        model.load_state_dict(find_model(args.ckpt))                       # :62-64
    else:
        g = torch.Generator("cpu").manual_seed(1)
        for _, p in model.named_parameters():
            if p.requires_grad and float(p.detach().abs().max()) == 0.0:
                p.copy_(torch.randn(p.shape, generator=g) * 0.02)
        model.mark_weights_dirty()
    model.eval()                                                           # :66
    diffusion = create_diffusion(str(args.num_sampling_steps))             # :67
    if cli.vae or args.get("pretrained_model_path") and os.path.isdir(os.path.join(args.pretrained_model_path, "vae")):
        root = cli.vae or os.path.join(args.pretrained_model_path, "vae")
        vae = latte_amd.AutoencoderKL.from_pretrained(root, latent_size=args.latent_size, max_frames=args.num_frames).to(device)
    else:
        from latte_amd.random_init import vae_decoder_state_dict
        vae = latte_amd.AutoencoderKL(latent_size=args.latent_size, max_frames=args.num_frames)
        vae.load_state_dict(vae_decoder_state_dict(0))
        vae.to(device)
    if args.get("use_fp16"):                                               # :72-75
        model.to(dtype=torch.float16)
        vae.to(dtype=torch.float16)
    z = torch.randn(1, args.num_frames, 4, args.latent_size, args.latent_size, device=device)    # :81-84
    if using_cfg:                                                          # :86-94
        z = torch.cat([z, z], 0)
        y = torch.randint(0, args.num_classes, (1,), device=device)
        y_null = torch.tensor([args.num_classes] * 1, device=device)
        y = torch.cat([y, y_null], dim=0)
        model_kwargs = dict(y=y, cfg_scale=args.cfg_scale, use_fp16=bool(args.get("use_fp16")))
        sample_fn = model.forward_with_cfg
    else:                                                                  # :95-98
        sample_fn = model.forward
        model_kwargs = dict(y=None, use_fp16=bool(args.get("use_fp16")))
    if args.sample_method == "ddim":                                       # :100-107
        samples = diffusion.ddim_sample_loop(sample_fn, z.shape, z, clip_denoised=False, model_kwargs=model_kwargs,
                                             progress=True, device=device)
    else:
        samples = diffusion.p_sample_loop(sample_fn, z.shape, z, clip_denoised=False, model_kwargs=model_kwargs,
                                          progress=True, device=device)
    if using_cfg:
        samples, _ = samples.chunk(2, dim=0)                               # :109
    b, f, c, h, w = samples.shape                                          # :112 '(b f) c h w'
    frames = vae.decode(samples.reshape(b * f, c, h, w) / 0.18215).sample  # :113-115
    video = ((frames.reshape(b, f, *frames.shape[1:]) * 0.5 + 0.5) * 255).add_(0.5).clamp_(0, 255) \
        .to(dtype=torch.uint8).cpu().permute(0, 1, 3, 4, 2).contiguous()   # :122
    out_dir = args.get("save_video_path") or "./sample_videos"
    os.makedirs(out_dir, exist_ok=True)                                    # :118-120
    path = os.path.join(out_dir, "sample.mp4")                             # :123
    latte_amd.write_mp4(path, video[0], fps=8)                             # :124-126 (fps 8)
    print("saved", path, tuple(video.shape))
    return video


if __name__ == "__main__":
    parser = argparse.ArgumentParser()
    parser.add_argument("--config", type=str, default=os.path.join(ROOT, "configs", "ffs_sample.yaml"))
    parser.add_argument("--ckpt", type=str, default="")
    parser.add_argument("--save_video_path", type=str, default="")
    parser.add_argument("--vae", type=str, default="")
    parser.add_argument("--steps", type=int, default=0)
    cli = parser.parse_args()
    conf = latte_amd.load_config(cli.config)                               # :136
    if cli.ckpt:
        conf.ckpt = cli.ckpt                                               # :137
    if cli.save_video_path:
        conf.save_video_path = cli.save_video_path                         # :138
    if cli.steps:
        conf.num_sampling_steps = cli.steps
    main(conf, cli)
