"""sample/sample_t2x.py of the reference on the MI355X engine: Latte-1 text-to-video.

  python tools/sample_t2x.py --config configs/t2v_sample.yaml [--random] [--steps N] [--layers L]

With a real checkpoint directory (`pretrained_model_path` holding transformer/, vae/, tokenizer/, text_encoder/) the flow is
the reference's (sample_t2x.py:21-140): T5 tokenizer + encoder from `transformers`, `LatteT2V.from_pretrained_2d`,
`AutoencoderKL.from_pretrained`, a DDIM scheduler, `LattePipeline(...)`, one video per prompt.  Offline there are no
weights: `--random` builds randomly initialised models and random prompt embeddings of the right shape, which exercises the
whole device path (denoiser, guidance loop, VAE decode, video hand-off) and times it.  Videos are written as .mp4
(Motion-JPEG samples, latte_amd.video_io).  Only the DDIM scheduler has a self-contained stand-in (latte_amd/schedulers.py); any diffusers
scheduler object can be passed to LattePipeline instead.
"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import latte_amd  # noqa: E402
from latte_amd.schedulers import DDIMScheduler  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", required=True)
    ap.add_argument("--random", action="store_true", help="random weights and prompt embeddings (no checkpoints offline)")
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--layers", type=int, default=28)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    args = latte_amd.load_config(a.config)
    torch.set_grad_enabled(False)
    assert torch.cuda.is_available(), "sample_t2x needs an MI355X"
    device = "cuda"
    cdt = "f16"                       # sample_t2x.py:29 .to(device, dtype=torch.float16): the only operand type of latte_amd.LatteT2V
    latent = args.image_size[0] // 8
    if args.sample_method != "DDIM":
        raise SystemExit("only the DDIM scheduler has an offline stand-in; pass a diffusers scheduler to LattePipeline otherwise")
    scheduler = DDIMScheduler(beta_start=args.beta_start, beta_end=args.beta_end, beta_schedule=args.beta_schedule, clip_sample=False)
    tokenizer = text_encoder = None
    if a.random:
        from latte_amd.random_init import t2v_state_dict, vae_decoder_state_dict
        transformer = latte_amd.LatteT2V(num_layers=a.layers, sample_size=latent, video_length=args.video_length,
                                         compute_dtype=cdt, max_batch=2).load_state_dict(t2v_state_dict(0, num_layers=a.layers))
        if args.enable_vae_temporal_decoder:                       # sample_t2x.py:31-32
            from latte_amd.random_init import vae_temporal_decoder_state_dict
            vae = latte_amd.AutoencoderKLTemporalDecoder(latent_size=latent, max_frames=14, compute_dtype="f16")
            vae.load_state_dict(vae_temporal_decoder_state_dict(0))
        else:
            vae = latte_amd.AutoencoderKL(latent_size=latent, max_frames=args.video_length, compute_dtype="f16")
            vae.load_state_dict(vae_decoder_state_dict(0))
    else:
        from transformers import T5EncoderModel, T5Tokenizer
        p = args.pretrained_model_path
        transformer = latte_amd.LatteT2V.from_pretrained_2d(p, subfolder="transformer", video_length=args.video_length,
                                                            compute_dtype=cdt, max_batch=2)
        if args.enable_vae_temporal_decoder:                       # sample_t2x.py:31-32
            vae = latte_amd.AutoencoderKLTemporalDecoder.from_pretrained(p, subfolder="vae_temporal_decoder", latent_size=latent,
                                                                         max_frames=14)
        else:
            vae = latte_amd.AutoencoderKL.from_pretrained(p, subfolder="vae", latent_size=latent, max_frames=args.video_length)
        tokenizer = T5Tokenizer.from_pretrained(p, subfolder="tokenizer")
        text_encoder = T5EncoderModel.from_pretrained(p, subfolder="text_encoder", torch_dtype=torch.float16).to(device).eval()
    pipe = latte_amd.LattePipeline(vae=vae, text_encoder=text_encoder, tokenizer=tokenizer, scheduler=scheduler,
                                   transformer=transformer).to(device)
    out_dir = a.out or args.save_img_path
    os.makedirs(out_dir, exist_ok=True)
    steps = a.steps or args.num_sampling_steps
    g = torch.Generator("cpu").manual_seed(int(args.seed or 0))
    for n, prompt in enumerate(args.text_prompt):
        print(f"Processing the ({prompt}) prompt")
        kw = {}
        if a.random:                                                  # stand-in for the T5 features of the prompt / of ""
            k = min(8 + len(prompt.split()), 120)
            kw = dict(prompt_embeds=torch.randn(1, k, 4096, generator=g), negative_prompt_embeds=torch.randn(1, k, 4096, generator=g))
        else:
            kw = dict(prompt=prompt)
        torch.cuda.synchronize()
        t0 = time.time()
        video = pipe(video_length=args.video_length, height=args.image_size[0], width=args.image_size[1],
                     num_inference_steps=steps, guidance_scale=args.guidance_scale,
                     enable_temporal_attentions=args.enable_temporal_attentions, num_images_per_prompt=1, mask_feature=True,
                     enable_vae_temporal_decoder=bool(args.enable_vae_temporal_decoder), generator=g, **kw).video
        torch.cuda.synchronize()
        dt = time.time() - t0
        path = os.path.join(out_dir, f"{n:03d}.mp4")
        latte_amd.write_mp4(path, video[0], fps=8)                    # sample_t2x.py:137 imageio.mimwrite(..., fps=8)
        print(f"  {steps} steps + decode in {dt:.2f} s ({steps / dt:.2f} steps/s incl. decode) -> {path} {tuple(video.shape)}")


if __name__ == "__main__":
    main()
