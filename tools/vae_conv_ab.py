#!/usr/bin/env python
"""Interleaved A/B of the convolution kernel choice inside the SD-VAE decode (16 frames) and, with `temporal`, inside a 14-frame chunk of
the temporal decoder: latte_debug_set_choice("conv_kernel", v) per decode, alternating in one process.  Prints ms per decode, the
convolution class time of a profiled decode, and the output difference against choice 0."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import latte_amd
from latte_amd import _lib
from latte_amd.random_init import vae_decoder_state_dict
lib = _lib.load_library()
choices = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "0,1,3").split(",")]
dev = torch.device("cuda")
vae = latte_amd.AutoencoderKL(latent_size=32, max_frames=16, compute_dtype="f16")
vae.load_state_dict(vae_decoder_state_dict(0))
vae.to(dev)
lat = torch.randn(1, 16, 4, 32, 32, device=dev) * 0.18215
z = (lat[0] / 0.18215).contiguous()
ref = None
best = {c: 1e9 for c in choices}
conv = {}
for rnd in range(4):
    for c in choices:
        assert lib.latte_debug_set_choice(b"conv_kernel", c) == 0
        out = vae.decode_video_uint8(lat)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            out = vae.decode_video_uint8(lat)
        torch.cuda.synchronize()
        best[c] = min(best[c], (time.perf_counter() - t0) / 3 * 1e3)
        if rnd == 0:
            vae.profile_decode(z)
            conv[c] = vae.profile_decode(z)["conv3x3"]
            o = out.float()
            if ref is None:
                ref = o
            else:
                print(f"conv_kernel {c}: uint8 output against conv_kernel {choices[0]}: max abs diff {float((o - ref).abs().max()):.0f}, mean abs {float((o - ref).abs().mean()):.4f}")
lib.latte_debug_set_choice(b"conv_kernel", 0)
for c in choices:
    print(f"conv_kernel {c}: {best[c]:7.2f} ms per 16-frame decode; convolutions {conv[c][0]:.3f} ms in {conv[c][1]} launches")
