"""Side measurement: BASELINE config 4's decoder -- AutoencoderKLTemporalDecoder on 16 frames of 64x64 latents in chunks of 14 + 2 frames
(sample/pipeline_latte.py:779-798), random weights; ms per video and the per-class table of latte_vae_profile_decode."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import latte_amd  # noqa: E402
from latte_amd.random_init import vae_temporal_decoder_state_dict  # noqa: E402

dev = torch.device("cuda")
vae = latte_amd.AutoencoderKLTemporalDecoder(latent_size=64, max_frames=14)
vae.load_state_dict(vae_temporal_decoder_state_dict(0))
vae.to(dev)
z = torch.randn(16, 4, 64, 64, generator=torch.Generator("cpu").manual_seed(4000)).to(dev)


def decode():
    return [vae.decode(z[i:i + 14].contiguous(), num_frames=min(14, 16 - i)).sample for i in range(0, 16, 14)]


decode()
torch.cuda.synchronize()
n = int(os.environ.get("LATTE_DECODE_REPS", "3"))
t0 = time.perf_counter()
for _ in range(n):
    decode()
torch.cuda.synchronize()
print(f"temporal decoder: {(time.perf_counter() - t0) / n * 1e3:.2f} ms per 16-frame 512x512 video")
if os.environ.get("LATTE_DECODE_PROFILE", "1") == "1":
    z14 = z[:14].contiguous()
    vae.profile_decode(z14)
    print("chunk14 classes (ms, launches):", {k: (round(v[0], 2), v[1]) for k, v in vae.profile_decode(z14).items()})
