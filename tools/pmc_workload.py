"""Workload for the rocprofv3 passes of tools/pmc_round.sh: Latte-XL/2 forwards at the bench batch (and optionally one
16-frame VAE decode), so that one trace holds every kernel class of the step with its real in-model cache state."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import latte_amd  # noqa: E402
from latte_amd.random_init import vae_decoder_state_dict  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n_fwd = int(sys.argv[2]) if len(sys.argv) > 2 else 2
mode = sys.argv[3] if len(sys.argv) > 3 else ""
with_vae = mode in ("vae", "vaeonly")
dtype = os.environ.get("LATTE_PMC_DTYPE", "f16")      # the headline operand type since round 4
dev = torch.device("cuda")
torch.manual_seed(0)
if mode != "vaeonly":
    m = latte_amd.Latte_models["Latte-XL/2"](input_size=32, num_frames=16, extras=1, compute_dtype=dtype, max_batch=B)
    g = torch.Generator("cpu").manual_seed(1)
    with torch.no_grad():
        for _, p in m.named_parameters():
            if p.requires_grad and float(p.detach().abs().max()) == 0.0:
                p.copy_(torch.randn(p.shape, generator=g) * 0.02)
    m = m.to(dev).eval()
    x = torch.randn(B, 16, 4, 32, 32, device=dev)
    t = torch.full((B,), 500, device=dev, dtype=torch.int64)
    for _ in range(n_fwd + 1):
        out = m(x, t)
    torch.cuda.synchronize()
    print("forward finite:", bool(torch.isfinite(out).all()))
if with_vae:
    vae = latte_amd.AutoencoderKL(latent_size=32, max_frames=16)
    vae.load_state_dict(vae_decoder_state_dict(0))
    vae.to(dev)
    lat = torch.randn(1, 16, 4, 32, 32, device=dev) * 0.18215
    for _ in range(2):
        u8 = vae.decode_video_uint8(lat)
    torch.cuda.synchronize()
    print("vae decode ok:", tuple(u8.shape))
