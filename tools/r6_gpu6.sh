set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_vae.py tests/test_vae_temporal.py -q -m gpu -x 2>&1 | tail -6
python - <<'PY'
import sys, json, torch
sys.argv=['bench.py']
import bench
from latte_amd._lib import load_library, check
lib=load_library()
dev=torch.device('cuda')
for k in (3, 0):
    check(lib.latte_debug_set_choice(b"conv_kernel", k))
    r=bench.vae_decode_rate(dev)
    print("conv_kernel",k,"vae ms",r["ms_per_video"],[ (t["class"],t["ms_per_video"],t.get("frac")) for t in r["roofline_table"]],flush=True)
for k in (3, 0):
    check(lib.latte_debug_set_choice(b"conv_kernel", k))
    import latte_amd
    from latte_amd.random_init import vae_temporal_decoder_state_dict
    import time
    vae = latte_amd.AutoencoderKLTemporalDecoder(latent_size=64, max_frames=14)
    vae.load_state_dict(vae_temporal_decoder_state_dict(0)); vae.to(dev)
    z = torch.randn(16, 4, 64, 64, device=dev)
    def decode():
        return [vae.decode(z[i:i + 14].contiguous(), num_frames=min(14, 16 - i)).sample for i in range(0, 16, 14)]
    decode(); torch.cuda.synchronize(); t0=time.perf_counter(); decode(); torch.cuda.synchronize()
    print("conv_kernel",k,"temporal decoder ms",round((time.perf_counter()-t0)*1e3,1),flush=True)
    prof=vae.profile_decode(z[:14].contiguous()); prof=vae.profile_decode(z[:14].contiguous())
    print("   chunk14 classes", {c:(round(v[0],2),v[1]) for c,v in prof.items()},flush=True)
    del vae
PY
