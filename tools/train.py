"""train.py of the reference, re-hosted on the MI355X engine (BASELINE config 5).

  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 tools/train.py --config configs/ffs_train.yaml
  python tools/train.py --config configs/ffs_train.yaml --max-steps 20 --log-every 5          (single GPU)

One process per GPU (train.py:55-66), the model replicated, `local_batch_size` latent clips per rank and step; per step
`LatteTrainer.train_step` = q_sample + forward + training_losses + backward (gradient slices all-reduced over RCCL bucket by bucket
under the backward) + clip_grad_norm_ + AdamW + update_ema (train.py:197-236).  Checkpoints are the reference's
`{"model": state_dict, "ema": state_dict}` (train.py:257-262) and load back through `find_model` / `--pretrained`.
The reference's VAE ENCODER step (train.py:205-211) is outside the engine: `data_path` holds latent clips (.npy [F, 4, h, w], already
scaled by 0.18215) or is "synthetic" (N(0, 1) latents: throughput and plumbing, not a model worth keeping).
"""
import argparse
import glob
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import latte_amd  # noqa: E402
from latte_amd import parallel  # noqa: E402


class LatentClips:
    """Rank-sharded, seeded access to the latent clips (DistributedSampler(shuffle=True, seed=global_seed), train.py:136-151)."""

    def __init__(self, path, frames, latent, rank, world, seed, num_classes):
        self.synthetic = path in (None, "", "synthetic")
        self.shape = (frames, 4, latent, latent)
        self.rank, self.world, self.seed, self.num_classes = rank, world, seed, num_classes
        self.files = [] if self.synthetic else sorted(glob.glob(os.path.join(path, "*.npy")))
        if not self.synthetic and not self.files:
            raise SystemExit(f"no .npy latent clips under {path}")

    def batch(self, step, n):
        g = torch.Generator("cpu").manual_seed(self.seed * 1000003 + step * self.world + self.rank)
        if self.synthetic:
            x = torch.randn(n, *self.shape, generator=g)
            y = torch.randint(0, max(self.num_classes, 1), (n,), generator=g)
            return x, y
        idx = torch.randint(0, len(self.files), (n,), generator=g).tolist()
        xs, ys = [], []
        for i in idx:
            a = np.load(self.files[i])
            assert a.shape == self.shape, f"{self.files[i]}: expected {self.shape}, got {a.shape}"
            xs.append(torch.from_numpy(a).float())
            name = os.path.basename(self.files[i])
            ys.append(int(name.split("_")[0]) if name.split("_")[0].isdigit() else 0)
        return torch.stack(xs), torch.tensor(ys)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", required=True)
    ap.add_argument("--max-steps", type=int, default=None)
    ap.add_argument("--log-every", type=int, default=None)
    ap.add_argument("--ckpt-every", type=int, default=None)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    args = latte_amd.load_config(a.config)
    rank, world, local = parallel.setup_distributed()
    assert torch.cuda.is_available(), "tools/train.py needs MI355X GPUs"
    device = torch.device("cuda", local)
    torch.cuda.set_device(device)
    if int(args.get("gradient_accumulation_steps") or 1) != 1:
        raise SystemExit("gradient_accumulation_steps must be 1")
    seed = int(args.global_seed)
    torch.manual_seed(seed)                                   # identical replicas (DDP broadcasts rank 0's weights, train.py:125)
    assert args.image_size % 8 == 0, "Image size must be divisible by 8 (for the VAE encoder)."   # train.py:88
    args.latent_size = args.image_size // 8
    nb = int(args.local_batch_size)
    args.max_batch = nb
    model = latte_amd.get_models(args).to(device)
    if args.get("pretrained"):
        sd = latte_amd.find_model(args.pretrained)
        own = model.state_dict()
        model.load_state_dict({**own, **{k: v for k, v in sd.items() if k in own}})          # train.py:109-122
    diffusion = latte_amd.create_diffusion(timestep_respacing="")                                # train.py:92
    trainer = latte_amd.LatteTrainer(model, diffusion, max_batch=nb, lr=float(args.learning_rate), clip_max_norm=float(args.clip_max_norm),
                                     start_clip_iter=int(args.start_clip_iter))
    # Replicas were initialised from the shared seed; from here every rank draws its own timesteps, noise and label-dropout
    # masks, as the reference does (train.py:62 `seed = args.global_seed + rank`): a global batch covers world * nb independent
    # draws, not nb draws replicated world times.
    torch.manual_seed(seed + rank)
    if args.get("resume_from_checkpoint"):
        raise SystemExit("resume_from_checkpoint (the reference's accelerate-style state resume, train.py:176-192) is not supported: "
                         "continue from a checkpoint with `pretrained: <results_dir>/checkpoints/<step>.pt`")
    first_step = 0
    if args.get("pretrained"):
        # train.py:195-196: the step counter continues from the checkpoint's file name (0100000.pt -> 100000), so gradient clipping
        # (start_clip_iter) and the checkpoint numbering carry on instead of restarting.  Only THAT counter: the checkpoint holds
        # model and ema, no optimiser state, so AdamW's moments start at zero and its bias correction at step 1 -- the engine
        # counts its applied updates itself (LatteTrainer docstring), as the reference's fresh torch.optim.AdamW does
        stem = os.path.basename(str(args.pretrained)).split(".")[0]
        if stem.isdigit():
            first_step = int(stem)
            trainer.train_steps = first_step
    data = LatentClips(args.get("data_path"), int(args.num_frames), args.latent_size, rank, world, seed, int(args.get("num_classes") or 0))
    out_dir = a.out or args.results_dir
    max_steps = a.max_steps or int(args.max_train_steps)
    log_every = a.log_every or int(args.log_every)
    ckpt_every = a.ckpt_every or int(args.ckpt_every)
    if rank == 0:
        os.makedirs(os.path.join(out_dir, "checkpoints"), exist_ok=True)
        print(f"Model Parameters: {sum(p.numel() for p in model.parameters()):,}; world {world}, local batch {nb}")
    parallel.barrier()
    running, t0, log_steps = 0.0, time.time(), 0
    stuck_logs = 0      # consecutive log lines whose whole interval was skipped updates at the floor scale (or with scaling off)
    seen_skips = 0.0
    for step in range(first_step + 1, max_steps + 1):
        x, y = data.batch(step, nb)
        out = trainer.train_step(x.to(device), y=y.to(device) if int(args.extras) == 2 else None)
        running += float(out["loss"].mean())                  # (the reference's loss.item(), train.py:239)
        log_steps += 1
        if step % log_every == 0:
            torch.cuda.synchronize()
            sps = log_steps / (time.time() - t0)
            avg = torch.tensor(running / log_steps, device=device)
            if world > 1:
                torch.distributed.all_reduce(avg)
                avg /= world
            # Overflow handling of the half-precision backward (LatteTrainer docstring): a non-finite gradient norm skips the update on
            # the device, silently.  The log line carries the counters, and a run that can no longer apply ANY update -- every step of
            # two log intervals skipped with the scale at its floor of 1 (or with dynamic scaling off) -- stops instead of burning its
            # budget on zero updates (a diverged model or an f16 forward overflow no loss scale cures).
            sc = trainer.scaler_state()
            new_skips = sc["skipped_updates"] - seen_skips
            seen_skips = sc["skipped_updates"]
            stuck = new_skips >= log_steps and (sc["loss_scale"] <= 1.0 or not sc["dynamic"])
            stuck_logs = stuck_logs + 1 if stuck else 0
            if rank == 0:
                print(f"(step={step:07d}) Train Loss: {float(avg):.4f}, Gradient Norm: {float(out['grad_norm']):.4f}, "
                      f"Train Steps/Sec: {sps:.2f}, samples/s: {sps * nb * world:.1f}, loss scale: {sc['loss_scale']:g}, "
                      f"skipped updates: {int(sc['skipped_updates'])} (+{int(new_skips)})", flush=True)
                if new_skips and not stuck:
                    print(f"  warning: {int(new_skips)} of the last {log_steps} updates were skipped (non-finite gradient norm)", flush=True)
            if stuck_logs >= 2:
                raise RuntimeError(f"training is stuck: every update of the last {2 * log_steps} steps was skipped (non-finite gradient norm) "
                                   f"with the loss scale at {sc['loss_scale']:g}" + ("" if sc["dynamic"] else " and dynamic scaling off")
                                   + " -- the model has diverged or an f16 activation overflows; lower the learning rate or train with compute_dtype='bf16'")
            running, t0, log_steps = 0.0, time.time(), 0
        if step % ckpt_every == 0 or step == max_steps:
            if rank == 0:
                path = os.path.join(out_dir, "checkpoints", f"{step:07d}.pt")
                torch.save({"model": {k: v.cpu() for k, v in trainer.model_state_dict().items()},
                            "ema": {k: v.cpu() for k, v in trainer.ema_state_dict().items()}}, path)
                print(f"Saved checkpoint to {path}", flush=True)
            parallel.barrier()
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
