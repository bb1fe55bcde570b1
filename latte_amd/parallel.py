"""Multi-GPU sampling: batch-of-samples sharding, one process per GPU (sample/sample_ddp.py:51-185).

The path shards by sample with NO data-path collective (the reference has none either: two barriers,
sample_ddp.py:113,180).  What this module adds to the reference's inline arithmetic:

* ``plan_shards``: the iteration / sample-index arithmetic of sample_ddp.py:116-176
  (``index = i * world + rank + total``), as a pure function so it can be tested without GPUs;
* ``sample_noise``: the initial latent of a sample drawn from its GLOBAL index, so a run produces the same
  videos for any world size (the reference seeds per rank, sample_ddp.py:63-65, which ties results to N);
* ``broadcast_temb_table``: the one payload collective of the design — rank 0 computes the timestep-embedding
  table ``[num_timesteps, hidden]`` fp32 (1.15 MB for 250 x 1152) through the engine and broadcasts it with
  ``torch.distributed`` (backend "nccl" = RCCL over xGMI on ROCm; "gloo" in the CPU tests); every rank installs
  it with ``latte_engine_set_temb_table``.
"""
import math
import os

import torch
import torch.distributed as dist


def setup_distributed(backend=None):
    """-> (rank, world_size, local_rank).  torchrun / torch.distributed.run environment; single process otherwise."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    return rank, world, local


def barrier():
    if dist.is_available() and dist.is_initialized():
        dist.barrier()


def plan_shards(num_samples, per_proc_batch_size, rank, world_size):
    """sample_ddp.py:116-176.  -> (total_samples, iterations, index_lists) where index_lists[it] are the GLOBAL
    sample indices this rank writes in iteration `it` (``i * world + rank + total``)."""
    n = int(per_proc_batch_size)
    global_batch = n * world_size
    total_samples = int(math.ceil(num_samples / global_batch) * global_batch)
    assert total_samples % world_size == 0, "total_samples must be divisible by world_size"
    per_gpu = total_samples // world_size
    assert per_gpu % n == 0, "samples_needed_this_gpu must be divisible by the per-GPU batch size"
    iterations = per_gpu // n
    lists, total = [], 0
    for _ in range(iterations):
        lists.append([i * world_size + rank + total for i in range(n)])
        total += global_batch
    return total_samples, iterations, lists


def sample_noise(global_index, shape, seed=0, device="cpu"):
    """Initial latent of ONE sample ([F, C, H, W]) as a function of its global index only."""
    g = torch.Generator("cpu").manual_seed((int(seed) * 1_000_003 + int(global_index)) & 0x7FFFFFFFFFFFFFFF)
    return torch.randn(*shape, generator=g).to(device)


def sample_label(global_index, num_classes, seed=0):
    g = torch.Generator("cpu").manual_seed((int(seed) * 2_000_003 + 7919 * int(global_index) + 1) & 0x7FFFFFFFFFFFFFFF)
    return int(torch.randint(0, num_classes, (1,), generator=g))


def broadcast_tensor(t, src=0):
    """In-place broadcast of `t` from rank `src` (no-op in a single process)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(t, src=src)
    return t


def broadcast_temb_table(model, diffusion, batch=1, src=0, guided=False):
    """Rank `src` computes the timestep-embedding table of `diffusion` on its engine, everyone receives it over
    RCCL and installs it.  Returns the table (device tensor [num_timesteps, hidden]).  `guided`: install it on the engine
    guided calls use (forward_with_cfg / the fused loop with cfg: a separate engine when the operand types differ, latte_amd.Latte
    docstring) -- the table lands on the engine the sampling loop will actually run."""
    from ._lib import check, load_library, ptr, stream_ptr
    lib = load_library()
    eng = model.engine(batch, guided=guided)
    dev = model.pos_embed.device
    rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
    table = torch.zeros(diffusion.num_timesteps, model.hidden_size, device=dev, dtype=torch.float32)
    with torch.cuda.device(dev):
        if rank == src:
            check(lib.latte_engine_temb_table(eng, diffusion._h, ptr(table), stream_ptr()))
        broadcast_tensor(table, src)
        check(lib.latte_engine_set_temb_table(eng, diffusion._h, ptr(table), stream_ptr()))
        torch.cuda.current_stream().synchronize()
    return table
