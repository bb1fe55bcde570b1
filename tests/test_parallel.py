"""Multi-GPU sharding logic on CPU: pure arithmetic + a world_size-2 gloo group (127.0.0.1)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from latte_amd import parallel


def test_plan_shards_matches_reference_arithmetic():
    # sample_ddp.py:116-176 for num_fvd_samples=2048, per_proc_batch_size=2, 8 ranks
    seen = set()
    for rank in range(8):
        total, iters, lists = parallel.plan_shards(2048, 2, rank, 8)
        assert total == 2048 and iters == 128 and len(lists) == 128
        assert lists[0] == [rank, 8 + rank] and lists[1] == [16 + rank, 24 + rank]
        for l in lists:
            seen.update(l)
    assert seen == set(range(2048))                       # every sample exactly once, no rank overlap
    total, iters, _ = parallel.plan_shards(10, 4, 0, 2)   # rounds up to a multiple of the global batch
    assert total == 16 and iters == 2


def test_noise_depends_on_global_index_only():
    a = parallel.sample_noise(37, (4, 4, 8, 8), seed=3)
    b = parallel.sample_noise(37, (4, 4, 8, 8), seed=3)
    c = parallel.sample_noise(38, (4, 4, 8, 8), seed=3)
    assert torch.equal(a, b) and not torch.equal(a, c)
    assert 0 <= parallel.sample_label(5, 101, seed=1) < 101
    assert parallel.sample_label(5, 101, seed=1) == parallel.sample_label(5, 101, seed=1)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    r, w, _ = parallel.setup_distributed("gloo")
    assert (r, w) == (rank, world)
    # the timestep-embedding table broadcast: rank 0 owns the values, everybody ends with them
    table = torch.arange(250 * 16, dtype=torch.float32).view(250, 16) if rank == 0 else torch.zeros(250, 16)
    parallel.broadcast_tensor(table, src=0)
    ok = torch.equal(table, torch.arange(250 * 16, dtype=torch.float32).view(250, 16))
    # sharded "sampling": each rank produces the noise of its own global indices; rank 0 gathers the index lists
    _, _, lists = parallel.plan_shards(8, 2, rank, world)
    mine = sorted(i for l in lists for i in l)
    gathered = [None] * world
    dist.all_gather_object(gathered, mine)
    parallel.barrier()
    # max-over-ranks timing reduction used by bench.py
    t = torch.tensor([1.0 + rank], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if rank == 0:
        torch.save({"ok": ok, "gathered": gathered, "tmax": float(t)}, os.path.join(out_dir, "r0.pt"))
    assert ok
    dist.destroy_process_group()


def test_world_size_2_gloo(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    res = torch.load(os.path.join(str(tmp_path), "r0.pt"))
    assert res["ok"] and res["tmax"] == 2.0
    assert sorted(res["gathered"][0] + res["gathered"][1]) == list(range(8))
    assert set(res["gathered"][0]).isdisjoint(res["gathered"][1])
