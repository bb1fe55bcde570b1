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


# ------------------------------------------------------------------------------------------------ GPU
import subprocess  # noqa: E402
import sys  # noqa: E402

import numpy as np  # noqa: E402
import pytest  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_sample_ddp_world1_reproduces_single_sample_latents(tmp_path):
    """tools/sample_ddp.py (the reference's sample_ddp.py flow, BASELINE config 3's driver) at world size 1 on the tiny
    guided class-conditional config: every latent it writes equals the SAME global index sampled alone in this process --
    noise and label are functions of the global sample index, and the engine is batch-composition invariant, so the result
    of a sample does not depend on how samples are spread over processes."""
    import latte_amd
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import sample_ddp
    cfg_path = os.path.join(ROOT, "configs", "tiny_sample.yaml")
    out = str(tmp_path)
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "sample_ddp.py"), "--config", cfg_path, "--num-samples", "4",
                        "--steps", "6", "--no-decode", "--out", out], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    files = sorted(f for f in os.listdir(out) if f.endswith("_latent.npy"))
    assert files == [f"{i:04d}_latent.npy" for i in range(4)]
    args = latte_amd.load_config(cfg_path)
    seed = int(args.seed)
    args.latent_size, args.max_batch = args.image_size // 8, 2
    torch.manual_seed(seed)
    model = latte_amd.get_models(args)
    sample_ddp.randomise_zero_init(model)
    model = model.to("cuda").eval()
    d = latte_amd.create_diffusion("6")
    shape = (args.num_frames, 4, args.latent_size, args.latent_size)
    for i in range(4):
        z = parallel.sample_noise(i, shape, seed, "cuda")[None]
        y = torch.tensor([parallel.sample_label(i, args.num_classes, seed), args.num_classes], device="cuda")
        x = torch.cat([z, z], 0)
        s = d.ddim_sample_loop(model.forward_with_cfg, x.shape, x, clip_denoised=False,
                               model_kwargs=dict(y=y, cfg_scale=float(args.cfg_scale)), device="cuda")
        got = torch.from_numpy(np.load(os.path.join(out, files[i])))
        assert torch.isfinite(got).all()
        assert torch.equal(got, s[0].cpu()), i


@pytest.mark.gpu
def test_sample_single_video_tool(tmp_path):
    """tools/sample.py (sample/sample.py re-hosted): one guided video through model, sampler, VAE decode and the file."""
    import latte_amd
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "sample.py"), "--config",
                        os.path.join(ROOT, "configs", "tiny_sample.yaml"), "--save_video_path", str(tmp_path), "--steps", "4"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    frames, fps = latte_amd.read_mp4(os.path.join(str(tmp_path), "sample.mp4"))   # the reference's file name (sample.py:123)
    assert tuple(frames.shape) == (4, 128, 128, 3) and fps == 8 and float(np.asarray(frames, dtype=np.float64).std()) > 0


def _rccl_worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import latte_amd
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from _util import engine_model, load_golden_model
    r, w, local = parallel.setup_distributed("nccl")               # RCCL over xGMI
    kw, sd, g = load_golden_model("tiny_classcond")
    dev = torch.device("cuda", local)
    m = engine_model(kw, sd, "bf16", device=dev)
    d = latte_amd.create_diffusion("10")
    x, y = torch.from_numpy(g["x"]).to(dev), torch.from_numpy(g["y"]).to(dev)
    own = d.ddim_sample_loop(m.forward, x.shape, x.clone(), clip_denoised=False, model_kwargs=dict(y=y))
    table = parallel.broadcast_temb_table(m, d, batch=x.shape[0])  # rank 0 computes, everybody installs
    got = d.ddim_sample_loop(m.forward, x.shape, x.clone(), clip_denoised=False, model_kwargs=dict(y=y))
    tables = [torch.zeros_like(table) for _ in range(world)]
    dist.all_gather(tables, table)
    ok = all(torch.equal(t.cpu(), tables[0].cpu()) for t in tables) and torch.equal(own, got)
    parallel.barrier()
    torch.save({"ok": bool(ok)}, os.path.join(out_dir, f"r{rank}.pt"))
    dist.destroy_process_group()


@pytest.mark.gpu
def test_broadcast_temb_table_rccl_two_gpus(tmp_path):
    """The one payload collective of the design over RCCL: needs two visible GPUs (skipped on a 1-GPU lease)."""
    if torch.cuda.device_count() < 2:
        pytest.skip("RCCL needs one device per rank: fewer than 2 GPUs visible")
    port = _free_port()
    mp.spawn(_rccl_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert all(torch.load(os.path.join(str(tmp_path), f"r{r}.pt"))["ok"] for r in range(2))


def _run_bench(extra, timeout):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["LATTE_BENCH_BACKEND"] = "gloo"      # two ranks on fewer than two GPUs: RCCL wants one device per rank
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"] + extra, env=env, capture_output=True,
                       text=True, timeout=timeout)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]          # ONE JSON line, from rank 0
    return json.loads(lines[0])


def test_bench_gpus_flag_launches_the_ranks_itself():
    """`python bench.py --gpus 2` with no launcher around it (the driver's N > 1 contract allows both forms) must start two
    ranks, rendezvous on 127.0.0.1 and report n_gpus = 2; --launch-check stops after the collective (no GPU here)."""
    res = _run_bench(["--launch-check"], 240)
    assert res == {"launch_check": True, "n_gpus": 2, "collective_ranks": 2, "collective_backend": "gloo"}


import pytest  # noqa: E402


@pytest.mark.gpu
def test_bench_two_ranks_full_path_on_the_gpu():
    """The whole N = 2 code path of bench.py (temb-table broadcast, barriers, max-over-ranks reduction, one JSON line) with
    both ranks on the lease's GPU over gloo: `--gpus 2` alone starts them.  Numbers are not asserted, the contract fields are."""
    res = _run_bench(["--steps", "2", "--warmup", "1", "--batch", "1", "--no-side", "--no-vae", "--no-cpu-baseline"], 900)
    assert res["n_gpus"] == 2 and res["collective_ranks"] == 2 and res["steps"] == 2 and res["warmup"] == 1
    assert res["config"]["global_batch"] == 2 and res["scaling"] == "weak" and res["finite"]
    assert res["value"] > 0 and res["roofline"]["frac"] > 0
