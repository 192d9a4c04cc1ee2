"""world_size-2 (and 3) gloo test of the multi-rank path of bench.py: every rank owns the
tiles with tile_id % N == rank, packs them as the kernels do, ONE gather brings equal-sized
shards to rank 0, which un-permutes them.  The pixel values on CPU come from the host compile
of the lane program (tests only); the permutations under test are the PRODUCT's own
(librt_mi355x: rt_shard_tile_count, rt_pack_tiles_host, rt_unpack_tiles_host -- the loops rt_render and
rt_render_progressive run on every call), checked against the numpy restatement of the kernels' layout as well."""
import os
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent


def _worker(rank, world, port, W, H, q):
    sys.path.insert(0, str(ROOT / "tests"))
    sys.path.insert(0, str(ROOT))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import importlib

    import lane_emul_binding as le
    import sharding_mirror as sm
    from __graft_entry__ import load_package
    rt = load_package()
    scenes = importlib.import_module("ray_tracer_amd.scenes")
    sc, cam = scenes.build_product(scenes.book_one(1, W / H), device=-1)
    counts = [rt.shard_tile_count(W, H, r, world) for r in range(world)]
    pad = max(counts)
    assert counts[rank] == len(sm.owned_tiles(W, H, rank, world))
    full, *_ = le.render(sc, cam, W, H, 2, 20, 5)  # sample streams are global: any rank computes the same pixels
    packed = rt.pack_tiles_host(full, rank, world, pad)            # the product's permutation ...
    assert np.array_equal(packed, sm.pack_tiles(full, rank, world, pad))  # ... is the kernels' layout
    mine = torch.from_numpy(packed.reshape(-1))
    glist = [torch.zeros_like(mine) for _ in range(world)] if rank == 0 else None
    dist.gather(mine, glist, dst=0)
    if rank == 0:
        gathered = torch.cat(glist).numpy()
        img = rt.unpack_tiles_host(gathered, pad, world, W, H)
        q.put(bool(np.array_equal(img, full)) and bool(np.array_equal(img, sm.unpack_tiles(gathered, pad, world, W, H))))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,W,H", [(2, 40, 24), (3, 35, 19)])
def test_tile_sharding_gather_unpack(world, W, H, rt, lane_emul):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + world
    procs = [ctx.Process(target=_worker, args=(r, world, port, W, H, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=180)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


def test_bench_gpus_n_without_a_launcher_relays_failure():
    """`python bench.py --gpus 2` with no WORLD_SIZE starts its own ranks (a child torch.distributed.run).  Without a GPU
    every rank fails: the launcher's non-zero exit code comes back and NO result line is printed -- never a silent
    single-rank run.  (The successful form of the same command is a -m gpu test.)"""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible: covered by tests/test_gpu_parity.py::test_bench_gpus_n_starts_its_own_ranks")
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--backend", "gloo", "--same-device", "--steps", "1",
                        "--warmup", "0", "--width", "64", "--height", "64", "--spp", "2", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=300, env=env, cwd=str(ROOT))
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.split("\n") if ln.startswith("{")], r.stdout
    assert "nproc-per-node" not in r.stdout  # the old usage message is gone
