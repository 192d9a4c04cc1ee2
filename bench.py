#!/usr/bin/env python3
"""bench.py -- Msamples/s of the book-one random-spheres scene, 1200x800x500 spp,
depth 100 (BASELINE.json configs[1]) on N MI355X GPUs of one node.

A "step" is one whole render of the image: every rank renders the 8x8 tiles with
tile_id % N == rank into a packed device buffer (hand-written HIP kernel through
the C ABI), one RCCL gather brings the shards to rank 0, one small kernel
un-permutes them and rank 0 copies the framebuffer to the host.  Total work is
fixed as N grows (strong scaling).  Inputs (the committed scene) are resident in
HBM before the timed region.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
"""
import argparse
import importlib
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md "Chip-level parameters"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--width", type=int, default=1200)
    ap.add_argument("--height", type=int, default=800)
    ap.add_argument("--spp", type=int, default=500)
    ap.add_argument("--depth", type=int, default=100)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--scene-seed", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target duration of the CPU baseline sample")
    ap.add_argument("--scene", default="book_one", choices=["book_one", "cornell", "cover"],
                    help="book_one is the headline workload; the others are for measuring BASELINE configs[2], [3]")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo = rehearsal of the N>1 path on a box with fewer GPUs (shards staged through host memory)")
    ap.add_argument("--same-device", action="store_true", help="rehearsal: every rank uses cuda:0")
    ap.add_argument("--check", action="store_true", help="rank 0 also renders the whole image alone and compares")
    ap.add_argument("--pipeline", action="store_true",
                    help="experiment: consecutive steps on two alternating streams (measured slower, see DESIGN.md section 8)")
    return ap.parse_args()


def cpu_baseline(desc, W, H, depth, seed, target_s):
    """The oracle in reference form (recursive unpruned BVH, 4x4 per sprite, uv on every hit),
    threaded like the example drivers, on a bounded sample of the same workload."""
    sys.path.insert(0, str(ROOT / "tests"))
    import oracle_binding
    o = oracle_binding.build_oracle(desc)
    cores = os.cpu_count() or 1
    t0 = time.perf_counter()
    o.render(W, H, 1, depth, seed, nthreads=cores)
    t1 = time.perf_counter() - t0
    spp = max(1, min(64, int(target_s / max(t1, 1e-3))))
    t0 = time.perf_counter()
    o.render(W, H, spp, depth, seed, nthreads=cores)
    dt = time.perf_counter() - t0
    return {"value": W * H * spp / dt / 1e6, "unit": "Msamples/s", "cores": cores, "kind": "port",
            "sample": f"book-one {W}x{H} at {spp} spp (of 500), depth {depth}, whole image, {dt:.1f} s; "
                      f"oracle reference form, {cores} threads dealt rows y % n like examples/book-one.rs:56-65"}


def main():
    a = parse()
    import numpy as np
    import torch
    import torch.distributed as dist
    from __graft_entry__ import load_package

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            raise SystemExit("launch N>1 with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
        raise SystemExit(f"--gpus {a.gpus} does not match WORLD_SIZE {world}")
    if a.same_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if a.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    rt = load_package()
    scenes = importlib.import_module("ray_tracer_amd.scenes")
    W, H, spp, depth = a.width, a.height, a.spp, a.depth
    if a.scene == "book_one":
        desc = scenes.book_one(a.scene_seed, W / H)
    elif a.scene == "cornell":
        desc = scenes.cornell(W / H)
    else:
        desc = scenes.cover(a.scene_seed, W / H)
    sc, cam = scenes.build_product(desc, device=local_rank)
    info = sc.info()

    n_tiles = [rt.shard_tile_count(W, H, r, world) for r in range(world)]
    pad_tiles = max(n_tiles)
    # --pipeline: two of everything a step writes and consecutive steps on alternating streams, so that the render of
    # step k+1 could fill the CUs while step k's last few 100-segment paths, its reduce, gather and unpack finish (the
    # library keeps two render slots per scene).  Measured SLOWER than one stream on the MI355X, so it is off by default
    n_buf = 2 if a.pipeline else 1
    mines = [torch.zeros(pad_tiles * 64 * 3, dtype=torch.float64, device=dev) for _ in range(n_buf)]
    gathereds = [torch.zeros(world * pad_tiles * 64 * 3, dtype=torch.float64, device=dev) if rank == 0 else None for _ in range(n_buf)]
    # default: the current stream, as ever; only the --pipeline experiment creates side streams
    render_streams = [torch.cuda.Stream(device=dev) for _ in range(n_buf)] if a.pipeline else [torch.cuda.current_stream()]
    mine = mines[0]
    # rank 0: double-buffered image + a copy stream, so the device->host copy of step k overlaps the render of
    # step k+1 (every copy still completes inside the timed region: sync() waits for all streams)
    images = [torch.zeros(H * W * 3, dtype=torch.float64, device=dev) for _ in range(2)] if rank == 0 else None
    host_images = [torch.zeros(H * W * 3, dtype=torch.float64).pin_memory() for _ in range(2)] if rank == 0 else None
    copy_stream = torch.cuda.Stream(device=dev) if rank == 0 else None
    copy_done = [torch.cuda.Event() for _ in range(2)] if rank == 0 else None
    step_no = [0]

    def step():
        k = step_no[0]
        step_no[0] += 1
        rs = render_streams[k % n_buf]
        mine_k, gathered_k = mines[k % n_buf], gathereds[k % n_buf]
        with torch.cuda.stream(rs):
            stream = rs.cuda_stream
            sc.render_tiles_device(cam, W, H, spp, depth, a.seed, (rank, world), mine_k.data_ptr(), None, stream)
            if world > 1 and a.backend == "nccl":
                glist = list(gathered_k.chunk(world)) if rank == 0 else None
                dist.gather(mine_k, glist, dst=0)  # the one collective of the path (RCCL over xGMI)
            elif world > 1:
                hm = mine_k.cpu()
                hl = [torch.zeros_like(hm) for _ in range(world)] if rank == 0 else None
                dist.gather(hm, hl, dst=0)
                if rank == 0:
                    gathered_k.copy_(torch.cat(hl))
            if rank == 0:
                b = k & 1
                src = gathered_k if world > 1 else mine_k
                rs.wait_event(copy_done[b])  # the copy that last read images[b] has finished
                rt.unpack_tiles_device(src.data_ptr(), pad_tiles, world, W, H, images[b].data_ptr(), stream)
                unpacked = torch.cuda.Event()
                unpacked.record(rs)
                copy_stream.wait_event(unpacked)
                with torch.cuda.stream(copy_stream):
                    host_images[b].copy_(images[b], non_blocking=True)
                    copy_done[b].record(copy_stream)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    sync()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    sync()
    dt = time.perf_counter() - t0
    last_kernel_ms = sc.last_kernel_ms()

    # per-launch kernel duration measured with HIP events on the launch stream: time each launch separately, untimed loop
    per_launch = []
    for _ in range(max(1, min(a.steps, 3))):
        stream = torch.cuda.current_stream().cuda_stream
        sc.render_tiles_device(cam, W, H, spp, depth, a.seed, (rank, world), mine.data_ptr(), None, stream)
        per_launch.append(sc.last_kernel_ms())
    kernel_avg_ms = float(np.mean(per_launch))
    launch_cfg = sc.last_launch_config()

    tmax = torch.tensor([dt, kernel_avg_ms], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax[0])
    kernel_avg_ms = float(tmax[1])

    if rank == 0:
        total_samples = W * H * spp
        value = a.steps * total_samples / dt / 1e6
        # algorithmic bytes per sample from the kernel's own traversal counters,
        # measured on the same image at reduced spp with the counting build of the same kernel
        cspp = min(spp, 64)
        _, cnt = sc.render(cam, W, H, cspp, depth, a.seed, counters=True)
        ns = max(1, cnt["samples"])
        bytes_per_sample = (cnt["nodes_visited"] * info["node_bytes"] + cnt["prims_tested"] * info["prim_bytes"]
                            + cnt["segments"] * info["material_bytes"]) / ns + 24.0 / spp
        launch_samples = total_samples / world
        achieved = bytes_per_sample * launch_samples / (kernel_avg_ms * 1e-3) / 1e9
        # HBM bytes per launch from the PMC counters (separate rocprofv3 --pmc passes of this same command,
        # tools/profile.sh; corrected as MI355X_MICROARCH.md prescribes); only valid for the profiled workload
        traffic, valu = None, None
        tf = ROOT / "profiles" / "pmc_traffic.json"
        if tf.exists() and (a.scene, W, H, spp, depth, world) == ("book_one", 1200, 800, 500, 100, 1):
            pmc = json.load(open(tf))
            traffic = pmc["hbm_bytes_per_launch"]
            valu = pmc.get("valu")  # what actually bounds the kernel (same profile): VALU issue
        res = {
            "metric": "Msamples/sec (pixels x spp), book-one 1200x800x500spp" if a.scene == "book_one" else f"Msamples/sec (pixels x spp), {a.scene}",
            "value": value, "unit": "Msamples/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": (f"book-one random-spheres {W}x{H}, {spp} spp, depth {depth} (BASELINE.json configs[1])"
                                    if a.scene == "book_one" else f"{a.scene} {W}x{H}, {spp} spp, depth {depth}"),
                       "scene_seed": a.scene_seed, "render_seed": a.seed, "n_spheres": info["n_prims"],
                       "bvh_nodes": info["n_nodes"], "sharding": f"8x8 tiles, tile_id % {world}, one RCCL gather",
                       "steps_pipelined_on_two_streams": bool(a.pipeline)},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "valu_pmc": valu,
                         "note": "algorithmic bytes are served from LDS/L1/L2 (scene < 100 KB): the kernel is VALU-issue bound, see DESIGN.md",
                         "kernel": "render_kernel", "kernel_ms": kernel_avg_ms, "launch": launch_cfg, "algorithmic_bytes_per_sample": bytes_per_sample,
                         "segments_per_sample": cnt["segments"] / ns, "nodes_per_sample": cnt["nodes_visited"] / ns,
                         "prims_per_sample": cnt["prims_tested"] / ns,
                         "simd_utilisation": {b: cnt[b + "_lane"] / max(1, 64 * cnt[b + "_wave"]) for b in ("node", "leaf", "shade")},
                         "block_executions_per_sample": {b: cnt[b + "_wave"] * 64 / ns for b in ("node", "leaf", "shade")},
                         "block_cycle_share": {b: cnt[b + "_cycles"] / max(1, cnt["node_cycles"] + cnt["leaf_cycles"] + cnt["shade_cycles"])
                                               for b in ("node", "leaf", "shade", "finish", "refill", "begin", "swap")},
                         "swap_at_shade": ({k[5:]: cnt[k] for k in cnt if k.startswith("swap_")} if cnt.get("swap_scattered") else None)},
            "wall_s": dt, "last_kernel_ms": last_kernel_ms,
        }
        if a.check:
            whole = sc.render(cam, W, H, spp, depth, a.seed)
            last = host_images[(step_no[0] - 1) & 1]
            res["image_matches_single_gpu"] = bool(np.array_equal(last.numpy().reshape(H, W, 3), whole))
        if not a.no_cpu_baseline and world == 1:
            res["cpu_baseline"] = cpu_baseline(desc, W, H, depth, a.seed, a.cpu_seconds)
        else:
            res["cpu_baseline"] = None
        print(json.dumps(res))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
