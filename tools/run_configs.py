#!/usr/bin/env python3
"""Run on the GPU box: every BASELINE.json config at full size through the C ABI (1 GPU) -> gpurun_out/configs.json."""
import importlib
import json
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from __graft_entry__ import load_package  # noqa: E402

rt = load_package()
scenes = importlib.import_module("ray_tracer_amd.scenes")
dev = torch.device("cuda", 0)
out = []


def run(name, desc, W, H, spp, depth, shard=(0, 1), reps=2):
    sc, cam = scenes.build_product(desc, device=0)
    n = rt.shard_tile_count(W, H, *shard)
    buf = torch.zeros(n * 64 * 3, dtype=torch.float64, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    best = None
    for _ in range(reps + 1):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        sc.render_tiles_device(cam, W, H, spp, depth, 1, shard, buf.data_ptr(), None, stream)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    samples = n * 64 * spp if shard[1] > 1 else W * H * spp
    rec = {"config": name, "width": W, "height": H, "spp": spp, "depth": depth, "shard": list(shard), "samples": samples,
           "seconds": best, "Msamples_per_s": samples / best / 1e6, "render_kernel_ms": sc.last_kernel_ms(),
           "finite": bool(torch.isfinite(buf).all()), "mean": float(buf.mean())}
    print(rec, flush=True)
    out.append(rec)


run("configs[0] book-one 400x225x50 depth 50", scenes.book_one(1, 400 / 225), 400, 225, 50, 50)
run("configs[1] book-one 1200x800x500", scenes.book_one(1, 1.5), 1200, 800, 500, 100)
run("configs[2] cornell-box 600x600x1000", scenes.cornell(1.0), 600, 600, 1000, 100)
run("configs[3] book-two cover 800x800x1000", scenes.cover(1, 1.0), 800, 800, 1000, 100)
run("configs[4] book-one 3840x2160x2000, shard 0 of 8", scenes.book_one(1, 3840 / 2160), 3840, 2160, 2000, 100, shard=(0, 8), reps=1)
json.dump(out, open(ROOT / "gpurun_out" / "configs.json", "w"), indent=1)
