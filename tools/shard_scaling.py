#!/usr/bin/env python3
"""Run on the GPU box (1 GPU): time one shard of the headline workload for N = 1, 2, 4, 8 -- the per-GPU work of a
strong-scaling run without the gather -- to predict the driver's multi-GPU scaling.  The first render of a shard learns its tile
order, the timed ones use it (`--ascending`: the ascending order, as before round 4)."""
import importlib
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from __graft_entry__ import load_package  # noqa: E402

rt = load_package()
scenes = importlib.import_module("ray_tracer_amd.scenes")
W, H, spp, depth = 1200, 800, 500, 100
sc, cam = scenes.build_product(scenes.book_one(1, W / H), device=0)
dev = torch.device("cuda", 0)
flags = rt.RT_FLAG_ASCENDING_TILES if "--ascending" in sys.argv else 0
base = None
for n in (1, 2, 4, 8):
    worst, worst_kern, kern = 0.0, 0.0, 0.0
    for r in (0, n - 1):
        cnt = rt.shard_tile_count(W, H, r, n)
        buf = torch.zeros(cnt * 64 * 3, dtype=torch.float64, device=dev)
        st = torch.cuda.current_stream().cuda_stream
        best = 1e9
        for _ in range(4):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            sc.render_tiles_device(cam, W, H, spp, depth, 1, (r, n), buf.data_ptr(), None, st, flags=flags)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            if dt < best:
                best, kern = dt, sc.last_kernel_ms()
        if best > worst:
            worst, worst_kern = best, kern
    base = base or worst
    print(f"N={n}: slowest shard {worst * 1e3:.2f} ms (render_kernel {worst_kern:.2f} ms, ideal {base * 1e3 / n:.2f} ms)"
          f"  -> ideal-gather speedup {base / worst:.2f}x", flush=True)
