#!/usr/bin/env python3
"""GPU box: the fixed per-launch part of render_kernel, estimated from the whole headline image (T1) and a 1/8 shard
(T8) as (8*T8 - T1)/7, for several path-depth caps.  If the end-of-launch tail is a few very long paths finishing
alone, it shrinks with the cap."""
import importlib
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from __graft_entry__ import load_package  # noqa: E402

rt = load_package()
scenes = importlib.import_module("ray_tracer_amd.scenes")
W, H, spp = 1200, 800, 500
sc, cam = scenes.build_product(scenes.book_one(1, W / H), device=0)
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream().cuda_stream
for depth in (100, 50, 20, 10, 5):
    t = {}
    for n in (1, 8):
        cnt = rt.shard_tile_count(W, H, 0, n)
        buf = torch.zeros(cnt * 64 * 3, dtype=torch.float64, device=dev)
        best = 1e9
        for _ in range(4):
            sc.render_tiles_device(cam, W, H, spp, depth, 1, (0, n), buf.data_ptr(), None, st)
            torch.cuda.synchronize()
            best = min(best, sc.last_kernel_ms())
        t[n] = best
    print(f"depth {depth:3d}: whole {t[1]:.2f} ms, 1/8 shard {t[8]:.2f} ms (ideal {t[1] / 8:.2f}), fixed part ~ {(8 * t[8] - t[1]) / 7:.2f} ms", flush=True)
