#!/usr/bin/env python3
"""GPU box, experiment: does the ORDER in which the tiles are handed out change the end-of-launch part of render_kernel?
book-one 1200x800x500, whole image and the 1/8 shard: ascending (default), reversed, and sorted by the mean path length per tile
that the CPU harness measured (profiles/r03_experiments/tile_order_{full,s8}.bin).  -> stdout"""
import importlib
import os
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from __graft_entry__ import load_package  # noqa: E402

rt = load_package()
scenes = importlib.import_module("ray_tracer_amd.scenes")
W, H, spp = 1200, 800, 500
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream().cuda_stream
E = ROOT / "profiles" / "r03_experiments"
for n, f in ((1, "tile_order_full.bin"), (8, "tile_order_s8.bin")):
    cnt = rt.shard_tile_count(W, H, 0, n)
    buf = torch.zeros(cnt * 64 * 3, dtype=torch.float64, device=dev)
    ref = None
    for name, env in (("ascending", {}), ("reversed", {"RT_TILE_ORDER": "reverse"}), ("deepest first", {"RT_TILE_ORDER_FILE": str(E / f)}), ("ascending", {})):
        for k in ("RT_TILE_ORDER", "RT_TILE_ORDER_FILE"):
            os.environ.pop(k, None)
        os.environ.update(env)
        sc, cam = scenes.build_product(scenes.book_one(1, W / H), device=0)
        ts = []
        for _ in range(5):
            sc.render_tiles_device(cam, W, H, spp, 100, 1, (0, n), buf.data_ptr(), None, st)
            torch.cuda.synchronize()
            ts.append(sc.last_kernel_ms())
        img = buf.clone()
        if ref is None:
            ref = img
        same = bool(torch.equal(ref, img))
        print(f"1/{n} {name:14s}: kernel min {min(ts):.3f} median {sorted(ts)[2]:.3f} ms, same pixels as ascending: {same}", flush=True)
        sc.close()
