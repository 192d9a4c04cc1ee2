#!/usr/bin/env python3
"""GPU box: render_kernel ms against samples per pixel on the headline image (and on a 1/8 shard): the intercept of the straight
line is the per-launch fixed part (launch, prologue, ramp, end-of-launch tail).  -> stdout"""
import importlib
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from __graft_entry__ import load_package  # noqa: E402

rt = load_package()
scenes = importlib.import_module("ray_tracer_amd.scenes")
W, H = 1200, 800
sc, cam = scenes.build_product(scenes.book_one(1, W / H), device=0)
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream().cuda_stream
for n in (1, 8):
    cnt = rt.shard_tile_count(W, H, 0, n)
    buf = torch.zeros(cnt * 64 * 3, dtype=torch.float64, device=dev)
    xs, ys = [], []
    for spp in (25, 50, 100, 200, 400, 800):
        best = 1e9
        for _ in range(4):
            sc.render_tiles_device(cam, W, H, spp, 100, 1, (0, n), buf.data_ptr(), None, st)
            torch.cuda.synchronize()
            best = min(best, sc.last_kernel_ms())
        xs.append(spp)
        ys.append(best)
        lc = sc.last_launch_config()
        print(f"shard 1/{n} spp {spp:4d}: {best:8.3f} ms  job_spp {lc.get('job_spp')} jobs {lc.get('n_jobs')}", flush=True)
    b, a = np.polyfit(xs[2:], ys[2:], 1)
    print(f"shard 1/{n}: {b * 1000:.3f} us per spp, intercept {a:.3f} ms (fit over spp >= 100)", flush=True)
