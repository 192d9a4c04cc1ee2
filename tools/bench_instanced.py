#!/usr/bin/env python3
"""GPU box: throughput of the kernel family with media over general boundaries (scenes.instanced: instanced nodes up to four
transform levels, media over a cube / a node / behind a TransformedGeometry) -- not a BASELINE config, a reference point."""
import importlib
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from __graft_entry__ import load_package  # noqa: E402

rt = load_package()
scenes = importlib.import_module("ray_tracer_amd.scenes")
W, H, spp = 800, 640, 256
sc, cam = scenes.build_product(scenes.instanced(W / H), device=0)
print(sc.info())
for _ in range(3):
    t0 = time.perf_counter()
    img, c = sc.render(cam, W, H, spp, 100, seed=1, counters=False), None
    dt = time.perf_counter() - t0
    print(f"instanced {W}x{H}x{spp}: {W * H * spp / dt / 1e6:.1f} Msamples/s wall, render_kernel {sc.last_kernel_ms():.1f} ms "
          f"({W * H * spp / sc.last_kernel_ms() / 1e3:.1f} Msamples/s), launch {sc.last_launch_config()}", flush=True)
_, c = sc.render(cam, W, H, 16, 100, seed=1, counters=True)
n = c["samples"]
print({k: round(c[k] / n, 2) for k in ("segments", "nodes_visited", "prims_tested")},
      {b: round(c[b + "_lane"] / max(1, 64 * c[b + "_wave"]), 2) for b in ("node", "leaf", "shade")})
