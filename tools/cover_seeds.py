#!/usr/bin/env python3
"""GPU box: the book-two cover scene (no fog, as the reference's cover.png) at 800x800x1000 for N scene seeds, converted like
examples/main.rs:113-121, rows top-down -> gpurun_out/cover_seeds.npz (uint8, one array per seed, + two render seeds of scene 1).
tests/golden/make_cover_stats.py turns them into the seed spread (sigma) of every region the picture is compared on."""
import importlib
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from __graft_entry__ import load_package  # noqa: E402

rt = load_package()
scenes = importlib.import_module("ray_tracer_amd.scenes")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 12


def to8(c):
    with np.errstate(invalid="ignore"):
        v = np.sqrt(c) * 255.0
    return np.where(np.isnan(v), 255.0, np.minimum(v, 255.0)).astype(np.uint8)


out = {}
for seed in range(1, N + 1):
    sc, cam = scenes.build_product(scenes.cover(seed, 1.0, with_fog=False), device=0)
    out[f"scene{seed}"] = to8(sc.render(cam, 800, 800, 1000, 100, seed=3)[::-1])
    if seed == 1:
        out["scene1_render4"] = to8(sc.render(cam, 800, 800, 1000, 100, seed=4)[::-1])
        out["scene1_4000spp"] = to8(sc.render(cam, 800, 800, 4000, 100, seed=5)[::-1])
    sc.close()
    print("scene seed", seed, flush=True)
np.savez_compressed(ROOT / "gpurun_out" / "cover_seeds.npz", **out)
print("wrote cover_seeds.npz")
