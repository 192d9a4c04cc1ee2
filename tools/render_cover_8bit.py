#!/usr/bin/env python3
"""GPU box: the book-two cover scene at the reference's size (800x800, 1000 spp, depth 100), with and without the
fog sprite, converted like examples/main.rs:113-121 (min(sqrt(c) * 255, 255) as u8), rows top-down -> gpurun_out/cover_8bit.npz"""
import importlib
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from __graft_entry__ import load_package  # noqa: E402

rt = load_package()
scenes = importlib.import_module("ray_tracer_amd.scenes")


def to8(c):
    with np.errstate(invalid="ignore"):
        v = np.minimum(np.sqrt(c) * 255.0, 255.0)
    return np.where(np.isnan(v), 0, v).astype(np.uint8)


out = {}
for fog in (False, True):
    for seed in (1, 2):
        sc, cam = scenes.build_product(scenes.cover(seed, 1.0, with_fog=fog), device=0)
        img = sc.render(cam, 800, 800, 1000, 100, seed=3)
        out["fog%d_seed%d" % (fog, seed)] = to8(img[::-1])
        print(fog, seed, float(img.mean()), flush=True)
np.savez_compressed(ROOT / "gpurun_out" / "cover_8bit.npz", **out)
