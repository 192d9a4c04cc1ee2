#!/usr/bin/env python3
"""GPU box, diagnostics for tests/test_cover_png.py: how much do the region means of the cover render (no fog,
800x800x1000) move with (a) the colour of the earth sphere this repo has to substitute and (b) the scene seed
(random floor heights, cloud of small spheres)?"""
import importlib
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
from __graft_entry__ import load_package  # noqa: E402

rt = load_package()
scenes = importlib.import_module("ray_tracer_amd.scenes")
import test_cover_png as t  # noqa: E402

REGIONS = ("blue_core", "orange_core", "metal_core")


def stats(desc, seed=3):
    sc, cam = scenes.build_product(desc, device=0)
    img8 = t.to8(sc.render(cam, 800, 800, 1000, 100, seed=seed)[::-1])
    out = {}
    for r in REGIONS:
        x0, y0, x1, y1 = t.FIX[r]["box"]
        out[r] = img8[y0:y1, x0:x1].reshape(-1, 3).astype(float).mean(0)
    return out


orig = scenes.earth_texture
for name, c in (("earth dark ocean", (25, 40, 90)), ("earth white", (250, 250, 250))):
    scenes.earth_texture = lambda w=1024, h=512, c=c: np.broadcast_to(np.array(c, dtype=np.uint8), (h, w, 3)).copy()
    print(name, {k: v.round(1).tolist() for k, v in stats(scenes.cover(1, 1.0, with_fog=False)).items()}, flush=True)
scenes.earth_texture = orig
rows = {r: [] for r in REGIONS}
for scene_seed in range(1, 13):
    s = stats(scenes.cover(scene_seed, 1.0, with_fog=False), seed=100 + scene_seed)
    for r in REGIONS:
        rows[r].append(s[r])
for r in REGIONS:
    a = np.array(rows[r])
    print(r, "over 12 scene seeds: mean", a.mean(0).round(2).tolist(), "std", a.std(0).round(2).tolist(), "min", a.min(0).round(1).tolist(),
          "max", a.max(0).round(1).tolist(), "| cover.png", [round(v, 1) for v in t.FIX[r]["mean"]], flush=True)
