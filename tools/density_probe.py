#!/usr/bin/env python3
"""GPU box, diagnostics for tests/test_cover_png.py: blue-sphere region of the cover render (no fog, 800x800x1000,
mean over 4 scene seeds) as a function of the density of the medium inside it (examples/main.rs:251 says 0.03)."""
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

x0, y0, x1, y1 = t.FIX["blue_core"]["box"]
print("cover.png", [round(v, 1) for v in t.FIX["blue_core"]["mean"]])
for density in (0.005, 0.01, 0.015, 0.02, 0.03, 0.05, 0.1, 0.2):
    acc = []
    for scene_seed in (1, 2, 3, 4):
        d = scenes.cover(scene_seed, 1.0, with_fog=False)
        # the medium geometry record of the blue sphere: ("medium", boundary geometry id, density)
        hits = [i for i, g in enumerate(d.geometries) if g[0] == "medium" and abs(g[2] - 0.03) < 1e-12]
        assert len(hits) == 1, hits
        g = list(d.geometries[hits[0]])
        g[2] = density
        d.geometries[hits[0]] = tuple(g)
        sc, cam = scenes.build_product(d, device=0)
        img8 = t.to8(sc.render(cam, 800, 800, 1000, 100, seed=50 + scene_seed)[::-1])
        acc.append(img8[y0:y1, x0:x1].reshape(-1, 3).astype(float).mean(0))
    a = np.array(acc)
    print("density", density, "blue_core", a.mean(0).round(1).tolist(), "+-", a.std(0).round(1).tolist(), flush=True)
