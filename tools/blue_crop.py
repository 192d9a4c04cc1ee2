#!/usr/bin/env python3
"""GPU box: render this repo's restatement of the cover scene (no fog, 800x800, spp from argv) and save the 8-bit
image -> gpurun_out/cover_ours_8bit.npy (for side-by-side crops against the reference's cover.png)."""
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

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
sc, cam = scenes.build_product(scenes.cover(1, 1.0, with_fog=False), device=0)
img = sc.render(cam, 800, 800, spp, 100, seed=3)[::-1]
np.save(ROOT / "gpurun_out" / "cover_ours_linear_crop.npy", img[470:710, 95:335].astype(np.float32))
np.save(ROOT / "gpurun_out" / "cover_ours_8bit.npy", t.to8(img))
print("saved", spp)
