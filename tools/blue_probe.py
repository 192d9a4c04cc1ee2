#!/usr/bin/env python3
"""GPU box: can the random floor explain the blue sphere of the reference's cover.png?

cover.png (examples/main.rs, unseeded) shows the glass-shell + density-0.03 medium sphere at 22.2 44.0 89.2 (mean
8-bit levels over tests/golden/cover_png_regions.json "blue_core"); this repo's restatement gave 23.6 41.2 88.1 +- (0.6,
0.8, 1.0) over 12 scene seeds.  The only random input near that sphere are the heights U[1,101) of the 100 x 100 floor
boxes (examples/main.rs:166-172): the sphere (centre (360,150,145), r 70) hangs 80 above y = 0, so the taller boxes reach
into it.  This probe renders the scene (no fog, 800x800x1000, as the picture) with
  (1) every box at one height (1, 26, 51, 76, 100.99),
  (2) scene seed 1 with ONE of the 5x5 boxes around the sphere set to 1 / 100.99,
  (3) four render seeds of one scene (noise floor of a region mean),
and prints the region means next to the picture's -> gpurun_out/blue_probe.json."""
import importlib
import json
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

REGIONS = ("blue_core", "blue_small", "orange_core", "metal_core", "floor_bottom")
SPP = int(sys.argv[1]) if len(sys.argv) > 1 else 1000


def stats(desc, seed=3):
    sc, cam = scenes.build_product(desc, device=0)
    img8 = t.to8(sc.render(cam, 800, 800, SPP, 100, seed=seed)[::-1])
    sc.close()
    out = {}
    for r in REGIONS:
        x0, y0, x1, y1 = t.FIX[r]["box"]
        out[r] = [round(float(v), 3) for v in img8[y0:y1, x0:x1].reshape(-1, 3).astype(float).mean(0)]
    return out


res = {"spp": SPP, "picture": {r: t.FIX[r]["mean"] for r in REGIONS}, "uniform": {}, "single_box": {}, "render_seeds": []}
nan = float("nan")
for h in (1.0, 26.0, 51.0, 76.0, 100.99):
    res["uniform"][str(h)] = stats(scenes.cover(1, 1.0, with_fog=False, heights=np.full((20, 20), h)))
    print("uniform", h, res["uniform"][str(h)]["blue_core"], flush=True)
for seed in (3, 4, 5, 6):
    res["render_seeds"].append(stats(scenes.cover(1, 1.0, with_fog=False), seed=seed))
    print("render seed", seed, res["render_seeds"][-1]["blue_core"], flush=True)
# sphere centre x 360 -> i = 13 (x in [300,400)), z 145 -> j = 11 (z in [100,200))
for i in range(11, 16):
    for j in range(9, 14):
        for h in (1.0, 100.99):
            hs = np.full((20, 20), nan)
            hs[i][j] = h
            k = f"{i},{j},{h}"
            res["single_box"][k] = stats(scenes.cover(1, 1.0, with_fog=False, heights=hs))
            print("box", k, res["single_box"][k]["blue_core"], flush=True)
json.dump(res, open(ROOT / "gpurun_out" / "blue_probe.json", "w"), indent=1)
print("picture", res["picture"]["blue_core"])
