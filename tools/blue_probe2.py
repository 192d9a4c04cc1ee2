#!/usr/bin/env python3
"""GPU box: height scan of the floor boxes right under the blue sphere (see tools/blue_probe.py).
The sphere's lowest point is y = 80; a box of height ~75-80 below it catches the light the ball focuses and sends it back."""
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

BOXES = {"blue_core": t.FIX["blue_core"]["box"], "blue_small": t.FIX["blue_small"]["box"], "ball_mid": (180, 560, 250, 600),
         "ball_low": (180, 620, 250, 660), "ball_bottom": (190, 670, 240, 690), "ball_top": (180, 500, 250, 540)}
pic = {"blue_core": [22.21, 43.98, 89.25], "blue_small": [18.46, 40.1, 86.24], "ball_mid": [20.1, 42.7, 88.6], "ball_low": [23.8, 42.3, 88.4],
       "ball_bottom": [52.2, 81.8, 150.2], "ball_top": [58.3, 80.7, 125.3]}
SPP = int(sys.argv[1]) if len(sys.argv) > 1 else 1000


def stats(hs, seed=3):
    sc, cam = scenes.build_product(scenes.cover(1, 1.0, with_fog=False, heights=hs), device=0)
    img8 = t.to8(sc.render(cam, 800, 800, SPP, 100, seed=seed)[::-1])
    sc.close()
    return {r: [round(float(v), 2) for v in img8[y0:y1, x0:x1].reshape(-1, 3).astype(float).mean(0)] for r, (x0, y0, x1, y1) in BOXES.items()}


nan = float("nan")
res = {"picture": pic, "scan": {}}
print("picture", pic, flush=True)
for (i, j) in ((13, 11), (12, 11), (13, 10), (13, 12), (14, 11), (12, 10)):
    for h in (20, 40, 60, 70, 75, 79, 82, 86, 92):
        hs = np.full((20, 20), nan)
        hs[i][j] = float(h)
        k = f"{i},{j},{h}"
        res["scan"][k] = stats(hs)
        print(k, res["scan"][k], flush=True)
# all four boxes under the ball at once
for h in (60, 70, 76, 79, 84):
    hs = np.full((20, 20), nan)
    for (i, j) in ((13, 11), (12, 11), (13, 10), (12, 10), (13, 12), (14, 11)):
        hs[i][j] = float(h)
    res["scan"][f"six,{h}"] = stats(hs)
    print("six", h, res["scan"][f"six,{h}"], flush=True)
json.dump(res, open(ROOT / "gpurun_out" / "blue_probe2.json", "w"), indent=1)
