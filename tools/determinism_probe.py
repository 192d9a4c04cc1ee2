#!/usr/bin/env python3
"""GPU box: the same launch repeated -- the algorithmic counters and the image must come out identical every time
(scheduling never changes a result).  Prints one line per repetition; exit code 1 on any variation."""
import hashlib
import importlib
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from __graft_entry__ import load_package  # noqa: E402

rt = load_package()
scenes = importlib.import_module("ray_tracer_amd.scenes")
fx = json.load(open(ROOT / "tests" / "golden" / "algo_counts_book_one_1200x800.json"))
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
sc, cam = scenes.build_product(scenes.book_one(fx["scene_seed"], fx["width"] / fx["height"]), device=0)
seen = set()
for counting in (True, False):
    for i in range(reps):
        out = sc.render(cam, fx["width"], fx["height"], fx["spp_measured"], fx["max_depth"], seed=fx["render_seed"], counters=counting)
        img, c = out if counting else (out, None)
        h = hashlib.sha256(img.tobytes()).hexdigest()[:16]
        key = (h, None if c is None else (c["samples"], c["segments"], c["nodes_visited"], c["prims_tested"]))
        seen.add((counting, key))
        print(counting, i, key, flush=True)
print("fixture", (fx["samples"], fx["segments"], fx["node_steps"], fx["prim_tests"]))
imgs = {k[0] for _, k in seen}
cnts = {k[1] for c, k in seen if c}
print("distinct images", len(imgs), "distinct counter sets", len(cnts))
sys.exit(0 if len(imgs) == 1 and len(cnts) == 1 else 1)
