#!/usr/bin/env python3
"""GPU box: the ways to run ONE render must draw the same picture, bit for bit, on random general scenes
(tests/test_random_scenes.py): rt_render  ==  rt_render_sharded over 2-8 clones  ==  rt_render_progressive over random sample
ranges  ==  rt_render under a workspace limit that forces several passes  ==  rt_render without the swap queues (RT_SWAP=0).
Index and scheduling logic only -- no oracle involved.   variant_fuzz.py [N [first seed]] -> stdout + gpurun_out/variant_fuzz.json"""
import importlib
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
from __graft_entry__ import load_package  # noqa: E402

rt = load_package()
scenes = importlib.import_module("ray_tracer_amd.scenes")
from test_random_scenes import random_scene, random_scene_r3  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
FIRST = int(sys.argv[2]) if len(sys.argv) > 2 else 700000
t0 = time.time()
bad = []
counts = {"sharded": 0, "progressive": 0, "passes": 0, "no_swap": 0}
for seed in range(FIRST, FIRST + N):
    rng = np.random.default_rng(seed)
    W, H, spp, depth = int(rng.integers(40, 200)), int(rng.integers(30, 150)), int(rng.integers(3, 40)), int(rng.integers(5, 60))
    d = random_scene_r3(scenes, seed) if seed % 2 else random_scene(scenes, seed)
    os.environ["RT_SWAP"] = "1"
    sc, cam = scenes.build_product(d, device=0)
    base = sc.render(cam, W, H, spp, depth, seed=seed)
    # sharded over clones on the same device
    k = int(rng.integers(2, 9))
    clones = [sc] + [sc.clone(0) for _ in range(k - 1)]
    img = rt.render_sharded(clones, cam, W, H, spp, depth, seed=seed)
    counts["sharded"] += 1
    if not np.array_equal(img, base):
        bad.append((seed, "sharded", k))
    for c in clones[1:]:
        c.close()
    # progressive over random ranges
    cuts = sorted(set([0, spp] + [int(c) for c in rng.integers(1, spp, int(rng.integers(1, 4)))]))
    sums = np.zeros((H, W, 3))
    for a, b in zip(cuts, cuts[1:]):
        sc.render_progressive(cam, W, H, spp, depth, seed, a, b, sums)
    counts["progressive"] += 1
    if not np.array_equal(sums / spp, base):
        bad.append((seed, "progressive", cuts))
    # several passes under a small workspace
    tiles = ((W + 7) // 8) * ((H + 7) // 8)
    sc.set_workspace_limit(int(tiles * 64 * 32 * max(1, spp // int(rng.integers(2, 6)))))
    img = sc.render(cam, W, H, spp, depth, seed=seed)
    counts["passes"] += 1
    if not np.array_equal(img, base):
        bad.append((seed, "passes"))
    sc.close()
    # without the swap queues
    os.environ["RT_SWAP"] = "0"
    sc, cam = scenes.build_product(d, device=0)
    img = sc.render(cam, W, H, spp, depth, seed=seed)
    counts["no_swap"] += 1
    if not np.array_equal(img, base):
        bad.append((seed, "no_swap"))
    sc.close()
    if (seed - FIRST) % 50 == 49:
        print(f"[{seed - FIRST + 1} / {N}] mismatches so far: {len(bad)}", flush=True)
os.environ["RT_SWAP"] = "1"
res = {"scenes": N, "first_seed": FIRST, "compared": counts, "mismatches": bad, "seconds": time.time() - t0}
print(res)
(ROOT / "gpurun_out").mkdir(exist_ok=True)
json.dump(res, open(ROOT / "gpurun_out" / "variant_fuzz.json", "w"), indent=1)
sys.exit(1 if bad else 0)
