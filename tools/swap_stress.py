#!/usr/bin/env python3
"""One-off stress of the swap-at-shade queues on the GPU box: many scenes / sizes, swap on vs off must agree bit for
bit, and every parked path must be pulled again.  Prints one summary line; exit code 1 on the first mismatch."""
import importlib
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
from test_random_scenes import random_scene  # noqa: E402

t_start = time.time()
n = 0


def both(sc, cam, W, H, spp, depth, seed, tag):
    global n
    os.environ["RT_SWAP"] = "0"
    a, ca = sc.render(cam, W, H, spp, depth, seed=seed, counters=True)
    os.environ["RT_SWAP"] = "1"
    b, cb = sc.render(cam, W, H, spp, depth, seed=seed, counters=True)
    ok = np.array_equal(a, b) and cb["swap_parked"] == cb["swap_pulled"] and all(
        ca[k] == cb[k] for k in ("samples", "segments", "nodes_visited", "prims_tested", "rng_draws"))
    n += 1
    if not ok:
        print("MISMATCH", tag, W, H, spp, depth, seed, "max diff", np.abs(a - b).max(), {k: (ca[k], cb[k]) for k in ca if ca[k] != cb[k]})
        sys.exit(1)


# random general scenes (matrices, cubes, media, textures, lens), many shapes
for s in range(60):
    d = random_scene(scenes, s)
    sc, cam = scenes.build_product(d, device=0)
    rng = np.random.default_rng(s)
    both(sc, cam, int(rng.integers(1, 160)), int(rng.integers(1, 120)), int(rng.integers(1, 24)), int(rng.integers(1, 60)), s, "random")
# the three book scenes at several sizes, repeated (timing-dependent interleavings differ run to run)
for name, mk in (("book_one", lambda a: scenes.book_one(1, a)), ("cornell", lambda a: scenes.cornell(a)), ("cover", lambda a: scenes.cover(1, a))):
    for (W, H, spp) in ((64, 64, 8), (200, 120, 16), (333, 211, 7), (640, 400, 24)):
        sc, cam = scenes.build_product(mk(W / H), device=0)
        for rep in range(3):
            both(sc, cam, W, H, spp, 100, 11 + rep, name)
# full-size headline image at 48 spp (two passes of the job queue per wave at least)
sc, cam = scenes.build_product(scenes.book_one(1, 1.5), device=0)
both(sc, cam, 1200, 800, 48, 100, 1, "book_one full")
print("swap stress: %d render pairs identical, %.1f s" % (n, time.time() - t_start))
