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
    la = sc.last_launch_config()["lds_nodes"]
    os.environ["RT_SWAP"] = "1"
    b, cb = sc.render(cam, W, H, spp, depth, seed=seed, counters=True)
    lb = sc.last_launch_config()["lds_nodes"]
    # (round 5: without the queues a tree may fit LDS with binary32 planes that with them only fits with binary16 ones -- another
    # tree of culling boxes: the same image and segments; node steps, primitive tests and with them the keyed free-flight draws of a
    # medium, one per medium TEST, follow the boxes)
    same_tree = la == lb
    ok = np.array_equal(a, b) and cb["swap_parked"] == cb["swap_pulled"] and all(
        ca[k] == cb[k] for k in (("samples", "segments", "nodes_visited", "prims_tested", "rng_draws") if same_tree else ("samples", "segments")))
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
# general prims only (rotated cubes / rectangles / spheres, no medium, no texture): the lean kernel family with
# 256-thread groups; odd sizes, shards that recombine, several passes through a small sample workspace
axes = ((1.0, 0.0, 0.0), (0.0, 1.0, 0.0), (0.0, 0.0, 1.0))
for s in range(24):
    rng = np.random.default_rng(1000 + s)
    d = scenes.SceneDesc()
    mats = [d.lambertian_rgb(rng.uniform(0.2, 0.9, 3)) for _ in range(3)] + [d.mat("metal", d.tex_solid((0.8, 0.8, 0.7)), 0.3), d.mat("dielectric", 1.5)]
    for i in range(int(rng.integers(1, 120))):
        M = scenes.mat4_multiplied(scenes.mat4_translation(tuple(rng.uniform(-6, 6, 3))), scenes.mat4_rotation(rng.uniform(0, 3.0), axes[int(rng.integers(3))]))
        g = (d.geom("rectangle", rng.uniform(0.5, 2.0), rng.uniform(0.5, 2.0)), d.geom("sphere", rng.uniform(0.3, 1.0)),
             d.geom("cube", rng.uniform(0.4, 1.5), rng.uniform(0.4, 1.5), rng.uniform(0.4, 1.5)))[int(rng.integers(3))]
        d.sprite(g, mats[int(rng.integers(len(mats)))], M)
    d.sprite(d.geom("sphere", 300.0), d.mat("diffuse_light", d.tex_solid((0.7, 0.8, 1.0))), None)
    d.camera = ((16.0, 6.0, 9.0), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), 0.7, 1.3, 18.0, 0.05 if s % 2 else 0.0)
    sc, cam = scenes.build_product(d, device=0)
    assert sc.info()["feature_mask"] & (rt.RT_FEAT_MEDIUM | rt.RT_FEAT_TEXTURED) == 0
    W, H, spp = int(rng.integers(1, 200)), int(rng.integers(1, 150)), int(rng.integers(1, 20))
    both(sc, cam, W, H, spp, 40, s, "lean")
    whole = sc.render(cam, W, H, spp, 40, seed=s)
    parts = sum(sc.render(cam, W, H, spp, 40, seed=s, shard=(r, 3)) for r in range(3))
    os.environ["RT_SAMPLE_WORKSPACE_MB"] = "1"
    passes = sc.render(cam, W, H, spp, 40, seed=s)
    del os.environ["RT_SAMPLE_WORKSPACE_MB"]
    if not (np.array_equal(whole, parts) and np.array_equal(whole, passes)):
        print("MISMATCH lean shards/passes", s, W, H, spp)
        sys.exit(1)
    n += 2
# full-size headline image at 48 spp (two passes of the job queue per wave at least)
sc, cam = scenes.build_product(scenes.book_one(1, 1.5), device=0)
both(sc, cam, 1200, 800, 48, 100, 1, "book_one full")
print("swap stress: %d render pairs identical, %.1f s" % (n, time.time() - t_start))
