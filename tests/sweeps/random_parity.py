#!/usr/bin/env python3
"""GPU box, one-off: many random general scenes (tests/test_random_scenes.py: non-rigid matrices, cubes, media,
textures, lens) through the C ABI against the oracle's iterative form.  -> gpurun_out/random_parity.json"""
import importlib
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
from __graft_entry__ import load_package  # noqa: E402

rt = load_package()
scenes = importlib.import_module("ray_tracer_amd.scenes")
import oracle_binding as oracle  # noqa: E402
from test_random_scenes import camera_scene, cubes_scene, random_scene, random_scene_r3, scaled_scene, wide_scene  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
FIRST = int(sys.argv[2]) if len(sys.argv) > 2 else 100  # first scene seed
DEPTH = int(sys.argv[3]) if len(sys.argv) > 3 else 40   # maxDepth of the renders
SCALE = int(os.environ.get("RANDOM_PARITY_SCALE", "1"))  # image edges and spp times this: many waves, full swap queues
GEN = sys.argv[4] if len(sys.argv) > 4 else "general"   # general: tests/test_random_scenes.py; book_one / cover: the example scenes by scene seed


def make_scene(seed, aspect):
    if GEN == "book_one":
        return scenes.book_one(seed, aspect)
    if GEN == "cover":
        return scenes.cover(seed, aspect)
    if GEN == "cubes":  # axis-aligned cubes: cube groups, the binary16 tree (RT_HALF_NODES=1 walks it whenever it exists)
        # (scene seeds with seed % 4 < 2 only: the others snap the cubes to a grid, faces coincide, and at an exact tie the reference's
        # answer depends on its random tree -- tests/test_random_scenes.py holds those to "every structure, one image" instead)
        d = cubes_scene(scenes, (seed // 2) * 4 + seed % 2)
        d.camera = d.camera[:4] + (aspect,) + d.camera[5:]
        return d
    if GEN == "wide":
        return wide_scene(scenes, seed)
    if GEN == "scaled":
        return scaled_scene(scenes, seed)
    if GEN == "camera":
        return camera_scene(scenes, seed, aspect)
    return random_scene_r3(scenes, seed) if seed % 2 else random_scene(scenes, seed)
threads = min(256, os.cpu_count() or 8)
t0 = time.time()
worst_mae, worst_max, exact, bad_pixels, pixels = 0.0, 0.0, 0, 0, 0
hist = {}
over_bar = []
differing = []  # (seed, W, H, spp, x, y, |diff|) of every pixel that is not bit-identical: candidates for a device-libm branch flip
for seed in range(FIRST, FIRST + N):
    # every second scene also carries deep transform chains and media inside the boundary of media (round 3)
    rng = np.random.default_rng(seed)
    W, H, spp = int(rng.integers(24, 96)) * SCALE, int(rng.integers(16, 72)) * SCALE, int(rng.integers(2, 12)) * SCALE
    d = make_scene(seed, W / H)
    sc, cam = scenes.build_product(d, device=0)
    img = sc.render(cam, W, H, spp, DEPTH, seed=seed)
    ref = oracle.build_oracle(d, bvh_seed=seed).render(W, H, spp, DEPTH, seed=seed, iterative=True, nthreads=threads)
    diff = np.abs(img - ref)
    finite = np.isfinite(ref).all() and np.isfinite(img).all()
    mae = float(diff.mean()) if finite else float("nan")
    nbad = int((diff.max(axis=2) > 1e-12).sum())
    worst_mae = max(worst_mae, mae)
    worst_max = max(worst_max, float(diff.max()))
    exact += int(np.array_equal(img, ref))
    bad_pixels += nbad
    pixels += W * H
    hist[nbad] = hist.get(nbad, 0) + 1
    for yy, xx in zip(*np.nonzero(diff.max(axis=2) > 0.0)):
        differing.append([seed, W, H, spp, int(xx), int(yy), float(diff[yy, xx].max())])
    every = 1000 if (SCALE == 1 and GEN != "wide") else 20
    if (seed - FIRST) % every == every - 1:  # a sign of life for long sweeps (gpurun takes seven silent minutes for a hang)
        (ROOT / "gpurun_out").mkdir(exist_ok=True)
        (ROOT / "gpurun_out" / "random_parity_progress.txt").write_text(f"{seed - FIRST + 1} of {N} scenes, {exact} bit-identical, {time.time() - t0:.0f} s\n")
        print(f"[{seed - FIRST + 1} / {N}] bit-identical {exact}", flush=True)
    if not finite or mae > 1e-4:
        print("OVER THE BAR: seed", seed, W, H, spp, mae, nbad, flush=True)
        over_bar.append([seed, W, H, spp, mae, nbad])
        if GEN == "general":
            sys.exit(1)
# A pixel that differs from the oracle may be one on which the oracle differs from ITSELF: the reference builds its trees with
# random axes (src/optimize.rs:374-409) and for a ray lying in a box's boundary plane the boxes of a pair of siblings decide
# otherwise than either's own (rt_lane.h ref_box_hit).  Every scene with differing pixels is rendered again under eight other
# trees: a pixel whose kernel value is the oracle's under one of them is the reference's own tree dependence, not a deviation.
explained, unexplained = [], []
for seed in sorted({p[0] for p in differing}):
    rng = np.random.default_rng(seed)
    W, H, spp = int(rng.integers(24, 96)) * SCALE, int(rng.integers(16, 72)) * SCALE, int(rng.integers(2, 12)) * SCALE
    d = make_scene(seed, W / H)
    sc, cam = scenes.build_product(d, device=0)
    img = sc.render(cam, W, H, spp, DEPTH, seed=seed)
    others = [oracle.build_oracle(d, bvh_seed=t).render(W, H, spp, DEPTH, seed=seed, iterative=True, nthreads=threads) for t in range(1, 9)]
    for p in differing:
        if p[0] != seed:
            continue
        x, y = p[4], p[5]
        (explained if any(np.array_equal(o[y, x], img[y, x]) for o in others) else unexplained).append(p)
res = {"scenes": N, "first_seed": FIRST, "max_depth": DEPTH, "generator": GEN, "scale": SCALE, "of_them_with_deep_chains_and_nested_media": N // 2 if GEN == "general" else 0, "bit_identical_scenes": exact, "pixels": pixels, "pixels_differing_by_more_than_1e-12": bad_pixels,
       "worst_mean_abs_error": worst_mae, "worst_abs_diff": worst_max, "differing_pixels_per_scene_histogram": {str(k): v for k, v in sorted(hist.items())},
       "scenes_over_the_bar": over_bar, "pixels_not_bit_identical": differing,
       "of_them_equal_to_the_oracle_under_another_of_8_trees": len(explained), "of_them_unexplained": unexplained, "seconds": time.time() - t0, "bar": "mean abs error <= 1e-4 per scene"}
print(res)
json.dump(res, open(ROOT / "gpurun_out" / ("random_parity.json" if (FIRST, DEPTH, GEN) == (100, 40, "general") else f"random_parity_{GEN}_from_{FIRST}_depth_{DEPTH}" + (f"_x{SCALE}" if SCALE != 1 else "") + ".json"), "w"), indent=1)
