#!/usr/bin/env python3
"""CPU, one-off: the 14 scenes of profiles/r03_random_parity_scaled_6000.json in which the kernels differ
from the oracle at world scale K >= 7e7 -- does the ORACLE agree with itself there?  Each scene is rendered by the oracle
under several `bvh_seed`s (the reference builds its trees with random axes, src/optimize.rs:374-409) and with the world as a
plain ObjectList (src/geometry.rs:76-116), and by the lane program (tests/lane_emul, what the kernels compute).  If the
oracle's own images differ on the recorded pixels, those pixels are decided by the reference's random tree (its binary64
AxisAlignedBoundingBox::hit on a ray lying exactly in a box's boundary plane, src/optimize.rs:61-82), not by the kernels.
-> profiles/r04_boundary_plane_probe.json"""
import importlib
import json
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
from __graft_entry__ import load_package  # noqa: E402

rt = load_package()
scenes = importlib.import_module("ray_tracer_amd.scenes")
import oracle_binding as oracle  # noqa: E402
import lane_emul_binding as lane  # noqa: E402
from test_random_scenes import scaled_scene  # noqa: E402

src = json.load(open(ROOT / "profiles" / "r03_random_parity_scaled_6000.json"))
DEPTH = src["max_depth"]
by_seed = {}
for seed, W, H, spp, x, y, dlt in src["pixels_not_bit_identical"]:
    by_seed.setdefault(seed, {"W": W, "H": H, "spp": spp, "pixels": []})["pixels"].append((x, y))
TREES = [None, 1, 2, 3, 4, 5, 6, 7, 8]  # None = the sweep's own (bvh_seed = scene seed)
out = []
for seed, info in sorted(by_seed.items()):
    W, H, spp = info["W"], info["H"], info["spp"]
    d = scaled_scene(scenes, seed)
    K = float(10.0 ** np.random.default_rng(seed + 999).uniform(-6.0, 9.0))
    imgs = {}
    for t in TREES:
        imgs[f"bvh_seed {seed if t is None else t}"] = oracle.build_oracle(d, bvh_seed=seed if t is None else t).render(W, H, spp, DEPTH, seed=seed, iterative=True, nthreads=8)
    imgs["object list"] = oracle.build_oracle(d, bvh_seed=seed, world="list").render(W, H, spp, DEPTH, seed=seed, iterative=True, nthreads=8)
    sc, cam = scenes.build_product(d, device=-1)  # commit only: the lane program below runs on the CPU
    kern = lane.render(sc, cam, W, H, spp, DEPTH, seed=seed)[0]
    names = list(imgs)
    base = imgs[names[0]]
    rec = {"seed": seed, "K": K, "W": W, "H": H, "spp": spp, "pixels": []}
    # every pixel on which ANY two of the oracle's own images differ
    spread = np.zeros((H, W), dtype=bool)
    for n in names[1:]:
        spread |= (imgs[n] != base).any(axis=2)
    rec["pixels_on_which_the_oracle_differs_from_itself"] = int(spread.sum())
    rec["kernel_program_vs_first_oracle_image_differing_pixels"] = int((kern != base).any(axis=2).sum())
    for (x, y) in info["pixels"]:
        vals = {n: imgs[n][y, x].tolist() for n in names}
        distinct = sorted({tuple(v) for v in vals.values()})
        rec["pixels"].append({"x": x, "y": y, "oracle_distinct_values": len(distinct),
                              "kernel_equals_one_of_them": tuple(kern[y, x].tolist()) in set(distinct),
                              "kernel_value": kern[y, x].tolist(), "oracle_values": [list(v) for v in distinct]})
    # and the other way round: is the kernels' image, on every pixel, one of the values the oracle itself produces?
    allowed = np.zeros((H, W), dtype=bool)
    for n in names:
        allowed |= (imgs[n] == kern).all(axis=2)
    rec["pixels_where_the_kernel_value_is_none_of_the_oracles"] = int((~allowed).sum())
    print(seed, f"K={K:.3g}", "oracle differs from itself on", rec["pixels_on_which_the_oracle_differs_from_itself"], "pixels;",
          "recorded pixels:", [(p["oracle_distinct_values"], p["kernel_equals_one_of_them"]) for p in rec["pixels"]],
          "kernel value none of the oracle's on", rec["pixels_where_the_kernel_value_is_none_of_the_oracles"], flush=True)
    out.append(rec)
json.dump({"source": "profiles/r03_random_parity_scaled_6000.json", "max_depth": DEPTH, "trees": [str(t) for t in TREES] + ["object list"],
           "scenes": out}, open(ROOT / "profiles" / "r04_boundary_plane_probe.json", "w"), indent=1)
