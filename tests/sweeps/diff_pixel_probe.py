#!/usr/bin/env python3
"""GPU box: the pixels of tests/sweeps/random_parity.py's sweep that are not bit-identical to the oracle (profiles/r03_random_parity.json,
`pixels_not_bit_identical`), looked at one by one: is the GPU's value the same from the timed build, the counting build and the
build without swap queues (then it is arithmetic, not scheduling); which SAMPLE of the pixel differs (per-sample radiance through
rt_render_progressive, one sample per pass, against the oracle's orc_render_pixel_samples) and by how much.
-> stdout + gpurun_out/diff_pixel_probe.json"""
import importlib
import json
import os
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
from test_random_scenes import camera_scene, random_scene, random_scene_r3, scaled_scene, wide_scene  # noqa: E402

# the sweep to look at: profiles/r03_random_parity.json, or the file named on the command line
sweep = json.load(open(sys.argv[1] if len(sys.argv) > 1 else ROOT / "profiles" / "r03_random_parity.json"))
cases, DEPTH, GEN = sweep["pixels_not_bit_identical"], int(sweep.get("max_depth", 40)), sweep.get("generator", "general")


def make_scene(seed, aspect):
    if GEN == "book_one":
        return scenes.book_one(seed, aspect)
    if GEN == "cover":
        return scenes.cover(seed, aspect)
    if GEN == "wide":
        return wide_scene(scenes, seed)
    if GEN == "scaled":
        return scaled_scene(scenes, seed)
    if GEN == "camera":
        return camera_scene(scenes, seed, aspect)
    return random_scene_r3(scenes, seed) if seed % 2 else random_scene(scenes, seed)
out = []
for seed, W, H, spp, x, y, diff in cases:
    d = make_scene(seed, W / H)
    orc = oracle.build_oracle(d, bvh_seed=seed)
    want = orc.pixel_samples(W, H, spp, DEPTH, seed, x, y, iterative=True)
    res = {"seed": seed, "W": W, "H": H, "spp": spp, "x": x, "y": y, "max_depth": DEPTH, "generator": GEN}
    for form in ("timed", "counting", "no_swap"):
        os.environ["RT_SWAP"] = "0" if form == "no_swap" else "1"
        sc, cam = scenes.build_product(d, device=0)
        if form == "counting":
            img, _ = sc.render(cam, W, H, spp, DEPTH, seed=seed, counters=True)
        else:
            img = sc.render(cam, W, H, spp, DEPTH, seed=seed)
        res[form + "_pixel"] = img[y, x].tolist()
        if form == "timed":
            sums, prev, got = np.zeros((H, W, 3)), np.zeros(3), []
            for s in range(spp):  # one sample per pass: the running sum's increments are the samples (exact while the sums are small)
                sc.render_progressive(cam, W, H, spp, DEPTH, seed, s, s + 1, sums)
                got.append((sums[y, x] - prev).tolist())
                prev = sums[y, x].copy()
            got = np.array(got)
            bad = [int(s) for s in range(spp) if not np.allclose(got[s], want[s], rtol=1e-12, atol=1e-15)]
            res["samples_that_differ"] = bad
            res["gpu_samples"] = got[bad].tolist()
            res["oracle_samples"] = want[bad].tolist()
        sc.close()
    os.environ["RT_SWAP"] = "1"
    res["oracle_pixel"] = orc.render(W, H, spp, DEPTH, seed=seed, region=(x, y, x + 1, y + 1), iterative=True)[y, x].tolist()
    res["all_forms_agree"] = res["timed_pixel"] == res["counting_pixel"] == res["no_swap_pixel"]
    print(json.dumps(res), flush=True)
    out.append(res)
(ROOT / "gpurun_out").mkdir(exist_ok=True)
json.dump(out, open(ROOT / "gpurun_out" / "diff_pixel_probe.json", "w"), indent=1)
