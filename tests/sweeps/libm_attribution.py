#!/usr/bin/env python3
"""GPU box: WHY do the pixels listed in gpurun_out/diff_pixel_probe.json (tests/sweeps/diff_pixel_probe.py) differ from the oracle?
(or the committed profiles/r03_experiments/diff_pixel_probe.json)  For the one sample of each pixel that differs, the host harness (tests/lane_emul.cpp, identical to the oracle on these scenes)
records the arguments its path passed to log / sin / atan2 / acos; tools/microbench/libm_probe.hip evaluates the DEVICE's
functions on the same arguments; the first call whose device result is not the host's bits is where the paths part (a 1-ulp
free-flight distance moves the scatter point, everything after it is a different path).  Also: how often device and host
differ on random arguments of the renderer's ranges.   -> stdout + gpurun_out/libm_attribution.json"""
import importlib
import json
import subprocess
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
from __graft_entry__ import load_package  # noqa: E402

OUT = ROOT / "gpurun_out"
OUT.mkdir(exist_ok=True)
exe = OUT / "libm_probe"
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-ffp-contract=off", "-fno-fast-math", "--offload-arch=gfx950",
                str(ROOT / "tools" / "microbench" / "libm_probe.hip"), "-o", str(exe)], check=True)
subprocess.run(["make", "-C", str(ROOT / "tests")], check=True, capture_output=True)

rt = load_package()
scenes = importlib.import_module("ray_tracer_amd.scenes")
import lane_emul_binding as le  # noqa: E402
from test_random_scenes import camera_scene, random_scene, random_scene_r3, scaled_scene, wide_scene  # noqa: E402

NAMES = ("log", "sin", "atan2", "acos")


def device(fn_a_b):
    """(n, 3) {fn, a, b} -> the device's results"""
    a = np.ascontiguousarray(fn_a_b, dtype=np.float64)
    a.tofile(OUT / "libm_in.bin")
    subprocess.run([str(exe), str(OUT / "libm_in.bin"), str(OUT / "libm_out.bin")], check=True)
    return np.fromfile(OUT / "libm_out.bin", dtype=np.float64)


def ulps(a, b):
    return np.abs(a.view(np.int64) - b.view(np.int64))


res = {"pixels": [], "random_arguments": {}}
probe = OUT / "diff_pixel_probe.json"  # fresh from tests/sweeps/diff_pixel_probe.py in the same call, else the committed copy
if not probe.exists():
    probe = ROOT / "profiles" / "r03_experiments" / "diff_pixel_probe.json"
for c in json.load(open(probe)):
    seed, W, H, spp, x, y = (c[k] for k in ("seed", "W", "H", "spp", "x", "y"))
    gen = c.get("generator", "general")
    d = (scenes.book_one(seed, W / H) if gen == "book_one" else scenes.cover(seed, W / H) if gen == "cover" else wide_scene(scenes, seed) if gen == "wide" else scaled_scene(scenes, seed) if gen == "scaled" else camera_scene(scenes, seed, W / H) if gen == "camera" else
         random_scene_r3(scenes, seed) if seed % 2 else random_scene(scenes, seed))
    sc, cam = scenes.build_product(d, device=-1)
    depth = int(c.get("max_depth", 40))
    t = le.trace_pixel(sc, cam, W, H, spp, depth, seed, x, y)
    entry = {"seed": seed, "x": x, "y": y, "samples_that_differ": c["samples_that_differ"], "calls_of_the_pixel": len(t)}
    # does the lane program on the HOST draw the oracle's pixel?  If not, the difference is not the device's (it is the lane program's)
    import oracle_binding as oracle
    hp, *_ = le.render(sc, cam, W, H, spp, depth, seed, region=(x, y, x + 1, y + 1))
    entry["host_harness_equals_oracle"] = bool(np.array_equal(hp[y, x], np.asarray(c["oracle_pixel"])))
    for s in c["samples_that_differ"]:
        calls = t[t[:, 0] == s]
        dev = device(calls[:, 1:4]) if len(calls) else np.zeros(0)
        host = np.ascontiguousarray(calls[:, 4])
        differ = np.nonzero(dev.view(np.int64) != host.view(np.int64))[0]
        entry["sample_%d" % s] = {"libm_calls": len(calls), "by_function": {NAMES[f]: int((calls[:, 1] == f).sum()) for f in range(4)},
                                  "calls_where_the_device_differs": [
                                      {"index": int(i), "fn": NAMES[int(calls[i, 1])], "a": float(calls[i, 2]), "b": float(calls[i, 3]), "host": float(host[i]),
                                       "device": float(dev[i]), "ulps": int(ulps(dev[i:i + 1], host[i:i + 1])[0])} for i in differ]}
    # the other samples of the pixel, which agree: none of their calls may differ (or the explanation would not hold)
    others = t[~np.isin(t[:, 0], c["samples_that_differ"])]
    if len(others):
        dev = device(others[:, 1:4])
        entry["calls_of_the_agreeing_samples"] = len(others)
        entry["of_them_device_differs"] = int((dev.view(np.int64) != np.ascontiguousarray(others[:, 4]).view(np.int64)).sum())
    print(json.dumps(entry), flush=True)
    res["pixels"].append(entry)

# how often does it happen at all?  keyed free-flight draws are log(u), u in (0, 1); the textures' arguments likewise
import math  # the host's C library (what the oracle and the harness call); numpy's vectorised functions are another implementation

rng = np.random.default_rng(5)
n = 500_000
for f, (a, b) in enumerate(((rng.uniform(0.0, 1.0, n), np.zeros(n)), (rng.uniform(0.0, 20.0 * np.pi, n), np.zeros(n)),
                            (rng.uniform(-1.0, 1.0, n), rng.uniform(-1.0, 1.0, n)), (rng.uniform(-1.0, 1.0, n), np.zeros(n)))):
    dev = device(np.stack([np.full(n, float(f)), a, b], axis=1))
    host = np.array([{0: math.log, 1: math.sin, 3: math.acos}[f](v) for v in a] if f != 2 else [math.atan2(p, q) for p, q in zip(a, b)])
    u = ulps(dev, host)
    res["random_arguments"][NAMES[f]] = {"n": n, "device_differs_from_host": int((u != 0).sum()), "max_ulps": int(u.max())}
    print(NAMES[f], res["random_arguments"][NAMES[f]], flush=True)
json.dump(res, open(OUT / "libm_attribution.json", "w"), indent=1)
