#!/usr/bin/env python3
"""CPU (oracle only): does an earlier form of ConstantMedium::hit explain the blue sphere of the reference's cover.png?

Renders the blue-sphere crop of the cover scene (no fog, picture coordinates x 95..335, y 470..710) with the oracle
under each ORC_HYP_* flag and compares region means / saves 8-bit crops.  Usage: blue_hypotheses.py <spp> [x0 y0 x1 y1]"""
import importlib
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import os  # noqa: E402
import subprocess  # noqa: E402
subprocess.run(["make", "-C", str(ROOT / "oracle"), "hyp"], check=True, capture_output=True)
os.environ["ORC_LIB"] = str(ROOT / "oracle" / "_build" / "librt_oracle_hyp.so")  # the probe-only library
from __graft_entry__ import load_package  # noqa: E402

load_package()
scenes = importlib.import_module("ray_tracer_amd.scenes")
import oracle_binding as ob  # noqa: E402
import test_cover_png as t  # noqa: E402

ob._sig("orc_set_hypothesis", None, [ob._VP, ob.C.c_uint])
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 100
box = tuple(int(v) for v in sys.argv[2:6]) if len(sys.argv) >= 6 else (95, 470, 335, 710)
x0, y0, x1, y1 = box
W = H = 800
out_dir = Path(sys.argv[6]) if len(sys.argv) > 6 else Path("/tmp/blue")
out_dir.mkdir(exist_ok=True)
pic = np.load("/tmp/blue/picture.npy") if Path("/tmp/blue/picture.npy").exists() else None
orc = ob.build_oracle(scenes.cover(1, 1.0, with_fog=False))
res = {}
HYPS = (("current", 0), ("inside_none", 4), ("inside_t_adds_t1", 1), ("iso_unnormalized", 8), ("no_inside_fresnel", 16), ("no_fresnel", 32),
        ("schlick_outside_angle", 64))
only = [h for h in sys.argv[7].split(",")] if len(sys.argv) > 7 else None
for name, flags in HYPS:
    if only and name not in only:
        continue
    ob.LIB.orc_set_hypothesis(orc.h, flags)
    t0 = time.time()
    img = orc.render(W, H, spp, 100, seed=5, region=(x0, H - y1, x1, H - y0), nthreads=8)
    px = t.to8(img[H - y1:H - y0, x0:x1][::-1])
    np.save(out_dir / f"hyp_{name}.npy", px)
    # blue_core of the fixture, clipped to the rendered box
    bx0, by0, bx1, by1 = t.FIX["blue_core"]["box"]
    cx0, cy0, cx1, cy1 = max(bx0, x0), max(by0, y0), min(bx1, x1), min(by1, y1)
    m = px[cy0 - y0:cy1 - y0, cx0 - x0:cx1 - x0].reshape(-1, 3).astype(float).mean(0)
    pm = pic[cy0:cy1, cx0:cx1].reshape(-1, 3).astype(float).mean(0) if pic is not None else None
    res[name] = {"mean": m.round(2).tolist(), "picture": None if pm is None else pm.round(2).tolist(), "s": round(time.time() - t0, 1)}
    print(name, res[name], flush=True)
json.dump(res, open(out_dir / "hyp_means.json", "w"), indent=1)
