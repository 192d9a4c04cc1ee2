#!/usr/bin/env python3
"""GPU box: quick safety check of a new scheduling feature before the test suite: the general scenes at small sizes, first with
the counting build (its watchdog turns a livelock into RT_ERR_DEVICE), then the timed build; both must equal the RT_SWAP=0 image."""
import importlib
import os
import subprocess
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from __graft_entry__ import load_package  # noqa: E402

if len(sys.argv) > 1 and sys.argv[1] == "noswap":
    os.environ["RT_SWAP"] = "0"
rt = load_package()
scenes = importlib.import_module("ray_tracer_amd.scenes")
CASES = (("cornell", scenes.cornell(1.0), 200, 200, 64), ("cover", scenes.cover(1, 1.0), 240, 240, 48), ("instanced", scenes.instanced(1.25), 200, 160, 32),
         ("cover_nofog", scenes.cover(2, 1.0, with_fog=False), 320, 320, 100))
out = {}
for name, desc, W, H, spp in CASES:
    sc, cam = scenes.build_product(desc, device=0)
    if os.environ.get("RT_SWAP") == "0":
        out[name] = sc.render(cam, W, H, spp, 100, seed=2)
    else:
        a, c = sc.render(cam, W, H, spp, 100, seed=2, counters=True)
        print(name, "counting build ok", c["samples"], flush=True)
        b = sc.render(cam, W, H, spp, 100, seed=2)
        assert np.array_equal(a, b), name
        out[name] = b
        print(name, "timed build ok", flush=True)
    sc.close()
np.savez(ROOT / "gpurun_out" / ("xchg_smoke_noswap.npz" if os.environ.get("RT_SWAP") == "0" else "xchg_smoke.npz"), **out)
if os.environ.get("RT_SWAP") != "0":
    subprocess.run([sys.executable, __file__, "noswap"], check=True)
    ref = np.load(ROOT / "gpurun_out" / "xchg_smoke_noswap.npz")
    for name in out:
        assert np.array_equal(out[name], ref[name]), name
    print("all images equal the RT_SWAP=0 renders")
