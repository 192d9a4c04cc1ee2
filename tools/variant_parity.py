#!/usr/bin/env python3
"""GPU box: a tuning variant of the library (RT_MI355X_LIB) must draw the pictures of the in-tree build, bit for bit.
  variant_parity.py render NAME   -> gpurun_out/variant_NAME.npz: small renders of the scene families, counting build first (its
                                     watchdog turns a livelock of a scheduling change into RT_ERR_DEVICE), then the timed build
  variant_parity.py compare A B.. -> every image of B.. equal to A's"""
import importlib
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
OUT = ROOT / "gpurun_out"

if sys.argv[1] == "compare":
    ref = np.load(OUT / f"variant_{sys.argv[2]}.npz")
    for other in sys.argv[3:]:
        got = np.load(OUT / f"variant_{other}.npz")
        for k in ref.files:
            assert np.array_equal(ref[k], got[k]), (other, k, int((ref[k] != got[k]).sum()))
        print(other, "equals", sys.argv[2], "on", len(ref.files), "images", flush=True)
    sys.exit(0)

from __graft_entry__ import load_package  # noqa: E402

rt = load_package()
scenes = importlib.import_module("ray_tracer_amd.scenes")
CASES = (("book_one", scenes.book_one(1, 1.5), 300, 200, 32), ("cornell", scenes.cornell(1.0), 200, 200, 64), ("cover", scenes.cover(1, 1.0), 240, 240, 48),
         ("instanced", scenes.instanced(1.25), 200, 160, 32), ("cover_nofog", scenes.cover(2, 1.0, with_fog=False), 320, 320, 64),
         ("deep_chains", scenes.deep_chains(), 160, 120, 32), ("nested_media", scenes.nested_media(), 160, 120, 32))
out = {}
print(rt.version(), flush=True)
for name, desc, W, H, spp in CASES:
    sc, cam = scenes.build_product(desc, device=0)
    a, c = sc.render(cam, W, H, spp, 100, seed=2, counters=True)
    b = sc.render(cam, W, H, spp, 100, seed=2)
    assert np.array_equal(a, b), name
    out[name] = b
    print(name, "counting and timed build agree;", "nodes/seg %.2f prims/seg %.2f" % (c["nodes_visited"] / c["segments"], c["prims_tested"] / c["segments"]), flush=True)
    sc.close()
OUT.mkdir(exist_ok=True)
np.savez(OUT / f"variant_{sys.argv[2]}.npz", **out)
