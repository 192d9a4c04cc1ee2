#!/usr/bin/env python3
"""GPU box: what is a node array in LDS worth against the same array in L2?  (The upper bound of a treelet: the top levels
of a tree that does not fit, VERDICT r2 #5 i.)  book-one (31 KB of nodes, spheres-only family) and a cover scene cut down to
16 floor boxes and 50 small spheres so that its tree fits beside four groups' stacks and queues (general-media family, everything
else as in the cover; the copy leaves the queues fewer entries -- both launch shapes are printed), each
rendered with the node copy and with RT_NO_LDS_NODES=1: render_kernel ms (rt_last_kernel_ms) -> stdout + gpurun_out/lds_nodes_ab.json"""
import importlib
import json
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from __graft_entry__ import load_package  # noqa: E402

rt = load_package()
scenes = importlib.import_module("ray_tracer_amd.scenes")


def small_cover():
    d = scenes.cover(1, 1.0)
    cubes, spheres = d.world[0][1], d.world[-1][1]
    drop = set(cubes) - {cubes[i * 20 + j] for i in range(11, 15) for j in range(3, 7)}  # keep 4 x 4 boxes in front of the camera
    drop |= set(spheres[50:])
    d.sprites = [sp for i, sp in enumerate(d.sprites) if i not in drop]  # the product takes the sprites flat; d.world is the oracle's
    d.world = None
    return d


out = {}
for name, desc, W, H, spp in (("book_one", scenes.book_one(1, 1.5), 1200, 800, 200), ("cover_small", small_cover(), 800, 800, 300)):
    res = {}
    for mode in ("lds", "l2", "lds", "l2"):
        os.environ["RT_NO_LDS_NODES"] = "1" if mode == "l2" else "0"
        sc, cam = scenes.build_product(desc, device=0)
        sc.render(cam, W, H, 8, 100, seed=1)
        sc.render(cam, W, H, spp, 100, seed=1)
        lc = sc.last_launch_config()
        res.setdefault(mode, []).append(round(sc.last_kernel_ms(), 3))
        res[mode + "_launch"] = {k: lc[k] for k in ("lds_nodes", "lds_bytes", "blocks_per_cu", "swap_cap") if k in lc}
        res["nodes"] = sc.info()["n_nodes"]
        sc.close()
    out[name] = res
    print(name, res, flush=True)
(ROOT / "gpurun_out").mkdir(exist_ok=True)
(ROOT / "gpurun_out" / "lds_nodes_ab.json").write_text(json.dumps(out, indent=1))
