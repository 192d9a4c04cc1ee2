#!/usr/bin/env python3
"""GPU box, round 5: what a small TREE scene gains from keeping its records (transforms, prims, materials) in LDS
(rt_launch_config.records_in_lds; RT_NO_LDS_RECORDS=1 switches it off): render_kernel ms of scenes.cube_row(5) (32 leaves, 62 transform
levels) and of the 42-leaf scene of tests/test_gpu_parity.py, 600 x 600 x 200 spp, interleaved."""
import importlib
import json
import os
import statistics
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
from __graft_entry__ import load_package  # noqa: E402

rt = load_package()
scenes = importlib.import_module("ray_tracer_amd.scenes")
import torch  # noqa: E402
from test_gpu_parity import _forty_leaves  # noqa: E402

W = H = 600
spp = 200
out = {"library": rt.version()}
for d in (scenes.cube_row(5), _forty_leaves(scenes, False), _forty_leaves(scenes, True)):
    res = {}
    for rnd in range(2):
        for mode in ("lds", "global"):
            if mode == "global":
                os.environ["RT_NO_LDS_RECORDS"] = "1"
            sc, cam = scenes.build_product(d, device=0)
            buf = torch.zeros(rt.shard_tile_count(W, H, 0, 1) * 64 * 3, dtype=torch.float64, device="cuda:0")
            ms = []
            for _ in range(5):
                sc.render_tiles_device(cam, W, H, spp, 100, 1, (0, 1), buf.data_ptr(), None, None)
                torch.cuda.synchronize()
                ms.append(sc.last_kernel_ms())
            lc = sc.last_launch_config()
            res.setdefault(mode, []).append({"ms": statistics.median(ms[1:]), "records_in_lds": lc["records_in_lds"], "lds_nodes": lc["lds_nodes"],
                                             "block": lc["block_threads"], "lds_bytes": lc["lds_bytes"], "swap_cap": lc["swap_cap"]})
            os.environ.pop("RT_NO_LDS_RECORDS", None)
    out[d.name] = res
    print(d.name, {m: [round(r["ms"], 2) for r in v] for m, v in res.items()}, res["lds"][0], flush=True)
json.dump(out, open(ROOT / "gpurun_out" / "r05_ab_small_trees.json", "w"), indent=1)
