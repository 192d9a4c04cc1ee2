#!/usr/bin/env python3
"""GPU box: who idles while the node block runs?  Counting build of render_kernel on the three BASELINE scenes (reduced spp):
lanes at a node / waiting at a leaf for the leaf quorum / finished and waiting for the shade quorum / without a path, per
node-block execution -> stdout + gpurun_out/idle_probe.json"""
import importlib
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from __graft_entry__ import load_package  # noqa: E402

rt = load_package()
scenes = importlib.import_module("ray_tracer_amd.scenes")
out = {}
for name, desc, W, H, spp in (("book_one", scenes.book_one(1, 1.5), 1200, 800, 32), ("cornell", scenes.cornell(1.0), 600, 600, 64),
                              ("cover", scenes.cover(1, 1.0), 800, 800, int(sys.argv[1]) if len(sys.argv) > 1 else 48)):
    sc, cam = scenes.build_product(desc, device=0)
    _, c = sc.render(cam, W, H, spp, 100, seed=1, counters=True)
    nw = max(1, c["node_wave"])
    r = {"node_exec_per_sample": c["node_wave"] * 64 / c["samples"], "at_node": c["node_lane"] / nw / 64, "idle_leaf": c["node_idle_leaf"] / nw / 64,
         "idle_done": c["node_idle_done"] / nw / 64, "idle_empty": c["node_idle_empty"] / nw / 64,
         "leaf_occ": c["leaf_lane"] / max(1, c["leaf_wave"]) / 64, "shade_occ": c["shade_lane"] / max(1, c["shade_wave"]) / 64,
         "launch": sc.last_launch_config()}
    out[name] = r
    print(name, {k: (round(v, 3) if isinstance(v, float) else v) for k, v in r.items() if k != "launch"}, flush=True)
    sc.close()
(ROOT / "gpurun_out" / "idle_probe.json").write_text(json.dumps(out, indent=1))
