#!/usr/bin/env python3
"""Run on the GPU box: measure the algorithmic counts (segments, node steps, primitive tests per sample)
of the BASELINE configs with the counting build of the render kernel -> gpurun_out/algo_counts_*.json.
These are the denominators of the roofline fraction."""
import importlib
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from __graft_entry__ import load_package  # noqa: E402

rt = load_package()
scenes = importlib.import_module("ray_tracer_amd.scenes")
CONFIGS = {
    "book_one_1200x800": (lambda: scenes.book_one(1, 1.5), 1200, 800),
    "cornell_600x600": (lambda: scenes.cornell(1.0), 600, 600),
    "cover_800x800": (lambda: scenes.cover(1, 1.0), 800, 800),
}
out_dir = ROOT / "gpurun_out"
out_dir.mkdir(exist_ok=True)
for name, (gen, W, H) in CONFIGS.items():
    sc, cam = scenes.build_product(gen(), device=0)
    info = sc.info()
    spp = 16
    _, c = sc.render(cam, W, H, spp, 100, seed=1, counters=True)
    ns = c["samples"]
    rec = {
        "config": name, "width": W, "height": H, "spp_measured": spp, "max_depth": 100, "scene_seed": 1, "render_seed": 1,
        "samples": ns, "segments": c["segments"], "node_steps": c["nodes_visited"], "prim_tests": c["prims_tested"],
        "segments_per_sample": c["segments"] / ns, "node_steps_per_sample": c["nodes_visited"] / ns,
        "prim_tests_per_sample": c["prims_tested"] / ns, "rng_draws_per_sample": c["rng_draws"] / ns,
        "node_bytes": info["node_bytes"], "prim_bytes": info["prim_bytes"], "material_bytes": info["material_bytes"],
        "algorithmic_bytes_per_sample_excl_framebuffer": (c["nodes_visited"] * info["node_bytes"] + c["prims_tested"] * info["prim_bytes"]
                                                          + c["segments"] * info["material_bytes"]) / ns,
        "n_prims": info["n_prims"], "n_hoisted": info["n_hoisted"], "n_nodes": info["n_nodes"], "bvh_max_depth": info["max_depth"],
        "source": "counting build of render_kernel (RT_FLAG_COUNTERS), tools/make_algo_counts.py on 1x MI355X",
    }
    json.dump(rec, open(out_dir / f"algo_counts_{name}.json", "w"), indent=1)
    print(name, {k: round(v, 3) for k, v in rec.items() if k.endswith("per_sample") or k.startswith("algorithmic")})
