#!/usr/bin/env python3
"""Round 5, VERDICT r4 #2, step 0: what would the cover family gain if its WHOLE node array lived in LDS?

The cover's tree (3406 nodes x 64 B) fits no LDS, so the question is put to the same kernel family on the same kind of scene at a
size whose tree does fit ONE 1024-thread workgroup's LDS (160 KB - 68 KB of stack - 24 KB of queues): scenes.cover() with a floor of
n x n cubes and m foam spheres instead of 20 x 20 and 1000.  Run under several builds (RT_MI355X_LIB) and switches:

    python tools/cover_lds_probe.py [--side 8] [--spheres 600] [--spp 200]

prints one JSON line: scene size, launch configuration, render_kernel ms (median of 5), with nodes in LDS and with RT_NO_LDS_NODES=1.
"""
import argparse
import importlib
import json
import os
import statistics
import sys
from math import radians
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def small_cover(scenes, side, n_spheres, seed=1):
    """scenes.cover() with side x side floor cubes (of the same 100-unit pitch, centred like the original's) and n_spheres foam spheres"""
    d = scenes.SceneDesc(name=f"cover-{side}x{side}-{n_spheres}")
    g = scenes.HostRng(seed)
    ground = d.lambertian_rgb((0.48, 0.83, 0.53))
    cubes = []
    for i in range(side):
        for j in range(side):
            w = 2000.0 / side
            x0, z0 = -1000.0 + i * w, -1000.0 + j * w
            y1 = g.gen_range(1.0, 101.0)
            cubes.append(d.sprite(d.geom("cube", w, y1, w), ground, scenes.mat4_translation((x0 + w / 2, y1 / 2.0, z0 + w / 2))))
    light = d.sprite(d.geom("rectangle", 300.0, 265.0), d.mat("diffuse_light", d.tex_solid((7.0, 7.0, 7.0))),
                     scenes.mat4_multiplied(scenes.mat4_translation((273.0, 554.0, 279.5)), scenes.mat4_rotation(radians(90.0), (1.0, 0.0, 0.0))))
    s50 = d.geom("sphere", 50.0)
    others = [light,
              d.sprite(s50, d.lambertian_rgb((0.7, 0.3, 0.1)), scenes.mat4_translation((400.0, 400.0, 200.0))),
              d.sprite(d.geom("sphere", 50.0), d.mat("dielectric", 1.5), scenes.mat4_translation((260.0, 150.0, 45.0))),
              d.sprite(d.geom("sphere", 50.0), d.mat("metal", d.tex_solid((0.8, 0.8, 0.9)), 1.0), scenes.mat4_translation((0.0, 150.0, 145.0))),
              d.sprite(d.geom("sphere", 70.0), d.mat("dielectric", 1.5), scenes.mat4_translation((360.0, 150.0, 145.0))),
              d.sprite(d.geom("medium", d.geom("sphere", 70.0 - 1e-6), 0.03), d.mat("isotropic", d.tex_solid((0.2, 0.4, 0.9))),
                       scenes.mat4_translation((360.0, 150.0, 145.0))),
              d.sprite(d.geom("medium", d.geom("sphere", 5000.0), 0.0001), d.mat("isotropic", d.tex_solid((1.0, 1.0, 1.0))), None)]
    d.textures.append(("image", scenes.earth_texture()))
    others.append(d.sprite(d.geom("sphere", 100.0), d.mat("lambertian", len(d.textures) - 1), scenes.mat4_translation((400.0, 200.0, 400.0))))
    white = d.lambertian_rgb((0.73, 0.73, 0.73))
    s10 = d.geom("sphere", 10.0)
    spheres = [d.sprite(s10, white, scenes.mat4_translation((g.gen_range(0.0, 165.0) - 100.0, g.gen_range(0.0, 165.0) + 270.0, g.gen_range(0.0, 165.0) + 395.0)))
               for _ in range(n_spheres)]
    d.world = [("bvh", cubes)] + others + [("bvh", spheres)]
    d.camera = ((555.0 / 2.0 + 200.0, 550.0 / 2.0, -600.0), (555.0 / 2.0, 555.0 / 2.0, 0.0), (0.0, 1.0, 0.0), radians(40.0), 1.0, 10.0, 0.0)
    return d


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--side", type=int, default=8)
    ap.add_argument("--spheres", type=int, default=600)
    ap.add_argument("--spp", type=int, default=200)
    ap.add_argument("--edge", type=int, default=800)
    ap.add_argument("--full", action="store_true", help="the real cover scene (scenes.cover(1)) instead of the scaled one")
    a = ap.parse_args()
    from __graft_entry__ import load_package
    rt = load_package()
    scenes = importlib.import_module("ray_tracer_amd.scenes")
    import torch
    desc = scenes.cover(1, 1.0) if a.full else small_cover(scenes, a.side, a.spheres)
    out = {"library": rt.version(), "scene": desc.name, "spp": a.spp, "edge": a.edge}
    ref = None
    for mode in ("default", "RT_NO_LDS_NODES=1"):
        if mode != "default":
            os.environ["RT_NO_LDS_NODES"] = "1"
        sc, cam = scenes.build_product(desc, device=0)
        info = sc.info()
        buf = torch.zeros(rt.shard_tile_count(a.edge, a.edge, 0, 1) * 64 * 3, dtype=torch.float64, device="cuda:0")
        ms = []
        for _ in range(6):
            sc.render_tiles_device(cam, a.edge, a.edge, a.spp, 100, 1, (0, 1), buf.data_ptr(), None, None)
            torch.cuda.synchronize()
            ms.append(sc.last_kernel_ms())
        lc = sc.last_launch_config()
        img = buf.cpu().numpy().copy()
        same = None if ref is None else bool((img == ref).all())
        ref = img if ref is None else ref
        out[mode] = {"kernel_ms_median": statistics.median(ms[1:]), "kernel_ms": [round(m, 2) for m in ms], "lds_nodes": lc["lds_nodes"],
                     "block_threads": lc["block_threads"], "blocks_per_cu": lc["blocks_per_cu"], "lds_bytes": lc["lds_bytes"], "swap_cap": lc["swap_cap"],
                     "same_image_as_first_mode": same}
        out["n_prims"], out["n_nodes"], out["max_depth"] = info["n_prims"], info["n_nodes"], info["max_depth"]
        os.environ.pop("RT_NO_LDS_NODES", None)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
