#!/usr/bin/env python3
"""GPU box: what the learnt deepest-first tile order (include/rt_mi355x.h RT_TILE_ORDER_*) is worth on one shard of N.

For each scene and N in (2, 4, 8): render shard 0 of N in ascending order (RT_FLAG_ASCENDING_TILES) and in the learnt order,
interleaved, `--repeats` times each; print the render_kernel milliseconds (HIP events inside the library) and the milliseconds of
the whole call (render + sums) and check that both orders give the same bits.  Writes JSON to --out."""
import argparse
import importlib
import json
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--repeats", type=int, default=8)
    ap.add_argument("--out", default="gpurun_out/tile_order_probe.json")
    ap.add_argument("--scenes", default="book_one,cornell,cover")
    ap.add_argument("--size", default="", help="W,H,spp instead of the scene's BASELINE size")
    ap.add_argument("--worlds", default="2,4,8", help="shard counts; 1 (a whole image) needs the test-hooks library and RT_TEST_TILE_ORDER_WHOLE=1")
    a = ap.parse_args()
    import torch
    from __graft_entry__ import load_package
    rt = load_package()
    scenes = importlib.import_module("ray_tracer_amd.scenes")
    dev = torch.device("cuda", 0)
    sizes = {"book_one": (1200, 800, 500), "cornell": (600, 600, 1000), "cover": (800, 800, 1000)}
    rows = []
    for name in a.scenes.split(","):
        W, H, spp = [int(v) for v in a.size.split(",")] if a.size else sizes[name]
        desc = {"book_one": lambda: scenes.book_one(1, W / H), "cornell": scenes.cornell, "cover": lambda: scenes.cover(1)}[name]()
        sc, cam = scenes.build_product(desc, device=0)
        for world in [int(w) for w in a.worlds.split(",")]:
            n = rt.shard_tile_count(W, H, 0, world)
            buf = [torch.zeros(n * 64 * 3, dtype=torch.float64, device=dev) for _ in range(2)]
            st = torch.cuda.current_stream()

            def once(flags, out):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(st)
                sc.render_tiles_device(cam, W, H, spp, 100, 1, (0, world), out.data_ptr(), None, st.cuda_stream, flags=flags)
                e1.record(st)
                mode = sc.last_launch_config()["tile_order"]
                torch.cuda.synchronize()
                return sc.last_kernel_ms(), e0.elapsed_time(e1), mode

            once(rt.RT_FLAG_ASCENDING_TILES, buf[0])  # warm
            _, _, mode = once(0, buf[1])                  # learns
            assert mode == rt.RT_TILE_ORDER_LEARNING, mode
            asc, lrn = [], []
            for _ in range(a.repeats):
                asc.append(once(rt.RT_FLAG_ASCENDING_TILES, buf[0])[:2])
                k, t, mode = once(0, buf[1])
                assert mode == rt.RT_TILE_ORDER_LEARNT, mode
                lrn.append((k, t))
            same = bool(torch.equal(buf[0], buf[1]))
            order, cost = sc.tile_order()
            row = {"scene": name, "width": W, "height": H, "spp": spp, "shard": [0, world], "owned_tiles": n,
                   "ascending_kernel_ms": float(np.median([x[0] for x in asc])), "learnt_kernel_ms": float(np.median([x[0] for x in lrn])),
                   "ascending_call_ms": float(np.median([x[1] for x in asc])), "learnt_call_ms": float(np.median([x[1] for x in lrn])),
                   "same_bits": same, "mean_path_length_deepest_tile": float(cost[order[0]]) / (64 * spp),
                   "mean_path_length_shallowest_tile": float(cost[order[-1]]) / (64 * spp)}
            row["kernel_gain"] = 1.0 - row["learnt_kernel_ms"] / row["ascending_kernel_ms"]
            rows.append(row)
            print(json.dumps(row), flush=True)
            assert same
    os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
    json.dump({"library": rt.version(), "repeats": a.repeats, "rows": rows,
               "env": {k: v for k, v in os.environ.items() if k.startswith("RT_")}}, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
