#!/usr/bin/env python3
"""GPU box, one-off: the traversal-side stages of a WAVEFRONT organisation of the book-two cover scene, measured.
tools/probe/rt_probe.hip (make -C tools/probe) holds the two kernels; the segments they work on are the real
segments of the scene's paths, recorded by the CPU lane program (16 processes).  -> gpurun_out/wavefront_probe.json
PROBE_WIDE4=1: the traverse stage over the scene's binary tree and over the same tree collapsed into 4-wide nodes, setting by
setting (-> gpurun_out/wide4_probe.json); PROBE_FULL=0 skips the whole-config megakernel run; PROBE_EDGE / PROBE_SPP / PROBE_REPLICATE
size the recording."""
import ctypes as C
import importlib
import json
import multiprocessing as mp
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
REP = int(os.environ.get("PROBE_REPLICATE", "16"))  # the recorded segments are gone through this many times per launch (steady state)
W = H = int(os.environ.get("PROBE_EDGE", "400"))
SPP = int(os.environ.get("PROBE_SPP", "6"))
DEPTH, SEED = 100, 1


def record(band):
    """the segments of rows [y0, y1) as the lane program walks them"""
    from __graft_entry__ import load_package
    load_package()
    scenes = importlib.import_module("ray_tracer_amd.scenes")
    lib = C.CDLL(str(ROOT / "tests" / "_build" / "liblane_emul.so"))
    lib.lane_emul_dump_segments.restype = C.c_long
    lib.lane_emul_dump_segments.argtypes = [C.c_void_p, C.c_void_p] + [C.c_int] * 4 + [C.c_uint64] + [C.c_int] * 4 + [C.POINTER(C.c_double), C.c_long]
    sc, cam = scenes.build_product(scenes.cover(1, 1.0), device=-1)
    y0, y1 = band
    cap = (y1 - y0) * W * SPP * 8
    out = np.zeros((cap, 11))
    n = lib.lane_emul_dump_segments(sc._h, C.addressof(cam.c), W, H, SPP, DEPTH, SEED, 0, y0, W, y1, out.ctypes.data_as(C.POINTER(C.c_double)), cap)
    assert 0 <= n <= cap, (n, cap)
    return out[:n].copy()


def main():
    os.environ["RT_MI355X_LIB"] = str(ROOT / "tools" / "probe" / "_build" / "librt_mi355x_travprobe.so")
    t0 = time.time()
    procs = 16
    bands = [(H * i // procs, H * (i + 1) // procs) for i in range(procs)]
    with mp.get_context("spawn").Pool(procs) as pool:
        rows = np.concatenate(pool.map(record, bands))
    n = len(rows)
    print(f"recorded {n} segments of {W * H * SPP} samples in {time.time() - t0:.1f} s", flush=True)
    # shuffle the paths' order as a pool of a wavefront tracer would see them: rays of all depths mixed, neighbours apart
    rng = np.random.default_rng(1)
    order_mixed = rng.permutation(n)
    rays = np.zeros(n, dtype=[("o", "f8", 3), ("d", "f8", 3), ("base", "u8"), ("k", "u4"), ("pad", "u4")])
    rays["o"], rays["d"] = rows[:, 1:4], rows[:, 4:7]
    rays["base"] = rows[:, 9].copy().view(np.uint64)
    rays["k"] = rows[:, 10].astype(np.uint32)
    want_t, want_prim = rows[:, 7], rows[:, 8]

    from __graft_entry__ import load_package
    rt = load_package()
    scenes = importlib.import_module("ray_tracer_amd.scenes")
    sc, cam = scenes.build_product(scenes.cover(1, 1.0), device=0)
    lib = C.CDLL(os.environ["RT_MI355X_LIB"])
    lib.rt_probe_traverse.restype = C.c_int
    lib.rt_probe_traverse.argtypes = [C.c_void_p, C.c_void_p] + [C.c_int] * 6 + [C.POINTER(C.c_double), C.c_void_p, C.POINTER(C.c_ulonglong), C.c_int]
    # the megakernel on the same image, for the same number of segments
    mk = []
    for _ in range(3):
        sc.render(cam, W, H, SPP, DEPTH, SEED)
        mk.append(sc.last_kernel_ms())
    _, cnt = sc.render(cam, W, H, SPP, DEPTH, SEED, counters=True)
    assert cnt["segments"] == n, (cnt["segments"], n)
    full = {}  # the megakernel's steady state: the BASELINE config itself, 800 x 800 x 1000 spp
    if os.environ.get("PROBE_FULL", "1") == "1":
        sc.render(cam, 800, 800, 1000, DEPTH, SEED)
        full_ms = sc.last_kernel_ms()
        _, c16 = sc.render(cam, 800, 800, 16, DEPTH, SEED, counters=True)
        seg_per_sample = c16["segments"] / c16["samples"]
        full = {"render_kernel_ms_800x800x1000": full_ms, "segments_per_sample": seg_per_sample,
                "ns_per_segment": full_ms * 1e6 / (800 * 800 * 1000 * seg_per_sample)}
        print("megakernel, whole config:", full, flush=True)
    res = {"scene": "book-two cover", "image": [W, H, SPP], "segments": n, "megakernel_steady_state": full, "megakernel_counters": {k: int(v) for k, v in cnt.items()},
           "megakernel_render_kernel_ms_same_image": min(mk), "megakernel_ns_per_segment": min(mk) * 1e6 / n, "runs": []}
    print("megakernel on this image:", min(mk), "ms =", min(mk) * 1e6 / n, "ns per segment", flush=True)
    hits = np.zeros(n, dtype=[("t", "f8"), ("prim", "u4"), ("pad", "u4")])
    for name, order in (("path order (pixel-major, a path's segments adjacent)", np.arange(n)), ("mixed (random order)", order_mixed)):
        r = np.ascontiguousarray(rays[order])
        # PROBE_WIDE4=1: every setting with the scene's binary tree and with the same tree collapsed into 4-wide nodes (rt_probe.hip)
        combos = [(f, v, k, 0) for f, v, k in ((8, 16, 8), (16, 16, 8), (32, 16, 8), (48, 16, 8), (32, 32, 8), (32, 16, 16))]
        if os.environ.get("PROBE_WIDE4", "0") == "1":
            combos = [(f, v, k, w) for f, v, k in ((16, 16, 8), (32, 16, 8), (32, 16, 16), (32, 12, 12)) for w in (0, 1)]
        for fetch_min, vote_leaf, node_keep, wide4 in combos:
            ms = (C.c_double * 5)()
            stats = (C.c_ulonglong * 6)()
            rc = lib.rt_probe_traverse(sc._h, r.ctypes.data, n, REP, fetch_min, vote_leaf, node_keep, 2, ms, hits.ctypes.data, stats, wide4)
            assert rc == 0, rc
            ok = np.array_equal(hits["t"], want_t[order]) and np.array_equal(hits["prim"].astype(np.float64), np.where(want_prim[order] > 4e9, 4294967295.0, want_prim[order]))
            st = [int(v) for v in stats]
            run = {"order": name, "tree": "4-wide (%d nodes, stack %d)" % (int(ms[3]), int(ms[4])) if wide4 else "binary", "fetch_min": fetch_min, "vote_leaf": vote_leaf, "node_keep": node_keep, "begin_ms": ms[0], "traverse_ms": ms[1], "blocks": int(ms[2]),
                   "replicate": REP, "results_equal_the_lane_program": bool(ok), "ns_per_segment": {"begin": ms[0] * 1e6 / (n * REP), "traverse": ms[1] * 1e6 / max(1, min(int(stats[5]), n * REP))},
                   "rays_traversed": min(int(stats[5]), n * REP),
                   "node_block_occupancy": st[1] / max(1, 64 * st[0]), "leaf_block_occupancy": st[3] / max(1, 64 * st[2]),
                   "fetch_occupancy": st[5] / max(1, 64 * st[4]), "block_executions": {"node": st[0], "leaf": st[2], "fetch": st[4]}}
            print(run, flush=True)
            res["runs"].append(run)
    # the whole-render figures these stand against: 800 x 800 x 1000 spp = 2.25 G segments in ~302 ms
    (ROOT / "gpurun_out").mkdir(exist_ok=True)
    json.dump(res, open(ROOT / "gpurun_out" / ("wide4_probe.json" if os.environ.get("PROBE_WIDE4", "0") == "1" else "wavefront_probe.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
