#!/usr/bin/env python3
"""GPU box: whole-image parity of the HIP path against the CPU oracle (iterative form) at the BASELINE image sizes.
The oracle runs on all host threads.  Prints one line per scene and writes gpurun_out/full_parity.json."""
import importlib
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
from __graft_entry__ import load_package  # noqa: E402

rt = load_package()
scenes = importlib.import_module("ray_tracer_amd.scenes")
import oracle_binding as oracle  # noqa: E402

threads = min(256, os.cpu_count() or 8)
out = []
for name, desc, W, H, spp in (("book-one 1200x800", scenes.book_one(1, 1.5), 1200, 800, 64),
                               ("cornell-box 600x600", scenes.cornell(1.0), 600, 600, 48),
                               ("book-two cover 800x800 (with fog)", scenes.cover(1, 1.0), 800, 800, 24)):
    sc, cam = scenes.build_product(desc, device=0)
    t0 = time.time()
    img = sc.render(cam, W, H, spp, 100, seed=1)
    t1 = time.time()
    ref = oracle.build_oracle(desc).render(W, H, spp, 100, seed=1, iterative=True, nthreads=threads)
    t2 = time.time()
    d = np.abs(img - ref)
    rec = {"scene": name, "width": W, "height": H, "spp": spp, "depth": 100, "samples": W * H * spp,
           "pixels_differing_at_all": int((d.max(axis=2) > 0).sum()), "pixels_differing_by_more_than_1e-12": int((d.max(axis=2) > 1e-12).sum()),
           "max_abs_diff": float(d.max()), "mean_abs_diff": float(d.mean()), "gpu_seconds": t1 - t0, "oracle_seconds": t2 - t1,
           "oracle_threads": threads}
    print(rec, flush=True)
    out.append(rec)
json.dump(out, open(ROOT / "gpurun_out" / "full_parity.json", "w"), indent=1)
