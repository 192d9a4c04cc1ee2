#!/usr/bin/env python3
"""Regenerate tests/golden/golden_v1.npz from the CPU oracle.

The Rust reference cannot be run here, so these vectors are outputs of the oracle
(itself pinned by the hand-derived KATs of tests/test_oracle_kat.py).  They let the GPU
tests check the HIP path without building the oracle, and pin the oracle against drift.
Inputs are fully described by (scene generator, seeds, W, H, spp, depth)."""
import importlib
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "tests"))
sys.path.insert(0, str(ROOT))
from __graft_entry__ import load_package  # noqa: E402

CASES = {
    # name: (generator, kwargs, W, H, spp, depth, render seed)
    "book_one": ("book_one", dict(scene_seed=1, aspect=1.5), 48, 32, 4, 50, 1),
    "book_one_deep": ("book_one", dict(scene_seed=3, aspect=2.0), 40, 20, 2, 100, 77),
    "cornell": ("cornell", dict(aspect=1.0), 32, 32, 4, 100, 1),
    "cover": ("cover", dict(scene_seed=1, aspect=1.0), 32, 32, 2, 100, 1),
}


def main():
    load_package()
    scenes = importlib.import_module("ray_tracer_amd.scenes")
    import oracle_binding as ob
    out = {}
    for name, (gen, kw, W, H, spp, depth, seed) in CASES.items():
        desc = getattr(scenes, gen)(**kw)
        o = ob.build_oracle(desc)
        out[name + "_iterative"] = o.render(W, H, spp, depth, seed, iterative=True, nthreads=8)
        out[name + "_recursive"] = o.render(W, H, spp, depth, seed, iterative=False, nthreads=8)
        out[name + "_params"] = np.array([W, H, spp, depth, seed], dtype=np.int64)
    desc = scenes.book_one(1, 1.5)
    out["book_one_samples_24_12"] = ob.build_oracle(desc).pixel_samples(48, 32, 16, 50, 1, 24, 12, iterative=True)
    np.savez_compressed(ROOT / "tests" / "golden" / "golden_v1.npz", **out)
    print("wrote", ROOT / "tests" / "golden" / "golden_v1.npz", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
