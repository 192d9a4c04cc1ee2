"""Golden vectors (tests/golden/golden_v1.npz, made by tests/golden/make_golden.py)."""
import importlib.util
from pathlib import Path

import numpy as np
import pytest

GOLD = Path(__file__).resolve().parent / "golden" / "golden_v1.npz"


def _cases():
    spec = importlib.util.spec_from_file_location("make_golden", GOLD.parent / "make_golden.py")
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m.CASES


CASES = _cases()


@pytest.mark.parametrize("name", sorted(CASES))
def test_oracle_reproduces_golden(name, scenes, oracle):
    g = np.load(GOLD)
    gen, kw, W, H, spp, depth, seed = CASES[name]
    assert g[name + "_params"].tolist() == [W, H, spp, depth, seed]
    o = oracle.build_oracle(getattr(scenes, gen)(**kw))
    assert np.array_equal(o.render(W, H, spp, depth, seed, iterative=True, nthreads=4), g[name + "_iterative"])
    assert np.array_equal(o.render(W, H, spp, depth, seed, iterative=False, nthreads=4), g[name + "_recursive"])


@pytest.mark.parametrize("name", sorted(CASES))
def test_lane_program_reproduces_golden(name, scenes, lane_emul):
    g = np.load(GOLD)
    gen, kw, W, H, spp, depth, seed = CASES[name]
    sc, cam = scenes.build_product(getattr(scenes, gen)(**kw), device=-1)
    img, *_ = lane_emul.render(sc, cam, W, H, spp, depth, seed)
    assert np.array_equal(img, g[name + "_iterative"])


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(CASES))
def test_gpu_reproduces_golden(name, scenes, gpu_device):
    """HIP path vs committed vectors, no oracle involved.  Tolerance: see test_gpu_parity.py."""
    g = np.load(GOLD)
    gen, kw, W, H, spp, depth, seed = CASES[name]
    sc, cam = scenes.build_product(getattr(scenes, gen)(**kw), device=gpu_device)
    img = sc.render(cam, W, H, spp, depth, seed)
    diff = np.abs(img - g[name + "_iterative"])
    assert diff.mean() <= 1e-4
    assert (diff.max(axis=2) > 1e-12).sum() <= 2, f"{(diff.max(axis=2) > 1e-12).sum()} pixels differ; max {diff.max()}"
    assert np.abs(img - g[name + "_recursive"]).mean() <= 1e-12
