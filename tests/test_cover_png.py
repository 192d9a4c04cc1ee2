"""The reference's published render (cover.png, examples/main.rs at 800x800x1000) as a statistical anchor.

The run behind the picture is unseeded, so no pixel can be compared; but the camera, the light, the four big
spheres and the medium inside the blue one are fixed, and region means of the 8-bit picture are reproducible to a
fraction of a level at the same 1000 spp.  tests/golden/cover_png_regions.json holds the region statistics of the
picture (made by tests/golden/make_cover_stats.py from /root/reference/cover.png); here the oracle (CPU) and the
HIP path (GPU) render this repo's restatement of the scene and must land on them.

Two findings recorded by these tests.  (1) The blue sphere (glass shell + density-0.03 medium) of the picture is 3.7 sigma
greener and 2.3 sigma less red than any render of today's source; floor randomness, the earth texture, sample count and seven
earlier forms of the code the source hints at were excluded or tried (profiles/r02_blue_sphere.md): the picture predates
today's ConstantMedium::hit (src/volume.rs:90 "written wrong originally", examples/main.rs:325).  The blue regions are
therefore pinned to THIS repo's own value within 3 sigma of its seed spread, and the comparison with the picture is an asserted,
measured disagreement -- parity of ConstantMedium + Isotropic against the Rust program stays unpinned.
(2) The picture was rendered WITHOUT the r = 5000 fog sprite that examples/main.rs now adds (its background is exactly 0 over 51,000 pixels x 1000 spp; with the fog the same pixels average ~55/255),
so the comparison uses scenes.cover(with_fog=False); the fog's effect is asserted separately.  Its pixel noise is
also ~0.66x that of a 1000 spp render here (high-pass std 4.5 vs 6.8 levels on the orange sphere), as if it had been
rendered with about twice the samples of today's literal; means and outlines are what can be, and is, compared.
"""
import json
from pathlib import Path

import numpy as np
import pytest

_ALL = json.loads((Path(__file__).resolve().parent / "golden" / "cover_png_regions.json").read_text())
FIX = _ALL["regions"]
OWN = _ALL["repo_values"]
BLUE = ("blue_core", "blue_small")
W = H = 800
SPP, DEPTH = 1000, 100


def to8(c):
    """examples/main.rs:113-121: (c.sqrt() * 255.0).min(255.0) as u8 (f64::min drops a NaN: 255)"""
    with np.errstate(invalid="ignore"):
        v = np.sqrt(c) * 255.0
    return np.where(np.isnan(v), 255.0, np.minimum(v, 255.0)).astype(np.uint8)


def oracle_region(orc, name, spp, box=None):
    x0, y0, x1, y1 = box or FIX[name]["box"]
    img = orc.render(W, H, spp, DEPTH, seed=5, region=(x0, H - y1, x1, H - y0), nthreads=8)  # the oracle's y is up
    return to8(img[H - y1:H - y0, x0:x1][::-1])


def check(name, px8):
    f = FIX[name]
    got = px8.reshape(-1, 3).astype(np.float64).mean(0)
    assert np.all(np.abs(got - np.array(f["mean"])) <= f["tolerance_levels"] + 1e-9), (name, got, f["mean"])
    return got


def check_own(name, px8):
    """the blue regions against this repo's own value: 3 sigma of the spread over scene seeds (+ 0.5 level of pixel noise)"""
    got = px8.reshape(-1, 3).astype(np.float64).mean(0)
    want, sigma = np.array(OWN[name]["mean"]), np.array(OWN[name]["sigma"])
    assert np.all(np.abs(got - want) <= 3.0 * sigma + 0.5), (name, got, want)
    return got


@pytest.fixture(scope="module")
def cover_oracle(scenes, oracle):
    return oracle.build_oracle(scenes.cover(1, 1.0, with_fog=False))


def test_oracle_light_and_background_are_exact(cover_oracle):
    """every sample of these pixels ends on the light (7,7,7 -> 255) or on nothing (0): exact at any spp"""
    for name in ("light", "background_mid", "background_right"):
        px = oracle_region(cover_oracle, name, 3)
        check(name, px)
        assert px.min() == px.max() == FIX[name]["min"][0]


def test_oracle_orange_sphere_matches_the_published_render(cover_oracle):
    # 30x30 pixels x 1000 spp: lambertian sphere lit by the floor's bounce of the rectangle light
    got = check("orange_small", oracle_region(cover_oracle, "orange_small", SPP))
    assert got[0] > got[1] > got[2]


@pytest.fixture(scope="module")
def oracle_blue_small(cover_oracle):
    # dielectric shell with an isotropic medium of density 0.03 inside (ConstantMedium::hit, src/volume.rs:46-100)
    return oracle_region(cover_oracle, "blue_small", SPP)


def test_oracle_blue_medium_sphere_is_where_this_repo_puts_it(oracle_blue_small):
    check_own("blue_small", oracle_blue_small)


def test_oracle_blue_medium_sphere_vs_the_published_render(oracle_blue_small):
    """on this 20x20 patch the seed spread is larger than on the 100x100 core: the picture (18.5 40.1 86.2) is inside 3.5
    levels here; the significant disagreement is asserted on the core region by the GPU test"""
    check("blue_small", oracle_blue_small)


def test_the_pictures_blue_sphere_is_outside_this_repos_spread():
    """the measured disagreement itself: >= 3 sigma in green, red on the other side"""
    d = (np.array(FIX["blue_core"]["mean"]) - np.array(OWN["blue_core"]["mean"])) / np.array(OWN["blue_core"]["sigma"])
    assert d[1] > 3.0 and d[0] < -2.0, d


def test_oracle_fog_of_the_current_driver_is_not_in_the_picture(scenes, oracle):
    """examples/main.rs:258-263 adds a density 1e-4 medium of radius 5000 around everything; a ray that starts inside
    it scatters along its way (src/volume.rs:76-95), so with it the empty background glows -- the picture's does not."""
    fog = oracle.build_oracle(scenes.cover(1, 1.0, with_fog=True))
    px = oracle_region(fog, "background_mid", SPP, box=(300, 200, 316, 216))
    assert px.reshape(-1, 3).astype(float).mean() > 20.0
    assert FIX["background_mid"]["max"] == [0, 0, 0]


@pytest.mark.gpu
def test_gpu_cover_matches_the_published_render(rt, scenes, gpu_device):
    """the whole picture at the reference's size and sample count through the C ABI (0.55 s on an MI355X)"""
    sc, cam = scenes.build_product(scenes.cover(1, 1.0, with_fog=False), device=gpu_device)
    img8 = to8(sc.render(cam, W, H, SPP, DEPTH, seed=3)[::-1])
    report = {}
    for name, f in FIX.items():
        x0, y0, x1, y1 = f["box"]
        report[name] = (check_own if name in BLUE else check)(name, img8[y0:y1, x0:x1])
    # the picture's blue sphere stays out of reach of today's source, measured on this very render (profiles/r02_blue_sphere.md):
    # if this ever stops holding, the medium path changed
    d = (np.array(FIX["blue_core"]["mean"]) - report["blue_core"]) / np.array(OWN["blue_core"]["sigma"])
    assert d[1] > 3.0 and d[0] < -1.5, d
    for name in ("light", "background_mid", "background_right"):
        x0, y0, x1, y1 = FIX[name]["box"]
        assert img8[y0:y1, x0:x1].min() == img8[y0:y1, x0:x1].max() == FIX[name]["min"][0]
    # the sharpest one: a deterministic, smooth, noise-free region agrees to a fraction of a level
    assert np.all(np.abs(report["orange_core"] - np.array(FIX["orange_core"]["mean"])) <= 0.5), report["orange_core"]
    # geometry, pixel for pixel: the outline of the saturated light (camera quirk Q1, the rotated rectangle's
    # matrix pair) and the box around the orange sphere
    lit = (img8[:200] == 255).all(2)
    want = _ALL["light_outline"]
    assert abs(int(lit.sum()) - want["pixels"]) <= 40, (int(lit.sum()), want["pixels"])      # 37,403 px: edge pixels only
    for y, (a, b) in want["rows"].items():
        xs = np.where(lit[int(y)])[0]
        assert abs(int(xs.min()) - a) <= 1 and abs(int(xs.max()) - b) <= 1, (y, xs.min(), xs.max(), a, b)
    ys, xs = np.where(img8[140:320, 60:240, 0] > 2)
    box = [int(xs.min()) + 60, int(ys.min()) + 140, int(xs.max()) + 60, int(ys.max()) + 140]
    assert all(abs(g - w) <= 1 for g, w in zip(box, _ALL["orange_bbox"])), (box, _ALL["orange_bbox"])


@pytest.mark.gpu
def test_gpu_fog_changes_the_background(rt, scenes, gpu_device):
    sc, cam = scenes.build_product(scenes.cover(1, 1.0, with_fog=True), device=gpu_device)
    img8 = to8(sc.render(cam, W, H, 200, DEPTH, seed=3)[::-1])
    x0, y0, x1, y1 = FIX["background_mid"]["box"]
    assert img8[y0:y1, x0:x1].astype(float).mean() > 20.0
