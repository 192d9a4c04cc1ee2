"""The reference's published render (cover.png, examples/main.rs at 800x800x1000) as a statistical anchor.

The run behind the picture is unseeded, so no pixel can be compared; but the camera, the light, the four big
spheres and the medium inside the blue one are fixed, and region means of the 8-bit picture are reproducible to a
fraction of a level at the same 1000 spp.  tests/golden/cover_png_regions.json holds the region statistics of the
picture (made by tests/golden/make_cover_stats.py from /root/reference/cover.png) and, as `repo_values`, the mean and sigma
of the same statistics over this repo's renders of 12 scene seeds (the random floor heights are the only input that
differs between two runs of the reference); here the oracle (CPU) and the HIP path (GPU) render this repo's restatement of
the scene and must land on the picture within 3 sigma of that spread.

What the picture pins (round 3, profiles/r03_cover_pins.md):
  * Lambertian, Metal, DiffuseLight, camera quirk Q1, the rotated rectangle: region means <= 2 levels, outlines +-1 px (round 1-2);
  * Sphere::hit + camera once more: twelve limb columns and the top row of the earth (texture-independent): identical +-1 px;
  * Dielectric WITHOUT the medium -- the clear glass ball (examples/main.rs:222-229): the lamp's doubly refracted image at
    the ball's bottom rim has the picture's outline pixel for pixel (Snell both ways, ior 1.5) and its area within 3 sigma; the
    dark lower half of the ball (black background through two refractions + the Fresnel reflection of the floor) has the
    picture's linear mean within 2 sigma; the lamp's Fresnel reflection on the blue ball's SHELL within 2.3 sigma (Schlick);
  * what it does not pin: the BODY of the blue ball (glass shell + density-0.03 medium) is 3 sigma greener and 2.4 sigma less
    red in the picture than in any render of today's source.  With Dielectric now pinned on its own, that offset is the medium's
    (ConstantMedium + Isotropic), and profiles/r02_blue_sphere.md argues the picture predates today's ConstantMedium::hit
    (src/volume.rs:90 "written wrong originally", examples/main.rs:325).  The comparison of the blue body with the picture is
    kept as an expected failure (xfail, not strict) -- documented, not a pass condition; the medium path is pinned by
    analytic transmittance tests instead (tests/test_medium_analytic.py).
The picture was rendered WITHOUT the r = 5000 fog sprite that examples/main.rs now adds (its background is exactly 0 over
51,000 pixels; with the fog the same pixels average ~55/255), so the comparison uses scenes.cover(with_fog=False); the fog's
effect is asserted separately.  Its pixel noise is ~0.66x that of a 1000 spp render here, so DARK regions are compared as
means of linearised values ((v + 0.5) / 255)^2, which do not depend on the noise level (the mean of 8-bit square roots does).
"""
import json
from pathlib import Path

import numpy as np
import pytest

_ALL = json.loads((Path(__file__).resolve().parent / "golden" / "cover_png_regions.json").read_text())
FIX = _ALL["regions"]
OWN = _ALL["repo_values"]["regions"]
BLUE = ("blue_core", "blue_small")
PINS = ("glass_dark", "glass_dark_small", "glass_core", "glass_upper", "glass_caustic", "blue_highlight")  # round 3: tolerance from the seed spread
W = H = 800
SPP, DEPTH = 1000, 100


def to8(c):
    """examples/main.rs:113-121: (c.sqrt() * 255.0).min(255.0) as u8 (f64::min drops a NaN: 255)"""
    with np.errstate(invalid="ignore"):
        v = np.sqrt(c) * 255.0
    return np.where(np.isnan(v), 255.0, np.minimum(v, 255.0)).astype(np.uint8)


def linear(px8):
    return ((px8.astype(np.float64) + 0.5) / 255.0) ** 2


def oracle_region(orc, name, spp, box=None):
    x0, y0, x1, y1 = box or FIX[name]["box"]
    img = orc.render(W, H, spp, DEPTH, seed=5, region=(x0, H - y1, x1, H - y0), nthreads=8)  # the oracle's y is up
    return to8(img[H - y1:H - y0, x0:x1][::-1])


def check(name, px8):
    f = FIX[name]
    got = px8.reshape(-1, 3).astype(np.float64).mean(0)
    assert np.all(np.abs(got - np.array(f["mean"])) <= f["tolerance_levels"] + 1e-9), (name, got, f["mean"])
    return got


def check_pin(name, px8, extra_sigma=0.0):
    """a region against the PICTURE.  The picture and the render under test are two single draws of the random floor, so their
    difference has sigma sqrt(2) x the spread over scene seeds: bound 3 sqrt(2) sigma (+ extra_sigma for the pixel noise of a
    reduced window); dark regions in linear space (module docstring)"""
    f, own = FIX[name], OWN[name]
    if f["compare"] == "linear":
        got, want, sigma = linear(px8).reshape(-1, 3).mean(0), np.array(f["linear_mean"]), np.array(own["linear_sigma"])
    else:
        got, want, sigma = px8.reshape(-1, 3).astype(np.float64).mean(0), np.array(f["mean"]), np.array(own["sigma"])
    z = (got - want) / sigma
    assert np.all(np.abs(z) <= 3.0 * np.sqrt(2.0) + extra_sigma), (name, got, want, z)
    return z


def check_own(name, px8):
    """the blue regions against this repo's own value: 3 sigma of the spread over scene seeds (+ 0.5 level of pixel noise)"""
    got = px8.reshape(-1, 3).astype(np.float64).mean(0)
    want, sigma = np.array(OWN[name]["mean"]), np.array(OWN[name]["sigma"])
    assert np.all(np.abs(got - want) <= 3.0 * sigma + 0.5), (name, got, want)
    return got


def lamp_image(img8):
    x0, y0, x1, y1 = _ALL["glass_lamp_image"]["window"]
    w = (img8[y0:y1, x0:x1] >= 250).all(2)
    ys, xs = np.where(w)
    return [int(w.sum()), int(xs.min()) + x0, int(xs.max()) + x0, int(ys.min()) + y0, int(ys.max()) + y0]


def check_lamp_image(got, count_slack):
    """the lamp seen through the glass ball: the picture's outline +-1 px, its area within 3 sigma of the seed spread"""
    want = _ALL["glass_lamp_image"]["value"]
    assert all(abs(g - w) <= 1 for g, w in zip(got[1:], want[1:])), (got, want)
    assert abs(got[0] - want[0]) <= 3.0 * _ALL["repo_values"]["glass_lamp_image"]["sigma"][0] + count_slack, (got, want)


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


def test_oracle_glass_ball_matches_the_published_render(cover_oracle):
    """Dielectric on its own, against the reference's pixels: (1) the lamp's image after two refractions through the clear
    glass ball, on a window tight around it at 300 spp (the image saturates at any sample count; the lit floor next to it
    must not, by noise); (2) a 40x16 patch of the ball's dark lower half at the
    picture's 1000 spp, linear mean within 3 sigma of the seed spread (+1 for 640 pixels' noise)"""
    x0, y0, x1, y1 = 384, 672, 450, 694
    img8 = np.zeros((H, W, 3), dtype=np.uint8)
    img8[y0:y1, x0:x1] = oracle_region(cover_oracle, None, 300, box=(x0, y0, x1, y1))
    check_lamp_image(lamp_image(img8), count_slack=8)
    check_pin("glass_dark_small", oracle_region(cover_oracle, "glass_dark_small", SPP), extra_sigma=1.0)


@pytest.fixture(scope="module")
def oracle_blue_small(cover_oracle):
    # dielectric shell with an isotropic medium of density 0.03 inside (ConstantMedium::hit, src/volume.rs:46-100)
    return oracle_region(cover_oracle, "blue_small", SPP)


def test_oracle_blue_medium_sphere_is_where_this_repo_puts_it(oracle_blue_small):
    """regression pin on this repo's own value (oracle = HIP path); NOT a statement about the reference"""
    check_own("blue_small", oracle_blue_small)


def test_oracle_fog_of_the_current_driver_is_not_in_the_picture(scenes, oracle):
    """examples/main.rs:258-263 adds a density 1e-4 medium of radius 5000 around everything; a ray that starts inside
    it scatters along its way (src/volume.rs:76-95), so with it the empty background glows -- the picture's does not."""
    fog = oracle.build_oracle(scenes.cover(1, 1.0, with_fog=True))
    px = oracle_region(fog, "background_mid", SPP, box=(300, 200, 316, 216))
    assert px.reshape(-1, 3).astype(float).mean() > 20.0
    assert FIX["background_mid"]["max"] == [0, 0, 0]


@pytest.fixture(scope="module")
def gpu_cover8(rt, scenes, gpu_device):
    """the whole picture at the reference's size and sample count through the C ABI (0.3 s on an MI355X)"""
    sc, cam = scenes.build_product(scenes.cover(1, 1.0, with_fog=False), device=gpu_device)
    return to8(sc.render(cam, W, H, SPP, DEPTH, seed=3)[::-1])


@pytest.mark.gpu
def test_gpu_cover_matches_the_published_render(gpu_cover8):
    img8 = gpu_cover8
    report = {}
    for name, f in FIX.items():
        x0, y0, x1, y1 = f["box"]
        px = img8[y0:y1, x0:x1]
        report[name] = check_own(name, px) if name in BLUE else (check_pin(name, px) if name in PINS else check(name, px))
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
def test_gpu_dielectric_pins_of_the_published_render(gpu_cover8):
    """Dielectric separated from the medium: the clear glass ball and the blue ball's shell against the
    reference's own pixels -- the lamp's doubly refracted image (outline +-1 px, area within 3 sigma), the dark half of the glass
    ball and the lamp's Fresnel reflection on the blue shell (linear means within 3 sigma of the seed spread)"""
    img8 = gpu_cover8
    check_lamp_image(lamp_image(img8), count_slack=2)
    for name in ("glass_dark", "blue_highlight"):
        x0, y0, x1, y1 = FIX[name]["box"]
        check_pin(name, img8[y0:y1, x0:x1])


@pytest.mark.gpu
def test_gpu_earth_outline_of_the_published_render(gpu_cover8):
    """the earth's limb against the black background (texture-independent: any lit texel is above the threshold): twelve
    columns on six rows and the first lit row, +-1 px -- Sphere::hit and the camera at the far end of the scene"""
    img8 = gpu_cover8
    e = _ALL["earth_outline"]
    cols = []
    for y in e["rows"]:
        on = np.where(img8[y, :330].max(1) > 3)[0]
        cols += [int(on.min()), int(on.max())]
    assert all(abs(g - w) <= 1 for g, w in zip(cols, e["columns"])), (cols, e["columns"])
    top = int(np.where(img8[300:480, 60:200].max(2).max(1) > 3)[0].min()) + 300
    assert abs(top - e["top_row"]) <= 1, (top, e["top_row"])


@pytest.mark.gpu
@pytest.mark.xfail(strict=False, reason="the picture's blue ball (glass shell + density-0.03 medium) is 3 sigma greener / 2.4 sigma less red "
                                        "than any render of today's source; Dielectric is pinned separately, so the offset is the medium's: "
                                        "cover.png predates today's ConstantMedium::hit (profiles/r02_blue_sphere.md, profiles/r03_cover_pins.md)")
def test_gpu_blue_ball_body_vs_the_published_render(gpu_cover8):
    """documented disagreement, not a pass condition: ConstantMedium + Isotropic stay unpinned against the Rust program"""
    x0, y0, x1, y1 = FIX["blue_core"]["box"]
    got = gpu_cover8[y0:y1, x0:x1].reshape(-1, 3).astype(np.float64).mean(0)
    z = (got - np.array(FIX["blue_core"]["mean"])) / np.array(OWN["blue_core"]["sigma"])
    assert np.all(np.abs(z) <= 3.0), z


@pytest.mark.gpu
def test_gpu_fog_changes_the_background(rt, scenes, gpu_device):
    sc, cam = scenes.build_product(scenes.cover(1, 1.0, with_fog=True), device=gpu_device)
    img8 = to8(sc.render(cam, W, H, 200, DEPTH, seed=3)[::-1])
    x0, y0, x1, y1 = FIX["background_mid"]["box"]
    assert img8[y0:y1, x0:x1].astype(float).mean() > 20.0
