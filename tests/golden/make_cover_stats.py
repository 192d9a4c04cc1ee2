#!/usr/bin/env python3
"""Region statistics of the reference's published render -> tests/golden/cover_png_regions.json.

/root/reference/cover.png (800x800) is the only output of the reference that ships with it: one run of
examples/main.rs at 1000 spp.  The run is unseeded, so it is no pixel golden, but most of the scene is fixed
(camera, light, the four big spheres, the medium inside the blue one); only the floor heights, the cloud of small
spheres and the earth texture file vary.  This script reads the PNG (zlib only; no image library in the image)
and stores mean / std of the 8-bit channels over hand-picked regions, in PNG coordinates (x right, y down).
Run it in the build container (it needs /root/reference); the tests only read the JSON.

Round 3: the clear glass sphere (examples/main.rs:222-229) and the earth's outline are added as pins of Dielectric and of
Sphere::hit + camera that owe nothing to the medium: the dark lower half of the ball (the black background refracted twice plus
the Fresnel reflection of the floor), the saturated image of the lamp at its bottom rim (double refraction: outline compared
pixel for pixel), the lamp's Fresnel reflection on the blue ball's shell, and twelve limb columns of the earth.  Regions marked
"linear" are compared as means of ((v + 0.5) / 255)^2: the mean of 8-bit square roots of a DARK region depends on the noise
level (this repo's 4000 spp render of glass_dark is 2.1-2.7 levels above its 1000 spp render; the picture's noise is 0.66 x
that of 1000 spp), the mean of the linearised values does not.
`repo_values`: mean and sigma of every region over this repo's renders of 12 scene seeds (the run behind the picture is
unseeded), from gpurun_out/cover_seeds.npz = `python tools/cover_seeds.py 12` on the GPU box (HIP path, no fog, 800x800x1000).
"""
import json
import re
import struct
import zlib
from pathlib import Path

import numpy as np

SRC = Path("/root/reference/cover.png")
DST = Path(__file__).resolve().parent / "cover_png_regions.json"

# name: (x0, y0, x1, y1), what it shows, how far a correct render may be off (8-bit levels, per channel mean)
REGIONS = {
    "light": ((220, 30, 400, 90), "the 7,7,7 rectangle light seen directly: saturates", 0.0),
    "background_mid": ((250, 150, 400, 350), "nothing behind it: exactly black (no fog in this render)", 0.0),
    "background_right": ((650, 100, 790, 250), "nothing behind it: exactly black", 0.0),
    "orange_core": ((117, 198, 177, 258), "lambertian (0.7,0.3,0.1) sphere lit by floor bounce", 1.5),
    "orange_small": ((132, 213, 162, 243), "30x30 centre of the same sphere (CPU-sized)", 2.0),
    "metal_core": ((650, 540, 710, 600), "metal (0.8,0.8,0.9) fuzz 1 sphere", 3.0),
    # 3 sigma of this repo's own spread over 12 scene seeds (0.6, 0.8, 1.0 levels; profiles/r02_blue_sphere.md) -- NOT widened
    # to make the picture fit: the picture is 3.7 sigma away in green and tests/test_cover_png.py asserts that disagreement
    "blue_core": ((165, 540, 265, 640), "dielectric shell + isotropic (0.2,0.4,0.9) medium, density 0.03", 3.0),
    "blue_small": ((205, 580, 225, 600), "20x20 centre of the same sphere (CPU-sized)", 3.5),
    "cluster": ((440, 260, 600, 400), "1000 white r=10 spheres at random places in a fixed box", 8.0),
    "floor_bottom": ((0, 700, 800, 800), "boxes of random height (1..101)", 14.0),
    # ---- round 3: Dielectric on its own (the clear glass ball, r = 50 at (260,150,45): centre (404, 606), radius 81 px) ----
    "glass_dark": ((352, 624, 440, 648), "lower half of the glass ball: black background through two refractions + Fresnel reflection", 0.0),
    "glass_dark_small": ((380, 628, 420, 644), "40x16 patch of the same (CPU-sized)", 0.0),
    "glass_core": ((364, 566, 444, 646), "80x80 centre of the glass ball: refracted floor (random heights) over refracted darkness", 0.0),
    "glass_upper": ((370, 560, 440, 595), "refracted floor boxes (random heights)", 0.0),
    "glass_caustic": ((372, 740, 474, 766), "the lamp focused by the ball onto the floor box under it (random height)", 0.0),
    "blue_highlight": ((197, 497, 250, 517), "the lamp's Fresnel reflection on the dielectric shell of the blue ball", 0.0),
    "whole": ((0, 0, 800, 800), "everything, incl. the earth texture this repo replaces by a synthetic one", 6.0),
}


def read_png_rgb(path):
    b = path.read_bytes()
    assert b[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat, hdr = 8, b"", None
    while pos < len(b):
        n, typ = struct.unpack(">I4s", b[pos:pos + 8])
        data = b[pos + 8:pos + 8 + n]
        pos += 12 + n
        if typ == b"IHDR":
            hdr = struct.unpack(">IIBBBBB", data)
        elif typ == b"IDAT":
            idat += data
        elif typ == b"IEND":
            break
    w, h, depth, ctype, _, _, interlace = hdr
    assert depth == 8 and interlace == 0 and ctype in (2, 6)
    bpp = 4 if ctype == 6 else 3
    raw, stride = zlib.decompress(idat), w * bpp
    out = np.zeros((h, stride), dtype=np.uint8)
    prev = np.zeros(stride, dtype=np.int32)
    p = 0
    for y in range(h):
        f = raw[p]
        line = np.frombuffer(raw[p + 1:p + 1 + stride], dtype=np.uint8).astype(np.int32)
        p += 1 + stride
        if f == 0:
            cur = line
        elif f == 2:
            cur = (line + prev) & 255
        else:
            cur = np.zeros(stride, dtype=np.int32)
            for i in range(stride):
                a = cur[i - bpp] if i >= bpp else 0
                up = prev[i]
                c = prev[i - bpp] if i >= bpp else 0
                if f == 1:
                    pred = a
                elif f == 3:
                    pred = (a + up) >> 1
                else:
                    pa, pb, pc = abs(up - c), abs(a - c), abs(a + up - 2 * c)
                    pred = a if (pa <= pb and pa <= pc) else (up if pb <= pc else c)
                cur[i] = (line[i] + pred) & 255
        out[y] = cur
        prev = cur
    return out.reshape(h, w, bpp)[:, :, :3]


LINEAR = ("glass_dark", "glass_dark_small", "glass_core", "glass_upper", "glass_caustic", "blue_highlight", "blue_core", "orange_core", "metal_core")
CRESCENT_WINDOW = (370, 660, 480, 700)  # the lamp's image inside the glass ball, at its bottom rim
EARTH_ROWS = (385, 400, 420, 440, 460, 475)


def linear(px8):
    return ((px8.astype(np.float64) + 0.5) / 255.0) ** 2


def crescent(im):
    """saturated pixels of the lamp's doubly refracted image: [count, x_min, x_max, y_min, y_max]"""
    x0, y0, x1, y1 = CRESCENT_WINDOW
    w = (im[y0:y1, x0:x1] >= 250).all(2)
    ys, xs = np.where(w)
    return [int(w.sum()), int(xs.min()) + x0, int(xs.max()) + x0, int(ys.min()) + y0, int(ys.max()) + y0]


def earth_outline(im):
    """first / last lit column (black background on both sides) of the earth on six rows, and its first lit row"""
    cols = []
    for y in EARTH_ROWS:
        on = np.where(im[y, :330].max(1) > 3)[0]
        cols += [int(on.min()), int(on.max())]
    top = int(np.where(im[300:480, 60:200].max(2).max(1) > 3)[0].min()) + 300
    return cols, top


def seed_spread(res):
    """this repo's own renders of 12 scene seeds -> mean / sigma per region (8-bit and linear), crescent, earth outline"""
    npz = Path(__file__).resolve().parents[2] / "gpurun_out" / "cover_seeds.npz"
    if not npz.exists():
        print("no", npz, "- keeping the repo_values already in", DST)
        return json.loads(DST.read_text())["repo_values"]
    z = np.load(npz)
    seeds = [z[k] for k in sorted(z.files, key=lambda k: (len(k), k)) if re.fullmatch(r"scene\d+", k)]
    out = {"renders": f"{len(seeds)} scene seeds, HIP path, no fog, 800x800x1000, render seed 3 (tools/cover_seeds.py)", "regions": {}}
    for name, f in res["regions"].items():
        x0, y0, x1, y1 = f["box"]
        m8 = np.array([s[y0:y1, x0:x1].reshape(-1, 3).astype(np.float64).mean(0) for s in seeds])
        ml = np.array([linear(s[y0:y1, x0:x1]).reshape(-1, 3).mean(0) for s in seeds])
        out["regions"][name] = {"mean": [round(v, 3) for v in m8.mean(0)], "sigma": [round(v, 3) for v in m8.std(0, ddof=1)],
                                "linear_mean": [round(v, 6) for v in ml.mean(0)], "linear_sigma": [round(v, 6) for v in ml.std(0, ddof=1)]}
    cr = np.array([crescent(s) for s in seeds], dtype=np.float64)
    out["glass_lamp_image"] = {"mean": [round(v, 2) for v in cr.mean(0)], "sigma": [round(v, 2) for v in cr.std(0, ddof=1)]}
    eo = np.array([earth_outline(s)[0] for s in seeds], dtype=np.float64)
    out["earth_outline"] = {"mean": [round(v, 2) for v in eo.mean(0)], "sigma": [round(v, 2) for v in eo.std(0, ddof=1)],
                            "top_row": sorted({earth_outline(s)[1] for s in seeds})}
    if "scene1_4000spp" in z.files:  # how much a region's 8-bit mean depends on the noise level
        x0, y0, x1, y1 = res["regions"]["glass_dark"]["box"]
        out["glass_dark_8bit_mean_1000_vs_4000spp"] = [[round(v, 2) for v in z[k][y0:y1, x0:x1].reshape(-1, 3).astype(np.float64).mean(0)]
                                                       for k in ("scene1", "scene1_4000spp")]
    return out


def main():
    im = read_png_rgb(SRC)
    assert im.shape == (800, 800, 3)
    res = {"source": "cover.png of aiifabbf/ray-tracer (README figure; examples/main.rs, 800x800, 1000 spp, unseeded)",
           "conversion": "min(sqrt(c) * 255, 255) as u8 (examples/main.rs:113-121); rows top-down",
           "coordinates": "x right, y down (PNG)", "regions": {}}
    for name, ((x0, y0, x1, y1), what, tol) in REGIONS.items():
        r = im[y0:y1, x0:x1].reshape(-1, 3).astype(np.float64)
        res["regions"][name] = {"box": [x0, y0, x1, y1], "what": what, "tolerance_levels": tol,
                                "mean": [round(v, 4) for v in r.mean(0)], "std": [round(v, 4) for v in r.std(0)],
                                "min": [int(v) for v in r.min(0)], "max": [int(v) for v in r.max(0)],
                                "linear_mean": [round(v, 6) for v in linear(im[y0:y1, x0:x1]).reshape(-1, 3).mean(0)],
                                "compare": "linear" if name in LINEAR else "8bit"}
    # geometry that does not depend on noise: the outline of the saturated light and the box around the orange sphere
    lit = (im[:200] == 255).all(2)
    res["light_outline"] = {"pixels": int(lit.sum()),
                            "rows": {str(y): [int(np.where(lit[y])[0].min()), int(np.where(lit[y])[0].max())] for y in (0, 20, 60, 100, 118)}}
    win = im[140:320, 60:240, 0] > 2
    ys, xs = np.where(win)
    res["orange_bbox"] = [int(xs.min()) + 60, int(ys.min()) + 140, int(xs.max()) + 60, int(ys.max()) + 140]
    # geometry that owes nothing to noise or to the random floor: the lamp's image inside the glass ball and the earth's limb
    res["glass_lamp_image"] = {"window": list(CRESCENT_WINDOW), "what": "pixels >= 250 in all channels: [count, x_min, x_max, y_min, y_max]",
                               "value": crescent(im)}
    cols, top = earth_outline(im)
    res["earth_outline"] = {"rows": list(EARTH_ROWS), "what": "first / last column with a channel > 3 per row (x < 330); first lit row",
                            "columns": cols, "top_row": top}
    res["repo_values"] = seed_spread(res)
    res["repo_values"]["note"] = ("cover.png predates today's ConstantMedium::hit (profiles/r02_blue_sphere.md, profiles/r03_cover_pins.md): "
                                  "every Dielectric-only pin agrees within 3 sigma, the blue ball's body does not (green +3 sigma, red -2.4)")
    DST.write_text(json.dumps(res, indent=1) + "\n")
    print("wrote", DST)


if __name__ == "__main__":
    main()
