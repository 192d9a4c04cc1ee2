"""The three example drivers restated in C++ over the facade (ray-tracer_amd/host): their
scenes must equal the Python scene descriptions sprite for sprite (CPU), and their P3 output
must be byte-identical to the Python-driven render of the same parameters (GPU)."""
import subprocess
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
BIN = ROOT / "ray-tracer_amd" / "host" / "bin"


@pytest.fixture(scope="module")
def binaries(rt):
    if not (BIN / "book_one").exists():
        subprocess.run(["make", "-C", str(ROOT / "ray-tracer_amd" / "host")], check=True, capture_output=True)
    return BIN


def g17(v):
    return "%.17g" % v


def describe_texture(d, t):
    tex = d.textures[t]
    if tex[0] == "solid":
        return "solid(%s)" % ",".join(g17(c) for c in tex[1])
    if tex[0] == "checker":
        return "checker(%s,%s)" % (describe_texture(d, tex[1]), describe_texture(d, tex[2]))
    a = np.ascontiguousarray(tex[1], dtype=np.uint8)
    s = 0
    for b in a.reshape(-1).tolist():
        s = (s * 1315423911 + b) & 0xFFFFFFFFFFFFFFFF
    return "image(%d,%d,%d)" % (a.shape[1], a.shape[0], s)


def describe(d, order=None):
    lines = []
    ident = [1.0, 0, 0, 0, 0, 1.0, 0, 0, 0, 0, 1.0, 0, 0, 0, 0, 1.0]
    for i in (order if order is not None else range(len(d.sprites))):
        gi, mi, M = d.sprites[i]
        g = d.geometries[gi]
        if g[0] == "medium":
            gs = "medium(sphere(%s),%s)" % (g17(d.geometries[g[1]][1]), g17(g[2]))
        else:
            gs = "%s(%s)" % (g[0], ",".join(g17(v) for v in g[1:]))
        m = d.materials[mi]
        if m[0] == "dielectric":
            ms = "dielectric(%s)" % g17(m[1])
        elif m[0] == "metal":
            ms = "metal(%s,%s)" % (describe_texture(d, m[1]), g17(m[2]))
        else:
            ms = "%s(%s)" % (m[0], describe_texture(d, m[1]))
        lines.append(" ".join([gs, ms] + [g17(v) for v in (M if M is not None else ident)]))
    return lines


def flat_order(world):
    out = []
    for e in world:
        if isinstance(e, tuple):
            out.extend(flat_order(e[1]))
        else:
            out.append(e)
    return out


@pytest.mark.parametrize("exe,gen,kw,wh", [
    ("book_one", "book_one", dict(scene_seed=5), (1600, 800)),
    ("cornell_box", "cornell", dict(), (800, 800)),
    ("cover", "cover", dict(scene_seed=5), (800, 800)),
])
def test_cpp_scene_equals_python_scene(binaries, rt, scenes, exe, gen, kw, wh):
    out = subprocess.run([str(binaries / exe), "--describe", "--scene-seed", "5"], check=True, capture_output=True, text=True).stdout
    lines = out.strip().split("\n")
    d = getattr(scenes, gen)(aspect=wh[0] / wh[1], **kw)
    order = flat_order(d.world) if d.world is not None else None
    assert lines[:-2] == describe(d, order)
    cam = rt.Camera(*d.camera)
    cam_vals = list(cam.c.eye) + list(cam.c.lower_left) + list(cam.c.horizontal) + list(cam.c.vertical) + [cam.c.lens_radius]
    assert lines[-2] == "camera " + " ".join(g17(v) for v in cam_vals)
    info = scenes.build_product(d, device=-1)[0].info()
    assert lines[-1] == "info prims=%d hoisted=%d nodes=%d child_prims=%d" % (info["n_prims"], info["n_hoisted"], info["n_nodes"],
                                                                                info["n_child_prims"])


@pytest.mark.gpu
@pytest.mark.parametrize("exe,gen,W,H,spp", [("book_one", "book_one", 96, 64, 4), ("cornell_box", "cornell", 48, 48, 4),
                                              ("cover", "cover", 40, 40, 2)])
def test_cpp_driver_ppm_is_byte_identical(binaries, rt, scenes, gpu_device, tmp_path, exe, gen, W, H, spp):
    out = tmp_path / "cpp.ppm"
    subprocess.run([str(binaries / exe), "--width", str(W), "--height", str(H), "--spp", str(spp), "--depth", "50", "--seed", "3",
                    "--scene-seed", "2", "--out", str(out)], check=True)
    kw = {} if gen == "cornell" else {"scene_seed": 2}
    sc, cam = scenes.build_product(getattr(scenes, gen)(aspect=W / H, **kw), device=gpu_device)
    ref = tmp_path / "py.ppm"
    rt.write_ppm_p3(ref, sc.render(cam, W, H, spp, 50, seed=3))
    assert out.read_bytes() == ref.read_bytes()
    assert out.read_text().startswith(f"P3\n{W} {H}\n255\n")


@pytest.mark.gpu
def test_cpp_driver_progressive_and_resume(binaries, gpu_device, tmp_path):
    """--passes / --checkpoint: interrupted after the first pass, resumed, byte-identical to the one-shot render."""
    common = ["--width", "64", "--height", "40", "--spp", "12", "--depth", "50", "--seed", "9", "--scene-seed", "3"]
    one = tmp_path / "one.ppm"
    subprocess.run([str(binaries / "book_one"), *common, "--out", str(one)], check=True)
    ck = tmp_path / "ck.bin"
    # first run: 5 of 12 spp, then "crash" (we just render a 5-spp prefix by asking for a checkpointed single pass of 5)
    part = tmp_path / "part.ppm"
    r = subprocess.run([str(binaries / "book_one"), *common, "--passes", "3", "--checkpoint", str(ck), "--out", str(part)],
                       check=True, capture_output=True, text=True)
    assert "4 / 12 spp" in r.stderr and "12 / 12 spp" in r.stderr
    assert part.read_bytes() == one.read_bytes()
    # truncate the checkpoint's progress to 4 spp worth of sums by re-running passes from a saved early state
    import struct
    raw = ck.read_bytes()
    hdr = list(struct.unpack("<12Q", raw[:96]))  # magic, version, w, h, spp, depth, seed, scene seed, scene / camera / kernels hash, s_done
    assert hdr[1:8] == [3, 64, 40, 12, 50, 9, 3] and hdr[11] == 12
    # build a genuine 4-spp checkpoint with the API and let the driver resume it
    import importlib
    import numpy as np
    from conftest import load_package
    rt = load_package()
    scenes = importlib.import_module("ray_tracer_amd.scenes")
    sc, cam = scenes.build_product(scenes.book_one(3, 64 / 40), device=gpu_device)
    sums = np.zeros((40, 64, 3))
    sc.render_progressive(cam, 64, 40, 12, 50, 9, 0, 4, sums)
    assert hdr[8] == sc.scene_hash()  # the header names the scene the sums belong to
    hdr[11] = 4
    ck.write_bytes(struct.pack("<12Q", *hdr) + sums.tobytes())
    res = tmp_path / "resumed.ppm"
    r = subprocess.run([str(binaries / "book_one"), *common, "--passes", "3", "--checkpoint", str(ck), "--out", str(res)],
                       check=True, capture_output=True, text=True)
    assert "resuming" in r.stderr and "at 4 / 12 spp" in r.stderr
    assert res.read_bytes() == one.read_bytes()
    assert not (tmp_path / "ck.bin.tmp").exists()  # replaced atomically
    # a checkpoint of ANOTHER scene (other --scene-seed) or another driver is refused and left alone
    before = ck.read_bytes()
    other = [a if a != "3" else "4" for a in common]
    assert "--scene-seed" in other and other[other.index("--scene-seed") + 1] == "4"
    r = subprocess.run([str(binaries / "book_one"), *other, "--passes", "3", "--checkpoint", str(ck), "--out", str(res)],
                       capture_output=True, text=True)
    assert r.returncode == 3 and "belongs to another render" in r.stderr and ck.read_bytes() == before
    r = subprocess.run([str(binaries / "cornell_box"), *common, "--passes", "3", "--checkpoint", str(ck), "--out", str(res)],
                       capture_output=True, text=True)
    assert r.returncode == 3 and ck.read_bytes() == before
    # ... and so are sums made by other kernels (header field 10 = hash of the library's device code, rt_version "kernels ...")
    alien = list(hdr)
    alien[10] ^= 1
    ck.write_bytes(struct.pack("<12Q", *alien) + sums.tobytes())
    r = subprocess.run([str(binaries / "book_one"), *common, "--passes", "3", "--checkpoint", str(ck), "--out", str(res)],
                       capture_output=True, text=True)
    assert r.returncode == 3 and "kernels" in r.stderr
    ck.write_bytes(before)
    # a truncated file is refused as well (never silently restarted from zero)
    ck.write_bytes(before[:len(before) // 2])
    r = subprocess.run([str(binaries / "book_one"), *common, "--passes", "3", "--checkpoint", str(ck), "--out", str(res)],
                       capture_output=True, text=True)
    assert r.returncode == 3 and "truncated" in r.stderr and ck.read_bytes() == before[:len(before) // 2]


@pytest.mark.gpu
@pytest.mark.parametrize("exe,W,H,spp", [("book_one", 100, 60, 4), ("cover", 40, 40, 2)])
def test_cpp_driver_gpus_flag_does_not_change_the_image(binaries, gpu_device, tmp_path, exe, W, H, spp):
    """--gpus N: one host thread and one committed scene per GPU, tile shards written into one image (no collective).
    On a one-GPU box the devices wrap around, so N = 2 and 3 rehearse the threading; the bytes must not depend on N."""
    outs = []
    for n in (1, 2, 3):
        out = tmp_path / f"g{n}.ppm"
        subprocess.run([str(binaries / exe), "--width", str(W), "--height", str(H), "--spp", str(spp), "--depth", "50", "--seed", "3",
                        "--scene-seed", "2", "--gpus", str(n), "--out", str(out)], check=True)
        outs.append(out.read_bytes())
    assert outs[0] == outs[1] == outs[2]


@pytest.mark.gpu
def test_cpp_cover_driver_writes_the_reference_png(binaries, rt, scenes, gpu_device, tmp_path):
    """--out x.png: the RGBA8 PNG path of examples/main.rs:105-135 (sqrt * 255, min 255, `as u8`, top-down rows)"""
    import sys
    sys.path.insert(0, str(Path(__file__).resolve().parent / "golden"))
    from make_cover_stats import read_png_rgb
    W, H, spp = 48, 40, 3
    out = tmp_path / "cover.png"
    subprocess.run([str(binaries / "cover"), "--width", str(W), "--height", str(H), "--spp", str(spp), "--depth", "50", "--seed", "3",
                    "--scene-seed", "2", "--out", str(out)], check=True)
    sc, cam = scenes.build_product(scenes.cover(2, W / H), device=gpu_device)
    img = sc.render(cam, W, H, spp, 50, seed=3)
    got = read_png_rgb(out)
    assert got.shape == (H, W, 3)
    assert np.array_equal(got, rt.tonemap_png8(img)[::-1])
    # like the reference (`cargo run --example main > image.png`) the cover driver's stdout is that same PNG
    piped = subprocess.run([str(binaries / "cover"), "--width", str(W), "--height", str(H), "--spp", str(spp), "--depth", "50",
                            "--seed", "3", "--scene-seed", "2"], check=True, stdout=subprocess.PIPE).stdout
    assert piped == out.read_bytes()
