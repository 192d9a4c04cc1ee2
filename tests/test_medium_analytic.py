"""ConstantMedium + Isotropic against ANALYTIC answers (Beer-Lambert), independent of both restatements.

cover.png cannot pin the medium path (tests/test_cover_png.py: the picture's blue ball predates today's source), and the
oracle and the lane program come from the same reading of src/volume.rs:46-100.  What neither reading can bend is the
physics the code implements: a ray that crosses a medium of density rho over a chord of length L (unit direction) is
scattered with probability 1 - exp(-rho L)  (`distance = -(1/rho) ln U`, hit iff distance <= L).  With a black Isotropic
albedo every scattered path contributes exactly 0 and every unscattered one the sky's emission 1, so a pixel is a binomial
mean with p = exp(-rho L), L from plain geometry done here in numpy (chord of a sphere / of an oriented box / distance to the
exit for an origin inside).  The per-pixel z-scores must look like a standard normal (mean ~ 0, mean square ~ 1): a wrong
free-flight law (rho in the wrong place, L measured on the wrong ray, the entering / inside branches swapped, a chord taken
between the wrong boundary hits) shifts them by tens of sigma.
Covered: ConstantMedium<Sphere> under a translation (kernel family "sphere media", RT_PRIM_MEDIUM_T), ConstantMedium<Cube>
under a rotation and ConstantMedium<Sphere> behind a TransformedGeometry (family "media over general boundaries",
RT_PRIM_MEDIUM_C), a camera INSIDE the medium (src/volume.rs:76-98), and a white furnace (albedo 1 inside emission 1: every
pixel exactly 1).  Each case runs on the oracle (CPU) and on the HIP path (-m gpu)."""
import math

import numpy as np
import pytest

W = H = 48
SUB = 6  # analytic sub-samples per pixel edge


def _unit(v):
    v = np.asarray(v, dtype=np.float64)
    return v / np.linalg.norm(v)


def camera_rays(desc, us, vs):
    """PerspectiveCamera::new / ray (src/camera.rs:25-59, 91-106) for a pinhole: origin, unit directions [n, 3]"""
    eye, center, up, fov, aspect, focus, lens = desc.camera
    assert lens == 0.0
    eye, center, up = (np.asarray(a, dtype=np.float64) for a in (eye, center, up))
    h = math.tan(fov / 2.0) * 2.0
    wd = aspect * h
    w = _unit(eye - center)
    u = np.cross(_unit(up), w)  # not normalised (quirk Q1); the cameras below keep up perpendicular to w
    v = np.cross(w, u)
    hor, ver = u * wd * focus, v * h * focus
    ll = eye - hor / 2.0 - ver / 2.0 - w * focus
    d = ll[None, :] + us[:, None] * hor[None, :] + vs[:, None] * ver[None, :] - eye[None, :]
    return eye, d / np.linalg.norm(d, axis=1, keepdims=True)


def sphere_chord(o, d, c, r):
    """length of the part of the ray (t >= 0) inside the sphere"""
    oc = o[None, :] - np.asarray(c)[None, :]
    b = (oc * d).sum(1)
    disc = b * b - ((oc * oc).sum(1) - r * r)
    sq = np.sqrt(np.maximum(disc, 0.0))
    t0, t1 = np.maximum(-b - sq, 0.0), np.maximum(-b + sq, 0.0)
    return np.where(disc > 0.0, t1 - t0, 0.0)


def box_chord(o, d, half, R, t):
    """length of the ray inside the box [-half, half]^3 placed by x -> R x + t (R a rotation)"""
    lo = (o[None, :] - t[None, :]) @ R  # R^T (o - t)
    ld = d @ R
    with np.errstate(divide="ignore", invalid="ignore"):
        ta, tb = (-half[None, :] - lo) / ld, (half[None, :] - lo) / ld
    tn, tf = np.minimum(ta, tb).max(1), np.maximum(ta, tb).min(1)
    return np.maximum(tf - np.maximum(tn, 0.0), 0.0)


def expected_image(desc, chord_fn, rho):
    """mean of exp(-rho L) over each pixel's footprint (examples/book-one.rs:70-72: u = (x + U) / W), SUB x SUB midpoints"""
    ys, xs, sy, sx = np.meshgrid(np.arange(H), np.arange(W), (np.arange(SUB) + 0.5) / SUB, (np.arange(SUB) + 0.5) / SUB, indexing="ij")
    us, vs = ((xs + sx) / W).ravel(), ((ys + sy) / H).ravel()
    o, d = camera_rays(desc, us, vs)
    return np.exp(-rho * chord_fn(o, d)).reshape(H, W, SUB * SUB).mean(2)


def rot_y(a):
    c, s = math.cos(a), math.sin(a)
    return np.array([[c, 0.0, s], [0.0, 1.0, 0.0], [-s, 0.0, c]])  # src/mat4.rs:61-78 about ey, as a 3x3 acting on columns


def make_scene(scenes, kind):
    """-> (SceneDesc, chord function, rho).  Sky: DiffuseLight (1,1,1) sphere around everything; medium: black Isotropic."""
    d = scenes.SceneDesc(name="medium-" + kind)
    d.sprite(d.geom("sphere", 500.0), d.mat("diffuse_light", d.tex_solid((1.0, 1.0, 1.0))), None)
    black = d.mat("isotropic", d.tex_solid((0.0, 0.0, 0.0)))
    cam_eye, cam_center = (0.3, 0.2, -6.0), (0.3, 0.2, 0.0)
    if kind == "sphere":  # examples/main.rs:241-263 shape: Sprite<ConstantMedium<Sphere>> under a translation
        rho, c, r = 0.7, (0.2, 0.1, 0.5), 1.3
        d.sprite(d.geom("medium", d.geom("sphere", r), rho), black, scenes.mat4_translation(c))
        fn = lambda o, dd: sphere_chord(o, dd, c, r)  # noqa: E731
    elif kind == "cube":  # the book's smoke box: ConstantMedium over Cube::new's six rectangles, rotated
        rho, ang, t = 0.9, 0.5, np.array([0.1, 0.3, 0.4])
        dims = np.array([1.8, 1.2, 2.2])
        M = scenes.mat4_multiplied(scenes.mat4_translation(t), scenes.mat4_rotation(ang, (0.0, 1.0, 0.0)))
        d.sprite(d.geom("medium", d.geom("cube", *dims), rho), black, M)
        R = rot_y(ang)
        fn = lambda o, dd: box_chord(o, dd, dims / 2.0, R, t)  # noqa: E731
    elif kind == "transformed":  # ConstantMedium<Sphere> behind a TransformedGeometry: the general-boundary path over a sphere
        rho, r = 1.1, 1.0
        inner, outer = (0.4, -0.2, 0.3), (-0.3, 0.5, 0.2)
        g = d.geom("transformed", d.geom("medium", d.geom("sphere", r), rho), scenes.mat4_translation(inner))
        d.sprite(g, black, scenes.mat4_multiplied(scenes.mat4_translation(outer), scenes.mat4_rotation(0.7, (0.0, 0.0, 1.0))))
        c = np.array(outer) + rot_z(0.7) @ np.array(inner)
        fn = lambda o, dd: sphere_chord(o, dd, c, r)  # noqa: E731
    elif kind == "inside":  # the camera sits inside the medium: `the origin is inside` branch, src/volume.rs:76-98
        rho, c, r = 0.25, (0.0, 0.0, -5.0), 4.0
        d.sprite(d.geom("medium", d.geom("sphere", r), rho), black, scenes.mat4_translation(c))
        fn = lambda o, dd: sphere_chord(o, dd, c, r)  # noqa: E731
    else:
        raise ValueError(kind)
    d.camera = (cam_eye, cam_center, (0.0, 1.0, 0.0), math.radians(35.0), 1.0, 10.0, 0.0)
    return d, fn, rho


def rot_z(a):
    c, s = math.cos(a), math.sin(a)
    return np.array([[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]])


def assert_beer_lambert(img, want, spp, label):
    got = img[..., 0]
    assert np.array_equal(img[..., 0], img[..., 1]) and np.array_equal(img[..., 0], img[..., 2])
    # pixels the medium does not cover: exactly the sky
    clear = want == 1.0
    far = clear.copy()  # ... and whose eight neighbours are clear too (the SUB x SUB footprint can miss a sliver a sample finds)
    for dy in (-1, 0, 1):
        for dx in (-1, 0, 1):
            far &= np.roll(np.roll(clear, dy, 0), dx, 1)
    assert np.all(got[far] == 1.0), label  # (no such pixel when the camera is inside the medium)
    # binomial means inside the medium's silhouette (pixels whose eight neighbours are inside too: on the silhouette the
    # SUB x SUB footprint is a poor integral)
    p = np.clip(want, 1e-9, 1.0 - 1e-9)
    sigma = np.sqrt(p * (1.0 - p) / spp)
    inner = want < 0.98
    core = inner.copy()
    for dy in (-1, 0, 1):
        for dx in (-1, 0, 1):
            core &= np.roll(np.roll(inner, dy, 0), dx, 1)
    part = ~far
    assert core.sum() > 200, (label, int(core.sum()))
    z = (got[core] - want[core]) / sigma[core]
    assert abs(z.mean()) < 4.0 / math.sqrt(core.sum()) + 0.05, (label, "mean z", z.mean())
    assert 0.8 < (z * z).mean() < 1.3, (label, "mean z^2", (z * z).mean())
    assert np.abs(z).max() < 6.0, (label, "max |z|", np.abs(z).max())
    # and the whole image's mean transmittance, edges included, to a few parts in a thousand
    assert abs(got.mean() - want.mean()) < 5.0 * math.sqrt((sigma[part] ** 2).sum()) / got.size + 2e-3, (label, got.mean(), want.mean())


KINDS = ("sphere", "cube", "transformed", "inside")


@pytest.mark.parametrize("kind", KINDS)
def test_oracle_transmittance_is_beer_lambert(scenes, oracle, kind):
    desc, fn, rho = make_scene(scenes, kind)
    spp = 256
    img = oracle.build_oracle(desc).render(W, H, spp, 6, seed=11, nthreads=8)
    assert_beer_lambert(img, expected_image(desc, fn, rho), spp, kind)


@pytest.mark.gpu
@pytest.mark.parametrize("kind", KINDS)
def test_gpu_transmittance_is_beer_lambert(rt, scenes, gpu_device, kind):
    desc, fn, rho = make_scene(scenes, kind)
    spp = 4096
    sc, cam = scenes.build_product(desc, device=gpu_device)
    img = sc.render(cam, W, H, spp, 6, seed=11)
    assert_beer_lambert(img, expected_image(desc, fn, rho), spp, kind)


def furnace(scenes):
    """albedo-1 Isotropic media (a sphere, a rotated cube) inside an emission-1 sky: energy is conserved exactly"""
    d = scenes.SceneDesc(name="medium-furnace")
    d.sprite(d.geom("sphere", 500.0), d.mat("diffuse_light", d.tex_solid((1.0, 1.0, 1.0))), None)
    white = d.mat("isotropic", d.tex_solid((1.0, 1.0, 1.0)))
    d.sprite(d.geom("medium", d.geom("sphere", 1.2), 2.0), white, scenes.mat4_translation((-1.0, 0.0, 0.0)))
    d.sprite(d.geom("medium", d.geom("cube", 1.5, 1.5, 1.5), 1.5), white,
             scenes.mat4_multiplied(scenes.mat4_translation((1.2, 0.0, 0.3)), scenes.mat4_rotation(0.4, (0.0, 1.0, 0.0))))
    d.camera = ((0.0, 0.3, -6.0), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), math.radians(40.0), 1.0, 10.0, 0.0)
    return d


def test_oracle_white_medium_in_a_white_furnace(scenes, oracle):
    img = oracle.build_oracle(furnace(scenes)).render(W, H, 32, 100, seed=3, nthreads=8, iterative=True)
    assert img.max() == 1.0 and img.mean() > 0.9999  # a path of 100 scatterings without leaving is the only way to lose energy


@pytest.mark.gpu
def test_gpu_white_medium_in_a_white_furnace(rt, scenes, gpu_device):
    sc, cam = scenes.build_product(furnace(scenes), device=gpu_device)
    img = sc.render(cam, W, H, 512, 100, seed=3)
    assert img.max() == 1.0 and img.mean() > 0.9999
