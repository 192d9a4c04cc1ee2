"""Scene descriptions of the reference's three example drivers, as neutral data.

Harness glue: a ``SceneDesc`` is plain lists (textures, materials, geometries,
sprites, camera) that tests/bench feed both to librt_mi355x (``build_product``)
and to the CPU oracle (tests/oracle_binding.py), so both sides see bit-identical
inputs.  Generators restate

  * ``randomScene()``  examples/book-one.rs:103-205, camera :35-47
  * cornell box        examples/cornell-box.rs:31-150
  * ``finalScene()``   examples/main.rs:156-330, camera :49-61

with the unseedable ``thread_rng()`` replaced by the host stream of
include/rt_rng.h (stream RT_RNG_SCENE_STREAM of ``scene_seed``).  ``./earthmap.jpg``
(examples/main.rs:266) is not in the reference repository and cannot be
downloaded here; ``earth_texture()`` is a deterministic procedural stand-in that
goes through the same nearest-texel rule.
"""
from __future__ import annotations

import math
import struct
from dataclasses import dataclass, field

import numpy as np

MASK = (1 << 64) - 1
GAMMA = 0x9E3779B97F4A7C15
SCENE_STREAM = (1 << 40) - 1


def mix64(z: int) -> int:
    z &= MASK
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & MASK
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & MASK
    return z ^ (z >> 31)


class HostRng:
    """Python mirror of include/rt_rng.h (checked against the C definition in tests)."""

    def __init__(self, seed: int, stream: int = SCENE_STREAM):
        self.base = (mix64(seed) + ((stream << 24) & MASK) * GAMMA) & MASK
        self.s0 = mix64((self.base + GAMMA) & MASK)
        self.s1 = mix64((self.base + 2 * GAMMA) & MASK)
        if (self.s0 | self.s1) == 0:
            self.s0 = GAMMA

    @staticmethod
    def _rotl(x: int, k: int) -> int:
        return ((x << k) | (x >> (64 - k))) & MASK

    def next_u64(self) -> int:  # xoroshiro128+ (a=24, b=16, c=37)
        s0, s1 = self.s0, self.s1
        r = (s0 + s1) & MASK
        s1 ^= s0
        self.s0 = self._rotl(s0, 24) ^ s1 ^ ((s1 << 16) & MASK)
        self.s1 = self._rotl(s1, 37)
        return r

    def gen_range(self, low: float, high: float) -> float:
        scale = high - low
        offset = low - scale
        while True:
            v12 = struct.unpack("<d", struct.pack("<Q", (self.next_u64() >> 12) | 0x3FF0000000000000))[0]
            res = v12 * scale + offset  # python floats: separate multiply and add
            if res < high:
                return res


@dataclass
class SceneDesc:
    textures: list = field(default_factory=list)    # ("solid", (r,g,b)) | ("checker", a, b) | ("image", ndarray HxWx3 u8)
    materials: list = field(default_factory=list)   # ("lambertian", tex) | ("metal", tex, fuzz) | ("dielectric", ior) | ("diffuse_light", tex) | ("isotropic", tex)
    geometries: list = field(default_factory=list)  # ("sphere", r) | ("rectangle", w, h) | ("cube", w, h, d) | ("medium", geom, density)
                                                    # | ("transformed", geom, M16) | ("bvh", [sprite ids]): a node used as a geometry
                                                    #   (instancing); its sprites must precede the sprite that carries it
    sprites: list = field(default_factory=list)     # (geom | None, mat | None, M16 list | None)
    world: list | None = None                       # nesting of sprite ids as the reference nests BVH nodes; None = all sprites flat
    camera: tuple = ()                              # (eye, center, up, fov, aspect, focus, lens)
    name: str = ""

    # builder helpers that de-duplicate like shared Arcs
    def tex_solid(self, rgb):
        self.textures.append(("solid", tuple(float(c) for c in rgb)))
        return len(self.textures) - 1

    def mat(self, *m):
        self.materials.append(tuple(m))
        return len(self.materials) - 1

    def geom(self, *g):
        self.geometries.append(tuple(g))
        return len(self.geometries) - 1

    def sprite(self, g, m, M=None):
        self.sprites.append((g, m, None if M is None else [float(v) for v in M]))
        return len(self.sprites) - 1

    def lambertian_rgb(self, rgb):
        return self.mat("lambertian", self.tex_solid(rgb))


# ---- Mat4 in python floats, same operation order as src/mat4.rs (cross-checked in tests) ----
def mat4_translation(t):
    m = [0.0] * 16
    m[0] = m[5] = m[10] = m[15] = 1.0
    m[12], m[13], m[14] = float(t[0]), float(t[1]), float(t[2])
    return m


def mat4_rotation(radians, axis):
    x, y, z = axis
    s, c = math.sin(radians), math.cos(radians)
    t = 1.0 - c
    return [x * x * t + c, y * x * t + z * s, z * x * t - y * s, 0.0,
            x * y * t - z * s, y * y * t + c, z * y * t + x * s, 0.0,
            x * z * t + y * s, y * z * t - x * s, z * z * t + c, 0.0,
            0.0, 0.0, 0.0, 1.0]


def mat4_multiplied(a, b):
    out = [0.0] * 16
    for col in range(4):
        b0, b1, b2, b3 = b[col * 4:col * 4 + 4]
        for row in range(4):
            out[col * 4 + row] = b0 * a[row] + b1 * a[4 + row] + b2 * a[8 + row] + b3 * a[12 + row]
    return out


def radians(deg):  # f64::to_radians
    return deg * (math.pi / 180.0)


# ------------------------------------------------------------------ book-one
def book_one(scene_seed: int = 1, aspect: float = 1.5) -> SceneDesc:
    """randomScene() of examples/book-one.rs:103-205 + camera :35-47 (aspect = W/H)."""
    d = SceneDesc(name="book-one")
    g = HostRng(scene_seed)
    sph1000, sph2000, sph02, sph1 = d.geom("sphere", 1000.0), d.geom("sphere", 2000.0), d.geom("sphere", 0.2), d.geom("sphere", 1.0)
    d.sprite(sph1000, d.lambertian_rgb((0.5, 0.5, 0.5)), mat4_translation((0.0, -1000.0, 0.0)))  # ground
    d.sprite(sph2000, d.mat("diffuse_light", d.tex_solid((0.5, 0.7, 1.0))), None)                # sky sphere
    for a in range(-11, 11):
        for b in range(-11, 11):
            which = g.gen_range(0.0, 1.0)
            cx = float(a) + 0.9 * g.gen_range(0.0, 1.0)
            cz = float(b) + 0.9 * g.gen_range(0.0, 1.0)
            dx, dy, dz = cx - 4.0, 0.2 - 0.2, cz - 0.0
            if math.sqrt(dx * dx + dy * dy + dz * dz) > 0.9:
                if which < 0.3:
                    alb = [g.gen_range(0.0, 1.0) for _ in range(3)]
                    alb = [c * c for c in alb]
                    m = d.lambertian_rgb(alb)
                elif which < 0.6:
                    alb = [g.gen_range(0.5, 1.0) for _ in range(3)]
                    fuzz = g.gen_range(0.0, 0.5)
                    m = d.mat("metal", d.tex_solid(alb), fuzz)
                else:
                    m = d.mat("dielectric", 1.5)  # one material per sprite, like the reference
                d.sprite(sph02, m, mat4_translation((cx, 0.2, cz)))
    d.sprite(sph1, d.lambertian_rgb((0.4, 0.2, 0.1)), mat4_translation((-4.0, 1.0, 0.0)))
    d.sprite(sph1, d.mat("metal", d.tex_solid((0.7, 0.6, 0.5)), 0.0), mat4_translation((4.0, 1.0, 0.0)))
    d.sprite(sph1, d.mat("dielectric", 1.5), mat4_translation((0.0, 1.0, 0.0)))
    d.camera = ((13.0, 2.0, 3.0), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), radians(20.0), float(aspect), 10.0, 0.05)
    return d


# ------------------------------------------------------------------ cornell
def cornell(aspect: float = 1.0) -> SceneDesc:
    """examples/cornell-box.rs:31-150."""
    d = SceneDesc(name="cornell-box")
    red = d.lambertian_rgb((0.65, 0.05, 0.05))
    white = d.lambertian_rgb((0.73, 0.73, 0.73))
    green = d.lambertian_rgb((0.12, 0.45, 0.15))
    light = d.mat("diffuse_light", d.tex_solid((15.0, 15.0, 15.0)))
    ex, ey = (1.0, 0.0, 0.0), (0.0, 1.0, 0.0)

    def tr(t, deg, axis):
        return mat4_multiplied(mat4_translation(t), mat4_rotation(radians(deg), axis))

    r555 = d.geom("rectangle", 555.0, 555.0)
    d.sprite(r555, green, tr((555.0, 555.0 / 2.0, 555.0 / 2.0), -90.0, ey))
    d.sprite(d.geom("rectangle", 555.0, 555.0), red, tr((0.0, 555.0 / 2.0, 555.0 / 2.0), 90.0, ey))
    d.sprite(d.geom("rectangle", 130.0, 105.0), light, tr((555.0 / 2.0, 554.0, 555.0 / 2.0), 90.0, ex))
    d.sprite(d.geom("rectangle", 555.0, 555.0), white, tr((555.0 / 2.0, 0.0, 555.0 / 2.0), -90.0, ex))
    d.sprite(d.geom("rectangle", 555.0, 555.0), white, tr((555.0 / 2.0, 555.0, 555.0 / 2.0), 90.0, ex))
    d.sprite(d.geom("rectangle", 555.0, 556.0), white, tr((555.0 / 2.0, 555.0 / 2.0, 555.0), 180.0, ey))
    d.sprite(d.geom("cube", 165.0, 165.0, 165.0), white, tr((212.5, 82.5, 147.5), -18.0, ey))
    d.sprite(d.geom("cube", 165.0, 330.0, 165.0), white, tr((347.5, 165.0, 377.5), 15.0, ey))
    d.camera = ((555.0 / 2.0, 555.0 / 2.0, -800.0), (555.0 / 2.0, 555.0 / 2.0, 0.0), (0.0, 1.0, 0.0), radians(40.0),
                float(aspect), 10.0, 0.0)
    return d


def cube_row(n_cubes: int, levels: int = 1, aspect: float = 1.0) -> SceneDesc:
    """n_cubes rotated boxes (6 leaves each) over a floor, under a light: small general scenes around the limits of the box-LIST walk
    (24 leaves) and of the records the LIST kernels keep in LDS (tests); levels > 1 nests every cube in that many TransformedGeometry
    levels (examples/cornell-box.rs:114-133 builds its boxes the same way, with one level)."""
    d = SceneDesc(name=f"cube-row-{n_cubes}x{levels}")
    white = d.lambertian_rgb((0.73, 0.73, 0.73))
    blue = d.lambertian_rgb((0.2, 0.3, 0.7))
    light = d.mat("diffuse_light", d.tex_solid((7.0, 7.0, 7.0)))
    ex, ey = (1.0, 0.0, 0.0), (0.0, 1.0, 0.0)
    d.sprite(d.geom("rectangle", 900.0, 900.0), white, mat4_multiplied(mat4_translation((0.0, 0.0, 0.0)), mat4_rotation(radians(-90.0), ex)))
    d.sprite(d.geom("rectangle", 400.0, 400.0), light, mat4_multiplied(mat4_translation((0.0, 500.0, 0.0)), mat4_rotation(radians(90.0), ex)))
    for i in range(n_cubes):
        g = d.geom("cube", 90.0, 120.0 + 25.0 * i, 90.0)
        for k in range(levels - 1):
            g = d.geom("transformed", g, mat4_multiplied(mat4_translation((3.0 * (k + 1), 0.0, -2.0 * k)), mat4_rotation(radians(4.0 + 3.0 * k), ey)))
        x = -60.0 * (n_cubes - 1) + 120.0 * i
        d.sprite(g, blue if i % 2 else white, mat4_multiplied(mat4_translation((x, 60.0 + 12.5 * i, 30.0 * (i % 3))), mat4_rotation(radians(-20.0 + 17.0 * i), ey)))
    d.camera = ((0.0, 260.0, -900.0), (0.0, 120.0, 0.0), (0.0, 1.0, 0.0), radians(40.0), float(aspect), 10.0, 0.0)
    return d


def slab_stack(n: int = 22, aspect: float = 1.0) -> SceneDesc:
    """n parallel rectangles one behind the other, every second one a mirror, lit from the front: a camera ray crosses the boxes of
    ALL of them, so the box-list walk pushes n - 1 entries -- the deepest a list scene's stack gets (tests)."""
    d = SceneDesc(name=f"slab-stack-{n}")
    white = d.lambertian_rgb((0.8, 0.8, 0.8))
    mirror = d.mat("metal", d.tex_solid((0.9, 0.9, 0.9)), 0.0)
    light = d.mat("diffuse_light", d.tex_solid((4.0, 4.0, 4.0)))
    ey = (0.0, 1.0, 0.0)
    d.sprite(d.geom("rectangle", 60.0, 60.0), light, mat4_multiplied(mat4_translation((0.0, 0.0, -40.0)), mat4_rotation(radians(180.0), ey)))
    for i in range(n - 1):
        size = 6.0 + 2.0 * i  # the nearer ones are smaller: every one is seen past the edges of those in front
        d.sprite(d.geom("rectangle", size, size), mirror if i % 2 else white,
                 mat4_multiplied(mat4_translation((0.3 * i, -0.2 * i, 3.0 * i)), mat4_rotation(radians(180.0 + 2.0 * i), ey)))  # (normals towards the camera)
    d.camera = ((0.0, 0.0, -30.0), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), radians(50.0), float(aspect), 10.0, 0.0)
    return d


# ------------------------------------------------------------------ book-two cover
def earth_texture(w: int = 1024, h: int = 512) -> np.ndarray:
    """Deterministic procedural stand-in for ./earthmap.jpg (absent upstream, examples/main.rs:266)."""
    yy, xx = np.mgrid[0:h, 0:w]
    lat = (yy.astype(np.int64) * 180) // h
    lon = (xx.astype(np.int64) * 360) // w
    land = (((lon // 30) + (lat // 20)) % 2 == 0) & (lat > 20) & (lat < 160)
    img = np.zeros((h, w, 3), dtype=np.uint8)
    img[..., 0] = np.where(land, 60 + (lat % 64), 20)
    img[..., 1] = np.where(land, 140 - (lat % 40), 60 + (lon % 50))
    img[..., 2] = np.where(land, 50, 160 + (lat % 80))
    ice = (lat <= 12) | (lat >= 168)
    img[ice] = (235, 240, 245)
    return img


def cover(scene_seed: int = 1, aspect: float = 1.0, with_fog: bool = True, heights=None) -> SceneDesc:
    """finalScene() of examples/main.rs:156-330 + camera :49-61.
    heights: optional 20x20 array overriding the random floor-box heights y1[i][j] (NaN = keep the random one); the
    generator is advanced all the same, so everything else of the scene stays as `scene_seed` makes it
    (used by the cover.png probes, tools/blue_probe.py, tools/fit_cover_floor.py)."""
    d = SceneDesc(name="book-two-cover")
    g = HostRng(scene_seed)
    ground = d.lambertian_rgb((0.48, 0.83, 0.53))
    cubes = []
    for i in range(20):
        for j in range(20):
            w = 100.0
            x0 = -1000.0 + float(i) * w
            y0 = 0.0
            z0 = -1000.0 + float(j) * w
            x1 = x0 + w
            y1 = g.gen_range(1.0, 101.0)
            if heights is not None and heights[i][j] == heights[i][j]:
                y1 = float(heights[i][j])
            z1 = z0 + w
            cubes.append(d.sprite(d.geom("cube", x1 - x0, y1 - y0, z1 - z0), ground,
                                  mat4_translation(((x0 + x1) / 2.0, (y0 + y1) / 2.0, (z0 + z1) / 2.0))))
    light = d.sprite(d.geom("rectangle", 300.0, 265.0), d.mat("diffuse_light", d.tex_solid((7.0, 7.0, 7.0))),
                     mat4_multiplied(mat4_translation((273.0, 554.0, 279.5)), mat4_rotation(radians(90.0), (1.0, 0.0, 0.0))))
    s50 = d.geom("sphere", 50.0)
    moving = d.sprite(s50, d.lambertian_rgb((0.7, 0.3, 0.1)), mat4_translation((400.0, 400.0, 200.0)))
    glass = d.sprite(d.geom("sphere", 50.0), d.mat("dielectric", 1.5), mat4_translation((260.0, 150.0, 45.0)))
    metal = d.sprite(d.geom("sphere", 50.0), d.mat("metal", d.tex_solid((0.8, 0.8, 0.9)), 1.0),
                     mat4_translation((0.0, 150.0, 145.0)))
    blue_surface = d.sprite(d.geom("sphere", 70.0), d.mat("dielectric", 1.5), mat4_translation((360.0, 150.0, 145.0)))
    blue_medium = d.sprite(d.geom("medium", d.geom("sphere", 70.0 - 1e-6), 0.03),
                           d.mat("isotropic", d.tex_solid((0.2, 0.4, 0.9))), mat4_translation((360.0, 150.0, 145.0)))
    fog = d.sprite(d.geom("medium", d.geom("sphere", 5000.0), 0.0001), d.mat("isotropic", d.tex_solid((1.0, 1.0, 1.0))),
                   None) if with_fog else None
    d.textures.append(("image", earth_texture()))
    earth = d.sprite(d.geom("sphere", 100.0), d.mat("lambertian", len(d.textures) - 1), mat4_translation((400.0, 200.0, 400.0)))
    white = d.lambertian_rgb((0.73, 0.73, 0.73))
    s10 = d.geom("sphere", 10.0)
    spheres = []
    for _ in range(1000):
        x = g.gen_range(0.0, 165.0)
        y = g.gen_range(0.0, 165.0)
        z = g.gen_range(0.0, 165.0)
        spheres.append(d.sprite(s10, white, mat4_translation((x - 100.0, y + 270.0, z + 395.0))))
    # nesting of examples/main.rs:316-327 (nested BVH nodes have no transform)
    d.world = [("bvh", cubes), light, moving, glass, metal, blue_surface, blue_medium, earth] + \
              ([fog] if with_fog else []) + [("bvh", spheres)]
    d.camera = ((555.0 / 2.0 + 200.0, 550.0 / 2.0, -600.0), (555.0 / 2.0, 555.0 / 2.0, 0.0), (0.0, 1.0, 0.0), radians(40.0),
                float(aspect), 10.0, 0.0)
    return d


# ------------------------------------------------------------------ product side
def build_product(desc: SceneDesc, device: int = 0):
    """Feed a SceneDesc through the C ABI; returns (Scene, Camera)."""
    import importlib
    import sys
    rt = sys.modules.get("ray_tracer_amd") or importlib.import_module("ray_tracer_amd")
    sc = rt.Scene()
    for t in desc.textures:
        if t[0] == "solid":
            sc.solid(t[1])
        elif t[0] == "checker":
            sc.checker(t[1], t[2])
        elif t[0] == "image":
            sc.image(t[1])
        else:
            raise ValueError(t[0])
    for m in desc.materials:
        getattr(sc, m[0])(*m[1:])
    # geometries are created on demand so that a node geometry ("bvh") finds its sprites already recorded; sprites are
    # recorded strictly in description order (their indices key the random draws of instanced media, include/rt_rng.h)
    gid = {}

    def geometry(i):
        if i is None:
            return None
        if i not in gid:
            gm = desc.geometries[i]
            if gm[0] == "medium":
                gid[i] = sc.constant_medium(geometry(gm[1]), gm[2])
            elif gm[0] == "transformed":
                gid[i] = sc.transformed(geometry(gm[1]), gm[2])
            elif gm[0] == "bvh":
                assert all(c < n_done[0] for c in gm[1]), "a node's sprites must be described before the sprite that carries the node"
                gid[i] = sc.bvh(gm[1])
            else:
                gid[i] = getattr(sc, gm[0])(*gm[1:])
        return gid[i]

    n_done = [0]
    for (gi, mi, M) in desc.sprites:
        got = sc.sprite(geometry(gi), mi, M)
        assert got == n_done[0]
        n_done[0] += 1
    sc.commit(device)
    cam = rt.Camera(*desc.camera)
    return sc, cam


# ------------------------------------------------------------------ instancing
def instanced(aspect: float = 1.0, seed: int = 1) -> SceneDesc:
    """Not one of the reference's examples: what its API allows beyond them.  A cluster of spheres, a rectangle and a cube
    (a BoundingVolumeHierarchyNode used as a sprite's GEOMETRY, src/sprite.rs:87-93) instanced three times under rotated /
    scaled transforms, one instance of instances (four transform levels above the cube faces), the book's Cornell smoke
    boxes (ConstantMedium over a rotated cube and over a node of two spheres, src/volume.rs:40-100), one of them instanced twice
    (media keyed per instance), a ConstantMedium behind a TransformedGeometry, a checker-textured medium, and a child that
    carries a material of its own (replaced by the instancing sprite's, src/sprite.rs:119-127)."""
    d = SceneDesc(name="instanced")
    g = HostRng(seed)
    ex, ey, ez = (1.0, 0.0, 0.0), (0.0, 1.0, 0.0), (0.0, 0.0, 1.0)

    def tr(t, rad=0.0, axis=ey):
        return mat4_multiplied(mat4_translation(t), mat4_rotation(rad, axis))

    white = d.lambertian_rgb((0.73, 0.73, 0.73))
    red = d.lambertian_rgb((0.65, 0.05, 0.05))
    glass = d.mat("dielectric", 1.5)
    steel = d.mat("metal", d.tex_solid((0.8, 0.8, 0.9)), 0.2)
    d.textures.append(("checker", d.tex_solid((0.1, 0.1, 0.4)), d.tex_solid((0.9, 0.9, 0.9))))
    checker = len(d.textures) - 1
    # the cluster: five small spheres, a rectangle, a cube (children; the red one's own material is never seen)
    kids = []
    s_small = d.geom("sphere", 0.4)
    for _ in range(5):
        kids.append(d.sprite(s_small, None, mat4_translation((g.gen_range(-1.0, 1.0), g.gen_range(-1.0, 1.0), g.gen_range(-1.0, 1.0)))))
    kids.append(d.sprite(d.geom("rectangle", 2.5, 1.0), red, tr((0.0, -1.2, 0.0), radians(-90.0), ex)))
    kids.append(d.sprite(d.geom("cube", 0.8, 1.2, 0.6), None, tr((0.9, 0.3, -0.8), 0.5, ey)))
    cluster = d.geom("bvh", kids)
    scale = [1.6, 0.0, 0.0, 0.0, 0.0, 0.7, 0.0, 0.0, 0.0, 0.0, 1.1, 0.0, 0.0, 0.0, 0.0, 1.0]  # non-rigid on purpose (quirk Q5)
    c1 = d.sprite(cluster, white, tr((-4.0, 1.5, 2.0), 0.3, ey))
    c2 = d.sprite(cluster, steel, mat4_multiplied(tr((0.0, 1.8, 3.5), -0.8, ez), scale))
    c3 = d.sprite(cluster, glass, tr((4.0, 1.5, 2.0), 1.1, ex))
    # an instance of instances: two clusters under one more node, placed twice
    pair = d.geom("bvh", [d.sprite(cluster, None, tr((-1.5, 0.0, 0.0), 0.2, ey)), d.sprite(cluster, None, tr((1.5, 0.4, 0.0), -0.4, ez))])
    p1 = d.sprite(pair, white, tr((-2.5, 5.0, 6.0), 0.6, ey))
    p2 = d.sprite(pair, steel, tr((3.0, 5.5, 7.0), -0.9, ex))
    # smoke: over a rotated cube, over a node of two spheres (instanced twice), behind a TransformedGeometry, textured
    smoke_dark = d.mat("isotropic", d.tex_solid((0.05, 0.05, 0.05)))
    smoke_light = d.mat("isotropic", d.tex_solid((0.95, 0.95, 0.95)))
    m1 = d.sprite(d.geom("medium", d.geom("cube", 1.6, 2.4, 1.6), 0.9), smoke_dark, tr((-2.0, 1.2, -1.0), radians(15.0), ey))
    two = d.geom("bvh", [d.sprite(d.geom("sphere", 0.8), None, mat4_translation((-0.5, 0.0, 0.0))),
                         d.sprite(d.geom("sphere", 0.6), None, mat4_translation((0.6, 0.2, 0.0)))])
    blob = d.geom("bvh", [d.sprite(d.geom("medium", two, 1.5), None, tr((0.0, 0.0, 0.0), 0.3, ez))])
    m2 = d.sprite(blob, smoke_light, tr((1.8, 0.9, -1.2), 0.4, ey))
    m3 = d.sprite(blob, smoke_dark, tr((3.6, 1.0, -0.2), -0.7, ex))
    m4 = d.sprite(d.geom("transformed", d.geom("medium", d.geom("sphere", 0.7), 1.2), tr((0.0, 0.3, 0.0), 0.9, ex)),
                  d.mat("isotropic", checker), tr((0.0, 0.8, -2.0), 0.2, ey))
    floor = d.sprite(d.geom("rectangle", 40.0, 40.0), d.mat("lambertian", checker), tr((0.0, 0.0, 0.0), radians(-90.0), ex))
    lamp = d.sprite(d.geom("rectangle", 6.0, 6.0), d.mat("diffuse_light", d.tex_solid((6.0, 6.0, 6.0))), tr((0.0, 9.0, 2.0), radians(90.0), ex))
    sky = d.sprite(d.geom("sphere", 80.0), d.mat("diffuse_light", d.tex_solid((0.25, 0.3, 0.4))), None)
    d.world = [c1, c2, c3, p1, p2, m1, m2, m3, m4, floor, lamp, sky]
    d.camera = ((0.0, 4.0, -11.0), (0.0, 2.0, 2.0), (0.0, 1.0, 0.0), radians(45.0), float(aspect), 10.0, 0.0)
    return d


def deep_chains(aspect: float = 1.0, seed: int = 1) -> SceneDesc:
    """Not one of the reference's examples: transform nesting beyond the four levels every kernel family unrolls (src/sprite.rs:87-93
    and src/geometry.rs:185-246 nest without bound; the product walks up to 15).  A sphere behind six TransformedGeometry levels, a
    cube at the bottom of four nested nodes (six levels above its faces), a rectangle under non-rigid matrices, a ConstantMedium
    five levels down, and one whose BOUNDARY is a cube behind five levels -- pure translations mixed in at every depth."""
    d = SceneDesc(name="deep-chains")
    g = HostRng(seed)
    ex, ey, ez = (1.0, 0.0, 0.0), (0.0, 1.0, 0.0), (0.0, 0.0, 1.0)

    def tr(t, rad=0.0, axis=ey):
        return mat4_multiplied(mat4_translation(t), mat4_rotation(rad, axis))

    def small(k):  # a small rigid or non-rigid step; every third one a pure translation
        t = (g.gen_range(-0.3, 0.3), g.gen_range(-0.2, 0.2), g.gen_range(-0.3, 0.3))
        if k % 3 == 2:
            return mat4_translation(t)
        m = tr(t, g.gen_range(-0.6, 0.6), (ex, ey, ez)[k % 3])
        if k % 4 == 1:  # non-rigid on purpose (quirk Q5)
            m = mat4_multiplied(m, [1.2, 0.0, 0.0, 0.0, 0.0, 0.85, 0.0, 0.0, 0.0, 0.0, 1.1, 0.0, 0.0, 0.0, 0.0, 1.0])
        return m

    white = d.lambertian_rgb((0.73, 0.73, 0.73))
    red = d.lambertian_rgb((0.65, 0.05, 0.05))
    glass = d.mat("dielectric", 1.5)
    steel = d.mat("metal", d.tex_solid((0.8, 0.8, 0.9)), 0.1)
    smoke = d.mat("isotropic", d.tex_solid((0.9, 0.9, 0.9)))
    dark = d.mat("isotropic", d.tex_solid((0.1, 0.1, 0.1)))
    # a sphere behind six TransformedGeometry levels (+ the sprite: 7)
    geo = d.geom("sphere", 0.7)
    for k in range(6):
        geo = d.geom("transformed", geo, small(k))
    s1 = d.sprite(geo, glass, tr((-2.5, 1.2, 0.5), 0.3, ey))
    # a cube at the bottom of four nested nodes (sprite, 4 x (node sprite), face: 6 levels above each rectangle)
    node = d.geom("bvh", [d.sprite(d.geom("cube", 0.9, 1.1, 0.8), None, small(1))])
    for k in range(3):
        node = d.geom("bvh", [d.sprite(node, None, small(k + 2))])
    s2 = d.sprite(node, steel, tr((0.2, 1.0, 0.8), -0.4, ey))
    # a rectangle under five levels, two of them non-rigid
    geo = d.geom("rectangle", 1.6, 1.0)
    for k in range(4):
        geo = d.geom("transformed", geo, small(k + 1))
    s3 = d.sprite(geo, red, tr((2.4, 1.3, 0.6), 0.5, ex))
    # a ConstantMedium five levels down (sphere boundary), and one whose boundary is a cube behind five levels
    geo = d.geom("medium", d.geom("sphere", 0.8), 1.4)
    for k in range(4):
        geo = d.geom("transformed", geo, small(k))
    s4 = d.sprite(geo, smoke, tr((-1.0, 2.8, 1.5), 0.2, ez))
    geo = d.geom("cube", 1.0, 1.4, 1.0)
    for k in range(5):
        geo = d.geom("transformed", geo, small(k + 2))
    s5 = d.sprite(d.geom("medium", geo, 1.1), dark, tr((1.6, 2.9, 1.2), -0.3, ey))
    floor = d.sprite(d.geom("rectangle", 30.0, 30.0), white, tr((0.0, 0.0, 0.0), radians(-90.0), ex))
    lamp = d.sprite(d.geom("rectangle", 5.0, 5.0), d.mat("diffuse_light", d.tex_solid((5.0, 5.0, 5.0))), tr((0.0, 7.0, 1.0), radians(90.0), ex))
    sky = d.sprite(d.geom("sphere", 60.0), d.mat("diffuse_light", d.tex_solid((0.3, 0.35, 0.45))), None)
    d.world = [s1, s2, s3, s4, s5, floor, lamp, sky]
    d.camera = ((0.0, 3.0, -9.0), (0.0, 1.6, 1.0), (0.0, 1.0, 0.0), radians(42.0), float(aspect), 10.0, 0.0)
    return d


def nested_media(aspect: float = 1.0, seed: int = 1) -> SceneDesc:
    """Not one of the reference's examples: ConstantMedium<T: Hit> with a ConstantMedium inside T (src/volume.rs:18-44 is generic
    over the boundary).  The outer medium's two boundary.hit calls each evaluate the inner one with a draw of its own, so the
    "boundary" it sees is a random surface inside the inner boundary.  (1) a medium whose boundary IS a medium over a sphere, on a
    sprite of the world's own list; (2) a medium over a node that holds a solid sphere and a medium over a rotated cube; (3) three
    levels behind a TransformedGeometry; (4) the same node as (2) instanced a second time (keys per instance)."""
    d = SceneDesc(name="nested-media")
    g = HostRng(seed)
    ex, ey, ez = (1.0, 0.0, 0.0), (0.0, 1.0, 0.0), (0.0, 0.0, 1.0)

    def tr(t, rad=0.0, axis=ey):
        return mat4_multiplied(mat4_translation(t), mat4_rotation(rad, axis))

    white = d.lambertian_rgb((0.73, 0.73, 0.73))
    light_smoke = d.mat("isotropic", d.tex_solid((0.9, 0.85, 0.8)))
    dark_smoke = d.mat("isotropic", d.tex_solid((0.15, 0.2, 0.3)))
    d.textures.append(("checker", d.tex_solid((0.1, 0.1, 0.4)), d.tex_solid((0.9, 0.9, 0.9))))
    checker_smoke = d.mat("isotropic", len(d.textures) - 1)
    # (1) medium over a medium over a sphere, directly on a world sprite
    m1 = d.sprite(d.geom("medium", d.geom("medium", d.geom("sphere", 1.0), 2.0), 0.8), light_smoke, mat4_translation((-2.6, 1.2, 0.4)))
    # (2) a node of {solid sphere, medium over a rotated cube} as the boundary
    kids = [d.sprite(d.geom("sphere", 0.7), None, mat4_translation((-0.5, 0.0, 0.0))),
            d.sprite(d.geom("medium", d.geom("cube", 1.1, 1.3, 0.9), 1.7), None, tr((0.6, 0.1, 0.1), g.gen_range(0.2, 0.6), ey))]
    node = d.geom("bvh", kids)
    m2 = d.sprite(d.geom("medium", node, 0.6), dark_smoke, tr((0.2, 1.3, 0.6), -0.4, ez))
    # (3) three levels behind a TransformedGeometry, textured
    deep = d.geom("medium", d.geom("medium", d.geom("medium", d.geom("sphere", 0.9), 3.0), 1.5), 0.7)
    m3 = d.sprite(d.geom("transformed", deep, tr((0.0, 0.2, 0.0), 0.5, ex)), checker_smoke, tr((2.7, 1.1, 0.8), 0.3, ey))
    # (4) the boundary node of (2) under a second medium geometry and another matrix
    m4 = d.sprite(d.geom("medium", node, 1.1), light_smoke, tr((0.3, 3.6, 1.8), 0.9, ex))
    floor = d.sprite(d.geom("rectangle", 30.0, 30.0), white, tr((0.0, 0.0, 0.0), radians(-90.0), ex))
    lamp = d.sprite(d.geom("rectangle", 5.0, 5.0), d.mat("diffuse_light", d.tex_solid((5.0, 5.0, 5.0))), tr((0.0, 7.5, 1.0), radians(90.0), ex))
    sky = d.sprite(d.geom("sphere", 60.0), d.mat("diffuse_light", d.tex_solid((0.3, 0.35, 0.45))), None)
    d.world = [m1, m2, m3, m4, floor, lamp, sky]
    d.camera = ((0.0, 2.6, -9.0), (0.0, 1.7, 1.0), (0.0, 1.0, 0.0), radians(42.0), float(aspect), 10.0, 0.0)
    return d
