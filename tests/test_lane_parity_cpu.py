"""The device lane program (ray-tracer_amd/csrc/rt_lane.h) compiled for the HOST and run
lane by lane over the committed flat scene, against the oracle.  This checks the
flattener, the SAH BVH with binary32 culling, the hoisting and every per-lane formula on
machines without a GPU; it claims nothing about the GPU itself (see test_gpu_parity.py).
Both sides use the host libm here, so agreement must be exact, pixel for pixel."""
import math

import numpy as np
import pytest


def both(scenes, oracle, lane_emul, desc, W, H, spp, depth, seed=1, **okw):
    sc, cam = scenes.build_product(desc, device=-1)
    img, cnt, high = lane_emul.render(sc, cam, W, H, spp, depth, seed)
    ref, ocnt = oracle.build_oracle(desc, **okw).render(W, H, spp, depth, seed, iterative=True, nthreads=8, counters=True)
    return img, ref, cnt, ocnt, high, sc


@pytest.mark.parametrize("W,H,spp,depth", [(60, 40, 4, 50), (33, 17, 3, 100), (8, 8, 1, 1), (5, 3, 2, 7)])
def test_book_one_exact(scenes, oracle, lane_emul, W, H, spp, depth):
    img, ref, cnt, ocnt, high, sc = both(scenes, oracle, lane_emul, scenes.book_one(1, W / H), W, H, spp, depth)
    assert np.array_equal(img, ref)
    assert cnt["segments"] == ocnt["segments"] and cnt["rng_draws"] == ocnt["rng_draws"]
    assert high <= sc.info()["max_depth"] + 1 <= 24
    # pruning + SAH: far fewer box tests than the reference's unpruned walk
    assert cnt["nodes_visited"] < ocnt["aabb_tests"]


def test_book_one_other_seeds_exact(scenes, oracle, lane_emul):
    for scene_seed, seed in [(2, 5), (7, 123456789)]:
        img, ref, *_ = both(scenes, oracle, lane_emul, scenes.book_one(scene_seed, 1.5), 48, 32, 2, 50, seed)
        assert np.array_equal(img, ref)


def test_cornell_exact(scenes, oracle, lane_emul):
    """Rectangles, rotated sprites, instanced cubes (two transform levels), area light."""
    img, ref, cnt, ocnt, *_ = both(scenes, oracle, lane_emul, scenes.cornell(1.0), 40, 40, 8, 100)
    assert np.array_equal(img, ref)
    assert cnt["segments"] == ocnt["segments"]
    assert img.max() > 1.0 and img.min() == 0.0  # light visible, black outside the box front


def test_cover_exact(scenes, oracle, lane_emul):
    """ConstantMedium x2 (keyed draws), Isotropic, image texture, 400 instanced cubes, 1000 spheres."""
    img, ref, cnt, ocnt, *_ = both(scenes, oracle, lane_emul, scenes.cover(1, 1.0), 40, 40, 4, 100)
    assert np.array_equal(img, ref)
    assert cnt["segments"] == ocnt["segments"]
    # keyed medium draws are only evaluated for media the pruned walk still reaches; a skipped
    # one could not have produced the nearest hit, so the count may drop but never a result
    assert cnt["rng_draws"] <= ocnt["rng_draws"]


def test_cube_groups_and_the_binary16_tree_exact(scenes, oracle, lane_emul, lane_devmath, monkeypatch):
    """Round 5: the cover's 400 floor boxes are CUBE GROUPS -- one leaf for the six faces of a cube, whose boxes the
    leaf step derives from twelve planes (rtl::trav_leaf_step) -- and its tree of 1406 nodes also exists with binary16 planes
    (RtNodeH), the form the device keeps in LDS.  Culling never reaches a result: with groups or without (RT_NO_CUBE_GROUPS=1),
    through the binary32 tree or the binary16 one, in both arithmetic forms, the image is the oracle's, bit for bit."""
    desc = scenes.cover(1, 1.0)
    W = H = 48
    sc, cam = scenes.build_product(desc, device=-1)
    assert sum(1 for p in range(sc.info()["n_prims"]) if sc.prim_group(p) == 6) == 400
    ref = oracle.build_oracle(desc).render(W, H, 4, 100, seed=3, iterative=True, nthreads=8)
    bad, n = lane_emul.half_tree_check(sc)
    assert bad == 0 and n == sc.info()["n_nodes"] == 1406
    imgs = {}
    for emul in (lane_emul, lane_devmath):
        for half in (False, True):
            emul.half_nodes(half)
            try:
                img, cnt, _ = emul.render(sc, cam, W, H, 4, 100, seed=3)
            finally:
                emul.half_nodes(False)
            assert np.array_equal(img, ref), (emul.name, half)
            imgs[(emul.name, half)] = cnt
    # the binary16 boxes are (slightly) larger: never fewer node steps or primitive tests than through the binary32 tree
    a, b = imgs[(lane_emul.name, False)], imgs[(lane_emul.name, True)]
    assert b["nodes_visited"] >= a["nodes_visited"] and b["prims_tested"] >= a["prims_tested"]
    assert b["nodes_visited"] <= 1.1 * a["nodes_visited"]  # ... and not many more: a step of 2^-10 of the coordinate
    # without groups: 2400 face leaves, the tree of round 4 -- the same image, and the groups test no more primitives than that tree does
    monkeypatch.setenv("RT_NO_CUBE_GROUPS", "1")
    sc0, cam0 = scenes.build_product(desc, device=-1)
    assert sc0.info()["n_nodes"] == 3406 and all(sc0.prim_group(p) == 1 for p in range(0, 3408, 97))
    img0, cnt0, _ = lane_emul.render(sc0, cam0, W, H, 4, 100, seed=3)
    assert np.array_equal(img0, ref)
    assert a["prims_tested"] <= 1.15 * cnt0["prims_tested"], (a, cnt0)
    monkeypatch.delenv("RT_NO_CUBE_GROUPS")
    # rotated cubes are not grouped (their faces' boxes are not the sides of one box); cubes far from the origin and tiny ones are
    d = scenes.SceneDesc()
    m = d.lambertian_rgb((0.7, 0.6, 0.5))
    for i, (size, at) in enumerate([(2.0, (0.0, 1.0, 0.0)), (1e-3, (3.0, 5e-4, 0.0)), (3.0, (4000.0, 1.5, 30.0))]):
        d.sprite(d.geom("cube", size, size, size), m, scenes.mat4_translation(at))
    d.sprite(d.geom("cube", 2.0, 2.0, 2.0), m, scenes.mat4_multiplied(scenes.mat4_translation((-3.0, 1.0, 0.0)), scenes.mat4_rotation(0.3, (0.0, 1.0, 0.0))))
    for i in range(24):  # (enough other leaves to keep the scene off the box list)
        d.sprite(d.geom("sphere", 0.3), m, scenes.mat4_translation((i - 12.0, 0.3, 4.0)))
    d.sprite(d.geom("sphere", 2000.0), d.mat("diffuse_light", d.tex_solid((0.6, 0.7, 1.0))), None)
    d.camera = ((0.0, 3.0, -12.0), (0.0, 1.0, 0.0), (0.0, 1.0, 0.0), 0.7, 1.0, 10.0, 0.0)
    sc, cam = scenes.build_product(d, device=-1)
    groups = [p for p in range(sc.info()["n_prims"]) if sc.prim_group(p) == 6]
    assert len(groups) == 3, groups
    for half in (False, True):
        lane_emul.half_nodes(half)
        try:
            img, _, _ = lane_emul.render(sc, cam, 40, 40, 4, 50, seed=2)
        finally:
            lane_emul.half_nodes(False)
        assert np.array_equal(img, oracle.build_oracle(d).render(40, 40, 4, 50, seed=2, iterative=True, nthreads=8))


def test_rounding_to_the_binary16_grid_is_outward():
    """rt::half_toward: the nearest binary16 value on the asked side, for values across the grid's whole range (subnormals included)"""
    import lane_emul_binding as le
    rng = np.random.default_rng(5)
    xs = np.concatenate([rng.uniform(-60000, 60000, 2000), rng.uniform(-1, 1, 2000) * 10.0 ** rng.uniform(-9, 0, 2000),
                         [0.0, -0.0, 1.0, -1.0, 2.0 ** -14, 2.0 ** -24, 2.0 ** -25, 65504.0 * 0.9, 1e-30, -1e-30, 1024.0, 1023.999]]).astype(np.float32)
    for x in xs:
        lo, hi = le.half_toward(x, False), le.half_toward(x, True)
        assert lo <= float(x) <= hi
        assert np.float16(lo) == np.float32(lo) and np.float16(hi) == np.float32(hi)  # both are binary16 values
        if float(np.float16(x)) == float(x):
            assert lo == hi == float(x)   # on the grid: itself
        else:
            assert hi == float(np.nextafter(np.float16(lo), np.float16(np.inf)))  # neighbours on the grid


def test_textured_and_rotated_primitives_exact(scenes, oracle, lane_emul):
    d = scenes.SceneDesc()
    black, white = d.tex_solid((0.05, 0.05, 0.05)), d.tex_solid((0.9, 0.9, 0.9))
    d.textures.append(("checker", black, white))
    checker = len(d.textures) - 1
    img8 = (np.arange(16 * 8 * 3) % 251).astype(np.uint8).reshape(8, 16, 3)
    d.textures.append(("image", img8))
    image = len(d.textures) - 1
    rot = scenes.mat4_multiplied(scenes.mat4_translation((0.0, 0.0, 6.0)), scenes.mat4_rotation(0.7, (0.0, 1.0, 0.0)))
    d.sprite(d.geom("sphere", 1.5), d.mat("lambertian", checker), rot)                       # general-matrix sphere
    d.sprite(d.geom("sphere", 1.0), d.mat("lambertian", image), scenes.mat4_translation((3.0, 0.0, 6.0)))
    d.sprite(d.geom("rectangle", 20.0, 20.0), d.mat("metal", checker, 0.3),
             scenes.mat4_multiplied(scenes.mat4_translation((0.0, -2.0, 6.0)), scenes.mat4_rotation(scenes.radians(-90.0), (1.0, 0.0, 0.0))))
    d.sprite(d.geom("cube", 1.0, 2.0, 1.0), d.mat("dielectric", 1.5),
             scenes.mat4_multiplied(scenes.mat4_translation((-3.0, 0.0, 5.0)), scenes.mat4_rotation(0.4, (0.0, 1.0, 0.0))))
    d.sprite(d.geom("medium", d.geom("sphere", 1.0), 0.8), d.mat("isotropic", d.tex_solid((0.2, 0.4, 0.9))),
             scenes.mat4_multiplied(scenes.mat4_translation((0.0, 2.5, 6.0)), scenes.mat4_rotation(1.0, (0.0, 0.0, 1.0))))
    d.sprite(d.geom("sphere", 60.0), d.mat("diffuse_light", d.tex_solid((1.0, 1.0, 1.0))), None)
    d.sprite(d.geom("sphere", 0.5), None, scenes.mat4_translation((1.0, 1.5, 4.0)))         # material None -> black
    d.camera = ((0.0, 0.5, -4.0), (0.0, 0.0, 6.0), (0.0, 1.0, 0.0), 0.9, 1.25, 10.0, 0.02)
    img, ref, *_ = both(scenes, oracle, lane_emul, d, 50, 40, 6, 60)
    assert np.array_equal(img, ref)


def test_oracle_tree_and_world_form_do_not_matter(scenes, oracle, lane_emul):
    d = scenes.cornell(1.0)
    sc, cam = scenes.build_product(d, device=-1)
    img, *_ = lane_emul.render(sc, cam, 24, 24, 4, 50, 3)
    for kw in ({"bvh_seed": 1}, {"bvh_seed": 1234}, {"world": "list"}):
        assert np.array_equal(img, oracle.build_oracle(d, **kw).render(24, 24, 4, 50, 3, iterative=True))


def test_per_sample_radiance_exact(scenes, oracle, lane_emul):
    d = scenes.book_one(1, 1.5)
    sc, cam = scenes.build_product(d, device=-1)
    W, H, spp, depth = 60, 40, 32, 50
    for (x, y) in [(30, 14), (10, 30), (45, 8)]:
        _, _, _, samples = lane_emul.render(sc, cam, W, H, spp, depth, 9, region=(x, y, x + 1, y + 1), sample_pixel=(x, y))
        ref = oracle.build_oracle(d).pixel_samples(W, H, spp, depth, 9, x, y, iterative=True)
        assert np.array_equal(samples, ref)


def test_max_depth_zero_is_black(scenes, lane_emul):
    sc, cam = scenes.build_product(scenes.book_one(1, 1.5), device=-1)
    img, *_ = lane_emul.render(sc, cam, 16, 8, 2, 0)
    assert np.array_equal(img, np.zeros((8, 16, 3)))


def test_bounded_ball_sampler_is_the_same_loop(lane_emul):
    """random_in_unit_sphere_bounded split over several calls (1, 2 or 3 iterations each) accepts the same point
    and leaves the same stream as the reference's single loop (src/util.rs:6-15); some streams need several calls."""
    most = 0
    for stream in range(400):
        for k in (1, 2, 3):
            calls, p, q, a, b = lane_emul.ball_check(7, stream, k)
            assert calls >= 1 and np.array_equal(p, q) and a == b, (stream, k)
            assert a[2] % 3 == 0 and (a[2] // 3 + k - 1) // k == calls  # draws = 3 per iteration; calls = ceil(iter / k)
            most = max(most, calls)
    assert most >= 4
    calls, p, q, a, b = lane_emul.ball_check(7, 5, 0)  # 0 = unbounded
    assert calls == 1 and np.array_equal(p, q) and a == b


def test_instanced_nodes_and_general_media_exact(scenes, oracle, lane_emul):
    """A BoundingVolumeHierarchyNode as a sprite's geometry (instancing, src/sprite.rs:87-93), instances of instances (four
    transform levels), ConstantMedium over a cube / over a node of spheres / behind a TransformedGeometry
    (src/volume.rs:40-100), media instanced twice (keyed per instance): the expanded flat scene against the oracle's
    recursive trait-object walk, bit for bit."""
    d = scenes.instanced(1.25)
    sc, cam = scenes.build_product(d, device=-1)
    info = sc.info()
    # 3 clusters x (5 + 1 + 6) + 2 pairs x 2 x 12 leaves, 4 media, floor, lamp, sky; boundary prims: 6 + 2 + 2 + 1
    assert info["n_prims"] == 3 * 12 + 2 * 24 + 4 + 3 and info["n_child_prims"] == 6 + 2 + 2 + 1
    assert info["feature_mask"] & rt_feat(scenes, "RT_FEAT_MEDIUM_GENERAL")
    img, cnt, high = lane_emul.render(sc, cam, 50, 40, 6, 60, 3)
    ref, ocnt = oracle.build_oracle(d).render(50, 40, 6, 60, 3, iterative=True, nthreads=8, counters=True)
    assert np.array_equal(img, ref)
    assert cnt["segments"] == ocnt["segments"]
    assert img.std() > 0.05
    # the tree and the world's form do not matter (F7), nested nodes included
    for kw in ({"bvh_seed": 99}, {"world": "list"}):
        assert np.array_equal(img, oracle.build_oracle(d, **kw).render(50, 40, 6, 60, 3, iterative=True, nthreads=8))


def open_boundary_media(scenes):
    """ConstantMedium<Rectangle>, ConstantMedium over a rotated sphere pair and over a scaled sphere: boundaries that do not
    enclose a volume in the sense of src/volume.rs:49 (a rectangle's normal is +z from both sides; general matrices move
    normals by M, quirk Q5).  For a ray that meets such a boundary from its back, volume.rs:80-98 scatters anywhere between
    the ray's ORIGIN and the boundary -- in front of every box the boundary has.  Solid objects in between must not cull it."""
    d = scenes.SceneDesc(name="open-boundary media")
    grey, red = d.lambertian_rgb((0.7, 0.7, 0.7)), d.lambertian_rgb((0.7, 0.1, 0.1))
    fog = [d.mat("isotropic", d.tex_solid(c)) for c in ((0.9, 0.9, 0.9), (0.2, 0.3, 0.9), (0.2, 0.8, 0.3))]
    ey = (0.0, 1.0, 0.0)
    # a rectangle whose +z looks away from the camera: every camera ray through it starts "inside"
    d.sprite(d.geom("medium", d.geom("rectangle", 6.0, 4.0), 0.05), fog[0], scenes.mat4_translation((-3.0, 0.0, 14.0)))
    # the same seen from its front (never hit: the restarted ray misses, volume.rs:75-77)
    d.sprite(d.geom("medium", d.geom("rectangle", 3.0, 3.0), 0.5), fog[1],
             scenes.mat4_multiplied(scenes.mat4_translation((4.0, 2.0, 10.0)), scenes.mat4_rotation(math.pi, ey)))
    # two spheres under a rotation inside the boundary, and a sphere squeezed by a non-rigid matrix
    pair = d.geom("bvh", [d.sprite(d.geom("sphere", 1.0), None, scenes.mat4_multiplied(scenes.mat4_translation((dx, 0.0, 0.0)), scenes.mat4_rotation(0.4, ey)))
                          for dx in (-0.7, 0.7)])
    d.sprite(d.geom("medium", pair, 0.8), fog[2], scenes.mat4_translation((3.0, -1.5, 9.0)))
    squeeze = [1.0, 0.0, 0.0, 0.0, 0.9, 0.25, 0.0, 0.0, 0.0, 0.0, 3.0, 0.0, 0.0, 0.0, 0.0, 1.0]
    d.sprite(d.geom("medium", d.geom("transformed", d.geom("sphere", 1.0), squeeze), 0.6), fog[1], scenes.mat4_translation((-4.0, 2.5, 11.0)))
    # solid things between the camera and the media
    for k, (x, y, z) in enumerate([(-3.5, 0.5, 6.0), (-2.0, -1.0, 8.0), (0.5, 0.3, 5.0), (3.0, -1.0, 6.5), (-4.5, 2.0, 7.0)]):
        d.sprite(d.geom("sphere", 0.6), red if k % 2 else grey, scenes.mat4_translation((x, y, z)))
    d.sprite(d.geom("sphere", 60.0), d.mat("diffuse_light", d.tex_solid((0.8, 0.8, 0.8))), None)
    d.camera = ((0.0, 0.5, -2.0), (0.0, 0.0, 10.0), (0.0, 1.0, 0.0), 0.9, 4 / 3, 10.0, 0.0)
    return d


def test_media_over_open_boundaries_exact(scenes, oracle, lane_emul):
    d = open_boundary_media(scenes)
    sc, cam = scenes.build_product(d, device=-1)
    info = sc.info()
    assert info["n_hoisted"] >= 4  # the four media (+ the enclosing light): tested for every segment, no box culls them
    img, cnt, high = lane_emul.render(sc, cam, 64, 48, 8, 40, 5)
    for kw in ({}, {"bvh_seed": 42}, {"world": "list"}):
        ref, ocnt = oracle.build_oracle(d, **kw).render(64, 48, 8, 40, 5, iterative=True, nthreads=8, counters=True)
        assert np.array_equal(img, ref)
    assert cnt["segments"] == ocnt["segments"] > 64 * 48 * 8 * 1.1  # the fog in front of the rectangle scatters


def test_list_walk_equals_tree_walk(scenes, oracle, lane_emul, monkeypatch):
    """General scenes of up to RT_LIST_MAX leaves are walked as a box list (rtl::trav_list_step) -- the Cornell box is one.
    Same boxes, same binary64 tests, same tie rule: the image equals the tree walk's and the oracle's bit for bit."""
    d = scenes.cornell(1.0)
    sc, cam = scenes.build_product(d, device=-1)
    info = sc.info()
    assert info["n_list"] == 18 == info["n_prims"] and info["n_nodes"] == 17 and info["node_bytes"] == 36
    img, cnt, high = lane_emul.render(sc, cam, 48, 48, 4, 60, 9)
    assert high <= info["n_list"] - 1
    assert cnt["nodes_visited"] == cnt["segments"] * 18  # every segment looks at every box once
    monkeypatch.setenv("RT_NO_LIST", "1")
    sc2, cam2 = scenes.build_product(d, device=-1)
    assert sc2.info()["n_list"] == 0 and sc2.info()["node_bytes"] == 64
    img2, cnt2, _ = lane_emul.render(sc2, cam2, 48, 48, 4, 60, 9)
    assert np.array_equal(img, img2) and cnt["segments"] == cnt2["segments"]
    assert np.array_equal(img, oracle.build_oracle(d).render(48, 48, 4, 60, 9, iterative=True, nthreads=8))
    monkeypatch.delenv("RT_NO_LIST")
    # spheres-only scenes keep the tree however small; 25 general leaves are one too many for the list
    few = scenes.SceneDesc()
    for k in range(5):
        few.sprite(few.geom("sphere", 0.5), few.lambertian_rgb((0.5, 0.5, 0.5)), scenes.mat4_translation((k, 0.0, 5.0)))
    few.camera = d.camera
    assert scenes.build_product(few, device=-1)[0].info()["n_list"] == 0
    for n, want in ((24, 24), (25, 0)):
        many = scenes.SceneDesc()
        for k in range(n):
            many.sprite(many.geom("rectangle", 1.0, 1.0), many.lambertian_rgb((0.5, 0.5, 0.5)),
                        scenes.mat4_multiplied(scenes.mat4_translation((k % 5, k // 5, 5.0)), scenes.mat4_rotation(0.3, (0.0, 1.0, 0.0))))
        many.camera = d.camera
        assert scenes.build_product(many, device=-1)[0].info()["n_list"] == want


def test_medium_traversal_form_equals_record_form(lane_emul):
    """medium_hit<false> takes the sign of normal . direction from the plain sum p . d when that sum is clearly non-zero
    and falls back to the reference's normalized(p / r) . d otherwise: hit / miss and t must equal the record form's on
    ordinary rays, on rays that graze the boundary (where the fallback runs) and at extreme scales."""
    rng = np.random.default_rng(3)
    n_hit = n_graze = 0
    for i in range(20000):
        r = float(10.0 ** rng.uniform(-3, 3)) if i % 3 else float(rng.uniform(0.5, 2.0))
        d = rng.normal(size=3)
        d /= np.linalg.norm(d)
        if i % 4 == 0:  # grazing: a ray tangent to the sphere, nudged in or out by ulps .. 1e-9
            q = np.cross(d, rng.normal(size=3))
            q /= np.linalg.norm(q)
            oc = q * r * (1.0 + float(rng.choice([-1, 1])) * float(10.0 ** rng.uniform(-16, -9))) - d * r * float(rng.uniform(0.0, 3.0))
            n_graze += 1
        elif i % 4 == 1:  # origin inside
            oc = rng.normal(size=3)
            oc *= r * float(rng.uniform(0.0, 0.999)) / np.linalg.norm(oc)
        else:
            oc = rng.normal(size=3) * r * 2.0
        if i % 7 == 0:
            d = d * float(10.0 ** rng.uniform(-3, 3))  # un-normalised directions (quirk Q5: sprites do not renormalise)
        a, b = lane_emul.medium_forms(oc, d, r, float(10.0 ** rng.uniform(-3, 1)), base=int(rng.integers(1 << 62)), segment=i % 50, slot=i % 7)
        assert a == b, (i, oc, d, r, a, b)
        n_hit += a[0]
    assert n_hit > 1500 and n_graze == 5000
    # scales at which the shortcut declines (guards on radius and on |p . d|): still the same answers
    for r, scale in ((1e-120, 1e-120), (1e120, 1e120), (1.0, 1e-160), (1.0, 1e160)):
        for k in range(50):
            d = rng.normal(size=3) * (scale if r == 1.0 else 1.0)
            oc = rng.normal(size=3) * r * 0.5
            a, b = lane_emul.medium_forms(oc, d, r, 0.5 / max(r, 1e-300) if r != 1.0 else 0.5, base=k)
            assert a == b


def rt_feat(scenes, name):
    import sys
    return getattr(sys.modules["ray_tracer_amd"], name)


def test_chain_depth_limit_is_reported(rt, scenes):
    """five transform levels above a sphere commit (round 2: an error; now the deep-chain family); sixteen are outside
    RT_MAX_CHAIN_DEEP: an error at commit, never a wrong picture"""
    def nested(levels):
        s = rt.Scene()
        node = s.bvh([s.sprite(s.sphere(1.0), None, scenes.mat4_rotation(0.1, (0.0, 1.0, 0.0)))])
        for _ in range(levels - 2):
            node = s.bvh([s.sprite(node, None, scenes.mat4_rotation(0.1, (0.0, 1.0, 0.0)))])
        s.sprite(node, None, scenes.mat4_rotation(0.1, (0.0, 1.0, 0.0)))
        return s
    s = nested(5)
    s.commit(-1)
    assert s.info()["feature_mask"] & rt.FEAT_DEEP_CHAIN
    s = nested(16)
    with pytest.raises(rt.RtError) as e:
        s.commit(-1)
    assert e.value.code == -4 and "transform levels" in str(e.value)


def test_scene_above_the_16_bit_reference_limit_exact(scenes, oracle, lane_emul):
    """40002 sphere sprites (39999 nodes): 32-bit node references and two-word stack entries; lane program vs oracle"""
    rng = np.random.default_rng(1)
    d = scenes.SceneDesc()
    g = d.geom("sphere", 0.3)
    mats = [d.lambertian_rgb(rng.uniform(0.1, 0.9, 3)) for _ in range(4)] + [d.mat("metal", d.tex_solid((0.8, 0.8, 0.8)), 0.1),
                                                                            d.mat("dielectric", 1.5)]
    n = 200
    for i in range(n):
        for j in range(n):
            d.sprite(g, mats[int(rng.integers(len(mats)))],
                     scenes.mat4_translation((i - n / 2 + rng.uniform(0, 0.3), 0.3, j - n / 2 + rng.uniform(0, 0.3))))
    d.sprite(d.geom("sphere", 1000.0), d.lambertian_rgb((0.5, 0.5, 0.5)), scenes.mat4_translation((0.0, -1000.0, 0.0)))
    d.sprite(d.geom("sphere", 3000.0), d.mat("diffuse_light", d.tex_solid((0.6, 0.7, 1.0))), None)
    d.camera = ((20.0, 6.0, 8.0), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), 0.5, 1.5, 20.0, 0.02)
    sc, cam = scenes.build_product(d, device=-1)
    info = sc.info()
    assert info["feature_mask"] & rt_feat(scenes, "RT_FEAT_WIDE") and info["n_nodes"] == 39999 and info["n_hoisted"] == 2
    img, cnt, high = lane_emul.render(sc, cam, 30, 20, 2, 30, 3)
    assert np.array_equal(img, oracle.build_oracle(d).render(30, 20, 2, 30, 3, iterative=True, nthreads=8))
    assert high <= info["max_depth"] + 1 <= 24


def test_deep_transform_chains_exact(scenes, oracle, lane_emul, rt):
    """More than four transform levels above a primitive (src/sprite.rs:87-93 nests without bound; round 2 returned
    RT_ERR_UNSUPPORTED): up to 15 are walked by the kernel family for general media, the levels beyond the fourth in the
    reference's own full 4x4 form.  Spheres, cube faces, a rectangle, a medium and a medium's boundary at depths 5-7."""
    d = scenes.deep_chains(1.25, seed=1)
    img, ref, cnt, ocnt, high, sc = both(scenes, oracle, lane_emul, d, 60, 48, 6, 40)
    assert sc.info()["feature_mask"] & rt.FEAT_DEEP_CHAIN
    assert np.array_equal(img, ref)
    assert cnt["segments"] == ocnt["segments"]
    assert img.std() > 0.05  # the objects are in the picture
    img2, ref2, *_ = both(scenes, oracle, lane_emul, scenes.deep_chains(1.0, seed=5), 40, 40, 4, 40, seed=9)
    assert np.array_equal(img2, ref2)


def test_sixteen_levels_are_still_an_error(scenes, rt):
    d = scenes.SceneDesc()
    geo = d.geom("sphere", 1.0)
    for _ in range(15):
        geo = d.geom("transformed", geo, scenes.mat4_translation((0.1, 0.0, 0.0)))
    d.sprite(geo, d.lambertian_rgb((0.5, 0.5, 0.5)), None)  # 15 + the sprite's own level = 16
    d.camera = ((0.0, 0.0, -5.0), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), 0.8, 1.0, 10.0, 0.0)
    with pytest.raises(rt.RtError) as e:
        scenes.build_product(d, device=-1)
    assert e.value.code == rt.ERR_UNSUPPORTED and "15" in str(e.value)


def test_media_inside_the_boundary_of_media_exact(scenes, oracle, lane_emul, rt):
    """ConstantMedium<T: Hit> with a ConstantMedium inside T (src/volume.rs:18-44; round 2 returned RT_ERR_UNSUPPORTED): a
    medium over a medium over a sphere, a node of {sphere, medium over a cube} as a boundary (twice, under two media), three
    levels behind a TransformedGeometry with a textured Isotropic.  Every inner evaluation draws a number keyed by the outer
    evaluation's key and pass (include/rt_rng.h): flattened boundary prims against the oracle's recursive walk, bit for bit."""
    d = scenes.nested_media(1.25, seed=1)
    img, ref, cnt, ocnt, high, sc = both(scenes, oracle, lane_emul, d, 60, 48, 8, 30)
    assert sc.info()["feature_mask"] & rt.RT_FEAT_MEDIUM_NESTED
    assert np.array_equal(img, ref)
    assert cnt["segments"] == ocnt["segments"]
    assert img.std() > 0.05
    for kw in ({"bvh_seed": 31}, {"world": "list"}):  # the oracle's tree / world form does not matter (keyed draws)
        assert np.array_equal(img, oracle.build_oracle(d, **kw).render(60, 48, 8, 30, 1, iterative=True, nthreads=8))
    img2, ref2, *_ = both(scenes, oracle, lane_emul, scenes.nested_media(1.0, seed=4), 40, 40, 4, 30, seed=11)
    assert np.array_equal(img2, ref2)


def test_directions_of_any_length_keep_the_culling_conservative(scenes, oracle, lane_emul):
    """A direction is never renormalised after a matrix (quirk Q5): a path that keeps scattering under a non-rigid matrix can
    carry |d| = 1e-38 (it happened, test below) -- an ordinary binary64 ray for the reference's binary64 boxes, but 1 / d and
    o / d leave binary32, and the culling boxes used to turn NaN / infinite and lose whole subtrees.  rt_lane.h trav_ray_constants
    makes such a reciprocal infinite, which drops the axis.  Here: the nearest hit of random rays, each at lengths 1e-150 ... 1e150
    and with single components scaled down to 1e-300 (a ray between two parallel mirrors loses one component bounce by bounce), through
    the lane program (hoisted prims, binary32 culling, binary64 tests, keyed medium draws) against the oracle's World::hit
    (orc_kat_world_hit), same bits of t, in the spheres-only, the list-walk and the general-media family.  (Below |d| = 1e-154
    d.d is denormal or zero and the reference's own roots are noise: not covered.)"""
    rng = np.random.default_rng(3)
    cases = (("book-one", scenes.book_one(1, 1.5), np.array([0.0, 1.0, 0.0]), 6.0), ("cornell", scenes.cornell(1.0), np.array([277.5, 277.5, 277.5]), 250.0),
             ("cover", scenes.cover(1, 1.0), np.array([200.0, 250.0, 200.0]), 300.0))
    for name, desc, centre, span in cases:
        sc, cam = scenes.build_product(desc, device=-1)
        orc = oracle.build_oracle(desc)
        hits = 0
        for i in range(60):
            o = np.ascontiguousarray(centre + rng.uniform(-span, span, 3))
            d = rng.standard_normal(3)
            variants = [d * 10.0 ** e for e in (0, -12, -30, -37, -38, -39, -45, -60, -100, -150, 12, 30, 38, 60, 150)]
            # ONE component (or two) sinking towards zero while |d| stays put: the ray between two parallel mirrors
            for e in (-17, -19, -30, -38, -42, -46, -100, -300):
                for axes in ((0,), (1,), (2,), (0, 1), (1, 2)):
                    v = d.copy()
                    v[list(axes)] *= 10.0 ** e
                    variants.append(v)
            for dd in variants:
                e = dd
                dd = np.ascontiguousarray(dd)
                out = np.zeros(10)
                hit = oracle.LIB.orc_kat_world_hit(orc.h, oracle.dp(o), oracle.dp(dd), 7, i, oracle.dp(out))
                got = lane_emul.world_hit(sc, cam, o, dd, 7, i)
                hits += hit
                assert (got is None) == (not hit), (name, i, e)
                if hit:
                    assert np.float64(got[0]).view(np.uint64) == out[0:1].view(np.uint64)[0], (name, i, e, got, out[0])
        assert hits > 500, name


def test_direction_shrinking_inside_a_scaled_medium_sweep_scene_78971(scenes, oracle, lane_emul, lane_devmath):
    """The scene of the 60 000-scene MI355X sweep that found it: a path scatters 35 times inside a medium under a non-rigid
    matrix, its direction shrinking 22-fold per bounce down to 1e-38 -- where the binary32 boxes used to lose the medium."""
    from test_random_scenes import random_scene_r3
    seed, W, H, spp = 78971, 60, 53, 2
    desc = random_scene_r3(scenes, seed)
    sc, cam = scenes.build_product(desc, device=-1)
    ref = oracle.build_oracle(desc, bvh_seed=seed).render(W, H, spp, 40, seed=seed, iterative=True, nthreads=8)
    for harness in (lane_emul, lane_devmath):
        img, *_ = harness.render(sc, cam, W, H, spp, 40, seed)
        assert np.array_equal(img, ref)
    assert ref[18, 23].sum() > 0.0  # the pixel the GPU had wrong (its first sample ended black in the oracle, lit on the device)


def test_component_sinking_between_parallel_mirrors_sweep_scene_115102(scenes, oracle, lane_emul):
    """The scene of the 50 000-scene depth-100 sweep: a ray caught between two parallel mirrors keeps |d| while one component
    sinks 3-fold per bounce, to 1e-38 at the 86th -- there o_y / d_y overflowed binary32, the mirror's box was culled and the path
    escaped to the light (oracle: black after 100 bounces)."""
    from test_random_scenes import random_scene
    seed, W, H, spp = 115102, 78, 53, 2
    desc = random_scene(scenes, seed)
    sc, cam = scenes.build_product(desc, device=-1)
    ref = oracle.build_oracle(desc, bvh_seed=seed).render(W, H, spp, 100, seed=seed, iterative=True, nthreads=8)
    img, *_ = lane_emul.render(sc, cam, W, H, spp, 100, seed)
    assert np.array_equal(img, ref)
    assert ref[0, 37].sum() == 0.0  # the pixel the GPU had lit
