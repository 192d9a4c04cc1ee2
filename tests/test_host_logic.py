"""CPU-side tests of the product's host logic: the C ABI surface, error behaviour,
Mat4 / camera / tone-map restatements, the flattener and the BVH.  No GPU needed."""
import ctypes as C
import math
import re
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent


def test_abi_exports_every_declared_symbol(rt):
    hdr = (ROOT / "include" / "rt_mi355x.h").read_text()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(rt_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 35
    lib = C.CDLL(str(rt.LIB_PATH))
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/rt_mi355x.h but not exported"
    assert declared == set(rt.ABI), "python binding table and header disagree"


def test_the_library_exports_the_header_and_nothing_else(rt):
    """`nm -D` of both libraries == the entry points include/rt_mi355x.h declares (csrc/Makefile's export map): the rt:: internals and
    the rt_launch_* / rt_pick_* / rt_kernel_* glue between the objects are local symbols."""
    import subprocess
    hdr = re.sub(r"/\*.*?\*/", "", (ROOT / "include" / "rt_mi355x.h").read_text(), flags=re.S)
    declared = set(re.findall(r"\b(rt_[a-z0-9_]+)\s*\(", hdr))
    for lib in (rt.LIB_PATH, rt.LIB_PATH.with_name("librt_mi355x_testhooks.so")):
        out = subprocess.run(["nm", "-D", "--defined-only", str(lib)], capture_output=True, text=True, check=True).stdout
        exported = {ln.split()[-1].split("@")[0] for ln in out.splitlines() if ln.strip()}
        assert exported == declared, (lib.name, sorted(exported - declared), sorted(declared - exported))


def _c_type(decl: str) -> str:
    """canonical spelling of a C parameter / return / field type (the declarator's name removed):
    `const double M[16]` -> `*const f64`, `rt_scene *const *` -> `*const *mut rt_scene`, `unsigned flags` -> `u32` ..."""
    decl = " ".join(decl.replace("*", " * ").split())
    array = re.search(r"\[\s*\d*\s*\]$", decl) is not None
    decl = re.sub(r"\s*\[\s*\d*\s*\]$", "", decl)
    toks = decl.split()
    base_words = {"const", "unsigned", "signed", "int", "char", "double", "float", "void", "size_t", "uint8_t", "uint32_t", "uint64_t", "int32_t",
                  "int64_t", "long", "short", "struct"}
    # drop the identifier: the last token when it is neither a type word, a `*`, nor a typedef name standing alone
    if len(toks) > 1 and toks[-1] not in base_words and toks[-1] != "*" and not (toks[-2] in ("const",) and len(toks) == 2):
        if not toks[-1].startswith("rt_") or toks[-2] in ("*",) or toks[-2].startswith("rt_") or toks[-2] in base_words:
            toks = toks[:-1]
    # split into the base type and the pointer levels (each with its own const-ness of the POINTEE)
    first_star = toks.index("*") if "*" in toks else len(toks)
    base, rest = toks[:first_star], toks[first_star:]
    base_const = "const" in base
    base = [t for t in base if t not in ("const", "struct")]
    name = " ".join(base)
    prim = {"int": "i32", "unsigned": "u32", "unsigned int": "u32", "uint32_t": "u32", "int32_t": "i32", "uint64_t": "u64", "int64_t": "i64",
            "size_t": "usize", "double": "f64", "float": "f32", "char": "c_char", "uint8_t": "u8", "unsigned char": "u8", "void": "void"}
    t = prim.get(name, name)
    levels = []  # const-ness of each pointer level's pointee, innermost first
    pointee_const = base_const
    for tok in rest:
        if tok == "*":
            levels.append(pointee_const)
            pointee_const = False
        elif tok == "const":
            pointee_const = True
    if array:
        levels.append(pointee_const)
    for c in levels:
        t = ("*const " if c else "*mut ") + t
    return t


def _rust_type(t: str) -> str:
    t = " ".join(t.split())
    m = re.fullmatch(r"\[(.+); (\d+)\]", t)
    if m:
        return "[%s; %s]" % (_rust_type(m.group(1)), m.group(2))
    for ptr in ("*const ", "*mut "):
        if t.startswith(ptr):
            return ptr + _rust_type(t[len(ptr):])
    return {"c_int": "i32", "c_uint": "u32", "c_double": "f64", "c_float": "f32", "c_void": "void", "c_char": "c_char"}.get(t, t)


def _split_params(params: str):
    params = params.strip()
    if params in ("", "void"):
        return []
    return [p.strip() for p in params.split(",") if p.strip()]  # (Rust allows a trailing comma)


def test_c_type_spelling():
    assert _c_type("const double M[16]") == "*const f64" and _c_type("double out[6]") == "*mut f64"
    assert _c_type("rt_scene *const *scenes") == "*const *mut rt_scene" and _c_type("const rt_scene *") == "*const rt_scene"
    assert _c_type("unsigned flags") == "u32" and _c_type("size_t n_pixels") == "usize" and _c_type("const char *path") == "*const c_char"
    assert _c_type("void *stream") == "*mut void" and _c_type("const void *p") == "*const void" and _c_type("uint64_t *out") == "*mut u64"
    assert _c_type("int") == "i32" and _c_type("rt_scene *") == "*mut rt_scene" and _c_type("const uint8_t *rgb") == "*const u8"


def test_rust_ffi_declares_every_entry_point():
    """ray-tracer_amd/rust/src/ffi.rs (uncompilable here: no Rust toolchain) against include/rt_mi355x.h, mechanically -- the one
    check that can stand in for `cargo check`: the same set of functions; for every function the same NUMBER of
    parameters, the same TYPE in every position (c_int / c_uint / c_double / usize / u64 / *const T / *mut T, arrays decaying to
    pointers) and the same return type; for every struct that crosses the boundary the same fields, in order, with the same types."""
    header = re.sub(r"/\*.*?\*/", "", (ROOT / "include" / "rt_mi355x.h").read_text(), flags=re.S)
    header = re.sub(r"//.*", "", header)
    declared = set(re.findall(r"\b(rt_[a-z0-9_]+)\s*\(", header))
    ffi = (ROOT / "ray-tracer_amd" / "rust" / "src" / "ffi.rs").read_text()
    ffi = re.sub(r"//.*", "", ffi)
    bound = set(re.findall(r"pub fn (rt_[a-z0-9_]+)\s*\(", ffi))
    assert declared == bound, (sorted(declared - bound), sorted(bound - declared))

    c_protos = {name: (ret, params) for ret, name, params in
                re.findall(r"^\s*([A-Za-z_][A-Za-z0-9_ \*]*?)\b(rt_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", header, flags=re.M | re.S)}
    assert set(c_protos) == declared, sorted(declared - set(c_protos))
    r_protos = {name: (params, ret) for name, params, ret in
                re.findall(r"pub fn (rt_[a-z0-9_]+)\s*\((.*?)\)\s*(?:->\s*([^;]+?))?\s*;", ffi, flags=re.S)}
    assert set(r_protos) == declared
    for name in sorted(declared):
        c_ret, c_params = c_protos[name]
        r_params, r_ret = r_protos[name]
        c_types = [_c_type(p) for p in _split_params(c_params)]
        r_types = [_rust_type(p.split(":", 1)[1]) for p in _split_params(r_params)]
        assert c_types == r_types, (name, c_types, r_types)
        c_r = _c_type(c_ret.strip() + " x")  # (a dummy declarator name, as for a parameter)
        r_r = _rust_type(r_ret) if r_ret else "void"
        assert c_r == r_r, (name, c_r, r_r)

    for struct in ("rt_camera", "rt_render_params", "rt_counters", "rt_launch_config", "rt_scene_info"):
        c_body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (struct, struct), header, flags=re.S).group(1)
        c_fields = []
        for decl in [d.strip() for d in c_body.split(";") if d.strip()]:
            m = re.fullmatch(r"(.+?)\s+([a-z_0-9, \[\]]+)", " ".join(decl.split()))
            ctype, names = m.group(1), m.group(2)
            for nm in [n.strip() for n in names.split(",")]:
                arr = re.fullmatch(r"([a-z_0-9]+)\[(\d+)\]", nm)
                base = _c_type(ctype + " x")
                c_fields.append((arr.group(1), "[%s; %s]" % (base, arr.group(2))) if arr else (nm, base))
        r_body = re.search(r"pub struct %s \{(.*?)\n\}" % struct, ffi, flags=re.S).group(1)
        r_fields = [(n, _rust_type(t)) for n, t in re.findall(r"pub ([a-z_0-9]+):\s*([^,\n]+),", r_body)]
        assert c_fields == r_fields, (struct, c_fields, r_fields)


def test_no_cpu_rendering_path(rt, scenes):
    """Without a HIP device the product must fail loudly, never fall back."""
    sc, cam = scenes.build_product(scenes.book_one(1, 1.5), device=-1)
    with pytest.raises(rt.RtError) as e:
        sc.render(cam, 16, 16, 1, 5)
    assert e.value.code == -3 and "no CPU rendering path" in str(e.value)
    if rt.device_count() == 0:
        s2 = rt.Scene()
        s2.sprite(s2.sphere(1.0), s2.lambertian(s2.solid((1, 1, 1))))
        with pytest.raises(rt.RtError) as e:
            s2.commit(0)
        assert e.value.code == -3


def test_only_tests_smoke_and_the_cpu_baseline_touch_the_oracle():
    """The oracle is test infrastructure (oracle/rt_oracle.h): outside tests/ and oracle/ only __graft_entry__.py (build of the
    checker, smoke()) and bench.py (the cpu_baseline leg and the eight checked pixels, both after the timed region) may name its
    binding, its library or its symbols; the package, the tools, the C++ and Rust sources never do -- and the shipped library exports
    no `orc_*` symbol and does not link the oracle."""
    import re
    import subprocess
    root = Path(__file__).resolve().parent.parent
    pat = re.compile(r"import\s+oracle_binding|from\s+oracle_binding|oracle_binding\.(?!py\b)\w|librt_oracle|\borc_[a-z]")
    allowed = {"__graft_entry__.py", "bench.py"}
    hits = []
    for f in root.rglob("*"):
        rel = f.relative_to(root)
        if not f.is_file() or rel.parts[0] in ("tests", "oracle", "gpurun_out", ".git", "profiles", "docs", "_ab") or f.suffix not in (
                ".py", ".sh", ".cpp", ".hpp", ".h", ".hip", ".rs", ".toml", "") and f.name != "Makefile":
            continue
        if f.suffix == "" and f.name != "Makefile":
            continue
        try:
            text = f.read_text()
        except UnicodeDecodeError:
            continue
        for i, line in enumerate(text.split("\n"), 1):
            if pat.search(line) and str(rel) not in allowed and not line.lstrip().startswith(("#", "//", "*")):
                hits.append(f"{rel}:{i}: {line.strip()[:100]}")
    assert not hits, hits
    lib = root / "ray-tracer_amd" / "lib" / "librt_mi355x.so"
    if lib.exists():
        syms = subprocess.run(["nm", "-D", str(lib)], capture_output=True, text=True).stdout
        assert " orc_" not in syms
        needed = subprocess.run(["readelf", "-d", str(lib)], capture_output=True, text=True).stdout
        assert "oracle" not in needed


def test_error_conventions(rt):
    s = rt.Scene()
    with pytest.raises(rt.RtError) as e:
        s.commit(-1)  # BoundingVolumeHierarchyNode::new(vec![]) -> None
    assert e.value.code == -2
    with pytest.raises(rt.RtError):
        s.lambertian(3)  # unknown texture
    with pytest.raises(rt.RtError):
        s.sprite(5, None)  # unknown geometry
    r = s.rectangle(1, 1)
    m = s.constant_medium(r, 0.1)  # ConstantMedium<T: Hit>: any boundary (src/volume.rs:40-43) ...
    m2 = s.constant_medium(m, 0.1)  # ... a medium included (round 3), three levels deep
    m3 = s.constant_medium(m2, 0.1)
    deep = rt.Scene()
    g4 = deep.constant_medium(deep.constant_medium(deep.constant_medium(deep.constant_medium(deep.sphere(1.0), 1.0), 1.0), 1.0), 1.0)
    deep.sprite(g4, None)
    with pytest.raises(rt.RtError) as e:
        deep.commit(-1)  # four ConstantMedium levels inside one another: outside RT_MAX_MEDIUM_NESTING, an error at commit
    assert e.value.code == -4 and "ConstantMedium levels" in str(e.value)
    assert m3 > m2 > m
    with pytest.raises(rt.RtError) as e:
        s.bvh([])  # BoundingVolumeHierarchyNode::new(vec![]) -> None
    assert e.value.code == -2
    with pytest.raises(rt.RtError):
        s.bvh([7])  # unknown sprite
    child = s.sprite(s.sphere(0.5), None)
    node = s.bvh([child])
    with pytest.raises(rt.RtError) as e:
        s.bvh([child])  # already moved into a node
    assert e.value.code == -5
    s.sprite(node, None)
    g = s.sphere(1.0)
    s.sprite(g, None)
    s.commit(-1)
    with pytest.raises(rt.RtError) as e:
        s.sphere(2.0)  # immutable after commit
    assert e.value.code == -5
    assert rt.lib().rt_device_count() >= 0


def test_mat4_matches_oracle_and_python_mirror(rt, oracle, scenes):
    rng = np.random.default_rng(0)
    for _ in range(20):
        t = rng.uniform(-10, 10, 3)
        ang = rng.uniform(-4, 4)
        ax = [(1.0, 0, 0), (0, 1.0, 0), (0, 0, 1.0), tuple(rng.uniform(-1, 1, 3))][rng.integers(4)]
        T, R = rt.Mat4.translation(t), rt.Mat4.rotation(ang, ax)
        M = T.multiplied(R)
        To, Ro, Mo = np.zeros(16), np.zeros(16), np.zeros(16)
        oracle.LIB.orc_kat_mat4_translation(oracle.dp(np.array(t)), oracle.dp(To))
        oracle.LIB.orc_kat_mat4_rotation(ang, oracle.dp(np.array(ax, dtype=np.float64)), oracle.dp(Ro))
        oracle.LIB.orc_kat_mat4_multiplied(oracle.dp(To), oracle.dp(Ro), oracle.dp(Mo))
        assert np.array_equal(T.a, To) and np.array_equal(R.a, Ro) and np.array_equal(M.a, Mo)
        assert np.array_equal(M.a, scenes.mat4_multiplied(scenes.mat4_translation(t), scenes.mat4_rotation(ang, ax)))
        inv, invo = M.inversed(), np.zeros(16)
        assert oracle.LIB.orc_kat_mat4_inversed(oracle.dp(Mo), oracle.dp(invo)) == 1
        assert np.array_equal(inv.a, invo)
        assert M.determinant() == oracle.LIB.orc_kat_mat4_determinant(oracle.dp(Mo))
    assert rt.Mat4(np.zeros(16)).inversed() is None
    assert np.array_equal(rt.Mat4.identity().a, np.eye(4).reshape(16))


def test_camera_matches_oracle(rt, oracle, scenes):
    for desc in (scenes.book_one(1, 1.5), scenes.cornell(1.0), scenes.cover(1, 1.0)):
        cam = rt.Camera(*desc.camera)
        f = np.zeros(9)
        oracle.LIB.orc_kat_camera_frame(oracle.build_oracle(desc).h, oracle.dp(f))
        assert np.array_equal(np.array(cam.c.lower_left), f[0:3])
        assert np.array_equal(np.array(cam.c.horizontal), f[3:6])
        assert np.array_equal(np.array(cam.c.vertical), f[6:9])


def test_tonemap_and_ppm_match_oracle(rt, oracle, tmp_path):
    rng = np.random.default_rng(1)
    img = rng.uniform(-0.2, 1.5, (7, 5, 3))
    img[0, 0] = [float("nan"), -0.0, float("inf")]
    img[1, 1] = [0.25, 1.0, 4.0]
    mine = rt.tonemap_rgb8(img)
    ref = np.zeros_like(mine)
    oracle.LIB.orc_tonemap_rgb8(oracle.dp(img), 35, ref.ctypes.data_as(C.POINTER(C.c_uint8)))
    assert np.array_equal(mine, ref)
    assert mine[1, 1].tolist() == [127, 255, 255] and mine[0, 0].tolist() == [255, 0, 255]
    a, b = tmp_path / "a.ppm", tmp_path / "b.ppm"
    rt.write_ppm_p3(a, img)
    assert oracle.LIB.orc_write_ppm_p3(str(b).encode(), oracle.dp(img), 5, 7) == 0
    assert a.read_bytes() == b.read_bytes()
    assert a.read_text().startswith("P3\n5 7\n255\n")


def _check_bvh(sc):
    info = sc.info()
    nodes = sc.nodes()
    n_prims, n_h = info["n_prims"], info["n_hoisted"]
    seen = []

    def walk(ref, box, depth):
        assert depth <= 24
        if ref < 0:
            p = ~ref
            n = sc.prim_group(p)  # 1, or 6: the head of a cube group, one leaf for six consecutive prims
            assert n in (1, 6)
            for q in range(p, p + n):
                assert sc.prim_group(q) == (n if q == p else 0)
                seen.append(q)
                b = sc.prim_bounds(q)
                assert np.all(b[:3] >= box[:3]) and np.all(b[3:] <= box[3:])
            return depth
        nd = nodes[ref]
        deepest = 0
        for c in range(2):
            cb = nd[c * 6:c * 6 + 6]
            assert np.all(cb[:3] >= box[:3]) and np.all(cb[3:] <= box[3:]), "child box escapes its parent"
            cull = nd[14 + c * 6:14 + c * 6 + 6]
            assert np.all(cull[:3] < cb[:3]) and np.all(cull[3:] > cb[3:]), "binary32 culling box must strictly contain the binary64 box"
            # padded by about 2^-21 of the coordinate scale, not more than 2^-19
            scale = np.maximum(np.maximum(np.abs(cb[:3]), np.abs(cb[3:])), cb[3:] - cb[:3])
            assert np.all(cb[:3] - cull[:3] <= scale * 2.0 ** -19 + 1e-29)
            deepest = max(deepest, walk(int(nd[12 + c]), cb, depth + 1))
        return deepest
    if n_prims > n_h:
        big = np.array([-np.inf] * 3 + [np.inf] * 3)
        root = 0 if len(nodes) else ~n_h
        d = walk(root, big, 0)
        assert d == info["max_depth"]
    assert sorted(seen) == list(range(n_h, n_prims)), "every non-hoisted prim must be a leaf exactly once"
    return info


def test_flatten_book_one(rt, scenes):
    desc = scenes.book_one(1, 1.5)
    sc, _ = scenes.build_product(desc, device=-1)
    info = _check_bvh(sc)
    assert info["n_prims"] == len(desc.sprites) and 480 <= info["n_prims"] <= 489
    assert info["n_hoisted"] == 2           # sky (r = 2000) and ground (r = 1000)
    assert info["feature_mask"] == rt.RT_FEAT_SPHERE_T  # every transform is a pure translation
    assert info["n_xforms"] == 0 and info["n_nodes"] == info["n_prims"] - info["n_hoisted"] - 1
    assert info["node_bytes"] == 64 and info["prim_bytes"] == 32
    # hoisted prims keep creation order: ground (sprite 0) then sky (sprite 1)
    assert np.allclose(sc.prim_bounds(0), [-1000, -2000, -1000, 1000, 0, 1000], rtol=1e-9)
    assert np.allclose(sc.prim_bounds(1), [-2000] * 3 + [2000] * 3, rtol=1e-9)


def test_flatten_cornell_and_cover(rt, scenes):
    sc, _ = scenes.build_product(scenes.cornell(), device=-1)
    info = _check_bvh(sc)
    # 6 rectangles + two cubes expanded into six rectangle leaves each (chain = sprite matrix, face matrix); the walls
    # span the scene but their boxes are flat: nothing is hoisted
    assert info["n_prims"] == 6 + 12 and info["n_child_prims"] == 0 and info["n_hoisted"] == 0
    assert info["n_xforms"] == 6 + 12 * 2
    assert info["feature_mask"] & rt.RT_FEAT_GENERAL
    d = scenes.cover(1)
    sc, _ = scenes.build_product(d, device=-1)
    info = _check_bvh(sc)
    assert info["n_prims"] == 2400 + 1 + 4 + 1 + 1 + 1 + 1000 and info["n_child_prims"] == 0
    assert info["feature_mask"] & rt.RT_FEAT_MEDIUM and info["feature_mask"] & rt.RT_FEAT_TEXTURED
    assert 1 <= info["n_hoisted"] <= 4  # the r = 5000 fog at least
    # round 5: the 400 floor boxes (Cube::new under a pure translation) are cube groups -- 400 leaves instead of 2400, a tree of
    # 1406 nodes instead of 3406 (45 KB with binary16 planes: it fits one workgroup's LDS beside the stack)
    heads = [p for p in range(info["n_prims"]) if sc.prim_group(p) == 6]
    assert len(heads) == 400 and heads[:3] == [1, 7, 13]
    assert info["n_nodes"] == (info["n_prims"] - info["n_hoisted"] - 5 * 400) - 1 == 1406
    # the Cornell box's two cubes are rotated (and the scene is a box list anyway): no groups there
    sc2, _ = scenes.build_product(scenes.cornell(), device=-1)
    assert all(sc2.prim_group(p) == 1 for p in range(18))


def test_flatten_edge_cases(rt, scenes):
    # single sprite: no BVH nodes at all
    s = rt.Scene()
    s.sprite(s.sphere(1.0), s.lambertian(s.solid((1, 1, 1))), scenes.mat4_translation((0, 0, 3)))
    assert s.commit(-1).info()["n_nodes"] == 0
    # geometry None and singular transforms are unhittable and dropped; material None is kept
    s = rt.Scene()
    g = s.sphere(1.0)
    s.sprite(None, None)
    s.sprite(g, None, [0.0] * 16)
    for i in range(5):
        s.sprite(g, None, scenes.mat4_translation((10.0 * i, 0, 0)))
    info = _check_bvh(s.commit(-1))
    assert info["n_prims"] == 5
    # five equal, well separated spheres: nothing is scene-spanning, nothing hoisted
    assert info["n_hoisted"] == 0 and info["n_nodes"] == 4
    # rotated sphere sprite takes the general-matrix path
    s = rt.Scene()
    s.sprite(s.sphere(1.0), None, scenes.mat4_rotation(0.3, (0.0, 1.0, 0.0)))
    assert s.commit(-1).info()["feature_mask"] & rt.RT_FEAT_GENERAL


def test_deep_tree_is_bounded(rt, scenes):
    """A pathological scene for SAH (nested shells) must still respect the stack bound."""
    s = rt.Scene()
    for i in range(200):
        s.sprite(s.sphere(1.5 ** (i * 0.25)), None, None)
    info = _check_bvh(s.commit(-1))
    assert info["max_depth"] <= 24


def test_shard_tile_counts(rt):
    for (w, h) in [(1200, 800), (100, 60), (7, 9), (8, 8)]:
        total = ((w + 7) // 8) * ((h + 7) // 8)
        for n in (1, 2, 3, 8):
            counts = [rt.shard_tile_count(w, h, r, n) for r in range(n)]
            assert sum(counts) == total and max(counts) - min(counts) <= 1
    with pytest.raises(rt.RtError):
        rt.shard_tile_count(10, 10, 2, 2)


def test_scene_generators_are_seeded(scenes):
    a, b, c = scenes.book_one(1), scenes.book_one(1), scenes.book_one(2)
    assert a.sprites == b.sprites and a.sprites != c.sprites
    assert 480 <= len(a.sprites) <= 489
    kinds = [m[0] for m in a.materials]
    assert 0.2 < kinds.count("lambertian") / len(kinds) < 0.4 and 0.3 < kinds.count("dielectric") / len(kinds) < 0.5
    cv = scenes.cover(1)
    heights = [cv.geometries[s[0]][2] for s in cv.sprites[:400]]
    assert all(1.0 <= hgt < 101.0 for hgt in heights)


def test_reference_width_follows_the_scene(rt):
    """up to 32767 prims and nodes: 16-bit node references; one more sprite and the scene commits with 32-bit ones
    (BoundingVolumeHierarchyNode::new takes any Vec, src/optimize.rs:366)"""
    def scene(n):
        s = rt.Scene()
        g = s.sphere(0.1)
        for i in range(n):
            s.sprite(g, None, [1.0, 0, 0, 0, 0, 1.0, 0, 0, 0, 0, 1.0, 0, float(i), 0.0, 0.0, 1.0])
        return s.commit(-1).info()
    a, b = scene(32767), scene(32768)
    assert not a["feature_mask"] & rt.RT_FEAT_WIDE and a["n_prims"] == 32767
    assert b["feature_mask"] & rt.RT_FEAT_WIDE and b["n_prims"] == 32768 and b["n_nodes"] == 32767 and b["max_depth"] <= 24


def test_png_writer_restates_main_rs(rt, tmp_path):
    """rt_write_png_rgba8 = examples/main.rs:105-135: channel (c.sqrt() * 255.0).min(255.0) as u8 (f64::min drops a
    NaN, the cast truncates and saturates), alpha 255, put_pixel(x, height - 1 - y); decoded here with zlib alone."""
    import struct
    import sys
    import zlib
    from pathlib import Path
    sys.path.insert(0, str(Path(__file__).resolve().parent / "golden"))
    from make_cover_stats import read_png_rgb
    rng = np.random.default_rng(0)
    img = rng.uniform(0.0, 1.3, (37, 53, 3))
    img[0, 0] = [np.nan, -1.0, np.inf]          # NaN / sqrt(-1) = NaN -> 255 (f64::min returns the other operand), inf -> 255
    img[1, 1] = [-0.0, 0.0, 1.0]                # 0, 0, 255
    img[2, 2] = [1e-300, 0.25, 0.999999]        # 0, 127 (127.5 truncates), 254
    path = tmp_path / "t.png"
    rt.write_png_rgba8(path, img)
    with np.errstate(invalid="ignore"):
        r = np.sqrt(img) * 255.0
    want = np.where(np.isnan(r), 255.0, np.minimum(r, 255.0))
    want = np.where(want <= 0.0, 0.0, want).astype(np.uint8)
    assert np.array_equal(read_png_rgb(path), want[::-1])                      # rows top-down
    assert np.array_equal(rt.tonemap_png8(img), want)
    assert want[0, 0].tolist() == [255, 255, 255] and want[1, 1].tolist() == [0, 0, 255] and want[2, 2].tolist() == [0, 127, 254]
    # container: signature, IHDR fields, per-chunk CRCs, zlib stream with a valid Adler-32 (zlib.decompress checks it)
    b = path.read_bytes()
    assert b[:8] == b"\x89PNG\r\n\x1a\n"
    pos, kinds = 8, []
    while pos < len(b):
        n, typ = struct.unpack(">I4s", b[pos:pos + 8])
        data = b[pos + 8:pos + 8 + n]
        assert struct.unpack(">I", b[pos + 8 + n:pos + 12 + n])[0] == zlib.crc32(typ + data)
        if typ == b"IHDR":
            assert struct.unpack(">IIBBBBB", data) == (53, 37, 8, 6, 0, 0, 0)
        if typ == b"IDAT":
            raw = zlib.decompress(data)
            assert len(raw) == 37 * (1 + 53 * 4) and raw[4::4][:53] == b"\xff" * 53   # filter 0 rows, alpha 255
        kinds.append(typ)
        pos += 12 + n
    assert kinds == [b"IHDR", b"IDAT", b"IEND"]
    with pytest.raises(rt.RtError):
        rt.write_png_rgba8(tmp_path / "no" / "dir.png", img)


def test_unit_ball_coordinate_from_bits_equals_the_reference_formula():
    """include/rt_rng.h rt_u64_to_pm1: the device builds random::<f64>() * 2.0 - 1.0 from the bits of the draw (1.m minus 1 or 2)
    instead of converting a 53-bit integer and multiplying; the two forms are the same double for every draw -- checked here on
    a million draws and on the edge patterns, in numpy's IEEE arithmetic."""
    rng = np.random.default_rng(5)
    x = rng.integers(0, 1 << 64, size=1_000_000, dtype=np.uint64)
    edge = np.array([0, 1, (1 << 11) - 1, 1 << 11, (1 << 63) - 1, 1 << 63, (1 << 63) + (1 << 11), (1 << 64) - 1, (1 << 64) - (1 << 11),
                     0x7FFFFFFFFFFFF800, 0x8000000000000800, 0x00000000000007FF, 0xFFFFFFFFFFFFF7FF], dtype=np.uint64)
    x = np.concatenate([x, edge])
    ref = (x >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0) * 2.0 - 1.0
    hi, lo = (x >> np.uint64(32)).astype(np.uint32), (x & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    mhi = np.uint32(0x3FF00000) | ((hi >> np.uint32(11)) & np.uint32(0xFFFFF))
    mlo = (hi << np.uint32(21)) | (lo >> np.uint32(11))
    one_m = ((mhi.astype(np.uint64) << np.uint64(32)) | mlo.astype(np.uint64)).view(np.float64)
    built = one_m - np.where(hi >> np.uint32(31) == 1, 1.0, 2.0)
    assert np.array_equal(ref, built)
    assert ref.min() >= -1.0 and ref.max() < 1.0


def test_lds_layout_is_aligned_and_fits_for_every_launch_shape(lane_emul):
    """ray-tracer_amd/csrc/rt_lds.h, the ONE layout host and kernel share (an odd queue capacity put the metal
    queue's binary64 arrays on a 4-mod-8 address; round 2's hang: a region the host did not size): over a sweep of stack depths,
    node-array sizes, reference widths and kernel families the regions are ordered, aligned (node copy / job state 16 B, queues
    8 B, every class queue 8 B), the capacity is even, and `groups_per_cu` workgroups fit one CU's 160 KiB whenever the
    capacity is above the 16-entry floor."""
    nq = 3  # the three class queues: the layout has no other (rt_lds.h)
    # (1024, 1, 2048): the family with sphere media / textures since round 5 -- one workgroup per CU, one SET of queues per eight waves
    for block, groups, front in ((512, 2, 0), (256, 4, 0), (256, 3, 0), (256, 4, 2048), (256, 3, 2048), (1024, 1, 2048)):  # front: the log table of the media families
        for entry_bytes in (2, 4, 8):
            for stack_entries in range(1, 25):
                for node_bytes in (0, 64, 576, 648, 1024, 9 * 64, 250 * 64, 484 * 64, 1000 * 64):
                    l = lane_emul.lds_layout(stack_entries, block, entry_bytes, node_bytes, groups, front)
                    assert l["aligned"] == 1, (block, entry_bytes, stack_entries, node_bytes, l)
                    assert l["cap"] % 2 == 0 and l["cap_effective"] % 2 == 0 and 16 <= l["cap"] <= 64
                    assert l["stack_off"] == front and l["node_off"] == front + stack_entries * block * entry_bytes
                    assert l["job_off"] >= l["node_off"] + node_bytes and l["swap_off"] == l["job_off"] + (block // 64) * 32
                    for cls in range(nq):  # every queue starts 8-byte aligned (its first 14 arrays are binary64)
                        assert (l["swap_off"] + 32 + cls * l["swap_class_bytes"]) % 8 == 0
                    sets = block // 512 if block >= 1024 else 1
                    assert l["total"] == l["swap_off"] + sets * (32 + nq * l["swap_class_bytes"])
                    assert (32 + nq * l["swap_class_bytes"]) % 8 == 0  # the second set's binary64 arrays stay aligned
                    if l["cap"] > 16 and block < 512:
                        assert l["total"] <= (160 * 1024 // groups) // 512 * 512, l
    # the book-two cover: a stack of 14 entries, 1406 nodes of 32 bytes, the log table, two sets of queues of 64: one CU's LDS holds it
    l = lane_emul.lds_layout(14, 1024, 4, 1406 * 32, 1, 2048)
    assert l["cap_effective"] == 64 and l["total"] == 2048 + 14 * 4096 + 1406 * 32 + 16 * 32 + 2 * (32 + 3 * 124 * 64) <= 160 * 1024
    # the case the advisor named: ~250 leaves with the node array in LDS used to get 39 entries; now an even 38
    l = lane_emul.lds_layout(12, 256, 4, 250 * 64, 4)
    assert l["cap"] % 2 == 0


def test_launch_plans_of_every_kernel_family(rt, scenes, monkeypatch):
    """rt_scene_plan_launch: what a render WOULD launch -- workgroup size, dynamic LDS, where the node array and the records live, the
    queues' capacity -- decided by the code a render runs (rt_api.cpp plan_launch), here without a device.  One scene per kernel
    family and per form; the switches move the plan the way the README says.  (Round 5 found a plan that gave the spheres-only
    family a binary16 tree it has no kernels for only on the GPU box, where the kernel refused the launch: this test is what
    would have found it here.)"""
    def plan(d):
        sc, _ = scenes.build_product(d, device=-1)
        return sc.plan_launch()
    b = plan(scenes.book_one(1, 1.5))       # spheres only: two 512-thread groups per CU, each with its copy of the 484 nodes
    assert (b["block_threads"], b["blocks_per_cu"], b["lds_nodes"], b["swap_cap"], b["records_in_lds"], b["kernel_features"]) == (512, 2, 1, 64, 0, 0)
    assert b["lds_bytes"] == 79648 <= 80 * 1024  # stack + 484 nodes of 64 bytes + job state + three queues of 64
    c = plan(scenes.cornell())               # lean general, box list: records + list in LDS, half-word stack entries
    assert (c["block_threads"], c["blocks_per_cu"], c["lds_nodes"], c["records_in_lds"], c["swap_cap"], c["kernel_features"]) == (256, 4, 1, 1, 64, 1)
    assert c["lds_bytes"] <= 40 * 1024
    v = plan(scenes.cover(1, 1.0))           # sphere media + textures: one 1024-thread group, the tree with binary16 planes, two sets of queues
    assert (v["block_threads"], v["blocks_per_cu"], v["lds_nodes"], v["records_in_lds"], v["swap_cap"], v["kernel_features"]) == (1024, 1, 2, 0, 64, 7)
    depth = scenes.build_product(scenes.cover(1, 1.0), device=-1)[0].info()["max_depth"]
    assert depth <= 15  # (round 4's tree of 3406 nodes: 16)
    # the log table, the stack, 1406 nodes of 32 bytes, sixteen waves' job state, two sets of queues: one CU's LDS holds it
    assert v["lds_bytes"] == 2048 + (depth + 1) * 4096 + 1406 * 32 + 16 * 32 + 2 * (32 + 3 * 124 * 64) <= 160 * 1024
    t = plan(scenes.cube_row(5))             # lean general, a TREE of 32 leaves: records and nodes in LDS
    assert (t["block_threads"], t["lds_nodes"], t["records_in_lds"]) == (256, 1, 1) and 32 <= t["swap_cap"] <= 64
    t9 = plan(scenes.cube_row(9))            # 56 leaves, 25 KB of records: nodes in LDS, records in global memory
    assert (t9["lds_nodes"], t9["records_in_lds"]) == (1, 0)
    g = plan(scenes.instanced())             # media over general boundaries: three 256-thread groups, 3 waves per SIMD
    assert (g["block_threads"], g["blocks_per_cu"], g["waves_per_simd"], g["records_in_lds"]) == (256, 3, 3, 0) and g["lds_nodes"] in (0, 1)
    n = plan(scenes.nested_media())          # media inside media: its own family
    assert n["kernel_features"] & 16 and n["lds_nodes"] in (0, 1) and n["swap"] == 1
    # a field of 1300 spheres: node array + stack + queues no longer fit half a CU's LDS -> global memory; NEVER the binary16 form
    # (only the family with sphere media / textures has kernels for it)
    rng = np.random.default_rng(3)
    d = scenes.SceneDesc()
    m = d.lambertian_rgb((0.5, 0.5, 0.5))
    for i in range(36):
        for j in range(36):
            d.sprite(d.geom("sphere", 0.3), m, scenes.mat4_translation((i - 18.0 + float(rng.uniform(0, 0.3)), 0.3, j - 18.0 + float(rng.uniform(0, 0.3)))))
    d.sprite(d.geom("sphere", 3000.0), d.mat("diffuse_light", d.tex_solid((0.6, 0.7, 1.0))), None)
    d.camera = ((20.0, 6.0, 8.0), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), 0.5, 1.5, 20.0, 0.02)
    big = plan(d)
    assert (big["kernel_features"], big["lds_nodes"], big["block_threads"]) == (0, 0, 512)
    monkeypatch.setenv("RT_HALF_NODES", "1")
    assert plan(d)["lds_nodes"] == 0 and plan(scenes.book_one(1, 1.5))["lds_nodes"] == 1 and plan(scenes.cube_row(5))["lds_nodes"] == 1
    monkeypatch.delenv("RT_HALF_NODES")
    # the switches
    for env, scene, field, want in (("RT_NO_HALF_NODES", scenes.cover(1, 1.0), "lds_nodes", 0), ("RT_NO_LDS_NODES", scenes.book_one(1, 1.5), "lds_nodes", 0),
                                    ("RT_NO_LDS_RECORDS", scenes.cornell(), "records_in_lds", 0), ("RT_NO_LDS_RECORDS", scenes.cube_row(5), "records_in_lds", 0)):
        monkeypatch.setenv(env, "1")
        assert plan(scene)[field] == want, (env, field)
        monkeypatch.delenv(env)
    monkeypatch.setenv("RT_SWAP", "0")
    w = plan(scenes.cover(1, 1.0))
    assert (w["swap"], w["swap_cap"], w["lds_nodes"], w["records_in_lds"]) == (0, 0, 1, 0)  # without the queues' 47 KB the 90 KB of binary32 nodes fit
    monkeypatch.delenv("RT_SWAP")
    monkeypatch.setenv("RT_NO_CUBE_GROUPS", "1")
    assert plan(scenes.cover(1, 1.0))["lds_nodes"] == 0  # 3406 nodes: 109 KB even with binary16 planes
    monkeypatch.delenv("RT_NO_CUBE_GROUPS")
    with pytest.raises(rt.RtError):
        rt.Scene().plan_launch()  # not committed


def test_scene_records_blob_of_small_general_scenes(scenes, lane_emul):
    """Small general scenes -- the box-LIST walk (<= 24 leaves) and, since round 5, trees of up to 64 leaves in the kernel families
    that have the form -- carry their transform / prim / material records a second time as ONE packed blob that their kernels copy
    into LDS (rtl::rec_at<true>): every array byte for byte at a 16-byte offset, at most RT_LIST_SCENE_MAX (16 KiB) in all.  A scene
    with more records has no blob (its records stay in global memory); neither has a scene of more than 64 leaves, of spheres only,
    or with media over general boundaries."""
    cases = ((scenes.cornell(), 18, True), (scenes.cube_row(2), 14, True), (scenes.cube_row(3, levels=1), 20, True),
             (scenes.cube_row(3, levels=4), 20, False),   # 20 leaves x 5 levels x 192 B: 19 KB
             (scenes.cube_row(5), 0, True),               # 32 leaves: a TREE scene, 14 KB of records
             (scenes.cube_row(9), 0, False),              # 56 leaves, 25 KB
             (scenes.book_one(1, 1.5), 0, False), (scenes.cover(1, 1.0), 0, False), (scenes.instanced(), 0, False))
    for d, want_list, want_blob in cases:
        sc, _ = scenes.build_product(d, device=-1)
        verdict, nbytes, n_list = lane_emul.scene_blob_check(sc)
        assert n_list == want_list, (d.name, n_list)
        assert (verdict == 0 and 0 < nbytes <= 16384 and nbytes % 16 == 0) if want_blob else (verdict == -1 and nbytes == 0), (d.name, verdict, nbytes)
        sc.close()
    # a tree scene with cube groups: the groups' records travel in the blob's prim_geo (two slots each behind the prims)
    d = scenes.SceneDesc()
    m = d.lambertian_rgb((0.7, 0.6, 0.5))
    for i in range(5):
        d.sprite(d.geom("cube", 1.0 + 0.1 * i, 1.0, 1.0), m, scenes.mat4_translation((2.0 * i - 4.0, 0.5, 0.0)))
    d.sprite(d.geom("sphere", 100.0), d.mat("diffuse_light", d.tex_solid((0.7, 0.8, 1.0))), None)
    d.camera = ((0.0, 3.0, -9.0), (0.0, 0.5, 0.0), (0.0, 1.0, 0.0), 0.8, 1.0, 10.0, 0.0)
    sc, _ = scenes.build_product(d, device=-1)
    assert sum(1 for q in range(sc.info()["n_prims"]) if sc.prim_group(q) == 6) == 5
    verdict, nbytes, n_list = lane_emul.scene_blob_check(sc)
    assert (verdict, n_list) == (0, 0) and nbytes > 0


def test_the_deepest_stack_of_a_list_scene(scenes, oracle, lane_emul):
    """scenes.slab_stack: parallel rectangles one behind the other -- a camera ray crosses the culling boxes of all of them, so the
    box-list walk pushes every box but the nearest (n_list - 1 entries, what rt_api.cpp sizes a list scene's stack for) and pops them
    against the shrinking best hit; the picture equals the oracle's."""
    d = scenes.slab_stack(22)
    sc, cam = scenes.build_product(d, device=-1)
    n_list = lane_emul.scene_blob_check(sc)[2]
    assert n_list == 20  # (22 leaves, two of them scene-filling: tested directly, not in the list)
    img, _, high_water = lane_emul.render(sc, cam, 64, 64, 4, 30, seed=2)
    assert high_water == n_list - 1
    assert np.array_equal(img, oracle.build_oracle(d).render(64, 64, 4, 30, seed=2, iterative=True, nthreads=8)) and img.mean() > 0.1
