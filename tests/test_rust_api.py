"""The Rust crate (ray-tracer_amd/rust/, uncompilable here: no Rust tool chain) against the reference's own constructor surface.

VERDICT r4 #1: the crate has to offer the API the reference's three examples are written against -- `Sphere::new(r).into()`,
`Lambertian::new(v).into()`, `Mat4::translation(v)`, `BoundingVolumeHierarchyNode::new(vec) -> Option<Self>` ... -- so that their
scene-building code is drop-in.  Two guards:

* CALL_SHAPES / IMPORTS below: every constructor call shape and import path the scene code of examples/book-one.rs:103-205,
  examples/cornell-box.rs:31-140 and examples/main.rs:156-330 uses, WRITTEN BY HAND from those lines (names + arity + what the
  signature must contain) -- each must exist in the crate's sources;
* where /root/reference is present (this container; not the GPU box), the same list is extracted mechanically from the reference's
  three files and checked as well, so the hand-written list cannot quietly miss a call.
"""
import re
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
CRATE = ROOT / "ray-tracer_amd" / "rust"
REFERENCE = Path("/root/reference")

# (module, type, function, number of arguments besides self, substrings the signature must contain)
CALL_SHAPES = [
    ("geometry", "Sphere", "new", 1, ["radius: f64", "-> Self"]),                       # src/geometry.rs:17
    ("geometry", "Rectangle", "new", 2, ["width: f64", "height: f64"]),                 # :136
    ("geometry", "Cube", "new", 3, ["-> Vec<TransformedGeometry<Rectangle>>"]),         # :254
    ("geometry", "TransformedGeometry", "new", 2, ["geometry: T", "M: Into<Mat4Cached>"]),  # :191
    ("volume", "ConstantMedium", "new", 2, ["boundary: Arc<T>", "density: f64"]),       # src/volume.rs:24
    ("material", "Lambertian", "new", 1, ["T: Into<Arc<dyn Texture>>"]),                # src/material.rs:33
    ("material", "Metal", "new", 2, ["T: Into<Arc<dyn Texture>>", "fuzziness: f64"]),   # :79
    ("material", "Dielectric", "new", 1, ["refractive: f64"]),                          # :128
    ("material", "DiffuseLight", "new", 1, ["T: Into<Arc<dyn Texture>>"]),              # :280
    ("material", "Isotropic", "new", 1, ["T: Into<Arc<dyn Texture>>"]),                 # :307
    ("material", "SolidColor", "new", 1, ["color: Vec3"]),                              # :206
    ("material", "CheckerTexture", "new", 2, ["black: T", "white: T"]),                 # :224
    ("material", "ImageTexture", "new", 1, ["mapping: T"]),                             # :252
    ("mat4", "Mat4", "identity", 0, ["-> Self"]),                                       # src/mat4.rs:21
    ("mat4", "Mat4", "translation", 1, ["offset: Vec3", "-> Self"]),                    # :36
    ("mat4", "Mat4", "rotation", 2, ["radians: f64", "axis: Vec3"]),                    # :52
    ("mat4", "Mat4", "multiplied", 1, ["&self", "other: &Self"]),                       # :85
    ("mat4", "Mat4", "inversed", 0, ["&self", "-> Option<Self>"]),                      # :184
    ("vec3", "Vec3", "new", 3, ["x: f64", "y: f64", "z: f64"]),                         # src/vec3.rs:13
    ("vec3", "Vec3", "ex", 0, ["-> Self"]), ("vec3", "Vec3", "ey", 0, ["-> Self"]), ("vec3", "Vec3", "ez", 0, ["-> Self"]),
    ("vec3", "Vec3", "length", 0, ["&self", "-> f64"]),
    ("vec3", "Vec3", "r", 0, ["&self"]), ("vec3", "Vec3", "g", 0, ["&self"]), ("vec3", "Vec3", "b", 0, ["&self"]),
    ("camera", "PerspectiveCamera", "new", 7, ["eye: Vec3", "center: Vec3", "up: Vec3", "fov: f64", "aspect: f64", "focusDistance: f64", "lensRadius: f64"]),  # src/camera.rs:25
    ("optimize", "BoundingVolumeHierarchyNode", "new", 1, ["objects: Vec<Arc<dyn Bound<AxisAlignedBoundingBox>>>", "-> Option<Self>"]),  # src/optimize.rs:366
    ("optimize", "AxisAlignedBoundingBox", "new", 2, ["min: Vec3", "max: Vec3"]),       # :27
    ("sprite", "Sprite", "builder", 0, ["-> SpriteBuilder<T, U>"]),                     # src/sprite.rs:61
    ("sprite", "Sprite", "new", 2, ["geometry: Option<Arc<T>>", "material: Option<Arc<U>>"]),
    ("sprite", "SpriteBuilder", "geometry", 1, ["mut self", "geometry: Arc<T>", "-> Self"]),   # :32
    ("sprite", "SpriteBuilder", "material", 1, ["mut self", "material: Arc<U>", "-> Self"]),   # :42
    ("sprite", "SpriteBuilder", "transform", 1, ["mut self", "M: Into<Mat4Cached>"]),          # :47
    ("sprite", "SpriteBuilder", "build", 0, ["self", "-> Sprite<T, U>"]),                      # :27
]
# `use ray_tracer::<module>::<Item>` of the examples, minus the two items only their sampling loops use (camera::Camera, render::color)
IMPORTS = [("camera", "PerspectiveCamera"), ("geometry", "Cube"), ("geometry", "Rectangle"), ("geometry", "Sphere"), ("mat4", "Mat4"),
           ("material", "CheckerTexture"), ("material", "Dielectric"), ("material", "DiffuseLight"), ("material", "ImageTexture"),
           ("material", "Isotropic"), ("material", "Lambertian"), ("material", "Material"), ("material", "Metal"),
           ("material", "SolidColor"), ("material", "Texture"), ("ray", "Hit"), ("sprite", "Sprite"), ("vec3", "Vec3"),
           ("volume", "ConstantMedium"), ("optimize", "AxisAlignedBoundingBox"), ("optimize", "Bound"),
           ("optimize", "BoundingVolumeHierarchyNode")]
# trait impls the examples' coercions and conversions rest on: (module, regex over the source)
IMPLS = [
    ("material", r"impl From<Vec3> for Arc<dyn Texture>"),                                       # Lambertian::new(Vec3::new(..))  (src/material.rs:48)
    ("mat4", r"impl From<Mat4> for Mat4Cached"),                                                 # .transform(Mat4::translation(..)) (src/mat4.rs:459)
    ("sprite", r"impl<T, U> Bound<AxisAlignedBoundingBox> for Sprite<T, U>"),                    # Arc::new(sprite) as Arc<dyn Bound<..>>
    ("geometry", r"impl<T> Bound<AxisAlignedBoundingBox> for TransformedGeometry<T>"),           # Arc::new(face) as Arc<dyn Bound<..>> (Cube)
    ("geometry", r"impl Bound<AxisAlignedBoundingBox> for Sphere"), ("geometry", r"impl Bound<AxisAlignedBoundingBox> for Rectangle"),
    ("optimize", r"impl Bound<AxisAlignedBoundingBox> for BoundingVolumeHierarchyNode<AxisAlignedBoundingBox>"),  # a node in a list / as a geometry
    ("volume", r"impl<T, B> Bound<B> for ConstantMedium<T>"),                                    # Sprite<ConstantMedium<Sphere>, Isotropic> in the world
    ("material", r"impl<T> Texture for ImageTexture<T>"),
    ("vec3", r"impl Mul<Vec3> for Vec3"), ("vec3", r"impl Sub<Vec3> for Vec3"), ("vec3", r"impl Mul<Vec3> for f64"),  # albedo * albedo, center - v, 0.9 * x
]


def _strip(src: str) -> str:
    src = re.sub(r"//[^\n]*", "", src)
    return re.sub(r"/\*.*?\*/", "", src, flags=re.S)


def _module(name: str) -> str:
    return _strip((CRATE / "src" / f"{name}.rs").read_text())


def _top_level_args(arglist: str):
    """split `a: T, b: Vec<(X, Y)>` at top-level commas"""
    out, depth, cur = [], 0, ""
    for ch in arglist:
        if ch in "(<[{":
            depth += 1
        elif ch in ")>]}":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


def _inherent_functions(src: str, type_name: str):
    """{fn name: (signature text, number of non-self parameters)} of every `impl .. Type .. {` block that is not a trait impl"""
    found = {}
    for m in re.finditer(r"\bimpl(?:<[^{]*?>)?\s+(%s)(?:<[^{]*?>)?\s*(?:where[^{]*)?\{" % re.escape(type_name), src):
        depth, i = 1, m.end()
        while depth and i < len(src):
            depth += {"{": 1, "}": -1}.get(src[i], 0)
            i += 1
        body = src[m.end():i]
        for f in re.finditer(r"pub(?:\(crate\))? fn (\w+)\s*(<[^(]*>)?\s*\((.*?)\)\s*(->\s*[^{;]+?)?\s*(where[^{]*)?\{", body, flags=re.S):
            name, generics, params, ret, where = f.groups()
            args = _top_level_args(" ".join(params.split()))
            n = len([a for a in args if not re.fullmatch(r"(&\s*)?(mut\s+)?self", a)])
            sig = " ".join(f"{generics or ''} ({' '.join(params.split())}) {ret or ''} {where or ''}".split())
            sig = sig.replace("-> ", "-> ").replace(" ,", ",")
            found[name] = (sig, n)
    return found


@pytest.mark.parametrize("module,type_name,fn,arity,must", CALL_SHAPES, ids=[f"{t}::{f}/{a}" for _, t, f, a, _ in CALL_SHAPES])
def test_every_constructor_call_shape_of_the_reference_examples_exists(module, type_name, fn, arity, must):
    fns = _inherent_functions(_module(module), type_name)
    assert fn in fns, f"{type_name}::{fn} is missing from ray-tracer_amd/rust/src/{module}.rs"
    sig, n = fns[fn]
    assert n == arity, (type_name, fn, sig)
    for piece in must:
        assert piece in sig, (type_name, fn, piece, sig)


def test_every_import_path_of_the_reference_examples_resolves():
    lib = _strip((CRATE / "src" / "lib.rs").read_text())
    cargo = (CRATE / "Cargo.toml").read_text()
    assert re.search(r'\[lib\]\s*name = "ray_tracer"', cargo), "`extern crate ray_tracer;` must resolve to this crate's library"
    for module, item in IMPORTS:
        assert re.search(r"pub mod %s;" % module, lib), module
        assert re.search(r"pub (?:struct|trait|enum|type) %s\b" % item, _module(module)), f"ray_tracer::{module}::{item}"
    for module, pattern in IMPLS:
        assert re.search(re.escape(pattern), _module(module)), f"{pattern} (src/{module}.rs)"


def test_the_crate_adds_no_second_conversion_into_an_arc():
    """`Sprite::builder().geometry(Sphere::new(1.0).into())` infers `Arc<Sphere>` through std's ONE `From<T> for Arc<T>`; an
    `impl From<Sphere> for Arc<..>` of the crate's own would make that `.into()` ambiguous (the examples would stop compiling)."""
    for f in (CRATE / "src").glob("*.rs"):
        for m in re.finditer(r"impl(?:<[^>]*>)?\s+(?:From<(\w+)[^>]*>\s+for\s+Arc<|Into<Arc<[^{]*>\s+for\s+(\w+))", _strip(f.read_text())):
            source = m.group(1) or m.group(2)
            assert source == "Vec3", (f.name, m.group(0))  # the one conversion upstream has too: a colour is a texture


def test_examples_use_only_what_the_crate_has():
    """every `Type::function(` of the three Rust examples is an inherent function of the crate with that arity"""
    types = {t: m for m, t, *_ in CALL_SHAPES}
    types.update({"HostRng": "util"})
    for ex in ("book_one.rs", "cornell_box.rs", "main.rs"):
        src = _strip((CRATE / "examples" / ex).read_text())
        for m in re.finditer(r"\b([A-Z]\w+)::(\w+)\s*\(", src):
            t, f = m.groups()
            if t in ("Arc", "Vec", "Some", "Object") or t not in types:
                continue
            fns = _inherent_functions(_module(types[t]), t)
            assert f in fns, (ex, t, f)
            depth, i = 1, m.end()
            while depth:
                depth += {"(": 1, ")": -1}.get(src[i], 0)
                i += 1
            assert len(_top_level_args(src[m.end():i - 1])) == fns[f][1], (ex, m.group(0), src[m.end():i - 1])


@pytest.mark.skipif(not (REFERENCE / "examples").is_dir(), reason="/root/reference is not present (it never is on the GPU box)")
def test_call_shapes_extracted_from_the_reference_examples_themselves():
    """Mechanical counterpart of CALL_SHAPES: every `Type::function(..)` call and every `use ray_tracer::..` path in the
    reference's three examples, outside what their sampling loops use, exists in the crate with the same number of arguments."""
    known = {(t, f): (m, a) for m, t, f, a, _ in CALL_SHAPES}
    imports = set(IMPORTS)
    loop_only = {("camera", "Camera"), ("render", "color")}
    for ex in ("book-one.rs", "cornell-box.rs", "main.rs"):
        src = _strip((REFERENCE / "examples" / ex).read_text())
        for module, item in re.findall(r"use ray_tracer::(\w+)::(\w+);", src):
            assert (module, item) in imports or (module, item) in loop_only, (ex, module, item)
        for m in re.finditer(r"\b([A-Z]\w+)::(\w+)\s*\(", src):
            t, f = m.groups()
            if t in ("Arc", "Vec3") and f in ("new", "clone") and t == "Arc":
                continue
            if t in ("Arc", "Self") or (t, f) == ("Vec3", "new"):
                continue
            if (t, f) not in known:
                # outside the crate (image::*, std) or part of the sampling loop
                assert t in ("DynamicImage", "Rgba", "ImageFormat"), (ex, t, f)
                continue
            depth, i = 1, m.end()
            while depth:
                depth += {"(": 1, ")": -1}.get(src[i], 0)
                i += 1
            assert len(_top_level_args(src[m.end():i - 1])) == known[(t, f)][1], (ex, m.group(0))
        # builder chains
        for meth in re.findall(r"\.(geometry|material|transform|build)\(", src):
            assert ("SpriteBuilder", meth) in known
