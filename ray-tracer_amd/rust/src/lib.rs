//! `ray_tracer` -- the builder API of aiifabbf/ray-tracer over librt_mi355x.so (include/rt_mi355x.h).
//!
//! The module tree, the type names, the constructors' signatures and the generic shape of `Sprite<T, U>` are the reference's
//! (src/lib.rs:1-12), so the scene-building code of its three examples compiles against this crate as it stands:
//! `Sphere::new(r).into()`, `Lambertian::new(v).into()`, `Mat4::translation(v)`, `Cube::new(w, h, d)`,
//! `BoundingVolumeHierarchyNode::new(vec) -> Option<Self>`, `Arc<dyn Bound<AxisAlignedBoundingBox>>` lists and all.
//! What is different underneath: `Hit`, `Material` and `Texture` do not intersect, scatter or sample anything on the host.
//! Each carries ONE required method that RECORDS the object through the C ABI (`gpu::Recorder`); the world -- the outermost
//! `BoundingVolumeHierarchyNode` -- commits that recording to a HIP device on its first `render` and the per-pixel sampling loop
//! of the drivers (threads + mpsc + `color`, examples/book-one.rs:52-88) becomes `world.render(&camera, w, h, spp, depth, seed)`.
//!
//! NOT COMPILED in this repository's build image (no Rust tool chain there); README.md says what guards it instead.
#![allow(non_snake_case)]

pub mod camera;
pub mod ffi;
pub mod geometry;
pub mod gpu;
pub mod mat4;
pub mod material;
pub mod optimize;
pub mod ray;
pub mod sprite;
pub mod util;
pub mod vec3;
pub mod volume;
