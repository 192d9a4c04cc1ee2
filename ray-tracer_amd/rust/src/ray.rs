//! `Hit` (src/ray.rs:45-47).  In the reference an object that can be hit computes `hit(&Ray) -> Option<HitRecord>`; here it
//! RECORDS itself, and the intersection runs in the HIP kernels (`rtl::prim_hit`, ray-tracer_amd/csrc/rt_lane.h).  The trait keeps
//! its name, its supertraits' shape (`Send + Sync + Debug` come in through `Bound`) and its place in every bound of the API, so
//! generic code over `T: Hit` -- `Sprite<T, U>`, `ConstantMedium<T>`, `TransformedGeometry<T>` -- reads as upstream.
//!
//! A user's own `impl Hit` has to describe itself through the closed set of the C ABI (spheres, rectangles, cubes, transformed
//! geometries, nodes, constant media): an intersection routine written in Rust cannot cross it.  See README.md, "user impls".
use crate::gpu::{Error, Recorder};

use std::fmt::Debug;

pub trait Hit: Send + Sync + Debug {
    /// Record this object as a GEOMETRY -- what a `Sprite`'s `geometry`, a `ConstantMedium`'s `boundary` or a
    /// `TransformedGeometry` holds -- and return its geometry id (`rt_add_geometry_*`).
    fn record_geometry(&self, recorder: &mut Recorder) -> Result<i32, Error>;

    /// Record this object as a MEMBER of a list: the `Vec<Arc<dyn Bound<_>>>` a `BoundingVolumeHierarchyNode` is made of.  Members
    /// are sprites in the C ABI; `out` receives the ids.  The default serves a bare geometry in a list: the sprite without
    /// material and with the identity transform (the reference's examples never do that: every member is a `Sprite`, a node or a
    /// face of a `Cube`, which all override this).
    fn record_member(&self, recorder: &mut Recorder, out: &mut Vec<i32>) -> Result<(), Error> {
        let geometry = self.record_geometry(recorder)?;
        out.push(recorder.sprite(geometry, -1, None)?);
        Ok(())
    }

    /// `Some((width, height, depth, face))` for face number `face` of `Cube::new(width, height, depth)`: how a node of exactly those
    /// six faces is recognised and recorded as ONE cube (`rt_add_geometry_cube`).
    fn cube_face(&self) -> Option<(f64, f64, f64, usize)> {
        None
    }
}
