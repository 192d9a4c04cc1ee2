//! `ConstantMedium<T>` (src/volume.rs:18-100): a boundary filled with a medium of constant density, scattered in by `Isotropic`.
//! Generic over any `T: Hit` as upstream -- a sphere (examples/main.rs:241-263), a cube node, another medium (up to three levels).
use crate::ffi;
use crate::gpu::{check, Error, Recorder};
use crate::optimize::Bound;
use crate::ray::Hit;

use std::sync::Arc;

#[derive(Debug, Clone)]
pub struct ConstantMedium<T> {
    boundary: Arc<T>,
    density: f64,
}
impl<T> ConstantMedium<T> {
    pub fn new(boundary: Arc<T>, density: f64) -> Self {
        ConstantMedium { boundary, density }
    }
    pub fn boundary(&self) -> &Arc<T> {
        &self.boundary
    }
    pub fn density(&self) -> f64 {
        self.density
    }
}
impl<T> Hit for ConstantMedium<T>
where
    T: Hit,
{
    fn record_geometry(&self, recorder: &mut Recorder) -> Result<i32, Error> {
        recorder.intern(self, "constant-medium", |r| {
            let boundary = self.boundary.record_geometry(r)?;
            check(unsafe { ffi::rt_add_geometry_constant_medium(r.raw(), boundary, self.density) })
        })
    }
}
impl<T, B> Bound<B> for ConstantMedium<T>
where
    T: Bound<B>,
    B: Hit,
{
    fn bound(&self) -> Option<B> {
        self.boundary.bound() // src/optimize.rs:508-516
    }
}
