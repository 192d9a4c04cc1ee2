//! `Sprite<T, U>` and its builder (src/sprite.rs:10-92): a geometry, a material and a transform.  Generic over both, as upstream,
//! so that `Sprite::builder().geometry(Sphere::new(1.0).into()).material(Lambertian::new(c).into())` infers
//! `Sprite<Sphere, Lambertian>` through the very same `Arc<T>` / `Arc<U>` parameters.
use crate::gpu::{Error, Recorder};
use crate::mat4::{Mat4, Mat4Cached};
use crate::material::Material;
use crate::optimize::{AxisAlignedBoundingBox, Bound};
use crate::ray::Hit;

use std::sync::Arc;

#[derive(Clone, Debug)]
pub struct Sprite<T, U> {
    geometry: Option<Arc<T>>,
    material: Option<Arc<U>>,
    transform: Mat4Cached,
}

pub struct SpriteBuilder<T, U> {
    sprite: Sprite<T, U>,
}

impl<T, U> SpriteBuilder<T, U> {
    pub fn build(self) -> Sprite<T, U> {
        self.sprite
    }
    pub fn geometry(mut self, geometry: Arc<T>) -> Self {
        self.sprite.geometry = Some(geometry);
        self
    }
    pub fn material(mut self, material: Arc<U>) -> Self {
        self.sprite.material = Some(material);
        self
    }
    pub fn transform<M>(mut self, transform: M) -> Self
    where
        M: Into<Mat4Cached>,
    {
        self.sprite.transform = transform.into();
        self
    }
}

impl<T, U> Sprite<T, U> {
    pub fn builder() -> SpriteBuilder<T, U> {
        SpriteBuilder { sprite: Sprite { geometry: None, material: None, transform: Mat4::identity().into() } }
    }
    pub fn new(geometry: Option<Arc<T>>, material: Option<Arc<U>>) -> Self {
        Sprite { geometry, material, transform: Mat4::identity().into() }
    }
    pub fn geometry(&self) -> &Option<Arc<T>> {
        &self.geometry
    }
    pub fn material(&self) -> &Option<Arc<U>> {
        &self.material
    }
    pub fn transform(&self) -> &Mat4Cached {
        &self.transform
    }
}

impl<T, U> Hit for Sprite<T, U>
where
    T: Hit,
    U: Material + 'static,
{
    /// A sprite used AS a geometry (a `Sprite` is `Hit`, so `Sprite<Sprite<..>, _>` type-checks upstream, src/sprite.rs:87-93):
    /// the node of this one sprite.
    fn record_geometry(&self, recorder: &mut Recorder) -> Result<i32, Error> {
        let mut ids = Vec::new();
        self.record_member(recorder, &mut ids)?;
        crate::optimize::record_node(recorder, &ids)
    }
    /// `rt_add_sprite(geometry, material, transform)`; a shared `Arc` geometry or material is one record
    fn record_member(&self, recorder: &mut Recorder, out: &mut Vec<i32>) -> Result<(), Error> {
        let geometry = match &self.geometry {
            Some(g) => g.record_geometry(recorder)?,
            None => -1, // never hit (src/sprite.rs:95,136)
        };
        let material = match &self.material {
            Some(m) => recorder.intern(&**m, "material", |r| m.record(r))?,
            None => -1,
        };
        out.push(recorder.sprite(geometry, material, Some(self.transform.origin().as_slice()))?);
        Ok(())
    }
}

impl<T, U> Bound<AxisAlignedBoundingBox> for Sprite<T, U>
where
    T: Bound<AxisAlignedBoundingBox>,
    U: Material + 'static,
{
    fn bound(&self) -> Option<AxisAlignedBoundingBox> {
        let inner = self.geometry.as_ref()?.bound()?;
        Some(inner.transformed(self.transform.origin())) // the 8 corners through M, src/optimize.rs:128-186
    }
}
