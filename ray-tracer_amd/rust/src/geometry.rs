//! `Sphere`, `Rectangle`, `TransformedGeometry<T>`, `Cube` with the reference's constructors (src/geometry.rs:12-286).
//! `Sphere::new(r).into()` is the standard library's `From<T> for Arc<T>`, as upstream: this crate adds no conversion of its own
//! (a second `Into<Arc<_>>` would make the `.into()` of `Sprite::builder().geometry(Sphere::new(1.0).into())` ambiguous).
use crate::ffi;
use crate::gpu::{check, Error, Recorder};
use crate::mat4::{Mat4, Mat4Cached};
use crate::optimize::{AxisAlignedBoundingBox, Bound};
use crate::ray::Hit;
use crate::vec3::Vec3;

#[derive(Clone, Debug)]
pub struct Sphere {
    radius: f64,
}
impl Sphere {
    /// centred at the origin of its own frame; a `Sprite`'s transform places it (src/geometry.rs:17-19)
    pub fn new(radius: f64) -> Self {
        Sphere { radius }
    }
    pub fn radius(&self) -> f64 {
        self.radius
    }
}
impl Hit for Sphere {
    fn record_geometry(&self, recorder: &mut Recorder) -> Result<i32, Error> {
        recorder.intern(self, "sphere", |r| check(unsafe { ffi::rt_add_geometry_sphere(r.raw(), self.radius) }))
    }
}
impl Bound<AxisAlignedBoundingBox> for Sphere {
    fn bound(&self) -> Option<AxisAlignedBoundingBox> {
        let r = Vec3::new(self.radius, self.radius, self.radius);
        Some(AxisAlignedBoundingBox::new(-r, r)) // src/optimize.rs:104-113
    }
}

/// in the plane z = 0 of its own frame, centred, normal +z (src/geometry.rs:127-180)
#[derive(Clone, Debug)]
pub struct Rectangle {
    width: f64,
    height: f64,
}
impl Rectangle {
    pub fn new(width: f64, height: f64) -> Self {
        Rectangle { width, height }
    }
    pub fn width(&self) -> f64 {
        self.width
    }
    pub fn height(&self) -> f64 {
        self.height
    }
}
impl Hit for Rectangle {
    fn record_geometry(&self, recorder: &mut Recorder) -> Result<i32, Error> {
        recorder.intern(self, "rectangle", |r| check(unsafe { ffi::rt_add_geometry_rectangle(r.raw(), self.width, self.height) }))
    }
}
impl Bound<AxisAlignedBoundingBox> for Rectangle {
    fn bound(&self) -> Option<AxisAlignedBoundingBox> {
        Some(AxisAlignedBoundingBox::new(
            Vec3::new(-self.width / 2.0, -self.height / 2.0, -1e-6),
            Vec3::new(self.width / 2.0, self.height / 2.0, 1e-6), // src/optimize.rs:115-126
        ))
    }
}

/// a geometry seen through a matrix, without a material (src/geometry.rs:185-246)
#[derive(Clone, Debug)]
pub struct TransformedGeometry<T> {
    geometry: T,
    transform: Mat4Cached,
    cube: Option<(f64, f64, f64, usize)>, // made by Cube::new: (width, height, depth, face)
}
impl<T> TransformedGeometry<T> {
    pub fn new<M>(geometry: T, transform: M) -> Self
    where
        M: Into<Mat4Cached>,
    {
        TransformedGeometry { geometry, transform: transform.into(), cube: None }
    }
    pub fn geometry(&self) -> &T {
        &self.geometry
    }
    pub fn transform(&self) -> &Mat4Cached {
        &self.transform
    }
}
impl<T> Hit for TransformedGeometry<T>
where
    T: Hit,
{
    fn record_geometry(&self, recorder: &mut Recorder) -> Result<i32, Error> {
        recorder.intern(self, "transformed", |r| {
            let inner = self.geometry.record_geometry(r)?;
            check(unsafe { ffi::rt_add_geometry_transformed(r.raw(), inner, self.transform.origin().as_slice().as_ptr()) })
        })
    }
    /// In a node's list -- the six faces of a `Cube` are the one use upstream -- a `TransformedGeometry` is the sprite with the same
    /// matrix and no material: `TransformedGeometry::hit` (src/geometry.rs:210-246) and `Sprite::hit` (src/sprite.rs:94-138)
    /// perform the same arithmetic, and both are children of the node, whose walk tests their 8-corner boxes.
    fn record_member(&self, recorder: &mut Recorder, out: &mut Vec<i32>) -> Result<(), Error> {
        let inner = self.geometry.record_geometry(recorder)?;
        out.push(recorder.sprite(inner, -1, Some(self.transform.origin().as_slice()))?);
        Ok(())
    }
    fn cube_face(&self) -> Option<(f64, f64, f64, usize)> {
        self.cube
    }
}
impl<T> Bound<AxisAlignedBoundingBox> for TransformedGeometry<T>
where
    T: Bound<AxisAlignedBoundingBox>,
{
    fn bound(&self) -> Option<AxisAlignedBoundingBox> {
        self.geometry.bound().map(|b| b.transformed(self.transform.origin())) // src/optimize.rs:188-241
    }
}

#[derive(Clone, Debug)]
pub struct Cube;
impl Cube {
    /// Six rectangles, outward normals: front, left, back, right, top, bottom (src/geometry.rs:254-286).  The examples wrap them
    /// into a node, `BoundingVolumeHierarchyNode::new(Cube::new(w, h, d).into_iter().map(..).collect())`, and that node is
    /// recorded as ONE `rt_add_geometry_cube(w, h, d)` (each face remembers where it comes from).
    pub fn new(width: f64, height: f64, depth: f64) -> Vec<TransformedGeometry<Rectangle>> {
        let quarter = |degrees: f64, axis: Vec3| Mat4::rotation(degrees.to_radians(), axis);
        let faces = vec![
            (Rectangle::new(width, height), Mat4::translation(Vec3::new(0.0, 0.0, depth / 2.0))),
            (Rectangle::new(depth, height), Mat4::translation(Vec3::new(-width / 2.0, 0.0, 0.0)).multiplied(&quarter(-90.0, Vec3::ey()))),
            (Rectangle::new(width, height), Mat4::translation(Vec3::new(0.0, 0.0, -depth / 2.0)).multiplied(&quarter(180.0, Vec3::ey()))),
            (Rectangle::new(depth, height), Mat4::translation(Vec3::new(width / 2.0, 0.0, 0.0)).multiplied(&quarter(90.0, Vec3::ey()))),
            (Rectangle::new(width, depth), Mat4::translation(Vec3::new(0.0, height / 2.0, 0.0)).multiplied(&quarter(-90.0, Vec3::ex()))),
            (Rectangle::new(width, depth), Mat4::translation(Vec3::new(0.0, -height / 2.0, 0.0)).multiplied(&quarter(90.0, Vec3::ex()))),
        ];
        faces
            .into_iter()
            .enumerate()
            .map(|(face, (rectangle, matrix))| {
                let mut t = TransformedGeometry::new(rectangle, matrix);
                t.cube = Some((width, height, depth, face));
                t
            })
            .collect()
    }
}
