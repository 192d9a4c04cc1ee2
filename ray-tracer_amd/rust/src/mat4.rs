//! `Mat4` (column-major, src/mat4.rs:5-17) with its constructors as ASSOCIATED functions, as the examples call them:
//! `Mat4::translation(v)`, `Mat4::rotation(radians, axis)`, `a.multiplied(&b)`.  The arithmetic is the library's
//! (`rt_mat4_*`: the gl-matrix formulas of src/mat4.rs:36-245 restated in C++), so a matrix built here has the bits of the
//! matrix the C++ and Python drivers build for the same scene.
use crate::ffi;
use crate::vec3::Vec3;

#[derive(Copy, Clone, Debug, PartialEq)]
pub struct Mat4 {
    data: [f64; 16],
}

impl Mat4 {
    pub fn identity() -> Self {
        let mut m = Mat4::zero();
        unsafe { ffi::rt_mat4_identity(m.data.as_mut_ptr()) };
        m
    }
    pub fn zero() -> Self {
        Mat4 { data: [0.0; 16] }
    }
    pub fn translation(offset: Vec3) -> Self {
        let mut m = Mat4::zero();
        unsafe { ffi::rt_mat4_translation(offset.to_array().as_ptr(), m.data.as_mut_ptr()) };
        m
    }
    pub fn rotation(radians: f64, axis: Vec3) -> Self {
        let mut m = Mat4::zero();
        unsafe { ffi::rt_mat4_rotation(radians, axis.to_array().as_ptr(), m.data.as_mut_ptr()) };
        m
    }
    /// self * other
    pub fn multiplied(&self, other: &Self) -> Self {
        let mut m = Mat4::zero();
        unsafe { ffi::rt_mat4_multiplied(self.data.as_ptr(), other.data.as_ptr(), m.data.as_mut_ptr()) };
        m
    }
    pub fn determinant(&self) -> f64 {
        unsafe { ffi::rt_mat4_determinant(self.data.as_ptr()) }
    }
    /// `None` iff the determinant is 0.0 (src/mat4.rs:184-190)
    pub fn inversed(&self) -> Option<Self> {
        let mut m = Mat4::zero();
        if unsafe { ffi::rt_mat4_inversed(self.data.as_ptr(), m.data.as_mut_ptr()) } == ffi::RT_OK {
            Some(m)
        } else {
            None
        }
    }
    pub fn as_slice(&self) -> &[f64] {
        &self.data
    }
    pub fn as_mut_slice(&mut self) -> &mut [f64] {
        &mut self.data
    }
}

/// A matrix with its inverse and determinant (src/mat4.rs:413-451): what `Sprite` and `TransformedGeometry` store.  The library
/// inverts the matrix itself at commit; this type exists so that `.transform(m)` keeps its `M: Into<Mat4Cached>` bound.
#[derive(Copy, Clone, Debug)]
pub struct Mat4Cached {
    origin: Mat4,
    inversed: Mat4,
    determinant: f64,
}

impl Mat4Cached {
    pub fn new(matrix: Mat4) -> Self {
        Mat4Cached { origin: matrix, inversed: matrix.inversed().unwrap_or_else(Mat4::zero), determinant: matrix.determinant() }
    }
    pub fn origin(&self) -> &Mat4 {
        &self.origin
    }
    /// `None` when the matrix is singular (src/mat4.rs:437-443)
    pub fn inversed(&self) -> Option<&Mat4> {
        if self.determinant == 0.0 {
            None
        } else {
            Some(&self.inversed)
        }
    }
    pub fn determinant(&self) -> f64 {
        self.determinant
    }
}
impl From<Mat4> for Mat4Cached {
    fn from(m: Mat4) -> Self {
        Mat4Cached::new(m)
    }
}
impl From<Mat4Cached> for Mat4 {
    fn from(m: Mat4Cached) -> Self {
        *m.origin()
    }
}
impl AsRef<Mat4> for Mat4Cached {
    fn as_ref(&self) -> &Mat4 {
        self.origin()
    }
}
