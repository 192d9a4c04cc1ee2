//! `PerspectiveCamera::new(eye, center, up, fov, aspect, focusDistance, lensRadius)` (src/camera.rs:25-59): the same seven
//! arguments, fov in radians.  The frame (lower-left corner, horizontal, vertical -- quirks Q1 / Q2 of the reference included) is
//! computed by the library (`rt_camera_perspective`), the rays by the kernels (`rtl::camera_ray`).
use crate::ffi;
use crate::vec3::Vec3;

#[derive(Clone, Debug)]
pub struct PerspectiveCamera {
    eye: Vec3,
    center: Vec3,
    up: Vec3,
    fov: f64,
    aspect: f64,
    focusDistance: f64,
    lensRadius: f64,
    raw: ffi::rt_camera,
}

impl PerspectiveCamera {
    pub fn new(eye: Vec3, center: Vec3, up: Vec3, fov: f64, aspect: f64, focusDistance: f64, lensRadius: f64) -> Self {
        let mut raw = ffi::rt_camera::default();
        let (e, c, u) = (eye.to_array(), center.to_array(), up.to_array());
        // (fails only on null pointers)
        unsafe { ffi::rt_camera_perspective(&mut raw, e.as_ptr(), c.as_ptr(), u.as_ptr(), fov, aspect, focusDistance, lensRadius) };
        PerspectiveCamera { eye, center, up, fov, aspect, focusDistance, lensRadius, raw }
    }
    pub fn eye(&self) -> &Vec3 {
        &self.eye
    }
    pub fn center(&self) -> &Vec3 {
        &self.center
    }
    pub fn up(&self) -> &Vec3 {
        &self.up
    }
    pub fn fov(&self) -> f64 {
        self.fov
    }
    pub fn aspect(&self) -> f64 {
        self.aspect
    }
    pub fn focusDistance(&self) -> f64 {
        self.focusDistance
    }
    pub fn lensRadius(&self) -> f64 {
        self.lensRadius
    }
    pub(crate) fn raw(&self) -> &ffi::rt_camera {
        &self.raw
    }
}
