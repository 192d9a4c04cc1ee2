//! Raw declarations, one to one with include/rt_mi355x.h.
#![allow(non_camel_case_types)]
use std::os::raw::{c_char, c_double, c_int, c_uint};

#[repr(C)]
pub struct rt_scene {
    _private: [u8; 0],
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct rt_camera {
    pub eye: [c_double; 3],
    pub lower_left: [c_double; 3],
    pub horizontal: [c_double; 3],
    pub vertical: [c_double; 3],
    pub lens_radius: c_double,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct rt_render_params {
    pub width: c_int,
    pub height: c_int,
    pub spp: c_int,
    pub max_depth: c_int,
    pub seed: u64,
    pub shard_index: c_int,
    pub shard_count: c_int,
    pub flags: c_uint,
}

pub const RT_OK: c_int = 0;
pub const RT_ERR_EMPTY: c_int = -2;

extern "C" {
    pub fn rt_last_error() -> *const c_char;
    pub fn rt_device_count() -> c_int;
    pub fn rt_scene_create() -> *mut rt_scene;
    pub fn rt_scene_destroy(s: *mut rt_scene);
    pub fn rt_add_texture_solid(s: *mut rt_scene, rgb: *const c_double) -> c_int;
    pub fn rt_add_texture_checker(s: *mut rt_scene, black: c_int, white: c_int) -> c_int;
    pub fn rt_add_texture_image_rgb8(s: *mut rt_scene, rgb: *const u8, w: c_int, h: c_int) -> c_int;
    pub fn rt_add_material_lambertian(s: *mut rt_scene, tex: c_int) -> c_int;
    pub fn rt_add_material_metal(s: *mut rt_scene, tex: c_int, fuzziness: c_double) -> c_int;
    pub fn rt_add_material_dielectric(s: *mut rt_scene, refractive: c_double) -> c_int;
    pub fn rt_add_material_diffuse_light(s: *mut rt_scene, tex: c_int) -> c_int;
    pub fn rt_add_material_isotropic(s: *mut rt_scene, tex: c_int) -> c_int;
    pub fn rt_add_geometry_sphere(s: *mut rt_scene, radius: c_double) -> c_int;
    pub fn rt_add_geometry_rectangle(s: *mut rt_scene, w: c_double, h: c_double) -> c_int;
    pub fn rt_add_geometry_cube(s: *mut rt_scene, w: c_double, h: c_double, d: c_double) -> c_int;
    pub fn rt_add_geometry_constant_medium(s: *mut rt_scene, boundary: c_int, density: c_double) -> c_int;
    pub fn rt_add_sprite(s: *mut rt_scene, geometry: c_int, material: c_int, m: *const c_double) -> c_int;
    pub fn rt_scene_commit(s: *mut rt_scene, device: c_int) -> c_int;
    pub fn rt_camera_perspective(
        out: *mut rt_camera,
        eye: *const c_double,
        center: *const c_double,
        up: *const c_double,
        fov: c_double,
        aspect: c_double,
        focus_distance: c_double,
        lens_radius: c_double,
    ) -> c_int;
    pub fn rt_render(
        s: *mut rt_scene,
        cam: *const rt_camera,
        p: *const rt_render_params,
        out_rgb: *mut c_double,
        counters: *mut std::ffi::c_void,
    ) -> c_int;
    pub fn rt_write_ppm_p3(path: *const c_char, rgb: *const c_double, w: c_int, h: c_int) -> c_int;
    pub fn rt_write_png_rgba8(path: *const c_char, rgb: *const c_double, w: c_int, h: c_int) -> c_int; // examples/main.rs:105-135
}
