//! Raw declarations, one to one with include/rt_mi355x.h (every entry point; tests/test_host_logic.py diffs the two
//! symbol lists).  NOT COMPILED in this repository's build image (no Rust toolchain there); see README.md.
#![allow(non_camel_case_types)]
use std::os::raw::{c_char, c_double, c_float, c_int, c_uint, c_void};

#[repr(C)]
pub struct rt_scene {
    _private: [u8; 0],
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct rt_camera {
    pub eye: [c_double; 3],
    pub lower_left: [c_double; 3],
    pub horizontal: [c_double; 3],
    pub vertical: [c_double; 3],
    pub lens_radius: c_double,
}

#[repr(C)]
#[derive(Clone, Copy, Debug)]
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

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct rt_counters {
    pub samples: u64,
    pub segments: u64,
    pub nodes_visited: u64,
    pub prims_tested: u64,
    pub rng_draws: u64,
    pub node_wave: u64,
    pub node_lane: u64,
    pub leaf_wave: u64,
    pub leaf_lane: u64,
    pub shade_wave: u64,
    pub shade_lane: u64,
    pub node_cycles: u64,
    pub leaf_cycles: u64,
    pub shade_cycles: u64,
    pub finish_cycles: u64,
    pub refill_cycles: u64,
    pub begin_cycles: u64,
    pub swap_class_mode: u64,
    pub swap_new_mode: u64,
    pub swap_parked: u64,
    pub swap_pulled: u64,
    pub swap_lock_busy: u64,
    pub swap_scattered: u64,
    pub swap_off_class: u64,
    pub swap_cycles: u64,
    pub node_idle_done: u64,
    pub node_idle_leaf: u64,
    pub node_idle_empty: u64,
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct rt_launch_config {
    pub blocks: c_int,
    pub block_threads: c_int,
    pub lds_bytes: c_uint,
    pub blocks_per_cu: c_int,
    pub n_cu: c_int,
    pub passes: c_int,
    pub n_jobs: c_int,
    pub job_spp: c_int,
    pub kernel_features: c_uint,
    pub lds_nodes: c_int,
    pub swap: c_int,
    pub workspace_bytes: usize,
    pub swap_cap: c_int,
    pub waves_per_simd: c_int,
    pub tile_order: c_int,
    pub records_in_lds: c_int,
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct rt_scene_info {
    pub n_prims: c_int,
    pub n_child_prims: c_int,
    pub n_hoisted: c_int,
    pub n_nodes: c_int,
    pub max_depth: c_int,
    pub n_materials: c_int,
    pub n_textures: c_int,
    pub n_xforms: c_int,
    pub node_bytes: c_int,
    pub prim_bytes: c_int,
    pub material_bytes: c_int,
    pub feature_mask: c_uint,
    pub device_bytes: usize,
    pub n_list: c_int,
}

pub const RT_OK: c_int = 0;
pub const RT_ERR_INVALID: c_int = -1;
pub const RT_ERR_EMPTY: c_int = -2;
pub const RT_ERR_DEVICE: c_int = -3;
pub const RT_ERR_UNSUPPORTED: c_int = -4;
pub const RT_ERR_STATE: c_int = -5;
pub const RT_TILE: c_int = 8;
pub const RT_FLAG_COUNTERS: c_uint = 1;
pub const RT_FLAG_DEFERRED_OUTPUT: c_uint = 2;
pub const RT_FLAG_ASCENDING_TILES: c_uint = 4;
pub const RT_TILE_ORDER_ASCENDING: c_int = 0;
pub const RT_TILE_ORDER_LEARNT: c_int = 1;
pub const RT_TILE_ORDER_LEARNING: c_int = 2;

extern "C" {
    pub fn rt_last_error() -> *const c_char;
    pub fn rt_version() -> *const c_char;
    pub fn rt_device_count() -> c_int;

    pub fn rt_mat4_identity(out: *mut c_double);
    pub fn rt_mat4_translation(offset: *const c_double, out: *mut c_double);
    pub fn rt_mat4_rotation(radians: c_double, axis: *const c_double, out: *mut c_double);
    pub fn rt_mat4_multiplied(a: *const c_double, b: *const c_double, out: *mut c_double);
    pub fn rt_mat4_determinant(m: *const c_double) -> c_double;
    pub fn rt_mat4_inversed(m: *const c_double, out: *mut c_double) -> c_int;

    pub fn rt_scene_create() -> *mut rt_scene;
    pub fn rt_scene_destroy(s: *mut rt_scene);
    pub fn rt_scene_clone(s: *const rt_scene, device: c_int) -> *mut rt_scene;

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
    pub fn rt_add_geometry_transformed(s: *mut rt_scene, geometry: c_int, m: *const c_double) -> c_int;
    pub fn rt_add_geometry_bvh(s: *mut rt_scene, sprites: *const c_int, n: c_int) -> c_int;
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
        counters: *mut rt_counters,
    ) -> c_int;
    pub fn rt_render_sharded(
        scenes: *const *mut rt_scene,
        n_scenes: c_int,
        cam: *const rt_camera,
        p: *const rt_render_params,
        out_rgb: *mut c_double,
    ) -> c_int;
    pub fn rt_render_progressive(
        s: *mut rt_scene,
        cam: *const rt_camera,
        p: *const rt_render_params,
        s_begin: c_int,
        s_end: c_int,
        sums: *mut c_double,
    ) -> c_int;
    pub fn rt_shard_tile_count(width: c_int, height: c_int, shard_index: c_int, shard_count: c_int) -> c_int;
    pub fn rt_render_tiles_device(
        s: *mut rt_scene,
        cam: *const rt_camera,
        p: *const rt_render_params,
        d_tiles_out: *mut c_void,
        d_counters: *mut c_void,
        stream: *mut c_void,
    ) -> c_int;
    pub fn rt_unpack_tiles_device(
        d_gathered: *const c_void,
        tiles_per_shard_padded: c_int,
        shard_count: c_int,
        width: c_int,
        height: c_int,
        d_image_out: *mut c_void,
        stream: *mut c_void,
    ) -> c_int;
    pub fn rt_pack_tiles_host(
        image: *const c_double,
        width: c_int,
        height: c_int,
        shard_index: c_int,
        shard_count: c_int,
        tiles_padded: c_int,
        tiles_out: *mut c_double,
    ) -> c_int;
    pub fn rt_unpack_tiles_host(
        gathered: *const c_double,
        tiles_per_shard_padded: c_int,
        shard_count: c_int,
        width: c_int,
        height: c_int,
        image_out: *mut c_double,
    ) -> c_int;
    pub fn rt_render_status(s: *mut rt_scene) -> c_int;
    pub fn rt_render_wait_output(s: *mut rt_scene, stream: *mut c_void) -> c_int;
    pub fn rt_scene_set_workspace_limit(s: *mut rt_scene, bytes: usize) -> c_int;
    pub fn rt_scene_trim(s: *mut rt_scene) -> c_int;
    pub fn rt_scene_workspace_bytes(s: *const rt_scene) -> usize;
    pub fn rt_last_kernel_ms(s: *mut rt_scene, ms: *mut c_float) -> c_int;
    pub fn rt_last_launch_config(s: *mut rt_scene, out: *mut rt_launch_config) -> c_int;
    pub fn rt_scene_plan_launch(s: *const rt_scene, out: *mut rt_launch_config) -> c_int;
    pub fn rt_scene_tile_order(s: *mut rt_scene, order_out: *mut u32, cost_out: *mut u64, capacity: c_int) -> c_int;

    pub fn rt_tonemap_rgb8(rgb: *const c_double, n_pixels: usize, out_rgb8: *mut u8);
    pub fn rt_write_ppm_p3(path: *const c_char, rgb: *const c_double, w: c_int, h: c_int) -> c_int;
    pub fn rt_tonemap_png8(rgb: *const c_double, n_pixels: usize, out_rgb8: *mut u8);
    pub fn rt_write_png_rgba8(path: *const c_char, rgb: *const c_double, w: c_int, h: c_int) -> c_int; // examples/main.rs:105-135

    pub fn rt_scene_get_info(s: *const rt_scene, out: *mut rt_scene_info) -> c_int;
    pub fn rt_scene_copy_nodes(s: *const rt_scene, out: *mut c_double, max_nodes: c_int) -> c_int;
    pub fn rt_scene_hash(s: *const rt_scene, out: *mut u64) -> c_int;
    pub fn rt_scene_prim_bounds(s: *const rt_scene, prim: c_int, out: *mut c_double) -> c_int;
    pub fn rt_scene_prim_group(s: *const rt_scene, prim: c_int) -> c_int;
    pub fn rt_probe_device_math(
        device: c_int,
        a: *const c_double,
        b: *const c_double,
        n: c_int,
        out_sqrt_a: *mut c_double,
        out_a_div_b: *mut c_double,
    ) -> c_int;
    pub fn rt_probe_device_libm(
        device: c_int,
        which: c_int,
        a: *const c_double,
        b: *const c_double,
        n: c_int,
        out: *mut c_double,
    ) -> c_int;
}
