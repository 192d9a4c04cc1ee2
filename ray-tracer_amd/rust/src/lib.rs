//! Safe wrapper over librt_mi355x that keeps the vocabulary of aiifabbf/ray-tracer:
//! `Sprite::builder().geometry(..).material(..).transform(..).build()`,
//! `BoundingVolumeHierarchyNode::new(sprites) -> Option<_>`, `PerspectiveCamera::new(..)`.
//! The per-pixel sampling loop of the example drivers (thread::spawn + mpsc + color())
//! becomes one `world.render(..)` call executed on the MI355X.
//!
//! NOT COMPILED in this repository's build image (no Rust toolchain there); see README.md.
#![allow(non_snake_case)]

pub mod ffi;

use std::collections::HashMap;
use std::ffi::{CStr, CString};
use std::sync::Arc;

#[derive(Debug)]
pub struct Error {
    pub code: i32,
    pub message: String,
}

fn check(rc: i32) -> Result<i32, Error> {
    if rc < 0 {
        let message = unsafe { CStr::from_ptr(ffi::rt_last_error()) }.to_string_lossy().into_owned();
        Err(Error { code: rc, message })
    } else {
        Ok(rc)
    }
}

#[derive(Clone, Copy, Debug, PartialEq)]
pub struct Vec3 {
    pub x: f64,
    pub y: f64,
    pub z: f64,
}
impl Vec3 {
    pub fn new(x: f64, y: f64, z: f64) -> Self {
        Self { x, y, z }
    }
    fn arr(&self) -> [f64; 3] {
        [self.x, self.y, self.z]
    }
}

/// column-major 4x4, same layout as the reference's `Mat4` (src/mat4.rs:5-17)
pub type Mat4 = [f64; 16];

/// Mat4::translation / rotation / multiplied / inversed of src/mat4.rs, evaluated by the library (same rounding as the
/// matrices the kernels use)
pub fn mat4_translation(t: Vec3) -> Mat4 {
    let mut m = [0.0; 16];
    unsafe { ffi::rt_mat4_translation(t.arr().as_ptr(), m.as_mut_ptr()) };
    m
}
pub fn mat4_rotation(radians: f64, axis: Vec3) -> Mat4 {
    let mut m = [0.0; 16];
    unsafe { ffi::rt_mat4_rotation(radians, axis.arr().as_ptr(), m.as_mut_ptr()) };
    m
}
pub fn mat4_multiplied(a: &Mat4, b: &Mat4) -> Mat4 {
    let mut m = [0.0; 16];
    unsafe { ffi::rt_mat4_multiplied(a.as_ptr(), b.as_ptr(), m.as_mut_ptr()) };
    m
}
pub fn mat4_inversed(a: &Mat4) -> Option<Mat4> {
    let mut m = [0.0; 16];
    if unsafe { ffi::rt_mat4_inversed(a.as_ptr(), m.as_mut_ptr()) } == ffi::RT_OK {
        Some(m)
    } else {
        None
    }
}

pub enum Texture {
    SolidColor(Vec3),
    Checker(Arc<Texture>, Arc<Texture>),
    /// the reference's ImageTexture closure is the nearest-texel lookup of an RGB8 image (examples/main.rs:267-280)
    ImageRgb8 { data: Vec<u8>, width: u32, height: u32 },
}
impl From<Vec3> for Arc<Texture> {
    fn from(c: Vec3) -> Self {
        Arc::new(Texture::SolidColor(c))
    }
}

pub enum Material {
    Lambertian(Arc<Texture>),
    Metal(Arc<Texture>, f64),
    Dielectric(f64),
    DiffuseLight(Arc<Texture>),
    Isotropic(Arc<Texture>),
}

pub enum Geometry {
    Sphere(f64),
    Rectangle(f64, f64),
    /// BoundingVolumeHierarchyNode::new(Cube::new(w, h, d)) as the examples wrap it
    Cube(f64, f64, f64),
    /// ConstantMedium::new(boundary, density): any boundary but another medium (src/volume.rs:18-44)
    ConstantMedium(Arc<Geometry>, f64),
    /// TransformedGeometry::new(geometry, M) (src/geometry.rs:185-246)
    Transformed(Arc<Geometry>, Mat4),
    /// BoundingVolumeHierarchyNode::new(sprites) used as a sprite's geometry: instancing (src/sprite.rs:87-93)
    Node(Vec<Sprite>),
}

pub struct Sprite {
    geometry: Option<Arc<Geometry>>,
    material: Option<Arc<Material>>,
    transform: Mat4,
}
pub struct SpriteBuilder {
    sprite: Sprite,
}
impl Sprite {
    pub fn builder() -> SpriteBuilder {
        let mut identity = [0.0; 16];
        identity[0] = 1.0;
        identity[5] = 1.0;
        identity[10] = 1.0;
        identity[15] = 1.0;
        SpriteBuilder { sprite: Sprite { geometry: None, material: None, transform: identity } }
    }
}
impl SpriteBuilder {
    pub fn geometry(mut self, g: Arc<Geometry>) -> Self {
        self.sprite.geometry = Some(g);
        self
    }
    pub fn material(mut self, m: Arc<Material>) -> Self {
        self.sprite.material = Some(m);
        self
    }
    pub fn transform(mut self, m: Mat4) -> Self {
        self.sprite.transform = m;
        self
    }
    pub fn build(self) -> Sprite {
        self.sprite
    }
}

pub struct PerspectiveCamera {
    raw: ffi::rt_camera,
}
impl PerspectiveCamera {
    /// same seven arguments as src/camera.rs:25-33 (fov in radians)
    pub fn new(eye: Vec3, center: Vec3, up: Vec3, fov: f64, aspect: f64, focusDistance: f64, lensRadius: f64) -> Self {
        let mut raw = ffi::rt_camera::default();
        let (e, c, u) = (eye.arr(), center.arr(), up.arr());
        unsafe {
            ffi::rt_camera_perspective(&mut raw, e.as_ptr(), c.as_ptr(), u.as_ptr(), fov, aspect, focusDistance, lensRadius);
        }
        Self { raw }
    }
}

/// The committed world.  `new` mirrors BoundingVolumeHierarchyNode::new: None for an empty list.
pub struct BoundingVolumeHierarchyNode {
    raw: *mut ffi::rt_scene,
}
unsafe impl Send for BoundingVolumeHierarchyNode {}
unsafe impl Sync for BoundingVolumeHierarchyNode {}

struct Recorder {
    raw: *mut ffi::rt_scene,
    seen: HashMap<usize, i32>, // Arc pointer identity -> id: a shared Arc is one record
}
impl Recorder {
    fn texture(&mut self, t: &Arc<Texture>) -> Result<i32, Error> {
        let key = Arc::as_ptr(t) as usize;
        if let Some(id) = self.seen.get(&key) {
            return Ok(*id);
        }
        let id = match t.as_ref() {
            Texture::SolidColor(c) => check(unsafe { ffi::rt_add_texture_solid(self.raw, c.arr().as_ptr()) })?,
            Texture::Checker(b, w) => {
                let (b, w) = (self.texture(b)?, self.texture(w)?);
                check(unsafe { ffi::rt_add_texture_checker(self.raw, b, w) })?
            }
            Texture::ImageRgb8 { data, width, height } => {
                check(unsafe { ffi::rt_add_texture_image_rgb8(self.raw, data.as_ptr(), *width as i32, *height as i32) })?
            }
        };
        self.seen.insert(key, id);
        Ok(id)
    }
    fn material(&mut self, m: &Arc<Material>) -> Result<i32, Error> {
        let key = Arc::as_ptr(m) as usize;
        if let Some(id) = self.seen.get(&key) {
            return Ok(*id);
        }
        let id = match m.as_ref() {
            Material::Lambertian(t) => {
                let t = self.texture(t)?;
                check(unsafe { ffi::rt_add_material_lambertian(self.raw, t) })?
            }
            Material::Metal(t, f) => {
                let t = self.texture(t)?;
                check(unsafe { ffi::rt_add_material_metal(self.raw, t, *f) })?
            }
            Material::Dielectric(r) => check(unsafe { ffi::rt_add_material_dielectric(self.raw, *r) })?,
            Material::DiffuseLight(t) => {
                let t = self.texture(t)?;
                check(unsafe { ffi::rt_add_material_diffuse_light(self.raw, t) })?
            }
            Material::Isotropic(t) => {
                let t = self.texture(t)?;
                check(unsafe { ffi::rt_add_material_isotropic(self.raw, t) })?
            }
        };
        self.seen.insert(key, id);
        Ok(id)
    }
    fn sprite(&mut self, s: &Sprite) -> Result<i32, Error> {
        let g = match &s.geometry {
            Some(g) => self.geometry(g)?,
            None => -1,
        };
        let m = match &s.material {
            Some(m) => self.material(m)?,
            None => -1,
        };
        check(unsafe { ffi::rt_add_sprite(self.raw, g, m, s.transform.as_ptr()) })
    }
    fn geometry(&mut self, g: &Arc<Geometry>) -> Result<i32, Error> {
        let key = Arc::as_ptr(g) as usize;
        if let Some(id) = self.seen.get(&key) {
            return Ok(*id);
        }
        let id = match g.as_ref() {
            Geometry::Sphere(r) => check(unsafe { ffi::rt_add_geometry_sphere(self.raw, *r) })?,
            Geometry::Rectangle(w, h) => check(unsafe { ffi::rt_add_geometry_rectangle(self.raw, *w, *h) })?,
            Geometry::Cube(w, h, d) => check(unsafe { ffi::rt_add_geometry_cube(self.raw, *w, *h, *d) })?,
            Geometry::ConstantMedium(b, density) => {
                let b = self.geometry(b)?;
                check(unsafe { ffi::rt_add_geometry_constant_medium(self.raw, b, *density) })?
            }
            Geometry::Transformed(inner, m) => {
                let inner = self.geometry(inner)?;
                check(unsafe { ffi::rt_add_geometry_transformed(self.raw, inner, m.as_ptr()) })?
            }
            Geometry::Node(children) => {
                let mut ids = Vec::with_capacity(children.len());
                for c in children {
                    ids.push(self.sprite(c)?);
                }
                check(unsafe { ffi::rt_add_geometry_bvh(self.raw, ids.as_ptr(), ids.len() as i32) })?
            }
        };
        self.seen.insert(key, id);
        Ok(id)
    }
}

impl BoundingVolumeHierarchyNode {
    /// Records every sprite and commits on HIP device `device`.
    /// `Ok(None)` for an empty list, exactly like the reference (src/optimize.rs:367-370).
    pub fn new(objects: &[Sprite], device: i32) -> Result<Option<Self>, Error> {
        let raw = unsafe { ffi::rt_scene_create() };
        let world = Self { raw }; // dropped (destroyed) on every early return
        let mut rec = Recorder { raw, seen: HashMap::new() };
        for s in objects {
            rec.sprite(s)?;
        }
        let rc = unsafe { ffi::rt_scene_commit(raw, device) };
        if rc == ffi::RT_ERR_EMPTY {
            return Ok(None);
        }
        check(rc)?;
        Ok(Some(world))
    }

    /// The whole loop of examples/book-one.rs:56-88.  Returns `buffer[y * width + x]`, y up.
    pub fn render(
        &self,
        camera: &PerspectiveCamera,
        width: usize,
        height: usize,
        subPixelSampleCount: usize,
        maxDepth: usize,
        seed: u64,
    ) -> Result<Vec<Vec3>, Error> {
        let mut rgb = vec![0.0f64; width * height * 3];
        let p = ffi::rt_render_params {
            width: width as i32,
            height: height as i32,
            spp: subPixelSampleCount as i32,
            max_depth: maxDepth as i32,
            seed,
            shard_index: 0,
            shard_count: 1,
            flags: 0,
        };
        check(unsafe { ffi::rt_render(self.raw, &camera.raw, &p, rgb.as_mut_ptr(), std::ptr::null_mut()) })?;
        Ok(rgb.chunks(3).map(|c| Vec3::new(c[0], c[1], c[2])).collect())
    }
}
impl BoundingVolumeHierarchyNode {
    /// A second committed copy on another device: what `Arc::clone(&world)` hands every worker thread upstream
    /// (examples/book-one.rs:57-59), one per GPU here.
    pub fn clone_to(&self, device: i32) -> Result<Self, Error> {
        let raw = unsafe { ffi::rt_scene_clone(self.raw, device) };
        if raw.is_null() {
            check(ffi::RT_ERR_DEVICE)?;
        }
        Ok(Self { raw })
    }

    /// Wait for the renders launched on this world and report a device error word (a kernel that refused to run or made
    /// no progress) as `Err` instead of a silent wrong image (`rt_render_status`).
    pub fn status(&self) -> Result<(), Error> {
        check(unsafe { ffi::rt_render_status(self.raw) })?;
        Ok(())
    }

    /// Limit (bytes, 0 = default) and release of the per-sample workspace a render keeps on the device
    /// (`rt_scene_set_workspace_limit`, `rt_scene_trim`); larger renders run in several passes, bit-identically.
    pub fn set_workspace_limit(&self, bytes: usize) -> Result<(), Error> {
        check(unsafe { ffi::rt_scene_set_workspace_limit(self.raw, bytes) })?;
        Ok(())
    }
    pub fn trim(&self) -> Result<(), Error> {
        check(unsafe { ffi::rt_scene_trim(self.raw) })?;
        Ok(())
    }
    pub fn workspace_bytes(&self) -> usize {
        unsafe { ffi::rt_scene_workspace_bytes(self.raw) }
    }

    /// The thread fan-out + mpsc gather of examples/book-one.rs:52-88 across GPUs: `worlds[i]` is a committed copy on
    /// its own device, tiles are dealt `tile_id % worlds.len()`, one host thread per copy inside the library.
    /// Bit-identical to `render` for any number of copies.
    pub fn render_sharded(
        worlds: &[&BoundingVolumeHierarchyNode],
        camera: &PerspectiveCamera,
        width: usize,
        height: usize,
        subPixelSampleCount: usize,
        maxDepth: usize,
        seed: u64,
    ) -> Result<Vec<Vec3>, Error> {
        let mut rgb = vec![0.0f64; width * height * 3];
        let raws: Vec<*mut ffi::rt_scene> = worlds.iter().map(|w| w.raw).collect();
        let p = ffi::rt_render_params {
            width: width as i32,
            height: height as i32,
            spp: subPixelSampleCount as i32,
            max_depth: maxDepth as i32,
            seed,
            shard_index: 0,
            shard_count: 1,
            flags: 0,
        };
        check(unsafe { ffi::rt_render_sharded(raws.as_ptr(), raws.len() as i32, &camera.raw, &p, rgb.as_mut_ptr()) })?;
        Ok(rgb.chunks(3).map(|c| Vec3::new(c[0], c[1], c[2])).collect())
    }

    /// Progressive form of `render`: continues the raw per-pixel sums (`width * height * 3`, y up) with samples
    /// `[s_begin, s_end)` of the `subPixelSampleCount`-sample render; divide by `subPixelSampleCount` after the last range.
    /// Save `sums` and `s_end` (and `scene_hash()`) to checkpoint; the result is bit-identical to one `render` call.
    pub fn render_progressive(
        &self,
        camera: &PerspectiveCamera,
        width: usize,
        height: usize,
        subPixelSampleCount: usize,
        maxDepth: usize,
        seed: u64,
        s_begin: usize,
        s_end: usize,
        sums: &mut [f64],
    ) -> Result<(), Error> {
        assert_eq!(sums.len(), width * height * 3);
        let p = ffi::rt_render_params {
            width: width as i32,
            height: height as i32,
            spp: subPixelSampleCount as i32,
            max_depth: maxDepth as i32,
            seed,
            shard_index: 0,
            shard_count: 1,
            flags: 0,
        };
        check(unsafe { ffi::rt_render_progressive(self.raw, &camera.raw, &p, s_begin as i32, s_end as i32, sums.as_mut_ptr()) })?;
        Ok(())
    }

    /// identifies the committed scene (checkpoints)
    pub fn scene_hash(&self) -> Result<u64, Error> {
        let mut h = 0u64;
        check(unsafe { ffi::rt_scene_hash(self.raw, &mut h) })?;
        Ok(h)
    }
}
impl Drop for BoundingVolumeHierarchyNode {
    fn drop(&mut self) {
        unsafe { ffi::rt_scene_destroy(self.raw) }
    }
}

/// P3 text exactly as examples/book-one.rs:28-30,90-100 prints it
pub fn write_ppm(path: &str, buffer: &[Vec3], width: usize, height: usize) -> Result<(), Error> {
    let rgb: Vec<f64> = buffer.iter().flat_map(|v| vec![v.x, v.y, v.z]).collect();
    let c = CString::new(path).unwrap();
    check(unsafe { ffi::rt_write_ppm_p3(c.as_ptr(), rgb.as_ptr(), width as i32, height as i32) })?;
    Ok(())
}

/// RGBA8 PNG exactly as examples/main.rs:105-135 builds it with the `image` crate
pub fn write_png(path: &str, buffer: &[Vec3], width: usize, height: usize) -> Result<(), Error> {
    let rgb: Vec<f64> = buffer.iter().flat_map(|v| vec![v.x, v.y, v.z]).collect();
    let c = CString::new(path).unwrap();
    check(unsafe { ffi::rt_write_png_rgba8(c.as_ptr(), rgb.as_ptr(), width as i32, height as i32) })?;
    Ok(())
}
