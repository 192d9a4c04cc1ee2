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
    ConstantMedium(Arc<Geometry>, f64),
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
            let g = match &s.geometry {
                Some(g) => rec.geometry(g)?,
                None => -1,
            };
            let m = match &s.material {
                Some(m) => rec.material(m)?,
                None => -1,
            };
            check(unsafe { ffi::rt_add_sprite(raw, g, m, s.transform.as_ptr()) })?;
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
