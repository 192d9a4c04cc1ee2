//! Materials and textures with the reference's constructors (src/material.rs:24-325): `Lambertian::new(albedo)`,
//! `Metal::new(albedo, fuzziness)`, `Dielectric::new(refractive)`, `DiffuseLight::new(emission)`, `Isotropic::new(albedo)`, each
//! generic over `T: Into<Arc<dyn Texture>>` so that a plain `Vec3` colour and an `Arc<dyn Texture>` both go in; `SolidColor`,
//! `CheckerTexture`, `ImageTexture`.  `Material::scatter` and `Texture::value` run on the device (`rtl::shade`,
//! `rtl::texture_value`); the traits' one required method here records the object (see ray.rs).
use crate::ffi;
use crate::gpu::{check, Error, Recorder};
use crate::vec3::Vec3;

use std::fmt::Debug;
use std::sync::Arc;

pub trait Material: Send + Sync + Debug {
    /// `rt_add_material_*`; returns the material id
    fn record(&self, recorder: &mut Recorder) -> Result<i32, Error>;
}

pub trait Texture: Send + Sync + Debug {
    /// `rt_add_texture_*`; returns the texture id
    fn record(&self, recorder: &mut Recorder) -> Result<i32, Error>;
}

fn texture_id(t: &Arc<dyn Texture>, recorder: &mut Recorder) -> Result<i32, Error> {
    recorder.intern(&**t, "texture", |r| t.record(r))
}

// ---------------------------------------------------------------- textures
#[derive(Clone, Debug)]
pub struct SolidColor {
    color: Vec3,
}
impl SolidColor {
    pub fn new(color: Vec3) -> Self {
        SolidColor { color }
    }
}
impl Texture for SolidColor {
    fn record(&self, recorder: &mut Recorder) -> Result<i32, Error> {
        check(unsafe { ffi::rt_add_texture_solid(recorder.raw(), self.color.to_array().as_ptr()) })
    }
}
/// a colour is a texture (src/material.rs:48-58): `Lambertian::new(Vec3::new(0.5, 0.5, 0.5))`
impl From<Vec3> for Arc<dyn Texture> {
    fn from(color: Vec3) -> Self {
        Arc::new(SolidColor { color })
    }
}
impl From<Vec3> for SolidColor {
    fn from(color: Vec3) -> Self {
        SolidColor { color }
    }
}

#[derive(Clone, Debug)]
pub struct CheckerTexture {
    black: Arc<dyn Texture>,
    white: Arc<dyn Texture>,
}
impl CheckerTexture {
    pub fn new<T>(black: T, white: T) -> Self
    where
        T: Into<Arc<dyn Texture>>,
    {
        CheckerTexture { black: black.into(), white: white.into() }
    }
}
impl Texture for CheckerTexture {
    fn record(&self, recorder: &mut Recorder) -> Result<i32, Error> {
        let black = texture_id(&self.black, recorder)?;
        let white = texture_id(&self.white, recorder)?;
        check(unsafe { ffi::rt_add_texture_checker(recorder.raw(), black, white) })
    }
}

/// `ImageTexture::new(mapping)` with the reference's closure `Fn(&(f64, f64)) -> Vec3` (src/material.rs:247-265).  A closure
/// cannot cross the C ABI, so it is TABULATED at commit: `mapping` is called once at the centre of every texel of a
/// `width x height` RGB8 table and the device does the nearest-texel lookup of examples/main.rs:267-280,
/// `(u * width) as u32, ((1 - v) * height) as u32`.  For that closure -- a nearest-texel lookup into an image -- a table of the
/// IMAGE's size reproduces it exactly: say so with `.resolution(image.width(), image.height())`; without it the table is
/// 2048 x 1024 and the picture is resampled once.
pub struct ImageTexture<T> {
    mapping: T,
    width: u32,
    height: u32,
}
impl<T> ImageTexture<T> {
    pub fn new(mapping: T) -> Self {
        ImageTexture { mapping, width: 2048, height: 1024 }
    }
    pub fn resolution(mut self, width: u32, height: u32) -> Self {
        self.width = width.max(1);
        self.height = height.max(1);
        self
    }
}
impl<T> Texture for ImageTexture<T>
where
    T: (Fn(&(f64, f64)) -> Vec3) + Send + Sync,
{
    fn record(&self, recorder: &mut Recorder) -> Result<i32, Error> {
        let (w, h) = (self.width as usize, self.height as usize);
        let mut rgb = vec![0u8; w * h * 3];
        let channel = |c: f64| (c * 255.0).round().max(0.0).min(255.0) as u8;
        for py in 0..h {
            for px in 0..w {
                let uv = ((px as f64 + 0.5) / w as f64, 1.0 - (py as f64 + 0.5) / h as f64);
                let c = (self.mapping)(&uv);
                let at = (py * w + px) * 3;
                rgb[at] = channel(c.r());
                rgb[at + 1] = channel(c.g());
                rgb[at + 2] = channel(c.b());
            }
        }
        check(unsafe { ffi::rt_add_texture_image_rgb8(recorder.raw(), rgb.as_ptr(), w as i32, h as i32) })
    }
}
impl<T> Debug for ImageTexture<T> {
    fn fmt(&self, f: &mut std::fmt::Formatter) -> std::fmt::Result {
        write!(f, "ImageTexture")
    }
}

// ---------------------------------------------------------------- materials
#[derive(Clone, Debug)]
pub struct Lambertian {
    albedo: Arc<dyn Texture>,
}
impl Lambertian {
    pub fn new<T>(albedo: T) -> Self
    where
        T: Into<Arc<dyn Texture>>,
    {
        Lambertian { albedo: albedo.into() }
    }
    pub fn albedo(&self) -> &Arc<dyn Texture> {
        &self.albedo
    }
}
impl Material for Lambertian {
    fn record(&self, recorder: &mut Recorder) -> Result<i32, Error> {
        let t = texture_id(&self.albedo, recorder)?;
        check(unsafe { ffi::rt_add_material_lambertian(recorder.raw(), t) })
    }
}

#[derive(Clone, Debug)]
pub struct Metal {
    albedo: Arc<dyn Texture>,
    fuzziness: f64,
}
impl Metal {
    pub fn new<T>(albedo: T, fuzziness: f64) -> Self
    where
        T: Into<Arc<dyn Texture>>,
    {
        Metal { albedo: albedo.into(), fuzziness }
    }
    pub fn albedo(&self) -> &Arc<dyn Texture> {
        &self.albedo
    }
    pub fn fuzziness(&self) -> f64 {
        self.fuzziness
    }
}
impl Material for Metal {
    fn record(&self, recorder: &mut Recorder) -> Result<i32, Error> {
        let t = texture_id(&self.albedo, recorder)?;
        check(unsafe { ffi::rt_add_material_metal(recorder.raw(), t, self.fuzziness) })
    }
}

#[derive(Clone, Debug)]
pub struct Dielectric {
    refractive: f64,
}
impl Dielectric {
    pub fn new(refractive: f64) -> Self {
        Dielectric { refractive }
    }
    pub fn refractive(&self) -> f64 {
        self.refractive
    }
}
impl Material for Dielectric {
    fn record(&self, recorder: &mut Recorder) -> Result<i32, Error> {
        check(unsafe { ffi::rt_add_material_dielectric(recorder.raw(), self.refractive) })
    }
}

#[derive(Clone, Debug)]
pub struct DiffuseLight {
    emission: Arc<dyn Texture>,
}
impl DiffuseLight {
    pub fn new<T>(emission: T) -> Self
    where
        T: Into<Arc<dyn Texture>>,
    {
        DiffuseLight { emission: emission.into() }
    }
}
impl Material for DiffuseLight {
    fn record(&self, recorder: &mut Recorder) -> Result<i32, Error> {
        let t = texture_id(&self.emission, recorder)?;
        check(unsafe { ffi::rt_add_material_diffuse_light(recorder.raw(), t) })
    }
}

#[derive(Clone, Debug)]
pub struct Isotropic {
    albedo: Arc<dyn Texture>,
}
impl Isotropic {
    pub fn new<T>(albedo: T) -> Self
    where
        T: Into<Arc<dyn Texture>>,
    {
        Isotropic { albedo: albedo.into() }
    }
}
impl Material for Isotropic {
    fn record(&self, recorder: &mut Recorder) -> Result<i32, Error> {
        let t = texture_id(&self.albedo, recorder)?;
        check(unsafe { ffi::rt_add_material_isotropic(recorder.raw(), t) })
    }
}
