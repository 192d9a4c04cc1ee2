//! What stands between the builder types and the C ABI: the error type and the `Recorder` every `Hit / Material / Texture`
//! writes itself into.  One `Recorder` lives for the duration of one commit (optimize.rs `BoundingVolumeHierarchyNode::render`).
use crate::ffi;

use std::collections::HashMap;
use std::ffi::CStr;
use std::os::raw::c_int;

/// A status code of include/rt_mi355x.h (`RT_ERR_*`, always negative) with the library's thread-local error text.
#[derive(Debug, Clone)]
pub struct Error {
    pub code: i32,
    pub message: String,
}
impl std::fmt::Display for Error {
    fn fmt(&self, f: &mut std::fmt::Formatter) -> std::fmt::Result {
        write!(f, "librt_mi355x error {}: {}", self.code, self.message)
    }
}
impl std::error::Error for Error {}

/// a negative return value of the C ABI -> `Err` carrying `rt_last_error()`
pub fn check(rc: c_int) -> Result<i32, Error> {
    if rc < 0 {
        let message = unsafe { CStr::from_ptr(ffi::rt_last_error()) }.to_string_lossy().into_owned();
        Err(Error { code: rc, message })
    } else {
        Ok(rc)
    }
}

/// The scene under construction.  Objects the reference shares through `Arc::clone` (one `Arc<Lambertian>` on six walls,
/// examples/cornell-box.rs:31-34) are recorded ONCE: `intern` keys a record by the object's address and kind, and every object of
/// a world stays alive, at its address, for as long as the world that holds its `Arc` is being recorded.
pub struct Recorder {
    raw: *mut ffi::rt_scene,
    seen: HashMap<(usize, &'static str), i32>,
}

impl Recorder {
    pub(crate) fn new(raw: *mut ffi::rt_scene) -> Self {
        Recorder { raw, seen: HashMap::new() }
    }
    /// the scene handle, for `ffi::rt_add_*` calls of an implementor outside this crate
    pub fn raw(&self) -> *mut ffi::rt_scene {
        self.raw
    }
    /// `object` recorded before under `kind`?  Its id; else `record` makes the record and the id is remembered.
    /// (`kind` keeps a struct apart from its own first field, which shares its address: `TransformedGeometry<Rectangle>`.)
    pub fn intern<T: ?Sized, F>(&mut self, object: &T, kind: &'static str, record: F) -> Result<i32, Error>
    where
        F: FnOnce(&mut Recorder) -> Result<i32, Error>,
    {
        let key = (object as *const T as *const () as usize, kind);
        if let Some(id) = self.seen.get(&key) {
            return Ok(*id);
        }
        let id = record(self)?;
        self.seen.insert(key, id);
        Ok(id)
    }
    /// `Sprite { geometry, material, transform }` -> its id.  -1 stands for `None` (src/sprite.rs:12-13); a null matrix for the identity.
    pub fn sprite(&mut self, geometry: i32, material: i32, transform: Option<&[f64]>) -> Result<i32, Error> {
        let m = transform.map_or(std::ptr::null(), |m| m.as_ptr());
        check(unsafe { ffi::rt_add_sprite(self.raw, geometry, material, m) })
    }
}
