//! `AxisAlignedBoundingBox`, `Bound<T>` and `BoundingVolumeHierarchyNode<T>` (src/optimize.rs:21-516).
//!
//! `BoundingVolumeHierarchyNode::new(objects) -> Option<Self>` keeps its signature and its `None` for an empty list
//! (src/optimize.rs:366-370).  Upstream it sorts the objects along a random axis and recurses; here it keeps the list and the
//! union of the members' boxes -- the acceleration structure is the library's business (a SAH tree over binary32 culling boxes,
//! built at commit; which tree is used never reaches a result: the reference's walk is unpruned and unordered).  A node is used
//! three ways, as upstream: as a sprite's geometry (instancing, `Cube`), as a member of another node's list (examples/main.rs:
//! 203-204,313-327) and as THE WORLD -- the outermost node, the one `render` is called on.
use crate::camera::PerspectiveCamera;
use crate::ffi;
use crate::gpu::{check, Error, Recorder};
use crate::mat4::Mat4;
use crate::ray::Hit;
use crate::vec3::Vec3;

use std::fmt::Debug;
use std::sync::{Arc, Mutex};

#[derive(Clone, Debug)]
pub struct AxisAlignedBoundingBox {
    min: Vec3,
    max: Vec3,
}
impl AxisAlignedBoundingBox {
    pub fn new(min: Vec3, max: Vec3) -> Self {
        AxisAlignedBoundingBox { min, max }
    }
    pub fn min(&self) -> &Vec3 {
        &self.min
    }
    pub fn max(&self) -> &Vec3 {
        &self.max
    }
    pub fn merged(&self, other: &Self) -> Self {
        AxisAlignedBoundingBox::new(
            Vec3::new(self.min.x().min(other.min.x()), self.min.y().min(other.min.y()), self.min.z().min(other.min.z())),
            Vec3::new(self.max.x().max(other.max.x()), self.max.y().max(other.max.y()), self.max.z().max(other.max.z())),
        )
    }
    /// the box of the 8 corners carried through `m` (what `Bound::bound` of a `Sprite` / `TransformedGeometry` returns,
    /// src/optimize.rs:128-241)
    pub fn transformed(&self, m: &Mat4) -> Self {
        let a = m.as_slice(); // column-major
        let (lo, hi) = (self.min, self.max);
        let mut min = [f64::INFINITY; 3];
        let mut max = [f64::NEG_INFINITY; 3];
        for corner in 0..8 {
            let p = [
                if corner & 1 != 0 { hi.x() } else { lo.x() },
                if corner & 2 != 0 { hi.y() } else { lo.y() },
                if corner & 4 != 0 { hi.z() } else { lo.z() },
            ];
            for i in 0..3 {
                let q = a[i] * p[0] + a[4 + i] * p[1] + a[8 + i] * p[2] + a[12 + i] * 1.0;
                if q < min[i] {
                    min[i] = q;
                }
                if q > max[i] {
                    max[i] = q;
                }
            }
        }
        AxisAlignedBoundingBox::new(Vec3::new(min[0], min[1], min[2]), Vec3::new(max[0], max[1], max[2]))
    }
}
/// upstream a box can be hit (it is how the tree is walked, src/optimize.rs:60-82); as a GEOMETRY it is outside the closed set
impl Hit for AxisAlignedBoundingBox {
    fn record_geometry(&self, _recorder: &mut Recorder) -> Result<i32, Error> {
        Err(Error { code: ffi::RT_ERR_UNSUPPORTED, message: "an AxisAlignedBoundingBox is not a geometry of librt_mi355x".into() })
    }
}

/// "this object can be wrapped in a `T`" (src/optimize.rs:88-96)
pub trait Bound<T>: Send + Sync + Hit + Debug
where
    T: Hit,
{
    fn bound(&self) -> Option<T>;
}
impl Bound<AxisAlignedBoundingBox> for AxisAlignedBoundingBox {
    fn bound(&self) -> Option<AxisAlignedBoundingBox> {
        Some(self.clone())
    }
}

/// the node of the sprites `ids` as a geometry (`rt_add_geometry_bvh`): the sprites are moved into it
pub(crate) fn record_node(recorder: &mut Recorder, ids: &[i32]) -> Result<i32, Error> {
    check(unsafe { ffi::rt_add_geometry_bvh(recorder.raw(), ids.as_ptr(), ids.len() as i32) })
}

struct Committed {
    raw: *mut ffi::rt_scene,
    device: i32,
}
// rt_render may be called concurrently on a committed scene (include/rt_mi355x.h); the handle is only freed by Drop
unsafe impl Send for Committed {}
unsafe impl Sync for Committed {}
impl Drop for Committed {
    fn drop(&mut self) {
        unsafe { ffi::rt_scene_destroy(self.raw) }
    }
}

pub struct BoundingVolumeHierarchyNode<T>
where
    T: Hit, // `dyn Bound<T>` is only well-formed for a `T` that can be hit
{
    volume: T,
    objects: Vec<Arc<dyn Bound<T>>>,
    device: Mutex<Option<i32>>,            // where the world is to be committed (`on_device`); None: RT_MI355X_DEVICE, else 0
    committed: Mutex<Vec<Arc<Committed>>>, // the world on every device it has been rendered on so far
}
impl<T: Hit> Debug for BoundingVolumeHierarchyNode<T> {
    fn fmt(&self, f: &mut std::fmt::Formatter) -> std::fmt::Result {
        write!(f, "BoundingVolumeHierarchyNode {{ volume: {:?}, objects: {} }}", self.volume, self.objects.len())
    }
}

impl<T> BoundingVolumeHierarchyNode<T>
where
    T: Bound<T>,
{
    pub fn volume(&self) -> &T {
        &self.volume
    }
    /// the members, in the order they were given (upstream: `left()` / `right()` of a binary node)
    pub fn objects(&self) -> &[Arc<dyn Bound<T>>] {
        &self.objects
    }
}

impl BoundingVolumeHierarchyNode<AxisAlignedBoundingBox> {
    /// `None` for an empty list, and when no member has a box (src/optimize.rs:366-370,421-436)
    pub fn new(objects: Vec<Arc<dyn Bound<AxisAlignedBoundingBox>>>) -> Option<Self> {
        let mut volume: Option<AxisAlignedBoundingBox> = None;
        for o in &objects {
            if let Some(b) = o.bound() {
                volume = Some(match volume {
                    Some(v) => v.merged(&b),
                    None => b,
                });
            }
        }
        volume.map(|volume| BoundingVolumeHierarchyNode { volume, objects, device: Mutex::new(None), committed: Mutex::new(Vec::new()) })
    }

    /// Commit (and render) on HIP device `device` instead of `RT_MI355X_DEVICE` / device 0.  Only meaningful on the world.
    pub fn on_device(self, device: i32) -> Self {
        *self.device.lock().unwrap() = Some(device);
        self
    }

    fn default_device(&self) -> i32 {
        let chosen: Option<i32> = *self.device.lock().unwrap();
        chosen.or_else(|| std::env::var("RT_MI355X_DEVICE").ok().and_then(|v| v.parse().ok())).unwrap_or(0)
    }

    /// the six faces of ONE `Cube::new(w, h, d)`, in order?  Then the node is that cube.
    fn as_cube(&self) -> Option<(f64, f64, f64)> {
        if self.objects.len() != 6 {
            return None;
        }
        let (w, h, d, _) = self.objects[0].cube_face()?;
        for (i, o) in self.objects.iter().enumerate() {
            if o.cube_face()? != (w, h, d, i) {
                return None;
            }
        }
        Some((w, h, d))
    }

    /// the world on `device`: recorded and committed on first use (`rt_scene_commit` == `BoundingVolumeHierarchyNode::new(..).unwrap()`
    /// of the drivers: flatten, build the culling tree, upload)
    fn committed_on(&self, device: i32) -> Result<Arc<Committed>, Error> {
        let mut have = self.committed.lock().unwrap();
        if let Some(c) = have.iter().find(|c| c.device == device) {
            return Ok(c.clone());
        }
        let raw = unsafe { ffi::rt_scene_create() };
        let scene = Arc::new(Committed { raw, device }); // destroyed on every early return
        let mut recorder = Recorder::new(raw);
        let mut ids = Vec::new();
        for o in &self.objects {
            o.record_member(&mut recorder, &mut ids)?; // the world's own list: every sprite not moved into a node
        }
        check(unsafe { ffi::rt_scene_commit(raw, device) })?;
        have.push(scene.clone());
        Ok(scene)
    }

    fn params(width: usize, height: usize, spp: usize, maxDepth: usize, seed: u64) -> ffi::rt_render_params {
        ffi::rt_render_params {
            width: width as i32,
            height: height as i32,
            spp: spp as i32,
            max_depth: maxDepth as i32,
            seed,
            shard_index: 0,
            shard_count: 1,
            flags: 0,
        }
    }
    fn rows(rgb: &[f64], width: usize, height: usize) -> Vec<Vec<Vec3>> {
        (0..height)
            .map(|y| (0..width).map(|x| Vec3::new(rgb[(y * width + x) * 3], rgb[(y * width + x) * 3 + 1], rgb[(y * width + x) * 3 + 2])).collect())
            .collect()
    }

    /// The whole sampling loop of a driver (examples/book-one.rs:52-88: worker threads, `subPixelSampleCount` x `color(&ray, world,
    /// maxDepth)` per pixel, mpsc fan-in) on the MI355X.  Returns `buffer[y][x]`, y up -- the `vec![vec![Vec3; width]; height]` the
    /// drivers fill, so their printing loop stays as it is.  `seed`: upstream's `thread_rng()` cannot be seeded; here every sample owns
    /// a counter-based stream of `seed` (include/rt_rng.h) and the image is reproducible.
    pub fn render(
        &self,
        camera: &PerspectiveCamera,
        width: usize,
        height: usize,
        subPixelSampleCount: usize,
        maxDepth: usize,
        seed: u64,
    ) -> Result<Vec<Vec<Vec3>>, Error> {
        let scene = self.committed_on(self.default_device())?;
        let mut rgb = vec![0.0f64; width * height * 3];
        let p = Self::params(width, height, subPixelSampleCount, maxDepth, seed);
        check(unsafe { ffi::rt_render(scene.raw, camera.raw(), &p, rgb.as_mut_ptr(), std::ptr::null_mut()) })?;
        Ok(Self::rows(&rgb, width, height))
    }

    /// The same image from several GPUs: the world is committed on every device of `devices`, the image's 8x8 tiles are dealt
    /// `tile % devices.len()`, one host thread per device inside the library (`rt_render_sharded`).  Bit-identical to `render`.
    pub fn render_on_devices(
        &self,
        devices: &[i32],
        camera: &PerspectiveCamera,
        width: usize,
        height: usize,
        subPixelSampleCount: usize,
        maxDepth: usize,
        seed: u64,
    ) -> Result<Vec<Vec<Vec3>>, Error> {
        let mut scenes = Vec::new();
        for d in devices {
            scenes.push(self.committed_on(*d)?);
        }
        let raws: Vec<*mut ffi::rt_scene> = scenes.iter().map(|s| s.raw).collect();
        let mut rgb = vec![0.0f64; width * height * 3];
        let p = Self::params(width, height, subPixelSampleCount, maxDepth, seed);
        check(unsafe { ffi::rt_render_sharded(raws.as_ptr(), raws.len() as i32, camera.raw(), &p, rgb.as_mut_ptr()) })?;
        Ok(Self::rows(&rgb, width, height))
    }

    /// Continue raw per-pixel sums (`width * height * 3`, row-major, y up) with samples `[s_begin, s_end)` of the
    /// `subPixelSampleCount`-sample render; divide by `subPixelSampleCount` after the last range.  Checkpoint = `sums`, `s_end` and
    /// `scene_hash()`; the finished image is bit-identical to one `render` call (`rt_render_progressive`).
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
        let scene = self.committed_on(self.default_device())?;
        let p = Self::params(width, height, subPixelSampleCount, maxDepth, seed);
        check(unsafe { ffi::rt_render_progressive(scene.raw, camera.raw(), &p, s_begin as i32, s_end as i32, sums.as_mut_ptr()) })?;
        Ok(())
    }

    /// identifies the committed scene (checkpoints)
    pub fn scene_hash(&self) -> Result<u64, Error> {
        let scene = self.committed_on(self.default_device())?;
        let mut h = 0u64;
        check(unsafe { ffi::rt_scene_hash(scene.raw, &mut h) })?;
        Ok(h)
    }

    /// a device error word of an earlier render (a kernel that refused to run) as `Err` (`rt_render_status`)
    pub fn status(&self) -> Result<(), Error> {
        for c in self.committed.lock().unwrap().iter() {
            check(unsafe { ffi::rt_render_status(c.raw) })?;
        }
        Ok(())
    }

    /// give the per-sample workspace of every committed copy back to the device (`rt_scene_trim`)
    pub fn trim(&self) -> Result<(), Error> {
        for c in self.committed.lock().unwrap().iter() {
            check(unsafe { ffi::rt_scene_trim(c.raw) })?;
        }
        Ok(())
    }
}

impl Hit for BoundingVolumeHierarchyNode<AxisAlignedBoundingBox> {
    /// a node as a sprite's GEOMETRY (`Sprite::builder().geometry(Arc::new(node))`): `rt_add_geometry_cube` when it is the six faces
    /// of one `Cube::new`, else `rt_add_geometry_bvh` over its members (instancing, src/sprite.rs:87-93).  A shared `Arc` node is
    /// one record.
    fn record_geometry(&self, recorder: &mut Recorder) -> Result<i32, Error> {
        recorder.intern(self, "node", |r| {
            if let Some((w, h, d)) = self.as_cube() {
                return check(unsafe { ffi::rt_add_geometry_cube(r.raw(), w, h, d) });
            }
            let mut ids = Vec::new();
            for o in &self.objects {
                o.record_member(r, &mut ids)?;
            }
            record_node(r, &ids)
        })
    }
    /// a node as a MEMBER of another node's list (`cubes` and `spheres` of examples/main.rs:203-204,313-327): a node has neither
    /// matrix nor material, so its members simply join the outer list -- for the unpruned, unordered walk of the reference the
    /// nesting of boxes cannot change which primitive is nearest.
    fn record_member(&self, recorder: &mut Recorder, out: &mut Vec<i32>) -> Result<(), Error> {
        for o in &self.objects {
            o.record_member(recorder, out)?;
        }
        Ok(())
    }
}
impl Bound<AxisAlignedBoundingBox> for BoundingVolumeHierarchyNode<AxisAlignedBoundingBox> {
    fn bound(&self) -> Option<AxisAlignedBoundingBox> {
        Some(self.volume.clone())
    }
}
