/*
 * rt_mi355x.h -- C ABI of librt_mi355x.so, the MI355X (gfx950) path tracer that
 * replaces the per-pixel sampling loop of aiifabbf/ray-tracer.
 *
 * The reference has no FFI or plugin interface; its de-facto operator API is the
 * trait/builder surface used by the example drivers.
 * Crossing a C boundary once per ray (render::color, src/render.rs:5) would erase
 * any GPU gain, so the boundary sits one level up: the scene is described through
 * calls that mirror the reference constructors one-to-one, committed once
 * (flattened into SoA device arrays + BVH), and rendered per image / per tile set.
 * Each entry point cites the reference interface it replaces.
 *
 * Conventions
 *   - plain C types only: pointers, sizes, doubles, ints.  No torch / HIP types.
 *   - status returns: 0 = ok, negative = error; rt_last_error() gives the text
 *     (thread-local).  Constructors return an id >= 0 or a negative error.
 *     The reference signals "nothing" with Option and its drivers unwrap()/panic
 *     (examples/book-one.rs:32); here nothing aborts across the boundary.
 *   - ids returned by rt_add_* stay valid until rt_scene_destroy; sharing an id
 *     between sprites is the reference's Arc::clone
 *     (examples/cornell-box.rs:62,70,78,96,118).
 *   - matrices: 16 doubles, column-major like src/mat4.rs:5-17 (m[12..14] =
 *     translation).  NULL = identity.
 *   - images: linear radiance, double, layout [y][x][3] with y UP (row 0 is the
 *     bottom row) exactly like `buffer[y][x]` in examples/book-one.rs:53,87.
 *   - all arithmetic on the path is IEEE binary64 like the reference.
 *   - the scene is immutable after rt_scene_commit.  rt_render may be called from any
 *     number of threads on one committed scene (the traits are Send + Sync upstream,
 *     src/ray.rs:85); calls on the same scene are serialised because they share its
 *     per-sample workspace -- commit one scene per device for parallel renders.
 *     rt_render_tiles_device is asynchronous: issue the calls of one scene on one
 *     stream (or order them yourself).
 */
#ifndef RT_MI355X_H
#define RT_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RT_OK 0
#define RT_ERR_INVALID (-1)     /* bad argument / id */
#define RT_ERR_EMPTY (-2)       /* BoundingVolumeHierarchyNode::new(vec![]) -> None (src/optimize.rs:367-370) */
#define RT_ERR_DEVICE (-3)      /* HIP failure or no GPU */
#define RT_ERR_UNSUPPORTED (-4) /* scene feature outside the closed enums */
#define RT_ERR_STATE (-5)       /* call order (e.g. render before commit) */

typedef struct rt_scene rt_scene;

const char *rt_last_error(void);
const char *rt_version(void);
/* number of visible HIP devices; 0 when there is none (never an error) */
int rt_device_count(void);

/* ---- Mat4 helpers: src/mat4.rs:21-28,36-47,52-80,85-143,146-181,184-243 ---- */
void rt_mat4_identity(double out[16]);
void rt_mat4_translation(const double offset[3], double out[16]);
void rt_mat4_rotation(double radians, const double axis[3], double out[16]);
void rt_mat4_multiplied(const double self[16], const double other[16], double out[16]); /* self * other */
double rt_mat4_determinant(const double m[16]);
int rt_mat4_inversed(const double m[16], double out[16]); /* RT_ERR_INVALID iff det == 0.0 */

/* ---- scene lifetime ---- */
rt_scene *rt_scene_create(void);
void rt_scene_destroy(rt_scene *);

/* ---- textures: SolidColor::new / CheckerTexture::new / ImageTexture::new
 *      (src/material.rs:200-215,217-245,247-265).  ImageTexture's closure cannot
 *      cross the boundary; the one closure the reference uses -- nearest texel of
 *      an RGB8 image, examples/main.rs:267-280 -- is the supported form. ---- */
int rt_add_texture_solid(rt_scene *, const double rgb[3]);
int rt_add_texture_checker(rt_scene *, int black_texture, int white_texture);
int rt_add_texture_image_rgb8(rt_scene *, const uint8_t *rgb, int width, int height);

/* ---- materials: Lambertian::new, Metal::new, Dielectric::new, DiffuseLight::new,
 *      Isotropic::new (src/material.rs:36-46,81-97,128-138,279-288,307-315) ---- */
int rt_add_material_lambertian(rt_scene *, int albedo_texture);
int rt_add_material_metal(rt_scene *, int albedo_texture, double fuzziness);
int rt_add_material_dielectric(rt_scene *, double refractive);
int rt_add_material_diffuse_light(rt_scene *, int emission_texture);
int rt_add_material_isotropic(rt_scene *, int albedo_texture);

/* ---- geometries: Sphere::new (src/geometry.rs:17-19), Rectangle::new (:139-144),
 *      BoundingVolumeHierarchyNode::new(Cube::new(w,h,d)) (src/geometry.rs:254-286 as
 *      wrapped by examples/cornell-box.rs:85-101), ConstantMedium::new(boundary, density)
 *      (src/volume.rs:25-30; boundary = any geometry but another medium: ConstantMedium<T: Hit>) ---- */
int rt_add_geometry_sphere(rt_scene *, double radius);
int rt_add_geometry_rectangle(rt_scene *, double width, double height);
int rt_add_geometry_cube(rt_scene *, double width, double height, double depth);
int rt_add_geometry_constant_medium(rt_scene *, int boundary_geometry, double density);
/* TransformedGeometry::new(geometry, M) (src/geometry.rs:185-246): a geometry under one more matrix, without a material */
int rt_add_geometry_transformed(rt_scene *, int geometry, const double transform[16]);
/* BoundingVolumeHierarchyNode::new(vec![sprites...]) used as a GEOMETRY -- the reference's instancing
 * (src/sprite.rs:87-93 with T = BoundingVolumeHierarchyNode, src/optimize.rs:339-343: children are Arc<dyn Bound>).
 * The listed sprites are moved into the node (like the Vec the constructor takes): they stop being part of the world's
 * own list and are hit only through sprites whose geometry is this node.  A child without a material is the reference's
 * TransformedGeometry.  Whatever material a child carries, Sprite::hit of the enclosing sprite replaces it
 * (src/sprite.rs:119-127).  RT_ERR_EMPTY for an empty list (None upstream).  Nesting: up to 15 transform levels above
 * a sphere / rectangle (more than 4: RT_FEAT_DEEP_CHAIN, served by the slower kernel family for general media).  At commit every instance is expanded into leaves that keep the matrices of all their levels --
 * the traversal stays one flat BVH and the arithmetic per level is the reference's. */
int rt_add_geometry_bvh(rt_scene *, const int *sprites, int n_sprites);

/* ---- Sprite::builder().geometry(g).material(m).transform(M).build()
 *      (src/sprite.rs:22-72).  geometry / material = -1 is the reference's None:
 *      such a sprite is never hit / shades black (src/sprite.rs:95,136; src/render.rs:18-20).
 *      A singular M makes the sprite unhittable (src/sprite.rs:131-134). ---- */
int rt_add_sprite(rt_scene *, int geometry, int material, const double transform[16]);

/* ---- world = BoundingVolumeHierarchyNode::new(all sprites).unwrap()
 *      (src/optimize.rs:366-440, examples/book-one.rs:32).  Nested
 *      BoundingVolumeHierarchyNode children without a transform
 *      (examples/main.rs:191,303) are the same flat set of sprites: the reference's
 *      traversal is unpruned and order-free (src/optimize.rs:469-498), so the
 *      nearest hit does not depend on the tree.  Flattens the scene, builds the
 *      acceleration structure and, when device >= 0, uploads it to that HIP device.
 *      device = -1 keeps it host-only (inspection / CPU-side tests).
 *      RT_ERR_EMPTY for an empty scene. ---- */
int rt_scene_commit(rt_scene *, int device);

/* ---- PerspectiveCamera::new(eye, center, up, fov, aspect, focusDistance, lensRadius)
 *      (src/camera.rs:25-59); fov in radians.  Keeps the reference's un-normalised
 *      `u` (Q1) and scalar lens offset (Q2). ---- */
typedef struct rt_camera {
    double eye[3];
    double lower_left[3];
    double horizontal[3];
    double vertical[3];
    double lens_radius;
} rt_camera;
int rt_camera_perspective(rt_camera *out, const double eye[3], const double center[3], const double up[3], double fov,
                          double aspect, double focus_distance, double lens_radius);

/* ---- the hot path: the per-pixel sampling loop of the drivers
 *      (examples/book-one.rs:56-88 = cornell-box.rs:159-189 = main.rs:71-101):
 *      for every pixel, spp times: jitter, camera.ray, color(ray, world, max_depth),
 *      accumulate, divide by spp.  Sample stream id = (y*W + x)*spp + s under
 *      `seed` (include/rt_rng.h). ---- */
typedef struct rt_render_params {
    int width, height;
    int spp;
    int max_depth; /* the literal 100 in the drivers (examples/book-one.rs:74) */
    uint64_t seed;
    /* tile sharding: the image is cut into 8x8 pixel
     * tiles, tile id = ty * tiles_x + tx; this call renders the tiles with
     * id % shard_count == shard_index.  {0, 1} renders the whole image. */
    int shard_index, shard_count;
    unsigned flags; /* RT_FLAG_* */
} rt_render_params;

#define RT_TILE 8                 /* tile edge in pixels */
#define RT_FLAG_COUNTERS 1u       /* also accumulate rt_counters (slower build of the kernel) */
#define RT_FLAG_DEFERRED_OUTPUT 2u /* rt_render_tiles_device only: see rt_render_wait_output */
#define RT_FLAG_ASCENDING_TILES 4u /* hand the owned tiles out in ascending order: neither use nor learn a tile order (below) */

typedef struct rt_counters {
    uint64_t samples, segments, nodes_visited, prims_tested, rng_draws;
    /* executions of each block of the wave-vote loop and the lanes active in them:
     * lane / (64 * wave) is the SIMD utilisation of that block */
    uint64_t node_wave, node_lane, leaf_wave, leaf_lane, shade_wave, shade_lane;
    /* shader-clock cycles spent in each block, summed over waves (diagnostic) */
    uint64_t node_cycles, leaf_cycles, shade_cycles, finish_cycles, refill_cycles, begin_cycles;
    /* swap-at-shade kernels only (0 otherwise): shade-block executions in a class mode / in new-sample mode, paths
     * parked in and pulled from the workgroup queues, queue locks found busy, paths scattered, and of those the
     * ones scattered outside the chosen class (could not be parked) */
    uint64_t swap_class_mode, swap_new_mode, swap_parked, swap_pulled, swap_lock_busy, swap_scattered, swap_off_class;
    uint64_t swap_cycles; /* of finish_cycles: classification + queue traffic of the swap (counting build) */
    /* lanes that sat idle while the node block ran, by what they were waiting for: a finished segment waiting for the shade
     * quorum, a leaf waiting for the leaf quorum, no path at all */
    uint64_t node_idle_done, node_idle_leaf, node_idle_empty;
} rt_counters;

/* Render into host memory: out_rgb[(y*W + x)*3 + c].  With shard_count > 1 only
 * this shard's pixels are written.  counters may be NULL. */
int rt_render(rt_scene *, const rt_camera *, const rt_render_params *, double *out_rgb, rt_counters *counters);

/* ---- the drivers' thread fan-out and mpsc gather (examples/book-one.rs:52-88) across the GPUs of one node ----
 * rt_scene_clone: a second committed copy of a recorded scene on another device (the immutable world every worker
 * thread shares through an Arc upstream, examples/book-one.rs:57-59).
 * rt_render_sharded: scenes[i] is a committed copy on device i' of its choice; the image is cut into 8x8 tiles dealt
 * tile_id % n_scenes, all shards are rendered concurrently (one host thread per scene, each on its scene's device and
 * stream) and every shard's pixels land in out_rgb.  Sample streams are global, so the image is bit-identical for any
 * n_scenes (params->shard_index / shard_count are ignored).  No collective is involved: this is the entry for callers
 * that own host memory (the Rust drivers); bench.py keeps the framebuffer on the devices and uses one RCCL gather. */
rt_scene *rt_scene_clone(const rt_scene *, int device);
int rt_render_sharded(rt_scene *const *scenes, int n_scenes, const rt_camera *, const rt_render_params *, double *out_rgb);

/* Progressive / resumable rendering (the reference has none: a render is all-or-nothing and a 16.6 Gsample
 * image takes a while).  Renders samples s in [s_begin, s_end) of every pixel of this shard -- the streams are
 * those of the full `params->spp` render -- and continues the raw per-pixel sums in `sums` ([y][x][3], y up)
 * in sample order.  s_begin == 0 starts from zero; call with consecutive ranges; save `sums` + s_end to
 * checkpoint.  After the last range divide by spp: the image is bit-identical to one rt_render call. */
int rt_render_progressive(rt_scene *, const rt_camera *, const rt_render_params *, int s_begin, int s_end, double *sums);

/* Device-resident variant for callers that own device memory and a stream
 * (plumbed as raw pointers; `stream` is a hipStream_t or NULL).
 * d_tiles_out receives this shard's tiles PACKED in ascending tile id:
 * [k][py][px][3] doubles, k-th owned tile, RT_TILE*RT_TILE pixels each.
 * Asynchronous: returns after enqueueing.  rt_shard_tile_count gives the number
 * of tiles (buffer size = count * 64 * 3 doubles).
 * A scene keeps two render slots (sample workspace, job counter, timing events): two calls on different streams run
 * concurrently on the device -- the second one's workgroups fill the CUs while the first one's last long paths, its
 * reduce and whatever the caller enqueued behind it finish; a third call is ordered behind the slot it reuses. */
int rt_shard_tile_count(int width, int height, int shard_index, int shard_count);
int rt_render_tiles_device(rt_scene *, const rt_camera *, const rt_render_params *, void *d_tiles_out, void *d_counters,
                           void *stream);
/* Pipelining consecutive renders.  A render is two kernels: render_kernel (bound by the VALUs) and reduce_kernel, which sums the
 * sample records in sample order (bound by HBM).  With RT_FLAG_DEFERRED_OUTPUT in params->flags rt_render_tiles_device enqueues
 * only render_kernel on `stream`; reduce_kernel follows it on a stream of the scene's own, so the next render on `stream` starts
 * while the sums of this one are still being read (each render uses its own workspace slot).  d_tiles_out is complete where
 * rt_render_wait_output(scene, any_stream) -- called after that rt_render_tiles_device and before the next one on this scene --
 * puts its wait.  Results are bit-identical; bench.py uses it (3.7 % of a step). */
int rt_render_wait_output(rt_scene *, void *stream);
/* Un-permute gathered shards into a row-major image on the device:
 * d_gathered = shard_count buffers of `tiles_per_shard_padded` tiles each,
 * back to back (what one RCCL gather of equal-sized chunks produces). */
int rt_unpack_tiles_device(const void *d_gathered, int tiles_per_shard_padded, int shard_count, int width, int height,
                           void *d_image_out, void *stream);
/* The same two permutations on HOST memory, for callers whose gather ends in host buffers (a gloo / MPI gather, a
 * checkpoint): rt_pack_tiles_host copies this shard's pixels of a row-major image [y][x][3] into the packed layout of
 * rt_render_tiles_device (`tiles_padded` >= rt_shard_tile_count tiles, the rest and pixels outside the image zero);
 * rt_unpack_tiles_host is rt_unpack_tiles_device on host pointers.  Pure index work, no arithmetic. */
int rt_pack_tiles_host(const double *image, int width, int height, int shard_index, int shard_count, int tiles_padded, double *tiles_out);
int rt_unpack_tiles_host(const double *gathered, int tiles_per_shard_padded, int shard_count, int width, int height, double *image_out);
/* Status of the renders launched on this scene so far: waits for them, then returns RT_OK or RT_ERR_DEVICE when a kernel
 * set the device error word (launched with fewer LDS bytes than its layout needs; the counting build's progress watchdog:
 * ray-tracer_amd/csrc/rt_lds.h) -- a persistent kernel cannot fail any other way than by hanging, so it reports instead.
 * rt_render / rt_render_progressive / rt_last_kernel_ms check it themselves; callers of the asynchronous
 * rt_render_tiles_device call this once their stream has drained (or whenever they want to wait). */
int rt_render_status(rt_scene *);
/* Per-sample workspace.  A render keeps one 32-byte record per sample of a pass in device memory (width x height x spp x 32 B
 * / shard_count for one pass: 15.36 GB for 1200x800x500), written once by render_kernel and summed in sample order by
 * reduce_kernel; a scene holds one such buffer per render slot in use (one unless two renders overlap), sized to the largest
 * render so far, reused by later ones.  The limit per slot is the smaller of 32 GiB and a quarter of the device's memory
 * unless set here (bytes; 0 restores the default) or through RT_SAMPLE_WORKSPACE_MB; renders that need more run in several
 * passes (bit-identical).  rt_scene_trim waits for the scene's renders and gives the buffers back;
 * rt_scene_workspace_bytes reports what the scene holds now. */
int rt_scene_set_workspace_limit(rt_scene *, size_t bytes);
int rt_scene_trim(rt_scene *);
size_t rt_scene_workspace_bytes(const rt_scene *);
/* time (ms) the last rt_render / rt_render_tiles_device kernel took on its stream,
 * measured with HIP events recorded around the launch; blocks until it finished */
int rt_last_kernel_ms(rt_scene *, float *ms);
/* how the last render on this scene was launched (diagnostics; the figures rocprofv3 prints for VGPRs / LDS of a
 * dispatch are not reliable for these kernels): persistent workgroups, threads per workgroup, dynamic LDS per
 * workgroup, resident workgroups per CU, CUs, passes over the sample workspace, jobs of the last pass and their size,
 * kernel family feature bits (1 general prims, 2 media, 4 textures), whether the node array (and, for small general scenes, every record) is in LDS and the
 * swap-at-shade queues are in use (and their capacity), and the bytes of per-sample workspace the render used */
typedef struct rt_launch_config {
    int blocks, block_threads;
    unsigned lds_bytes;
    int blocks_per_cu, n_cu;
    int passes, n_jobs, job_spp;
    unsigned kernel_features;
    int lds_nodes, swap;          /* lds_nodes: 0 the node array stays in global memory; 1 every workgroup keeps a copy in LDS; 2 a copy with binary16
                                     planes (32-byte nodes: a tree whose binary32 form does not fit and this one does -- the book-two cover) */
    size_t workspace_bytes;
    int swap_cap, waves_per_simd; /* entries per swap queue; waves per SIMD the kernel family is compiled for */
    int tile_order;               /* RT_TILE_ORDER_* */
    int records_in_lds;           /* 1: a small general scene (box-list walk) whose transform / prim / material records the kernel keeps in LDS */
} rt_launch_config;
int rt_last_launch_config(rt_scene *, rt_launch_config *out);
/* What a render of this committed scene WOULD launch, decided by the same code a render runs, without a device: block_threads,
 * lds_bytes, lds_nodes, swap, swap_cap, waves_per_simd, kernel_features, records_in_lds are filled (blocks_per_cu = the workgroups per
 * CU the family's full occupancy asks for; the other fields 0).  Works on a scene committed with device = -1. */
int rt_scene_plan_launch(const rt_scene *, rt_launch_config *out);
/* A render of a SHARD (shard_count > 1) hands its tiles to the waves deepest first: the few 100-segment paths that finish a launch
 * alone (a fixed ~1 ms, 12 % of a 1/8 shard of book-one) then start early and the launch ends on shallow tiles.  The order is learnt:
 * the first render of a view (camera, size, shard, depth) adds up the path lengths per tile beside its sums and sorts the tiles (in eight
 * steps of depth, ascending within a step); later
 * renders of the same view use that order as soon as it is complete (never waiting for it).  The image does not depend on the order
 * (the reference deals rows to threads in no particular order either, examples/book-one.rs:56-65).  A whole image (shard_count 1)
 * is always rendered in ascending order (neighbouring tiles share rays: faster there). */
#define RT_TILE_ORDER_ASCENDING 0
#define RT_TILE_ORDER_LEARNT 1   /* this render used a learnt order */
#define RT_TILE_ORDER_LEARNING 2 /* this render ran in ascending order and learns the order of its view */
/* the learnt order of the scene's current view, once complete (waits for the device): n owned tiles written to order_out (owned-tile
 * indices, first handed out first) and their summed path lengths to cost_out (indexed by owned tile); 0 when there is none */
int rt_scene_tile_order(rt_scene *, uint32_t *order_out, uint64_t *cost_out, int capacity);

/* ---- output: tone map + P3 text of the drivers (examples/book-one.rs:28-30,90-100):
 *      gamma 2, clamp high, truncate, NaN / negative -> 255 (Q13); rows top-down ---- */
void rt_tonemap_rgb8(const double *rgb, size_t n_pixels, uint8_t *out_rgb8);
int rt_write_ppm_p3(const char *path, const double *rgb, int width, int height);
/* the PNG path of examples/main.rs:105-135: RGBA8, alpha 255, put_pixel(x, height - 1 - y), channel =
 * (c.sqrt() * 255.0).min(255.0) as u8 (NaN -> 255 through f64::min, the cast saturates); non-interlaced,
 * uncompressed (stored deflate blocks) -- any PNG reader decodes the same pixels the `image` crate would write */
void rt_tonemap_png8(const double *rgb, size_t n_pixels, uint8_t *out_rgb8);
int rt_write_png_rgba8(const char *path, const double *rgb, int width, int height);

/* ---- inspection of the committed flat scene (tests, algorithmic-byte accounting) ---- */
typedef struct rt_scene_info {
    int n_prims;        /* sprites that can be hit (hoisted + BVH leaves) */
    int n_child_prims;  /* faces inside cube instances */
    int n_hoisted;      /* scene-spanning prims tested directly for every segment */
    int n_nodes;
    int max_depth;      /* deepest leaf */
    int n_materials, n_textures, n_xforms;
    int node_bytes, prim_bytes, material_bytes; /* bytes one traversal step / test reads */
    unsigned feature_mask;                      /* RT_FEAT_* present in the scene */
    size_t device_bytes;                        /* total resident bytes after upload */
    int n_list;         /* > 0: a small general scene walked as a LIST of this many leaf boxes, all tested in lock step by the
                         * wave, instead of the tree (the tree is still built and reported); nodes_visited then counts boxes */
} rt_scene_info;
int rt_scene_get_info(const rt_scene *, rt_scene_info *out);
/* copy of the flat BVH: 28 doubles per node
 * {lo0[3],hi0[3],lo1[3],hi1[3], child0,child1, culllo0[3],cullhi0[3],culllo1[3],cullhi1[3], -,-};
 * child >= 0 inner node index, child < 0 leaf ~prim; cull* = the binary32 culling boxes */
int rt_scene_copy_nodes(const rt_scene *, double *out, int max_nodes);
/* 64-bit FNV-1a over the committed flat scene (prims, transform chains, materials, textures, texels): identifies the
 * scene a checkpoint of rt_render_progressive belongs to */
int rt_scene_hash(const rt_scene *, uint64_t *out);
/* world-space AABB of prim i: {lo[3], hi[3]} */
int rt_scene_prim_bounds(const rt_scene *, int prim, double out[6]);
/* leaves of the acceleration structure: 1 = prim i is a leaf of its own; 6 = prim i is the head of a CUBE GROUP, one leaf that stands
 * for prims i .. i + 5, the six faces of one Cube::new (src/geometry.rs:254-286) whose culling boxes are the sides of one box (the
 * 400 floor boxes of examples/main.rs:161-201); 0 = prim i is one of the other five faces of a group */
int rt_scene_prim_group(const rt_scene *, int prim);

/* ---- device self-test used by the GPU parity tests: evaluates sqrt, div on the
 *      device for n inputs so the host can check they are correctly rounded ---- */
int rt_probe_device_math(int device, const double *a, const double *b, int n, double *out_sqrt_a, double *out_a_div_b);
/* ... and the kernels' transcendentals -- the host libm's functions restated for the device (csrc/rt_libm.h), standing
 * for `f64::ln` (src/volume.rs:59-60,81-82), `sin` (src/material.rs:238), `acos` and `atan2` (src/geometry.rs:35-39):
 * which = 0 log(a), 1 sin(a), 2 acos(a), 3 atan2(a, b); the GPU tests compare the results with glibc's bit for bit.
 * Also 4 cos(a), 5 pow(a, b): restated and pinned the same way (`f64::cos`, `powf`: src/material.rs:140-143), not called by the
 * kernels (csrc/rt_lane.h schlick_reflects says why) */
int rt_probe_device_libm(int device, int which, const double *a, const double *b, int n, double *out);

#define RT_FEAT_SPHERE_T 1u   /* translation-only sphere sprites */
#define RT_FEAT_GENERAL 2u    /* sprites with a general matrix / rectangles / cubes */
#define RT_FEAT_MEDIUM 4u     /* ConstantMedium */
#define RT_FEAT_TEXTURED 8u   /* checker / image textures (uv needed) */
#define RT_FEAT_LENS 16u
#define RT_FEAT_WIDE 64u           /* more than 32767 prims or nodes: 32-bit node references, two LDS words per stack entry */
#define RT_FEAT_MEDIUM_GENERAL 32u /* a ConstantMedium whose boundary is not a plain sphere under a pure translation */
#define RT_FEAT_MEDIUM_NESTED 256u  /* a ConstantMedium inside the boundary of another (up to three levels) */
#define RT_FEAT_DEEP_CHAIN 128u    /* more than four transform levels above a primitive (up to 15): served by the same kernel family */

#ifdef __cplusplus
}
#endif
#endif /* RT_MI355X_H */
