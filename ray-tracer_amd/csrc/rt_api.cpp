// rt_api.cpp -- the C ABI of librt_mi355x.so (include/rt_mi355x.h): scene
// recording, commit (flatten + upload), render launches, output helpers.
// There is no CPU rendering path here: without a HIP device rt_render* fail
// with RT_ERR_DEVICE.

#include "../../include/rt_mi355x.h"
#include "../../include/rt_rng.h"
#include "rt_host.h"
#include "rt_lds.h"
#include "rt_scene_priv.h"
#include "rt_types.h"

#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

extern "C" int rt_launch_render(const RtLaunch *L, unsigned features, int lens, int count, int ldsnodes, int blocks,
                                unsigned lds_bytes, void *stream);
extern "C" int rt_launch_reduce(const double *samples, double *tiles, int n_owned_tiles, int s_count, int first_pass, int last_pass,
                                int spp, int width, int height, int shard_index, int shard_count, int narrow_blocks,
                                unsigned long long *tile_cost, void *stream);
extern "C" int rt_launch_tile_order(unsigned long long *tile_cost, int n, int levels, unsigned int *order, void *stream);
extern "C" int rt_kernel_block_size(unsigned features);
extern "C" int rt_kernel_waves_per_simd(unsigned features);
extern "C" int rt_persistent_blocks(unsigned features, int lens, int count, int ldsnodes, unsigned lds_bytes, int *blocks_per_cu,
                                    int *n_cu);
extern "C" int rt_launch_unpack(const double *gathered, int tiles_per_shard, int shard_count, int width, int height,
                                double *image, void *stream);
extern "C" int rt_launch_probe_math(const double *a, const double *b, int n, double *out_sqrt, double *out_div, void *stream);
extern "C" int rt_launch_probe_libm(int which, const double *a, const double *b, int n, double *out, void *stream);

namespace {

thread_local std::string g_err;
int fail(int code, const std::string &msg) {
    g_err = msg;
    return code;
}
int hip_fail(hipError_t e, const char *what) {
    g_err = std::string(what) + ": " + hipGetErrorString(e);
    return RT_ERR_DEVICE;
}
#define HIP_TRY(call)                                   \
    do {                                                \
        hipError_t e_ = (call);                         \
        if (e_ != hipSuccess) return hip_fail(e_, #call); \
    } while (0)

template <class T>
int upload(const std::vector<T> &v, void **dptr, size_t *total) {
    *dptr = nullptr;
    if (v.empty()) return RT_OK;
    const size_t bytes = v.size() * sizeof(T);
    HIP_TRY(hipMalloc(dptr, bytes));
    HIP_TRY(hipMemcpy(*dptr, v.data(), bytes, hipMemcpyHostToDevice));
    *total += bytes;
    return RT_OK;
}

// f(index into the packed tiles, index into the row-major image) for every pixel of the shard inside the image
template <class F>
void for_each_shard_pixel(int width, int height, int shard_index, int shard_count, int n_owned, F &&f) {
    const int tx_n = (width + RT_TILE - 1) / RT_TILE;
    for (int k = 0; k < n_owned; ++k) {
        const int tile = shard_index + k * shard_count;
        const int tx = tile % tx_n, ty = tile / tx_n;
        for (int lane = 0; lane < RT_TILE_PIXELS; ++lane) {
            const int x = tx * RT_TILE + (lane & 7), y = ty * RT_TILE + (lane >> 3);
            if (x >= width || y >= height) continue;
            f(((size_t)k * RT_TILE_PIXELS + (size_t)lane) * 3, ((size_t)y * (size_t)width + (size_t)x) * 3);
        }
    }
}

} // namespace

extern "C" {

const char *rt_last_error(void) { return g_err.c_str(); }
#ifndef RT_SOURCE_HASH
#define RT_SOURCE_HASH "unknown"
#endif
#ifndef RT_KERNEL_HASH
#define RT_KERNEL_HASH "unknown"
#endif
const char *rt_version(void) { return "rt_mi355x 0.3 (gfx950, f64) kernels " RT_KERNEL_HASH " src " RT_SOURCE_HASH; }

int rt_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

void rt_mat4_identity(double out[16]) { rt::mat4_identity(out); }
void rt_mat4_translation(const double offset[3], double out[16]) { rt::mat4_translation(offset, out); }
void rt_mat4_rotation(double radians, const double axis[3], double out[16]) { rt::mat4_rotation(radians, axis, out); }
void rt_mat4_multiplied(const double self[16], const double other[16], double out[16]) { rt::mat4_multiplied(self, other, out); }
double rt_mat4_determinant(const double m[16]) { return rt::mat4_determinant(m); }
int rt_mat4_inversed(const double m[16], double out[16]) {
    return rt::mat4_inversed(m, out) ? RT_OK : fail(RT_ERR_INVALID, "singular matrix (det == 0): Mat4::inversed is None");
}

rt_scene *rt_scene_create(void) { return new rt_scene; }
void rt_scene_destroy(rt_scene *s) {
    if (!s) return;
    s->release_device();
    delete s;
}

static int check_open(rt_scene *s) {
    if (s->committed) return fail(RT_ERR_STATE, "scene is immutable after rt_scene_commit");
    return RT_OK;
}
// every recording call holds the scene's mutex (rt_scene_clone copies the description under it)
#define RT_RECORDING(s)                                      \
    if (!(s)) return fail(RT_ERR_INVALID, "null scene");     \
    std::lock_guard<std::mutex> recording_lock_((s)->mu);    \
    if (int e_ = check_open(s)) return e_
static bool tex_ok(const rt_scene *s, int t) { return t >= 0 && (size_t)t < s->ir.textures.size(); }

int rt_add_texture_solid(rt_scene *s, const double rgb[3]) {
    RT_RECORDING(s);
    if (!rgb) return fail(RT_ERR_INVALID, "rgb is null");
    rt::TextureIR t{};
    t.kind = RT_TEX_SOLID;
    for (int i = 0; i < 3; ++i) t.rgb[i] = rgb[i];
    s->ir.textures.push_back(t);
    return (int)s->ir.textures.size() - 1;
}
int rt_add_texture_checker(rt_scene *s, int black, int white) {
    RT_RECORDING(s);
    if (!tex_ok(s, black) || !tex_ok(s, white)) return fail(RT_ERR_INVALID, "checker: unknown texture id");
    rt::TextureIR t{};
    t.kind = RT_TEX_CHECKER;
    t.a = black;
    t.b = white;
    s->ir.textures.push_back(t);
    return (int)s->ir.textures.size() - 1;
}
int rt_add_texture_image_rgb8(rt_scene *s, const uint8_t *rgb, int w, int h) {
    RT_RECORDING(s);
    if (!rgb || w <= 0 || h <= 0) return fail(RT_ERR_INVALID, "image texture: null data or empty size");
    rt::TextureIR t{};
    t.kind = RT_TEX_IMAGE;
    t.w = w;
    t.h = h;
    t.texels.assign(rgb, rgb + (size_t)w * (size_t)h * 3);
    s->ir.textures.push_back(t);
    return (int)s->ir.textures.size() - 1;
}

static int add_material(rt_scene *s, uint32_t kind, int tex, double param, bool needs_tex) {
    RT_RECORDING(s);
    if (needs_tex && !tex_ok(s, tex)) return fail(RT_ERR_INVALID, "material: unknown texture id");
    s->ir.materials.push_back(rt::MaterialIR{kind, tex, param});
    return (int)s->ir.materials.size() - 1;
}
int rt_add_material_lambertian(rt_scene *s, int tex) { return add_material(s, RT_MAT_LAMBERTIAN, tex, 0.0, true); }
int rt_add_material_metal(rt_scene *s, int tex, double fuzz) { return add_material(s, RT_MAT_METAL, tex, fuzz, true); }
int rt_add_material_dielectric(rt_scene *s, double refractive) { return add_material(s, RT_MAT_DIELECTRIC, -1, refractive, false); }
int rt_add_material_diffuse_light(rt_scene *s, int tex) { return add_material(s, RT_MAT_DIFFUSE_LIGHT, tex, 0.0, true); }
int rt_add_material_isotropic(rt_scene *s, int tex) { return add_material(s, RT_MAT_ISOTROPIC, tex, 0.0, true); }

static int add_geometry_locked(rt_scene *s, rt::GeometryKind k, double a, double b, double c, int boundary) { // caller holds s->mu
    rt::GeometryIR g{};
    g.kind = k;
    g.p[0] = a;
    g.p[1] = b;
    g.p[2] = c;
    g.boundary = boundary;
    s->ir.geometries.push_back(g);
    return (int)s->ir.geometries.size() - 1;
}
static int add_geometry(rt_scene *s, rt::GeometryKind k, double a, double b, double c) {
    RT_RECORDING(s);
    return add_geometry_locked(s, k, a, b, c, -1);
}
int rt_add_geometry_sphere(rt_scene *s, double r) { return add_geometry(s, rt::GEO_SPHERE, r, 0, 0); }
int rt_add_geometry_rectangle(rt_scene *s, double w, double h) { return add_geometry(s, rt::GEO_RECTANGLE, w, h, 0); }
int rt_add_geometry_cube(rt_scene *s, double w, double h, double d) { return add_geometry(s, rt::GEO_CUBE, w, h, d); }
int rt_add_geometry_constant_medium(rt_scene *s, int boundary, double density) {
    RT_RECORDING(s);
    if (boundary < 0 || (size_t)boundary >= s->ir.geometries.size()) return fail(RT_ERR_INVALID, "medium: unknown boundary geometry");
    return add_geometry_locked(s, rt::GEO_MEDIUM, density, 0, 0, boundary);
}
int rt_add_geometry_transformed(rt_scene *s, int geometry, const double M[16]) {
    RT_RECORDING(s);
    if (geometry < 0 || (size_t)geometry >= s->ir.geometries.size()) return fail(RT_ERR_INVALID, "transformed: unknown geometry id");
    const int id = add_geometry_locked(s, rt::GEO_TRANSFORMED, 0, 0, 0, geometry);
    if (id < 0) return id;
    if (M)
        std::memcpy(s->ir.geometries[(size_t)id].M, M, sizeof(double) * 16);
    else
        rt::mat4_identity(s->ir.geometries[(size_t)id].M);
    return id;
}
int rt_add_geometry_bvh(rt_scene *s, const int *sprites, int n) {
    RT_RECORDING(s);
    if (n < 0 || (n > 0 && !sprites)) return fail(RT_ERR_INVALID, "bvh: bad sprite list");
    if (n == 0) return fail(RT_ERR_EMPTY, "BoundingVolumeHierarchyNode::new(vec![]) is None (src/optimize.rs:367-370)");
    for (int i = 0; i < n; ++i) {
        if (sprites[i] < 0 || (size_t)sprites[i] >= s->ir.sprites.size()) return fail(RT_ERR_INVALID, "bvh: unknown sprite id");
        if (s->ir.sprites[(size_t)sprites[i]].owned) return fail(RT_ERR_STATE, "bvh: sprite already moved into another node");
        for (int j = 0; j < i; ++j)
            if (sprites[j] == sprites[i]) return fail(RT_ERR_INVALID, "bvh: sprite listed twice");
    }
    const int id = add_geometry_locked(s, rt::GEO_BVH, 0, 0, 0, -1);
    if (id < 0) return id;
    s->ir.geometries[(size_t)id].children.assign(sprites, sprites + n);
    for (int i = 0; i < n; ++i) s->ir.sprites[(size_t)sprites[i]].owned = true;
    return id;
}

int rt_add_sprite(rt_scene *s, int geometry, int material, const double M[16]) {
    RT_RECORDING(s);
    if (geometry >= (int)s->ir.geometries.size()) return fail(RT_ERR_INVALID, "sprite: unknown geometry id");
    if (material >= (int)s->ir.materials.size()) return fail(RT_ERR_INVALID, "sprite: unknown material id");
    rt::SpriteIR sp{};
    sp.geometry = geometry < 0 ? -1 : geometry;
    sp.material = material < 0 ? -1 : material;
    if (M)
        std::memcpy(sp.M, M, sizeof sp.M);
    else
        rt::mat4_identity(sp.M);
    s->ir.sprites.push_back(sp);
    return (int)s->ir.sprites.size() - 1;
}

int rt_scene_commit(rt_scene *s, int device) {
    if (!s) return fail(RT_ERR_INVALID, "null scene");
    std::lock_guard<std::mutex> recording_lock(s->mu);
    if (s->committed) return fail(RT_ERR_STATE, "scene already committed");
    std::string err;
    int rc = rt::flatten_scene(s->ir, &s->flat, &err);
    if (rc != RT_OK) return fail(rc, err);
    if (device >= 0) {
        int n = 0;
        if (hipGetDeviceCount(&n) != hipSuccess || device >= n)
            return fail(RT_ERR_DEVICE, "no such HIP device; librt_mi355x has no CPU rendering path");
        HIP_TRY(hipSetDevice(device));
        s->device = device;
        size_t total = 0;
        if ((rc = upload(s->flat.nodes, &s->d_nodes, &total))) return rc;
        if ((rc = upload(s->flat.nodes_half, &s->d_nodes_half, &total))) return rc; // the tree with binary16 planes, for an LDS copy
        if ((rc = upload(s->flat.prim_meta, &s->d_prim_meta, &total))) return rc;
        if ((rc = upload(s->flat.prim_geo, &s->d_prim_geo, &total))) return rc;
        if ((rc = upload(s->flat.prim_extra, &s->d_prim_extra, &total))) return rc;
        if ((rc = upload(s->flat.xform_store, &s->d_xforms, &total))) return rc; // boxes in reverse, then the records
        if ((rc = upload(s->flat.materials, &s->d_materials, &total))) return rc;
        if ((rc = upload(s->flat.textures, &s->d_textures, &total))) return rc;
        if ((rc = upload(s->flat.image_blob, &s->d_blob, &total))) return rc;
        if ((rc = upload(s->flat.scene_blob, &s->d_scene_blob, &total))) return rc; // list mode: the records the kernels keep in LDS
        s->device_bytes = total;
    }
    s->committed = true;
    return RT_OK;
}

int rt_camera_perspective(rt_camera *out, const double eye[3], const double center[3], const double up[3], double fov,
                          double aspect, double focus, double lens) {
    if (!out || !eye || !center || !up) return fail(RT_ERR_INVALID, "camera: null argument");
    RtCameraD c;
    rt::camera_perspective(&c, eye, center, up, fov, aspect, focus, lens);
    for (int i = 0; i < 3; ++i) {
        out->eye[i] = c.eye[i];
        out->lower_left[i] = c.lower_left[i];
        out->horizontal[i] = c.horizontal[i];
        out->vertical[i] = c.vertical[i];
    }
    out->lens_radius = c.lens_radius;
    return RT_OK;
}

static int tiles_x_of(int w) { return (w + RT_TILE - 1) / RT_TILE; }
static int tiles_y_of(int h) { return (h + RT_TILE - 1) / RT_TILE; }

int rt_shard_tile_count(int width, int height, int shard_index, int shard_count) {
    if (width <= 0 || height <= 0 || shard_count <= 0 || shard_index < 0 || shard_index >= shard_count)
        return fail(RT_ERR_INVALID, "bad image size or shard");
    const long total = (long)tiles_x_of(width) * tiles_y_of(height);
    if (shard_index >= total) return 0;
    return (int)((total - shard_index + shard_count - 1) / shard_count);
}

static int check_params(const rt_scene *s, const rt_camera *cam, const rt_render_params *p) {
    if (!s || !cam || !p) return fail(RT_ERR_INVALID, "null argument");
    if (!s->committed) return fail(RT_ERR_STATE, "rt_scene_commit has not been called");
    if (s->device < 0) return fail(RT_ERR_DEVICE, "scene was committed host-only (device = -1); librt_mi355x has no CPU rendering path");
    if (p->width <= 0 || p->height <= 0 || p->spp <= 0 || p->max_depth < 0) return fail(RT_ERR_INVALID, "bad width/height/spp/max_depth");
    if (p->shard_count <= 0 || p->shard_index < 0 || p->shard_index >= p->shard_count) return fail(RT_ERR_INVALID, "bad shard");
    if ((uint64_t)p->width * (uint64_t)p->height * (uint64_t)p->spp >= (1ull << 40))
        return fail(RT_ERR_INVALID, "width*height*spp must stay below 2^40 sample streams (include/rt_rng.h)");
    return RT_OK;
}

static void fill_launch(const rt_scene *s, const rt_camera *cam, const rt_render_params *p, int n_owned, RtLaunch *L) {
    std::memset(L, 0, sizeof *L);
    L->nodes = (const RtNode *)s->d_nodes;
    L->prim_meta = (const RtPrimMeta *)s->d_prim_meta;
    L->prim_geo = (const RtPrimGeo *)s->d_prim_geo;
    L->prim_extra = (const RtPrimExtra *)s->d_prim_extra;
    L->xforms = s->d_xforms ? (const RtXform *)((const unsigned char *)s->d_xforms + s->flat.xform_store_offset()) : nullptr;
    L->materials = (const RtMaterial *)s->d_materials;
    L->xforms_global = L->xforms;
    L->textures = (const RtTexture *)s->d_textures;
    L->image_blob = (const uint8_t *)s->d_blob;
    L->n_nodes = (int)s->flat.nodes.size();
    L->n_list = s->flat.n_list;
    L->stack_entries = s->flat.n_list ? s->flat.n_list - 1 : std::min(RT_STACK_DEPTH, s->flat.max_depth + 1);
    L->root = s->flat.root;
    L->n_hoisted = s->flat.n_hoisted;
    L->world_mid = s->flat.world_mid ? 1 : 0;
    L->n_prims = s->flat.n_leaf_prims;
    for (int i = 0; i < 3; ++i) {
        L->cam.eye[i] = cam->eye[i];
        L->cam.lower_left[i] = cam->lower_left[i];
        L->cam.horizontal[i] = cam->horizontal[i];
        L->cam.vertical[i] = cam->vertical[i];
    }
    L->cam.lens_radius = cam->lens_radius;
    L->width = p->width;
    L->height = p->height;
    L->spp = p->spp;
    L->max_depth = p->max_depth;
    L->seed_mix = rt_mix64(p->seed);
    L->tiles_x = tiles_x_of(p->width);
    L->tiles_y = tiles_y_of(p->height);
    L->shard_index = p->shard_index;
    L->shard_count = p->shard_count;
    L->n_owned_tiles = n_owned;
    L->job_counter = nullptr; // set per render slot
}

static unsigned kernel_features(const rt_scene *s) {
    unsigned f = 0;
    if (s->flat.feature_mask & RT_FEAT_GENERAL) f |= 1u;
    if (s->flat.feature_mask & RT_FEAT_MEDIUM) f |= 2u;
    if (s->flat.feature_mask & RT_FEAT_TEXTURED) f |= 4u;
    if (s->flat.feature_mask & (RT_FEAT_MEDIUM_GENERAL | RT_FEAT_DEEP_CHAIN)) f |= 8u; // + 2: that family is compiled with media
    if (s->flat.feature_mask & RT_FEAT_DEEP_CHAIN) f |= 2u | 4u;
    if (s->flat.feature_mask & RT_FEAT_MEDIUM_NESTED) f |= 16u | 8u | 4u | 2u | 1u; // the family with the nested evaluation (MEDIUM = 3)
    if (s->flat.wide) f |= 1u; // 32-bit references exist in the general kernel families only
    return f;
}

// bytes of per-sample workspace ONE render slot of this scene may hold; more samples are rendered in several passes.
// rt_scene_set_workspace_limit, else RT_SAMPLE_WORKSPACE_MB, else the smaller of 32 GiB and a quarter of the device's memory
// (an MI355X has 288 GB: 32 GiB; the headline 1200x800x500 render needs 15.36 GB and runs in one pass).  The workspace is
// allocated at the size a render needs (never the cap), grows on demand, is reused by later renders and is given back by
// rt_scene_trim / rt_scene_destroy.
static size_t sample_workspace_cap(rt_scene *s) {
    if (s->workspace_limit) return s->workspace_limit;
    const char *e = std::getenv("RT_SAMPLE_WORKSPACE_MB");
    if (e && *e) return (size_t)std::strtoull(e, nullptr, 10) << 20;
    if (!s->workspace_default) { // asked once per scene: hipMemGetInfo is not free and may wait for the device
        size_t cap = (size_t)32 << 30, free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && total_b / 4 < cap) cap = total_b / 4;
        s->workspace_default = cap;
    }
    return s->workspace_default;
}

// the device error word of a slot (rt_lds.h RT_DEV_*): read, cleared, turned into RT_ERR_DEVICE.  The slot's work has finished.
static int check_device_status(rt_scene::RenderSlot &sl) {
    if (!sl.d_job_counter) return RT_OK;
    unsigned st = 0;
    HIP_TRY(hipMemcpy(&st, (const unsigned *)sl.d_job_counter + 16, sizeof st, hipMemcpyDeviceToHost));
    if (st == RT_DEV_OK) return RT_OK;
    (void)hipMemset((unsigned *)sl.d_job_counter + 16, 0, sizeof st);
    std::string what;
    if (st & RT_DEV_ERR_LDS_LAYOUT) what += " render_kernel was launched with fewer LDS bytes than its layout needs (rt_lds.h) and refused to run;";
    if (st & RT_DEV_ERR_WATCHDOG) what += " a wave made no progress within its trip bound (counting build watchdog);";
    return fail(RT_ERR_DEVICE, "device error word " + std::to_string(st) + ":" + what + " the image is incomplete");
}

#define RT_TILE_ORDER_LEVELS 8
// what a learnt tile order belongs to: the view (never 0)
static uint64_t view_key(const rt_camera *cam, const rt_render_params *p) {
    uint64_t h = 0xCBF29CE484222325ull;
    auto eat = [&](const void *q, size_t n) {
        for (size_t i = 0; i < n; ++i) h = (h ^ ((const unsigned char *)q)[i]) * 0x100000001B3ull;
    };
    eat(cam, sizeof *cam); // thirteen doubles, no padding
    const int v[5] = {p->width, p->height, p->shard_index, p->shard_count, p->max_depth};
    eat(v, sizeof v);
    return h | 1ull;
}

// What a render of this scene launches: the form of its kernel family, what lives in the workgroup's LDS, the launch's dynamic LDS.
// Pure host logic (no HIP call): render_range uses it, rt_scene_plan_launch exposes it (tests on machines without a GPU).
struct LaunchPlan {
    unsigned block = 0, entry_bytes = 0, table = 0, front = 0, in_lds = 0, swap_cap = 0, lds_bytes = 0, groups_per_cu = 0;
    int swap = 0, list = 0, wide = 0, reclds = 0, ldsnodes = 0, half = 0, lds_mode = 0;
};
static int plan_launch(const rt_scene *s, unsigned feat, int stack_entries_in, bool half_nodes_uploaded, LaunchPlan *out) {
    LaunchPlan P;
    const unsigned stack_entries = (unsigned)stack_entries_in;
    // dynamic LDS (rt_lds.h: the one layout host and kernel share): the traversal stack, the waves' job state, a copy of the
    // node array when the family's full occupancy still fits the CU's 160 KiB with it (book-one: 31 KB of nodes + 12 KB of
    // stack), and the swap-at-shade queues with as many entries as are left (16 at least)
    const unsigned block = P.block = (unsigned)rt_kernel_block_size(feat);
    const int wide = P.wide = s->flat.wide ? 1 : 0;
    const unsigned entry_bytes = P.entry_bytes = s->flat.n_list > 0 ? 2u : (wide ? 8u : 4u); // (rt_kernels.hip StackOf)
    const unsigned node_bytes = (unsigned)(s->flat.nodes.size() * sizeof(RtNode));
    const char *no_lds = std::getenv("RT_NO_LDS_NODES");
    const bool lds_off = no_lds && *no_lds == '1';
    // RT_SWAP=0 selects the kernels without the queues (A/B runs)
    const char *swap_env = std::getenv("RT_SWAP");
    const int swap = P.swap = !(swap_env && *swap_env == '0') || (feat & 16u) != 0u; // (the nested-media family exists with the queues only)
    // keep the kernel family's full occupancy resident: that many workgroups per CU share its LDS
    const unsigned groups_per_cu = P.groups_per_cu = std::max(1u, (unsigned)rt_kernel_waves_per_simd(feat) * 256u / block);
    const unsigned lds_share = (RT_LDS_PER_CU / groups_per_cu) & ~(RT_LDS_GRANULE - 1u);
    const int list = P.list = s->flat.n_list > 0; // the box list (< 1 KB) always lives in LDS
    const unsigned min_cap = swap ? (block >= 512u ? (unsigned)RT_SWAP_CAP : 32u) : 0u; // the node copy must leave room for this
    // the families with media / textures keep the log table at the front; a small general scene's records follow it when they fit the
    // workgroup's share beside the stack, the box list or node array and the smallest queues (rtl::rec_at<true>: Cornell box 138 -> 126 ms)
    const unsigned table = P.table = rt_lds_front_bytes((feat & ~1u) != 0u);
    const unsigned blob_bytes = (unsigned)s->flat.scene_blob.size();
    // (a small TREE scene -- the host packed a blob for it, FlatScene::scene_blob -- takes that form when its node array fits as well:
    // the lean general family and the one with sphere media / textures have kernels for it)
    const char *no_rec = std::getenv("RT_NO_LDS_RECORDS"); // A/B
    const bool tree_form = !list && !wide && node_bytes > 0u && !(feat & (8u | 16u)) && (feat & 1u) != 0u && !lds_off;
    const int reclds = P.reclds = (list || tree_form) && swap && blob_bytes > 0u && !(no_rec && *no_rec == '1') &&
                                  rt_lds_layout(stack_entries, block, entry_bytes, node_bytes, min_cap, table + rt_lds_scene_room(blob_bytes)).total <= lds_share;
    const unsigned front = P.front = table + (reclds ? rt_lds_scene_room(blob_bytes) : 0u);
    const char *no_half = std::getenv("RT_NO_HALF_NODES");
    const char *want_half = std::getenv("RT_HALF_NODES"); // 1: the binary16 form wherever it exists, also when the binary32 nodes would fit (sweeps, A/B)
    const unsigned half_bytes = (unsigned)(s->flat.nodes_half.size() * sizeof(RtNodeH));
    // the family with sphere media / textures has kernels for a tree with binary16 planes (RtNodeH, half the bytes); a small tree with its
    // records in LDS keeps binary32 nodes there too (!reclds)
    const bool family_has_half = (feat & ~1u) != 0u && !(feat & (8u | 16u));
    const bool half_possible = !list && !reclds && !wide && swap && half_bytes > 0u && family_has_half && !lds_off && !(no_half && *no_half == '1') &&
                               half_nodes_uploaded && rt_lds_layout(stack_entries, block, entry_bytes, half_bytes, min_cap, front).total <= lds_share;
    int ldsnodes = list || (!wide && node_bytes > 0u && !lds_off && !(half_possible && want_half && *want_half == '1') &&
                            rt_lds_layout(stack_entries, block, entry_bytes, node_bytes, min_cap, front).total <= lds_share);
    // the binary32 nodes do not fit: the same tree with binary16 planes may (the book-two cover: 1406 nodes over cube groups, 45 KB beside
    // a 56 KB stack).  RT_NO_HALF_NODES=1: never.
    const int half = P.half = !ldsnodes && half_possible;
    if (half) ldsnodes = 1;
    if (reclds && !list && !ldsnodes) return fail(RT_ERR_DEVICE, "internal: a tree scene's records in LDS without its nodes");
    if (half && !family_has_half) return fail(RT_ERR_DEVICE, "internal: binary16 nodes for a kernel family without that form");
    P.ldsnodes = ldsnodes;
    const unsigned in_lds = P.in_lds = ldsnodes ? (half ? half_bytes : node_bytes) : 0u;
    unsigned swap_cap = swap ? rt_swap_cap_that_fits(stack_entries, block, entry_bytes, in_lds, groups_per_cu, front) : 0u;
    if (const char *lim = std::getenv("RT_SWAP_CAP_LIMIT")) // A/B runs: fewer entries per class queue than would fit (families of 256 threads)
        if (block < 512u && std::atoi(lim) >= 16) swap_cap = std::min(swap_cap, (unsigned)std::atoi(lim)) & ~1u;
    P.swap_cap = swap_cap;
    const RtLdsLayout lay = rt_lds_layout(stack_entries, block, entry_bytes, in_lds, swap ? rt_swap_cap_effective(block, swap_cap) : 0u, front);
    if (!rt_lds_layout_aligned(lay)) return fail(RT_ERR_DEVICE, "internal: misaligned LDS layout");
    P.lds_bytes = lay.total;
    if (P.lds_bytes > RT_LDS_PER_CU) return fail(RT_ERR_UNSUPPORTED, "the scene's traversal stack does not fit a CU's LDS");
    P.lds_mode = (ldsnodes ? 1 : 0) | (swap ? 2 : 0) | (wide ? 4 : 0) | (list ? 8 : 0) | (reclds ? 16 : 0) | (half ? 32 : 0);
    *out = P;
    return RT_OK;
}

// Render samples [s_begin, s_end) of every owned pixel.  accumulate: the tile buffer already holds the
// raw sums of samples [0, s_begin) and is continued in sample order; finalize: divide by spp at the end.
static int render_range(rt_scene *s, const rt_camera *cam, const rt_render_params *p, int s_begin, int s_end, bool accumulate,
                        bool finalize, void *d_tiles_out, void *d_counters, void *stream) {
    if (int e = check_params(s, cam, p)) return e;
    if (!d_tiles_out) return fail(RT_ERR_INVALID, "null output buffer");
    if (s_begin < 0 || s_end > p->spp || s_begin >= s_end) return fail(RT_ERR_INVALID, "bad sample range");
    HIP_TRY(hipSetDevice(s->device));
    const int n_owned = rt_shard_tile_count(p->width, p->height, p->shard_index, p->shard_count);
    if (n_owned < 0) return n_owned;
    hipStream_t st = (hipStream_t)stream;
    std::lock_guard<std::mutex> lock(s->mu);
    // slot 0 unless its last render is still running (then the two renders overlap on the device)
    int slot_index = 0;
    if (s->slots[0].used && hipEventQuery(s->slots[0].done) == hipErrorNotReady) slot_index = s->last_slot == 0 ? 1 : 0;
    rt_scene::RenderSlot &sl = s->slots[slot_index];
    s->last_slot = slot_index;
    sl.events_used = 0;
    sl.deferred = false;
    s->timed = true;
    if (n_owned == 0) return RT_OK;
    const size_t tile_doubles = (size_t)n_owned * RT_TILE_PIXELS * 3;
    if (p->max_depth == 0) {
        // color(ray, world, 0) is black before anything is traced (src/render.rs:6-8): sums stay as they are
        if (!accumulate) HIP_TRY(hipMemsetAsync(d_tiles_out, 0, tile_doubles * sizeof(double), st));
        return RT_OK;
    }
    // pass size: as many samples per pixel as the workspace cap allows (32-byte record per sample)
    const size_t bytes_per_spp = (size_t)n_owned * RT_TILE_PIXELS * 4 * sizeof(double);
    const int n_spp = s_end - s_begin;
    int chunk = (int)std::min<size_t>((size_t)n_spp, std::max<size_t>(1, sample_workspace_cap(s) / bytes_per_spp));
    // slot indices are 32-bit
    chunk = (int)std::min<size_t>((size_t)chunk, (size_t)0xFFFFFFFFu / ((size_t)n_owned * RT_TILE_PIXELS));
    if (chunk < 1) return fail(RT_ERR_INVALID, "image too large for one shard");
    const size_t need = bytes_per_spp * (size_t)chunk;
    if (!sl.done) HIP_TRY(hipEventCreateWithFlags(&sl.done, hipEventDisableTiming));
    if (!sl.d_job_counter) { // word 0: the job counter (zeroed per launch); word 16: the device error word (sticky, rt_lds.h RT_DEV_*)
        HIP_TRY(hipMalloc(&sl.d_job_counter, 256));
        HIP_TRY(hipMemset(sl.d_job_counter, 0, 256));
    }
    if (sl.used) HIP_TRY(hipStreamWaitEvent(st, sl.done, 0)); // the slot's previous render (any stream) has to be through
    if (need > sl.samples_bytes) {
        if (sl.d_samples) {
            HIP_TRY(hipDeviceSynchronize()); // an earlier render (any stream) may still read the old workspace
            HIP_TRY(hipFree(sl.d_samples));
            sl.d_samples = nullptr;
            sl.samples_bytes = 0;
        }
        HIP_TRY(hipMalloc(&sl.d_samples, need));
        sl.samples_bytes = need;
    }
    const bool count = (p->flags & RT_FLAG_COUNTERS) && d_counters;
    const unsigned feat = kernel_features(s);
    const int lens = cam->lens_radius != 0.0;
    RtLaunch L;
    fill_launch(s, cam, p, n_owned, &L);
    // which form of the kernel family, what lives in LDS and how large the launch's dynamic LDS is: plan_launch (above), the ONE place
    // that decides it -- rt_scene_plan_launch answers the same question without a device
    LaunchPlan P;
    if (int e = plan_launch(s, feat, L.stack_entries, s->d_nodes_half != nullptr, &P)) return e;
    const unsigned block = P.block;
    const int swap = P.swap, reclds = P.reclds, ldsnodes = P.ldsnodes, half = P.half;
    const unsigned swap_cap = P.swap_cap, lds_bytes = P.lds_bytes;
    if (reclds) { // these five fields carry the arrays' byte offsets in the LDS instead of addresses
        const uint32_t *off = s->flat.scene_blob_off;
        L.scene_blob = (const unsigned char *)s->d_scene_blob;
        L.scene_bytes = (unsigned)s->flat.scene_blob.size();
        L.scene_lds_off = P.table;
        if (RT_LIST_LDS_ARRAYS & 1) L.xforms = (const RtXform *)(uintptr_t)(P.table + off[0]);
        if (RT_LIST_LDS_ARRAYS & 2) L.prim_geo = (const RtPrimGeo *)(uintptr_t)(P.table + off[1]);
        if (RT_LIST_LDS_ARRAYS & 4) L.prim_meta = (const RtPrimMeta *)(uintptr_t)(P.table + off[2]);
        if (RT_LIST_LDS_ARRAYS & 8) L.prim_extra = (const RtPrimExtra *)(uintptr_t)(P.table + off[3]);
        if (RT_LIST_LDS_ARRAYS & 16) L.materials = (const RtMaterial *)(uintptr_t)(P.table + off[4]);
    }
    if (half) L.nodes = (const RtNode *)s->d_nodes_half; // RtNodeH records: the kernel's HALF instantiation reads them as such
    L.swap_cap = (int)swap_cap;
    L.lds_bytes = lds_bytes;
#if defined(RT_TEST_HOOKS) // librt_mi355x_testhooks.so only (Makefile): the shipped library reads no RT_TEST_* variable
    if (const char *t = std::getenv("RT_TEST_LDS_SHORT")) // claim fewer bytes than the layout needs -> the kernel must refuse
        if (*t == '1') L.lds_bytes = lds_bytes - 64u;
#endif
    const int lds_mode = P.lds_mode;
    int per_cu = 0, n_cu = 0, rc = 0;
    const unsigned occ_key = feat | (lens ? 32u : 0u) | (count ? 64u : 0u) | ((unsigned)lds_mode << 7); // feat uses bits 0-4
    if (s->occ_key == occ_key && s->occ_lds == lds_bytes) {
        per_cu = s->occ_per_cu;
        n_cu = s->occ_n_cu;
    } else {
        rc = rt_persistent_blocks(feat, lens, count, lds_mode, lds_bytes, &per_cu, &n_cu);
        if (rc != 0) return hip_fail((hipError_t)rc, "occupancy query");
        if (per_cu < 1) // never clamp silently: a persistent grid sized on a wrong occupancy runs at half speed
            return fail(RT_ERR_DEVICE, "occupancy query says the render kernel does not fit a CU with " + std::to_string(lds_bytes) +
                                           " bytes of LDS per workgroup");
        s->occ_key = occ_key;
        s->occ_lds = lds_bytes;
        s->occ_per_cu = per_cu;
        s->occ_n_cu = n_cu;
    }
    L.samples = (double *)sl.d_samples;
    L.job_counter = (unsigned int *)sl.d_job_counter;
    L.status = (unsigned int *)sl.d_job_counter + 16;
    // counting build: a wave may go this many times round its loop without finishing or starting a segment.  Every trip
    // advances >= 1 of 64 lanes by one step of a traversal of <= nodes + prims steps, or drains a queue: 1024 x that is far
    // beyond anything a healthy launch does (book-one: < 100 trips between two shade blocks)
    L.watchdog_trips = 0u;
    if (count) {
        L.watchdog_trips = 1024u * (unsigned)(s->flat.nodes.size() + s->flat.prim_meta.size() + 64u);
#if defined(RT_TEST_HOOKS)
        if (const char *t = std::getenv("RT_TEST_WATCHDOG_TRIPS")) L.watchdog_trips = (unsigned)std::strtoul(t, nullptr, 10);
#endif
    }
    L.counters = count ? (RtCounters *)d_counters : nullptr;
    // hand-out order of the owned tiles (include/rt_mi355x.h RT_TILE_ORDER_*): shards only, learnt once per view
    int order_mode = RT_TILE_ORDER_ASCENDING;
    uint64_t order_key = 0;
    rt_scene::TileOrder &to = s->tile_order;
    // costs are compared in eight steps of the largest one: tiles of about the same depth keep their ascending order, in which
    // neighbouring tiles share the rays' neighbourhoods (the exact order costs a LONG launch more than its end gains: a 1/8 shard of
    // 3840x2160x2000 266 -> 271 ms, in eight steps 266 -> 266; the short shards gain the same either way, docs/experiments.md 1.4)
    int order_levels = RT_TILE_ORDER_LEVELS;
    bool order_whole = false;
#if defined(RT_TEST_HOOKS) // experiments (tools/tile_order_probe.py)
    if (const char *t = std::getenv("RT_TEST_TILE_ORDER_LEVELS")) order_levels = std::atoi(t);
    if (const char *t = std::getenv("RT_TEST_TILE_ORDER_WHOLE")) order_whole = *t == '1';
#endif
    // (sample ranges of a progressive render are renders of the same view: the first range learns, the later ones use the order)
    if ((p->shard_count > 1 || order_whole) && !(p->flags & RT_FLAG_ASCENDING_TILES) && n_owned >= 64 && n_owned <= 65536) {
        order_key = view_key(cam, p);
        if (to.key == order_key && to.n == n_owned) {
            if (!to.complete && hipEventQuery(to.ready) == hipSuccess) to.complete = true;
            if (to.complete) {
                L.tile_order = (const unsigned int *)to.d_order;
                order_mode = RT_TILE_ORDER_LEARNT;
            } // else: the render that learns it is still in flight -- ascending, and no second learner
        } else {
            if (n_owned > to.capacity) {
                HIP_TRY(hipDeviceSynchronize()); // an earlier render may still read the old table
                if (to.d_cost) HIP_TRY(hipFree(to.d_cost));
                if (to.d_order) HIP_TRY(hipFree(to.d_order));
                to.d_cost = to.d_order = nullptr;
                to.capacity = 0;
                HIP_TRY(hipMalloc(&to.d_cost, ((size_t)n_owned + 1) * sizeof(uint64_t))); // + the largest cost (tile_top_kernel)
                HIP_TRY(hipMalloc(&to.d_order, (size_t)n_owned * sizeof(uint32_t)));
                to.capacity = n_owned;
            }
            if (!to.ready) HIP_TRY(hipEventCreateWithFlags(&to.ready, hipEventDisableTiming));
            to.key = 0; // no view owns the table until this render's `ready` has been recorded
            to.complete = false;
            order_mode = RT_TILE_ORDER_LEARNING;
        }
    }
    const bool learning = order_mode == RT_TILE_ORDER_LEARNING;
    const int n_pass = (n_spp + chunk - 1) / chunk;
    int last_blocks = 0;
    while ((int)sl.events.size() < 2 * n_pass) {
        hipEvent_t e;
        HIP_TRY(hipEventCreate(&e));
        sl.events.push_back(e);
    }
    for (int pass = 0; pass < n_pass; ++pass) {
        L.s0 = s_begin + pass * chunk;
        L.s_count = std::min(chunk, s_end - L.s0);
        // job size: at most RT_JOB_SPP_MAX samples per pixel, smaller when the pass is small so that
        // every resident wave still draws >= ~32 jobs (end-of-launch tail <= ~3 %)
        {
            const long long waves = (long long)per_cu * n_cu * (block / 64);
            const long long want_jobs = waves * 32;
            long long js = ((long long)L.s_count * n_owned + want_jobs - 1) / want_jobs;
            L.job_spp = (int)std::max<long long>(1, std::min<long long>(RT_JOB_SPP_MAX, js));
        }
        L.jobs_per_tile = (L.s_count + L.job_spp - 1) / L.job_spp;
        const long long n_jobs = (long long)L.jobs_per_tile * n_owned;
        if (n_jobs > 0x7FFFFFFFll) return fail(RT_ERR_INVALID, "too many jobs in one pass");
        L.n_jobs = (int)n_jobs;
        const long long waves_per_block = block / 64;
        const long long want = (n_jobs + waves_per_block - 1) / waves_per_block;
        const int blocks = (int)std::min<long long>((long long)per_cu * n_cu, want > 0 ? want : 1);
        last_blocks = blocks;
        HIP_TRY(hipMemsetAsync(sl.d_job_counter, 0, sizeof(unsigned int), st));
        HIP_TRY(hipEventRecord(sl.events[(size_t)(2 * pass)], st));
        rc = rt_launch_render(&L, feat, lens, count, lds_mode, blocks, lds_bytes, stream);
        if (rc != 0) return hip_fail((hipError_t)rc, "render_kernel launch");
        HIP_TRY(hipEventRecord(sl.events[(size_t)(2 * pass + 1)], st));
        sl.events_used = 2 * (pass + 1);
        // RT_FLAG_DEFERRED_OUTPUT: the LAST pass's sums run on the scene's own stream behind the render kernel, so the caller's
        // stream is free for its next render at once (reduce_kernel is bound by HBM, render_kernel by the VALUs: they overlap);
        // the output is complete where rt_render_wait_output puts its wait
        const bool defer = (p->flags & RT_FLAG_DEFERRED_OUTPUT) != 0u && pass == n_pass - 1;
        hipStream_t rs = st;
        if (defer) {
            if (!s->post_stream) HIP_TRY(hipStreamCreateWithFlags(&s->post_stream, hipStreamNonBlocking));
            HIP_TRY(hipStreamWaitEvent(s->post_stream, sl.events[(size_t)(2 * pass + 1)], 0));
            rs = s->post_stream;
        }
        // deferred: ONE 256-thread group per CU (grid-stride) = one wave per SIMD = the 32 VGPRs that four render waves of 120 leave
        // free, so the next render_kernel's two workgroups per CU are resident beside it (book-one 1200x800x500: 64.75 ms per step
        // against 65.85 with the sums on the render stream; two groups per CU, or a reduce_kernel of 36 VGPRs: 65.2)
        if (learning && pass == 0) {
            // the table is written behind EVERY earlier render of this scene (a reader of the old order, or its learner)
            for (rt_scene::RenderSlot &o : s->slots)
                if (o.used && o.done) HIP_TRY(hipStreamWaitEvent(rs, o.done, 0));
            HIP_TRY(hipMemsetAsync(to.d_cost, 0, (size_t)n_owned * sizeof(uint64_t), rs));
        }
        rc = rt_launch_reduce(L.samples, (double *)d_tiles_out, n_owned, L.s_count, pass == 0 && !accumulate,
                              pass == n_pass - 1 && finalize, p->spp, p->width,
                              p->height, p->shard_index, p->shard_count, defer ? n_cu : 0,
                              learning ? (unsigned long long *)to.d_cost : nullptr, (void *)rs);
        if (rc != 0) return hip_fail((hipError_t)rc, "reduce_kernel launch");
        if (learning && pass == n_pass - 1) {
            rc = rt_launch_tile_order((unsigned long long *)to.d_cost, n_owned, order_levels, (unsigned int *)to.d_order, (void *)rs);
            if (rc != 0) return hip_fail((hipError_t)rc, "tile_order_kernel launch");
            HIP_TRY(hipEventRecord(to.ready, rs));
            to.key = order_key;
            to.n = n_owned;
        }
        if (defer) {
            HIP_TRY(hipEventRecord(sl.done, rs));
            sl.used = true;
            sl.deferred = true;
        }
    }
    if (!sl.deferred) {
        HIP_TRY(hipEventRecord(sl.done, st));
        sl.used = true;
    }
    rt_launch_config &lc = s->last_launch;
    lc.blocks = last_blocks;
    lc.block_threads = (int)block;
    lc.lds_bytes = lds_bytes;
    lc.blocks_per_cu = per_cu;
    lc.n_cu = n_cu;
    lc.passes = n_pass;
    lc.n_jobs = L.n_jobs;
    lc.job_spp = L.job_spp;
    lc.kernel_features = feat;
    lc.lds_nodes = ldsnodes ? (half ? 2 : 1) : 0; // 2: as RtNodeH (binary16 planes)
    lc.swap = swap;
    lc.workspace_bytes = need;
    lc.swap_cap = (int)swap_cap;
    lc.waves_per_simd = rt_kernel_waves_per_simd(feat);
    lc.tile_order = order_mode;
    lc.records_in_lds = reclds;
    return RT_OK;
}

int rt_render_tiles_device(rt_scene *s, const rt_camera *cam, const rt_render_params *p, void *d_tiles_out, void *d_counters,
                           void *stream) {
    if (!p) return fail(RT_ERR_INVALID, "null argument");
    return render_range(s, cam, p, 0, p->spp, false, true, d_tiles_out, d_counters, stream);
}

int rt_render_progressive(rt_scene *s, const rt_camera *cam, const rt_render_params *p, int s_begin, int s_end, double *sums) {
    if (int e = check_params(s, cam, p)) return e;
    if (!sums) return fail(RT_ERR_INVALID, "null sums buffer");
    if (s_begin < 0 || s_end > p->spp || s_begin >= s_end) return fail(RT_ERR_INVALID, "bad sample range");
    std::lock_guard<std::mutex> render_lock(s->render_mu);
    HIP_TRY(hipSetDevice(s->device));
    const int n_owned = rt_shard_tile_count(p->width, p->height, p->shard_index, p->shard_count);
    if (n_owned <= 0) return n_owned;
    const size_t n_doubles = (size_t)n_owned * RT_TILE_PIXELS * 3;
    std::vector<double> host(n_doubles, 0.0);
    auto each_pixel = [&](auto &&f) { for_each_shard_pixel(p->width, p->height, p->shard_index, p->shard_count, n_owned, f); };
    const bool accumulate = s_begin > 0;
    if (accumulate) each_pixel([&](size_t h, size_t g) {
        host[h] = sums[g];
        host[h + 1] = sums[g + 1];
        host[h + 2] = sums[g + 2];
    });
    double *d_out = nullptr;
    HIP_TRY(hipMalloc((void **)&d_out, n_doubles * sizeof(double)));
    int rc = RT_OK;
    hipError_t e = hipMemcpy(d_out, host.data(), n_doubles * sizeof(double), hipMemcpyHostToDevice);
    if (e != hipSuccess) rc = hip_fail(e, "sums upload");
    rt_render_params q = *p;
    q.flags &= ~RT_FLAG_COUNTERS;
    if (rc == RT_OK) rc = render_range(s, cam, &q, s_begin, s_end, accumulate, false, d_out, nullptr, nullptr);
    if (rc == RT_OK && (e = hipStreamSynchronize(nullptr)) != hipSuccess) rc = hip_fail(e, "render_kernel execution");
    if (rc == RT_OK) {
        std::lock_guard<std::mutex> lock(s->mu);
        rc = check_device_status(s->slots[s->last_slot]);
    }
    if (rc == RT_OK && (e = hipMemcpy(host.data(), d_out, n_doubles * sizeof(double), hipMemcpyDeviceToHost)) != hipSuccess)
        rc = hip_fail(e, "sums download");
    (void)hipFree(d_out);
    if (rc != RT_OK) return rc;
    each_pixel([&](size_t h, size_t g) {
        sums[g] = host[h];
        sums[g + 1] = host[h + 1];
        sums[g + 2] = host[h + 2];
    });
    return RT_OK;
}

int rt_last_kernel_ms(rt_scene *s, float *ms) {
    if (!s || !ms) return fail(RT_ERR_INVALID, "null argument");
    std::lock_guard<std::mutex> lock(s->mu);
    if (!s->timed) return fail(RT_ERR_STATE, "no render has been launched on this scene");
    HIP_TRY(hipSetDevice(s->device));
    float total = 0.0f;
    const rt_scene::RenderSlot &sl = s->slots[s->last_slot];
    for (int i = 0; i + 1 < sl.events_used; i += 2) {
        float t = 0.0f;
        HIP_TRY(hipEventSynchronize(sl.events[(size_t)i + 1]));
        HIP_TRY(hipEventElapsedTime(&t, sl.events[(size_t)i], sl.events[(size_t)i + 1]));
        total += t;
    }
    *ms = total;
    if (sl.used && hipEventQuery(sl.done) == hipSuccess)
        if (int e = check_device_status(const_cast<rt_scene::RenderSlot &>(sl))) return e;
    return RT_OK;
}

int rt_render_status(rt_scene *s) {
    if (!s) return fail(RT_ERR_INVALID, "null argument");
    std::lock_guard<std::mutex> lock(s->mu);
    if (s->device < 0) return RT_OK;
    HIP_TRY(hipSetDevice(s->device));
    int rc = RT_OK;
    for (rt_scene::RenderSlot &sl : s->slots) {
        if (!sl.used) continue;
        HIP_TRY(hipEventSynchronize(sl.done));
        if (int e = check_device_status(sl)) rc = e;
    }
    return rc;
}

int rt_render_wait_output(rt_scene *s, void *stream) {
    if (!s) return fail(RT_ERR_INVALID, "null argument");
    std::lock_guard<std::mutex> lock(s->mu);
    if (!s->timed) return fail(RT_ERR_STATE, "no render has been launched on this scene");
    HIP_TRY(hipSetDevice(s->device));
    const rt_scene::RenderSlot &sl = s->slots[s->last_slot];
    if (sl.used) HIP_TRY(hipStreamWaitEvent((hipStream_t)stream, sl.done, 0)); // a no-op when the output was not deferred
    return RT_OK;
}

int rt_scene_set_workspace_limit(rt_scene *s, size_t bytes) {
    if (!s) return fail(RT_ERR_INVALID, "null argument");
    std::lock_guard<std::mutex> lock(s->mu);
    s->workspace_limit = bytes;
    return RT_OK;
}

int rt_scene_trim(rt_scene *s) {
    if (!s) return fail(RT_ERR_INVALID, "null argument");
    std::lock_guard<std::mutex> render_lock(s->render_mu);
    std::lock_guard<std::mutex> lock(s->mu);
    if (s->device < 0) return RT_OK;
    HIP_TRY(hipSetDevice(s->device));
    for (rt_scene::RenderSlot &sl : s->slots) {
        if (sl.used) HIP_TRY(hipEventSynchronize(sl.done)); // a render in flight still writes its workspace
        if (sl.d_samples) HIP_TRY(hipFree(sl.d_samples));
        sl.d_samples = nullptr;
        sl.samples_bytes = 0;
    }
    return RT_OK;
}

size_t rt_scene_workspace_bytes(const rt_scene *s) {
    if (!s) return 0;
    return s->slots[0].samples_bytes + s->slots[1].samples_bytes;
}

int rt_last_launch_config(rt_scene *s, rt_launch_config *out) {
    if (!s || !out) return fail(RT_ERR_INVALID, "null argument");
    std::lock_guard<std::mutex> lock(s->mu);
    if (!s->timed) return fail(RT_ERR_STATE, "no render has been launched on this scene");
    *out = s->last_launch;
    return RT_OK;
}

int rt_scene_tile_order(rt_scene *s, uint32_t *order_out, uint64_t *cost_out, int capacity) {
    if (!s || !order_out || !cost_out) return fail(RT_ERR_INVALID, "null argument");
    std::lock_guard<std::mutex> lock(s->mu);
    const rt_scene::TileOrder &to = s->tile_order;
    if (to.key == 0 || to.n == 0) return 0;
    if (capacity < to.n) return fail(RT_ERR_INVALID, "rt_scene_tile_order: capacity below the number of owned tiles");
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(hipEventSynchronize(to.ready));
    HIP_TRY(hipMemcpy(order_out, to.d_order, (size_t)to.n * sizeof(uint32_t), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(cost_out, to.d_cost, (size_t)to.n * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return to.n;
}

int rt_render(rt_scene *s, const rt_camera *cam, const rt_render_params *p, double *out_rgb, rt_counters *counters) {
    if (int e = check_params(s, cam, p)) return e;
    if (!out_rgb) return fail(RT_ERR_INVALID, "null output buffer");
    std::lock_guard<std::mutex> render_lock(s->render_mu); // one render at a time per scene (shared workspace)
    HIP_TRY(hipSetDevice(s->device));
    const int n_owned = rt_shard_tile_count(p->width, p->height, p->shard_index, p->shard_count);
    if (n_owned < 0) return n_owned;
    if (counters) std::memset(counters, 0, sizeof *counters);
    if (n_owned == 0) return RT_OK;
    const size_t n_doubles = (size_t)n_owned * RT_TILE_PIXELS * 3;
    double *d_out = nullptr;
    RtCounters *d_cnt = nullptr;
    HIP_TRY(hipMalloc((void **)&d_out, n_doubles * sizeof(double)));
    rt_render_params q = *p;
    int rc = RT_OK;
    if (counters) {
        if (hipMalloc((void **)&d_cnt, sizeof(RtCounters)) != hipSuccess || hipMemset(d_cnt, 0, sizeof(RtCounters)) != hipSuccess)
            rc = fail(RT_ERR_DEVICE, "counter allocation failed");
        q.flags |= RT_FLAG_COUNTERS;
    } else {
        q.flags &= ~RT_FLAG_COUNTERS;
    }
    std::vector<double> host(n_doubles);
    if (rc == RT_OK) rc = rt_render_tiles_device(s, cam, &q, d_out, d_cnt, nullptr);
    if (rc == RT_OK) {
        hipError_t e = hipStreamSynchronize(nullptr);
        if (e != hipSuccess) rc = hip_fail(e, "render_kernel execution");
    }
    if (rc == RT_OK) {
        std::lock_guard<std::mutex> lock(s->mu);
        rc = check_device_status(s->slots[s->last_slot]);
    }
    if (rc == RT_OK) {
        hipError_t e = hipMemcpy(host.data(), d_out, n_doubles * sizeof(double), hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = hip_fail(e, "framebuffer copy");
    }
    if (rc == RT_OK && counters) {
        RtCounters c;
        hipError_t e = hipMemcpy(&c, d_cnt, sizeof c, hipMemcpyDeviceToHost);
        if (e != hipSuccess)
            rc = hip_fail(e, "counter copy");
        else {
            counters->samples = c.samples;
            counters->segments = c.segments;
            counters->nodes_visited = c.nodes_visited;
            counters->prims_tested = c.prims_tested;
            counters->rng_draws = c.rng_draws;
            counters->node_wave = c.node_wave;
            counters->node_lane = c.node_lane;
            counters->leaf_wave = c.leaf_wave;
            counters->leaf_lane = c.leaf_lane;
            counters->shade_wave = c.shade_wave;
            counters->shade_lane = c.shade_lane;
            counters->node_cycles = c.node_cycles;
            counters->leaf_cycles = c.leaf_cycles;
            counters->shade_cycles = c.shade_cycles;
            counters->finish_cycles = c.finish_cycles;
            counters->refill_cycles = c.refill_cycles;
            counters->begin_cycles = c.begin_cycles;
            counters->swap_class_mode = c.swap_class_mode;
            counters->swap_new_mode = c.swap_new_mode;
            counters->swap_parked = c.swap_parked;
            counters->swap_pulled = c.swap_pulled;
            counters->swap_lock_busy = c.swap_lock_busy;
            counters->swap_scattered = c.swap_scattered;
            counters->swap_off_class = c.swap_off_class;
            counters->swap_cycles = c.swap_cycles;
            counters->node_idle_done = c.node_idle_done;
            counters->node_idle_leaf = c.node_idle_leaf;
            counters->node_idle_empty = c.node_idle_empty;
        }
    }
    (void)hipFree(d_out);
    if (d_cnt) (void)hipFree(d_cnt);
    if (rc != RT_OK) return rc;
    // scatter the packed tiles into the caller's [y][x][3] image
    for_each_shard_pixel(p->width, p->height, p->shard_index, p->shard_count, n_owned, [&](size_t t, size_t g) {
        out_rgb[g] = host[t];
        out_rgb[g + 1] = host[t + 1];
        out_rgb[g + 2] = host[t + 2];
    });
    return RT_OK;
}

rt_scene *rt_scene_clone(const rt_scene *src, int device) {
    if (!src) {
        fail(RT_ERR_INVALID, "null scene");
        return nullptr;
    }
    rt_scene *s = new rt_scene;
    bool committed;
    {
        std::lock_guard<std::mutex> lock(const_cast<rt_scene *>(src)->mu); // rt_add_* on the source take it too
        s->ir = src->ir;
        committed = src->committed;
    }
    if (committed && rt_scene_commit(s, device) != RT_OK) { // g_err is set
        rt_scene_destroy(s);
        return nullptr;
    }
    return s;
}

int rt_render_sharded(rt_scene *const *scenes, int n, const rt_camera *cam, const rt_render_params *p, double *out_rgb) {
    if (!scenes || n <= 0 || !cam || !p || !out_rgb) return fail(RT_ERR_INVALID, "null argument or no scenes");
    for (int i = 0; i < n; ++i)
        if (int e = check_params(scenes[i], cam, p)) return e;
    std::vector<int> rc((size_t)n, RT_OK);
    std::vector<std::string> msg((size_t)n);
    std::vector<std::thread> workers;
    for (int i = 0; i < n; ++i)
        workers.emplace_back([&, i]() {
            rt_render_params q = *p;
            q.shard_index = i;
            q.shard_count = n;
            rc[(size_t)i] = rt_render(scenes[i], cam, &q, out_rgb, nullptr); // shards own disjoint pixels of out_rgb
            if (rc[(size_t)i] != RT_OK) msg[(size_t)i] = rt_last_error();   // thread-local text -> carry it over
        });
    for (std::thread &t : workers) t.join();
    for (int i = 0; i < n; ++i)
        if (rc[(size_t)i] != RT_OK) return fail(rc[(size_t)i], "shard " + std::to_string(i) + ": " + msg[(size_t)i]);
    return RT_OK;
}

int rt_pack_tiles_host(const double *image, int width, int height, int shard_index, int shard_count, int tiles_padded, double *tiles_out) {
    if (!image || !tiles_out) return fail(RT_ERR_INVALID, "null argument");
    const int n_owned = rt_shard_tile_count(width, height, shard_index, shard_count);
    if (n_owned < 0) return n_owned;
    if (tiles_padded < n_owned) return fail(RT_ERR_INVALID, "tiles_padded is smaller than the shard's tile count");
    std::fill(tiles_out, tiles_out + (size_t)tiles_padded * RT_TILE_PIXELS * 3, 0.0);
    for_each_shard_pixel(width, height, shard_index, shard_count, n_owned, [&](size_t t, size_t g) {
        tiles_out[t] = image[g];
        tiles_out[t + 1] = image[g + 1];
        tiles_out[t + 2] = image[g + 2];
    });
    return RT_OK;
}

int rt_unpack_tiles_host(const double *gathered, int tiles_per_shard_padded, int shard_count, int width, int height, double *image_out) {
    if (!gathered || !image_out || tiles_per_shard_padded <= 0 || shard_count <= 0 || width <= 0 || height <= 0)
        return fail(RT_ERR_INVALID, "bad argument");
    for (int r = 0; r < shard_count; ++r) {
        const int n_owned = rt_shard_tile_count(width, height, r, shard_count);
        if (n_owned > tiles_per_shard_padded) return fail(RT_ERR_INVALID, "tiles_per_shard_padded is smaller than a shard's tile count");
        const double *src = gathered + (size_t)r * (size_t)tiles_per_shard_padded * RT_TILE_PIXELS * 3;
        for_each_shard_pixel(width, height, r, shard_count, n_owned, [&](size_t t, size_t g) {
            image_out[g] = src[t];
            image_out[g + 1] = src[t + 1];
            image_out[g + 2] = src[t + 2];
        });
    }
    return RT_OK;
}

int rt_unpack_tiles_device(const void *d_gathered, int tiles_per_shard_padded, int shard_count, int width, int height,
                           void *d_image_out, void *stream) {
    if (!d_gathered || !d_image_out || tiles_per_shard_padded <= 0 || shard_count <= 0 || width <= 0 || height <= 0)
        return fail(RT_ERR_INVALID, "bad argument");
    int rc = rt_launch_unpack((const double *)d_gathered, tiles_per_shard_padded, shard_count, width, height, (double *)d_image_out,
                              stream);
    if (rc != 0) return hip_fail((hipError_t)rc, "unpack_kernel launch");
    return RT_OK;
}

void rt_tonemap_rgb8(const double *rgb, size_t n_pixels, uint8_t *out) {
    for (size_t i = 0; i < n_pixels * 3; ++i) out[i] = rt::tonemap_channel(rgb[i]);
}

int rt_write_ppm_p3(const char *path, const double *rgb, int width, int height) { // examples/book-one.rs:28-30,90-100
    if (!path || !rgb || width <= 0 || height <= 0) return fail(RT_ERR_INVALID, "bad argument");
    FILE *f = std::fopen(path, "w");
    if (!f) return fail(RT_ERR_INVALID, std::string("cannot open ") + path);
    std::fprintf(f, "P3\n%d %d\n255\n", width, height);
    for (int y = height - 1; y >= 0; --y) // rows top-down while y is up
        for (int x = 0; x < width; ++x) {
            const double *px = rgb + ((size_t)y * (size_t)width + (size_t)x) * 3;
            std::fprintf(f, "%u %u %u\n", (unsigned)rt::tonemap_channel(px[0]), (unsigned)rt::tonemap_channel(px[1]),
                         (unsigned)rt::tonemap_channel(px[2]));
        }
    std::fclose(f);
    return RT_OK;
}

void rt_tonemap_png8(const double *rgb, size_t n_pixels, uint8_t *out) {
    for (size_t i = 0; i < n_pixels * 3; ++i) out[i] = rt::png_channel(rgb[i]);
}

int rt_write_png_rgba8(const char *path, const double *rgb, int width, int height) { // examples/main.rs:105-135
    if (!path || !rgb || width <= 0 || height <= 0) return fail(RT_ERR_INVALID, "bad argument");
    std::vector<uint8_t> rgba((size_t)width * (size_t)height * 4);
    for (int y = 0; y < height; ++y) // put_pixel(x, height - 1 - y, ..): rows top-down while y is up
        for (int x = 0; x < width; ++x) {
            const double *px = rgb + ((size_t)y * (size_t)width + (size_t)x) * 3;
            uint8_t *o = rgba.data() + ((size_t)(height - 1 - y) * (size_t)width + (size_t)x) * 4;
            o[0] = rt::png_channel(px[0]);
            o[1] = rt::png_channel(px[1]);
            o[2] = rt::png_channel(px[2]);
            o[3] = 255;
        }
    const std::vector<uint8_t> png = rt::encode_png_rgba8(rgba.data(), width, height);
    FILE *f = std::fopen(path, "wb");
    if (!f) return fail(RT_ERR_INVALID, std::string("cannot open ") + path);
    const bool ok = std::fwrite(png.data(), 1, png.size(), f) == png.size();
    std::fclose(f);
    return ok ? RT_OK : fail(RT_ERR_INVALID, std::string("short write to ") + path);
}

int rt_scene_get_info(const rt_scene *s, rt_scene_info *out) {
    if (!s || !out) return fail(RT_ERR_INVALID, "null argument");
    if (!s->committed) return fail(RT_ERR_STATE, "scene not committed");
    std::memset(out, 0, sizeof *out);
    out->n_prims = s->flat.n_leaf_prims;
    out->n_child_prims = (int)s->flat.prim_meta.size() - s->flat.n_leaf_prims;
    out->n_hoisted = s->flat.n_hoisted;
    out->n_nodes = (int)s->flat.host_nodes.size();
    out->n_list = s->flat.n_list;
    out->max_depth = s->flat.max_depth;
    out->n_materials = (int)s->flat.materials.size();
    out->n_textures = (int)s->flat.textures.size();
    out->n_xforms = (int)s->flat.xforms.size();
    out->node_bytes = s->flat.n_list ? RT_LIST_BOX_FLOATS * (int)sizeof(float) : (int)sizeof(RtNode); // per node step | per list box
    out->prim_bytes = (int)sizeof(RtPrimGeo);
    out->material_bytes = (int)sizeof(RtMaterial);
    out->feature_mask = s->flat.feature_mask;
    out->device_bytes = s->device_bytes;
    return RT_OK;
}

int rt_scene_copy_nodes(const rt_scene *s, double *out, int max_nodes) {
    if (!s || !out) return fail(RT_ERR_INVALID, "null argument");
    if (!s->committed) return fail(RT_ERR_STATE, "scene not committed");
    const int n = (int)s->flat.host_nodes.size();
    for (int i = 0; i < n && i < max_nodes; ++i) {
        const rt::HostNode &nd = s->flat.host_nodes[(size_t)i];
        double *o = out + (size_t)i * 28;
        for (int c = 0; c < 2; ++c) {
            float clo[3], chi[3];
            rt::cull_box(nd.box[c], clo, chi); // what the device array holds (a list-mode scene keeps these boxes per leaf)
            for (int k = 0; k < 3; ++k) {
                o[c * 6 + k] = nd.box[c].lo[k];
                o[c * 6 + 3 + k] = nd.box[c].hi[k];
            }
            o[12 + c] = (double)nd.child[c];
            // the binary32 culling box actually traversed
            for (int k = 0; k < 3; ++k) {
                o[14 + c * 6 + k] = clo[k];
                o[14 + c * 6 + 3 + k] = chi[k];
            }
        }
        o[26] = o[27] = 0.0;
    }
    return n;
}

int rt_scene_hash(const rt_scene *s, uint64_t *out) {
    if (!s || !out) return fail(RT_ERR_INVALID, "null argument");
    if (!s->committed) return fail(RT_ERR_STATE, "scene not committed");
    uint64_t h = 0xcbf29ce484222325ull;
    auto eat = [&](const void *p, size_t n) {
        const unsigned char *b = (const unsigned char *)p;
        for (size_t i = 0; i < n; ++i) h = (h ^ b[i]) * 0x100000001b3ull;
    };
    const rt::FlatScene &f = s->flat;
    eat(f.prim_meta.data(), f.prim_meta.size() * sizeof(RtPrimMeta));
    eat(f.prim_geo.data(), f.prim_geo.size() * sizeof(RtPrimGeo));
    eat(f.prim_extra.data(), f.prim_extra.size() * sizeof(RtPrimExtra));
    eat(f.xforms.data(), f.xforms.size() * sizeof(RtXform));
    eat(f.xform_boxes.data(), f.xform_boxes.size() * sizeof(RtXformBox));
    for (const RtMaterial &m : f.materials) { // field by field: the struct has padding
        eat(&m.kind, sizeof m.kind);
        eat(&m.tex, sizeof m.tex);
        eat(&m.solid, sizeof m.solid);
        eat(&m.param, sizeof m.param);
        eat(m.rgb, sizeof m.rgb);
    }
    for (const RtTexture &t : f.textures) {
        eat(&t.kind, sizeof t.kind);
        eat(&t.a, sizeof t.a);
        eat(&t.b, sizeof t.b);
        eat(&t.w, sizeof t.w);
        eat(&t.h, sizeof t.h);
        eat(t.rgb, sizeof t.rgb);
    }
    eat(f.image_blob.data(), f.image_blob.size());
    *out = h;
    return RT_OK;
}

int rt_scene_prim_bounds(const rt_scene *s, int prim, double out[6]) {
    if (!s || !out) return fail(RT_ERR_INVALID, "null argument");
    if (!s->committed) return fail(RT_ERR_STATE, "scene not committed");
    if (prim < 0 || prim >= s->flat.n_leaf_prims) return fail(RT_ERR_INVALID, "prim out of range");
    const rt::Aabb &b = s->flat.prim_bounds[(size_t)prim];
    for (int i = 0; i < 3; ++i) {
        out[i] = b.lo[i];
        out[3 + i] = b.hi[i];
    }
    return RT_OK;
}

int rt_scene_plan_launch(const rt_scene *s, rt_launch_config *out) {
    if (!s || !out) return fail(RT_ERR_INVALID, "null argument");
    if (!s->committed) return fail(RT_ERR_STATE, "scene not committed");
    const unsigned feat = kernel_features(s);
    const int stack_entries = s->flat.n_list ? s->flat.n_list - 1 : std::min(RT_STACK_DEPTH, s->flat.max_depth + 1); // (fill_launch)
    LaunchPlan P;
    // (a scene committed without a device has not uploaded its binary16 tree: planned as if it had)
    if (int e = plan_launch(s, feat, stack_entries, s->device < 0 ? !s->flat.nodes_half.empty() : s->d_nodes_half != nullptr, &P)) return e;
    std::memset(out, 0, sizeof *out);
    out->block_threads = (int)P.block;
    out->lds_bytes = P.lds_bytes;
    out->blocks_per_cu = (int)P.groups_per_cu; // what the family's full occupancy asks for (a render asks the runtime)
    out->kernel_features = feat;
    out->lds_nodes = P.ldsnodes ? (P.half ? 2 : 1) : 0;
    out->swap = P.swap;
    out->swap_cap = (int)P.swap_cap;
    out->waves_per_simd = rt_kernel_waves_per_simd(feat);
    out->records_in_lds = P.reclds;
    return RT_OK;
}

int rt_scene_prim_group(const rt_scene *s, int prim) {
    if (!s) return fail(RT_ERR_INVALID, "null argument");
    if (!s->committed) return fail(RT_ERR_STATE, "scene not committed");
    if (prim < 0 || prim >= s->flat.n_leaf_prims) return fail(RT_ERR_INVALID, "prim out of range");
    return (size_t)prim < s->flat.group_len.size() ? s->flat.group_len[(size_t)prim] : 1;
}

int rt_probe_device_math(int device, const double *a, const double *b, int n, double *out_sqrt, double *out_div) {
    if (!a || !b || !out_sqrt || !out_div || n <= 0) return fail(RT_ERR_INVALID, "bad argument");
    int cnt = 0;
    if (hipGetDeviceCount(&cnt) != hipSuccess || device < 0 || device >= cnt) return fail(RT_ERR_DEVICE, "no such HIP device");
    HIP_TRY(hipSetDevice(device));
    double *d[4] = {nullptr, nullptr, nullptr, nullptr};
    const size_t bytes = (size_t)n * sizeof(double);
    for (auto &p : d) HIP_TRY(hipMalloc((void **)&p, bytes));
    HIP_TRY(hipMemcpy(d[0], a, bytes, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d[1], b, bytes, hipMemcpyHostToDevice));
    int rc = rt_launch_probe_math(d[0], d[1], n, d[2], d[3], nullptr);
    if (rc != 0) return hip_fail((hipError_t)rc, "probe kernel launch");
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out_sqrt, d[2], bytes, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(out_div, d[3], bytes, hipMemcpyDeviceToHost));
    for (auto &p : d) (void)hipFree(p);
    return RT_OK;
}

int rt_probe_device_libm(int device, int which, const double *a, const double *b, int n, double *out) {
    if (!a || !b || !out || n <= 0 || which < 0 || which > 5) return fail(RT_ERR_INVALID, "bad argument");
    int cnt = 0;
    if (hipGetDeviceCount(&cnt) != hipSuccess || device < 0 || device >= cnt) return fail(RT_ERR_DEVICE, "no such HIP device");
    HIP_TRY(hipSetDevice(device));
    double *d[3] = {nullptr, nullptr, nullptr};
    const size_t bytes = (size_t)n * sizeof(double);
    for (auto &p : d) HIP_TRY(hipMalloc((void **)&p, bytes));
    HIP_TRY(hipMemcpy(d[0], a, bytes, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d[1], b, bytes, hipMemcpyHostToDevice));
    int rc = rt_launch_probe_libm(which, d[0], d[1], n, d[2], nullptr);
    if (rc != 0) return hip_fail((hipError_t)rc, "probe kernel launch");
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out, d[2], bytes, hipMemcpyDeviceToHost));
    for (auto &p : d) (void)hipFree(p);
    return RT_OK;
}

} // extern "C"
