// rt_types.h -- flat scene layout shared by the host flattener (rt_host.cpp) and
// the HIP kernels (rt_kernels.hip).  Plain PODs, 16-byte aligned so every record
// is fetched with whole global_load_dwordx4 instructions.
#ifndef RT_TYPES_H
#define RT_TYPES_H

#include <stdint.h>

#define RT_TILE_EDGE 8
#define RT_TILE_PIXELS 64
#define RT_STACK_DEPTH 24   /* traversal stack entries per lane; the builder bounds the tree depth to this */
#define RT_MAX_HOISTED 4    /* scene-spanning prims tested up front instead of through the BVH */
#define RT_JOB_SPP_MAX 32   /* samples per pixel in one job at most (job = one 8x8 tile x job_spp samples) */
#define RT_NO_MATERIAL 0xFFFFFFFFu
#define RT_MAT_KIND_NONE 0xFFu /* in RtPrimMeta::kind bits 8-15: the prim has no material */

// 16-bit child references (stack entries pack one next to a truncated f32 tnear)
#define RT_REF_LEAF 0x8000u       /* bit 15: leaf, low 15 bits = prim index; else inner node index */
#define RT_REF_MAX 0x7FFFu
#define RT_CUR_DONE 0x10000u      /* traversal finished, lane waits for the shade block */
#define RT_CUR_DEAD 0x20000u      /* no work left for this lane */

// primitive kinds (leaves of the acceleration structure)
enum RtPrimKind : uint32_t {
    RT_PRIM_SPHERE_T = 0, // Sprite<Sphere> whose transform is a pure translation: geo = {cx, cy, cz, r}
    RT_PRIM_SPHERE_M = 1, // Sprite<Sphere>, general matrix: geo = {r}, xform
    RT_PRIM_RECT_M = 2,   // Sprite<Rectangle> / TransformedGeometry<Rectangle>: geo = {w, h}, xform
    RT_PRIM_GROUP_M = 3,  // Sprite<BVH of TransformedGeometry<Rectangle>> (Cube): aux = first child, geo[0] = count, xform
    RT_PRIM_MEDIUM_T = 4, // Sprite<ConstantMedium<Sphere>>, translation: geo = {cx, cy, cz, r}, extra = {density}, aux = rng slot
    RT_PRIM_MEDIUM_M = 5, // general matrix: geo = {r, density}, xform, aux = rng slot
};

enum RtMaterialKind : uint32_t {
    RT_MAT_LAMBERTIAN = 0,
    RT_MAT_METAL = 1,
    RT_MAT_DIELECTRIC = 2,
    RT_MAT_DIFFUSE_LIGHT = 3,
    RT_MAT_ISOTROPIC = 4,
};

enum RtTextureKind : uint32_t { RT_TEX_SOLID = 0, RT_TEX_CHECKER = 1, RT_TEX_IMAGE = 2 };

// BVH2 node, binary32 CULLING boxes of both children stored in the parent, planes
// interleaved as (child0, child1) pairs.  Boxes are rounded outward and padded (see rt_host.cpp): they only ever
// decide which f64 primitive tests are skipped, never a result.
struct alignas(16) RtNode {
    float lo_x[2], lo_y[2], lo_z[2]; // [child]
    float hi_x[2], hi_y[2], hi_z[2];
    uint32_t child[2];               // 16-bit references, see RT_REF_*
    uint32_t pad[2];
}; // 64 B

struct alignas(16) RtPrimMeta {
    uint32_t kind;     // bits 0-7: RT_PRIM_*; bits 8-15: kind of its material (RT_MAT_*, RT_MAT_KIND_NONE without one), so
                       // the shade block classifies a hit with one load
    uint32_t material; // RT_NO_MATERIAL = the reference's `material: None`
    uint32_t xform;    // index into xforms (kinds *_M)
    uint32_t aux;      // GROUP: first child prim; MEDIUM: rng slot
}; // 16 B
struct alignas(16) RtPrimGeo {
    double g[4];
}; // 32 B
struct alignas(16) RtPrimExtra {
    double e[2];
}; // 16 B

// rows 0..2 of M and M^-1 (row-major, 4 coefficients each: x y z w); row 3 of a
// Mat4 never reaches a Vec3 (src/vec4.rs:97-103 drops w).
struct alignas(16) RtXform {
    double m[12];
    double inv[12];
}; // 192 B

struct alignas(16) RtMaterial {
    uint32_t kind;
    uint32_t tex;
    uint32_t solid; // 1 if tex is a SolidColor: rgb below is its colour, no texture fetch, no uv
    uint32_t pad;
    double param;   // Metal fuzziness / Dielectric refractive
    double rgb[3];
}; // 48 B

struct alignas(16) RtTexture {
    uint32_t kind;
    uint32_t a, b;  // checker: black / white texture ids
    uint32_t w, h;  // image size
    uint32_t data;  // byte offset of the RGB8 texels in the image blob
    uint32_t pad[2];
    double rgb[3];
    double pad2;
}; // 64 B

struct RtCameraD {
    double eye[3], lower_left[3], horizontal[3], vertical[3];
    double lens_radius;
};

struct RtCounters {
    unsigned long long samples, segments, nodes_visited, prims_tested, rng_draws;
    // scheduling statistics of the wave-vote loop: executions of each block and
    // the lanes that were active in them (lane / (64 * wave) = SIMD utilisation)
    unsigned long long node_wave, node_lane, leaf_wave, leaf_lane, shade_wave, shade_lane;
    // s_memtime cycles spent inside each block, summed over waves (counting build only)
    unsigned long long node_cycles, leaf_cycles, shade_cycles, finish_cycles, refill_cycles, begin_cycles;
    // swap-at-shade diagnostics (see include/rt_mi355x.h)
    unsigned long long swap_class_mode, swap_new_mode, swap_parked, swap_pulled, swap_lock_busy, swap_scattered, swap_off_class, swap_cycles;
};

// kernel arguments (passed by value)
struct RtLaunch {
    const RtNode *nodes;
    const RtPrimMeta *prim_meta;
    const RtPrimGeo *prim_geo;
    const RtPrimExtra *prim_extra;
    const RtXform *xforms;
    const RtMaterial *materials;
    const RtTexture *textures;
    const uint8_t *image_blob;
    int32_t n_nodes;
    int32_t stack_entries; // LDS stack entries per lane for this scene (tree depth + 1, <= RT_STACK_DEPTH)
    uint32_t root;      // 16-bit reference of the BVH root, or RT_CUR_DONE when every prim is hoisted
    int32_t n_hoisted;  // prims [0, n_hoisted) are tested directly for every segment
    int32_t n_prims;
    RtCameraD cam;
    int32_t width, height, spp, max_depth;
    uint64_t seed_mix; // rt_mix64(seed)
    int32_t tiles_x, tiles_y;
    int32_t shard_index, shard_count, n_owned_tiles;
    // this pass: samples [s0, s0 + s_count) of every owned pixel
    int32_t s0, s_count;
    int32_t job_spp, jobs_per_tile, n_jobs; // job_spp adapts so that every wave sees >= ~32 jobs
    unsigned int *job_counter; // zeroed before the launch
    double *samples;           // per-sample radiance of this pass: [owned tile][s - s0][pixel][4] (32-byte records)
    RtCounters *counters;      // may be null
};

#endif
