// rt_types.h -- flat scene layout shared by the host flattener (rt_host.cpp) and
// the HIP kernels (rt_kernels.hip).  Plain PODs, 16-byte aligned so every record
// is fetched with whole global_load_dwordx4 instructions.
#ifndef RT_TYPES_H
#define RT_TYPES_H

#include <stdint.h>

#define RT_TILE_EDGE 8
#define RT_TILE_PIXELS 64
#define RT_STACK_DEPTH 24   /* traversal stack entries per lane; the builder bounds the tree depth to this */
#define RT_MAX_HOISTED 4    /* scene-filling prims tested up front instead of through the BVH */
#define RT_LIST_MAX 24      /* general scenes of up to this many BVH leaves are walked as a box LIST (rtl::trav_list_step) */
#define RT_LIST_PRIM_BITS 6  /* bits of a LIST kernel's half-word stack entry that hold the leaf's prim index (rt_kernels.hip LdsStackList): a LIST scene has at most 1 << 6 leaf prims, hoisted ones included */
#define RT_LIST_SCENE_MAX 16384 /* a small general scene -- a LIST scene, or (round 5) a tree of at most RT_RECLDS_TREE_MAX leaves -- whose records (transforms, prims, materials) fit this many bytes AND the workgroup's LDS share keeps them in LDS */
#define RT_RECLDS_TREE_MAX 64   /* leaf prims of a TREE scene that may keep its records in LDS (kernel families: general prims without media over general boundaries) */
#ifndef RT_LIST_LDS_ARRAYS
#define RT_LIST_LDS_ARRAYS 0x1F /* which of them are READ there: 1 xforms, 2 prim_geo, 4 prim_meta, 8 prim_extra, 16 materials (A/B knob) */
#endif
#define RT_LIST_BOX_FLOATS 9 /* one list box: {lo, hi, lo} per axis -- entry plane at [s], exit plane at [s + 1], s = sign bit of 1/d */
static_assert((1 << RT_LIST_PRIM_BITS) >= RT_LIST_MAX + RT_MAX_HOISTED, "every leaf of a LIST scene needs an index that fits the stack entry");
#define RT_MAX_CHAIN 4      /* transform levels above one leaf (Sprite > BVH > Sprite > TransformedGeometry ...) every kernel family unrolls */
#define RT_MAX_CHAIN_DEEP 15 /* levels the family for general media / deep chains walks (the ones beyond RT_MAX_CHAIN in a run-time loop) */
#define RT_MAX_MEDIUM_NESTING 3 /* ConstantMedium levels inside one another (a medium in the boundary of a medium in ...) */
#define RT_JOB_SPP_MAX 32   /* samples per pixel in one job at most (job = one 8x8 tile x job_spp samples) */
#define RT_MAX_RECORDS (1u << 24) /* records per scene array: the kernels build 32-bit byte offsets with a 24-bit multiply */
#define RT_NO_MATERIAL 0xFFFFFFFFu
#define RT_MAT_KIND_NONE 0xFFu /* in RtPrimMeta::kind bits 8-15: the prim has no material */

// Child references.  Scenes of up to 32767 prims and nodes use 16-bit references (a stack entry packs one next to a
// truncated f32 tnear in ONE LDS word); larger ones 32-bit references (two words per entry) -- struct RtRef16 / RtRef32.
// A traversal cursor is a reference, or kDone (traversal finished, the lane waits for the shade block), or kDead (no
// work left for this lane): node < kLeaf <= leaf < kDone < kDead.
#define RT_REF_LEAF 0x8000u       /* 16-bit form: bit 15 = leaf, low 15 bits = prim index; else inner node index */
#define RT_REF_MAX 0x7FFFu
#define RT_CUR_DONE 0x10000u
#define RT_CUR_DEAD 0x20000u
#define RT_REF_LEAF_W 0x80000000u /* 32-bit form */
#define RT_REF_MAX_W 0x7FFFFFF0u
#define RT_CUR_DONE_W 0xFFFFFFFEu
#define RT_CUR_DEAD_W 0xFFFFFFFFu
struct RtRef16 {
    static constexpr uint32_t kLeaf = RT_REF_LEAF, kMask = RT_REF_MAX, kDone = RT_CUR_DONE, kDead = RT_CUR_DEAD;
};
struct RtRef32 {
    static constexpr uint32_t kLeaf = RT_REF_LEAF_W, kMask = 0x7FFFFFFFu, kDone = RT_CUR_DONE_W, kDead = RT_CUR_DEAD_W;
};

// primitive kinds (leaves of the acceleration structure).  Every leaf is ONE sphere, rectangle or medium under a
// chain of up to RT_MAX_CHAIN transforms (outermost first): a Sprite whose geometry is a BoundingVolumeHierarchyNode of
// further sprites / TransformedGeometry (src/sprite.rs:87-93, src/optimize.rs:339-343) is expanded at commit, each of
// its leaves keeping the transforms of every level -- Cube::new's six faces (src/geometry.rs:254-286) become six
// rectangle leaves with the chain {sprite matrix, face matrix}.  The ray goes down the chain matrix by matrix and the
// hit record comes back up it, in the reference's arithmetic (src/sprite.rs:101-126 at every level).
enum RtPrimKind : uint32_t {
    RT_PRIM_SPHERE_T = 0, // Sprite<Sphere> whose only transform is a pure translation: geo = {cx, cy, cz, r}, no chain
    RT_PRIM_SPHERE_C = 1, // sphere under a chain: geo = {r}
    RT_PRIM_RECT_C = 2,   // rectangle under a chain: geo = {w, h}
    RT_PRIM_MEDIUM_T = 4, // Sprite<ConstantMedium<Sphere>>, pure translation: geo = {cx, cy, cz, r}, extra = {density}, aux = rng key
    RT_PRIM_MEDIUM_C = 5, // ConstantMedium<any boundary> under a chain: geo = {density, first boundary prim, count}, aux = rng key;
                          // the boundary's prims (spheres / rectangles with their own chains below the medium) follow the
                          // leaf prims in the prim arrays and are never BVH leaves
};
// RtPrimMeta::kind: bits 0-7 RT_PRIM_*, bits 8-15 kind of the material (RT_MAT_*, RT_MAT_KIND_NONE), bits 16-19 chain
// length, bits 20-23 one bit per level of the first RT_MAX_CHAIN: that level is a pure translation (only its offset is used:
// o' = o + inv_t, d' = d, p = p' + t, n = n' -- what the 4x4 products give for such a matrix, rounding for rounding)
#define RT_META_CHAIN_SHIFT 16
#define RT_META_TMASK_SHIFT 20
// bit 24: this prim is the HEAD of a cube group -- six consecutive rectangle prims, the faces of one Cube::new (src/geometry.rs:254-286)
// whose culling boxes are (all but) the six sides of one axis-aligned box -- and the acceleration structure's leaf stands for all
// six: RtPrimMeta::aux of the head is the index in prim_geo of the group's RtCubeGroup record (two RtPrimGeo slots behind every prim).
// The faces stay ordinary RT_PRIM_RECT_C prims: hit records, tie rule and the careful scan know nothing of groups.
#define RT_META_GROUP_BIT (1u << 24)

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
    uint32_t child[2];               // references in the scene's form, see RT_REF_*
    uint32_t pad[2];
}; // 64 B
static_assert(sizeof(RtNode) == 64 && __builtin_offsetof(RtNode, child) == 48, "rtl::trav_node_step reads the planes and children by byte offset");

// The same node with binary16 planes, for a copy in LDS when the binary32 form does not fit (rt_api.cpp render_range): the planes are
// the binary32 culling planes rounded OUTWARD once more to the binary16 grid (rt_host.cpp half_nodes), so every box contains the
// binary32 box it stands for and the slab test is the same arithmetic on a slightly larger box: `fma((float)plane16, 1/d, n)` is
// one v_fma_mix_f32 -- the conversion is free.  Only built when every plane is finite and below 60000 in magnitude.
struct alignas(16) RtNodeH {
    uint16_t lo_x[2], lo_y[2], lo_z[2]; // binary16 bit patterns, [child]
    uint16_t hi_x[2], hi_y[2], hi_z[2];
    uint16_t child[2];                  // 16-bit references (RT_REF_*)
    uint32_t pad;
}; // 32 B
static_assert(sizeof(RtNodeH) == 32 && __builtin_offsetof(RtNodeH, hi_x) == 12 && __builtin_offsetof(RtNodeH, child) == 24,
              "rtl::trav_node_step reads the planes and children by byte offset");

// A cube group (RT_META_GROUP_BIT): the six faces' culling boxes are the six sides of ONE box, so they are derived from twelve planes
// instead of being stored (and walked) as six leaves under five nodes.  Per axis: the box's outer planes and, just inside them, the
// inner planes that close the slab of the face on that side: face (axis a, low side) has the box [outer_lo_a, inner_lo_a] on axis a
// and [outer_lo_b, outer_hi_b] on the other two.  Every derived box CONTAINS the face's own binary32 culling box (rt_host.cpp checks
// it plane by plane before it forms the group), so it culls by the same rule as a leaf box of the tree: never a result.
// slot = 2 * axis + side (0 low, 1 high).
struct alignas(16) RtCubeGroup {
    float outer_lo[3], outer_hi[3];
    float inner_lo[3], inner_hi[3];
    uint32_t first_prim;
    uint32_t faces; // three bits per slot: which of the six prims (first_prim + 0 .. 5) is the face of slot s = (faces >> 3 s) & 7
    uint32_t pad[2];
}; // 64 B = two RtPrimGeo slots
static_assert(sizeof(RtCubeGroup) == 64, "a cube group takes two prim_geo slots");

struct alignas(16) RtPrimMeta {
    uint32_t kind;     // see RT_META_*: prim kind, material kind (the shade block classifies a hit with one load), chain
    uint32_t material; // RT_NO_MATERIAL = the reference's `material: None`
    uint32_t xform;    // index of the OUTERMOST level of the chain in xforms; the levels of one leaf are consecutive
    uint32_t aux;      // MEDIUM: key of its free-flight draws (include/rt_rng.h)
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
// The reference's own bounding box of the object a transform level belongs to, in the frame ABOVE the level (the 8
// transformed corners of the inner bound, src/optimize.rs:128-241) -- present when that object is the child of a
// BoundingVolumeHierarchyNode (a sprite of the world or of a node geometry, a face of a Cube), whose walk tests it in binary64
// (src/optimize.rs:61-82,469-498); lo[0] is NaN otherwise (a TransformedGeometry: nobody tests its bound).  Read only by
// rtl::chain_boxes_admit, for segments that run (all but) in an axis plane.  The boxes live in FRONT of the transform array, in
// reverse: the box of xforms[i] is ((const RtXformBox *)xforms)[-1 - i] -- no kernel argument of their own, and the transform
// records keep the stride and alignment they were measured with (as a 256-byte record with the box inside, the hot `inv`
// halves sat in every other 128-byte line and the Cornell box lost 6 %).
struct RtXformBox {
    double lo[3], hi[3];
}; // 48 B

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
    unsigned long long node_idle_done, node_idle_leaf, node_idle_empty;
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
    // Small general scenes whose records fit -- box-LIST scenes, trees of up to RT_RECLDS_TREE_MAX leaves -- (lds_mode bit 4, kernels with RECLDS): the scene's records live in the workgroup's LDS.  prim_meta, prim_geo, prim_extra, xforms and
    // materials above then hold the arrays' BYTE OFFSETS in the LDS instead of addresses (rtl::rec_at<true>), and the kernel
    // copies `scene_bytes` bytes from `scene_blob` -- the five arrays packed by the host in that layout -- to offset
    // `scene_lds_off` at entry.  `xforms_global` is the transform array in global memory in every mode: the per-level boxes in
    // front of it are read there by the one cold path that needs them (rtl::chain_boxes_admit).
    const unsigned char *scene_blob;
    const RtXform *xforms_global;
    uint32_t scene_bytes, scene_lds_off;
    int32_t n_nodes;       // RtNode records behind `nodes` (in list mode: the packed box list, rounded up to whole records)
    int32_t n_list;        // 0: BVH walk.  > 0: `nodes` holds this many list boxes (RT_LIST_BOX_FLOATS binary32 each), leaf i = prim n_hoisted + i
    int32_t stack_entries; // LDS stack entries per lane for this scene (tree depth + 1, <= RT_STACK_DEPTH)
    int32_t swap_cap;      // entries per class queue of the swap-at-shade queues (rt_kernels.hip)
    uint32_t root;      // reference of the BVH root in the scene's form, or its kDone when every prim is hoisted
    int32_t n_hoisted;  // prims [0, n_hoisted) are tested directly for every segment
    int32_t world_mid;  // 1: every world-space sphere (RT_PRIM_SPHERE_T / RT_PRIM_MEDIUM_T) has |centre|, |radius| <= 2^100 (rtl::world_roots_rcp)
    int32_t n_prims;
    RtCameraD cam;
    int32_t width, height, spp, max_depth;
    uint64_t seed_mix; // rt_mix64(seed)
    int32_t tiles_x, tiles_y;
    int32_t shard_index, shard_count, n_owned_tiles;
    // this pass: samples [s0, s0 + s_count) of every owned pixel
    int32_t s0, s_count;
    int32_t job_spp, jobs_per_tile, n_jobs; // job_spp adapts so that every wave sees >= ~32 jobs
    uint32_t lds_bytes;        // dynamic LDS this launch provides: the kernel checks it against rt_lds_layout() (rt_lds.h)
    uint32_t watchdog_trips;   // counting build: loop trips a wave may make without finishing or starting a segment (0 = no bound)
    unsigned int *status;      // device error word (RT_DEV_*), sticky until the host reads it
    unsigned int *job_counter; // zeroed before the launch
    double *samples;           // per-sample radiance of this pass: [owned tile][s - s0][pixel][4] (32-byte records)
    RtCounters *counters;      // may be null
    // hand-out order of the owned tiles: null = ascending; else the k-th tile handed out is owned tile tile_order[k] (a
    // permutation of 0 .. n_owned - 1, learnt from the path lengths of an earlier render of the same view: rt_api.cpp).
    // A sample's slot and stream belong to its TILE, so the image does not depend on the order.
    const unsigned int *tile_order;
};

#endif
