// rt_types.h -- flat scene layout shared by the host flattener (rt_host.cpp) and
// the HIP kernels (rt_kernels.hip).  Plain PODs, 16-byte aligned so every record
// is fetched with whole global_load_dwordx4 instructions.
#ifndef RT_TYPES_H
#define RT_TYPES_H

#include <stdint.h>

#define RT_TILE_EDGE 8
#define RT_TILE_PIXELS 64
#define RT_STACK_DEPTH 24 /* traversal stack entries per lane; the builder bounds the tree depth to this */
#define RT_NO_MATERIAL 0xFFFFFFFFu

// primitive kinds (leaves of the acceleration structure)
enum RtPrimKind : uint32_t {
    RT_PRIM_SPHERE_T = 0, // Sprite<Sphere> whose transform is a pure translation: g = {cx, cy, cz, r}
    RT_PRIM_SPHERE_M = 1, // Sprite<Sphere>, general matrix: g = {r}, xform
    RT_PRIM_RECT_M = 2,   // Sprite<Rectangle> / TransformedGeometry<Rectangle>: g = {w, h}, xform
    RT_PRIM_GROUP_M = 3,  // Sprite<BVH of TransformedGeometry<Rectangle>> (Cube): aux = first child, g[0] = count, xform
    RT_PRIM_MEDIUM_T = 4, // Sprite<ConstantMedium<Sphere>>, translation: g = {cx, cy, cz, r}, aux = slot | density in g2
    RT_PRIM_MEDIUM_M = 5, // general matrix: g = {r, density}, xform
};

enum RtMaterialKind : uint32_t {
    RT_MAT_LAMBERTIAN = 0,
    RT_MAT_METAL = 1,
    RT_MAT_DIELECTRIC = 2,
    RT_MAT_DIFFUSE_LIGHT = 3,
    RT_MAT_ISOTROPIC = 4,
};

enum RtTextureKind : uint32_t { RT_TEX_SOLID = 0, RT_TEX_CHECKER = 1, RT_TEX_IMAGE = 2 };

// BVH2 node with the two child boxes stored in the parent: one fetch decides both
// children.  child >= 0: inner node index; child < 0: leaf, prim = ~child.
struct alignas(16) RtNode {
    double lo0[3], hi0[3];
    double lo1[3], hi1[3];
    int32_t child0, child1;
    int32_t pad[6];
}; // 128 B

struct alignas(16) RtPrim {
    uint32_t kind;
    uint32_t material; // RT_NO_MATERIAL = the reference's `material: None`
    uint32_t xform;    // index into xforms (kinds *_M)
    uint32_t aux;      // GROUP: first child prim; MEDIUM: rng slot
    double g[4];
    double g2[2];      // MEDIUM_T: {density, -}; padding otherwise
}; // 64 B

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
    unsigned long long samples, segments, nodes_visited, prims_tested, rng_draws, wave_iterations, lane_iterations;
};

// kernel arguments (passed by value)
struct RtLaunch {
    const RtNode *nodes;
    const RtPrim *prims;
    const RtXform *xforms;
    const RtMaterial *materials;
    const RtTexture *textures;
    const uint8_t *image_blob;
    int32_t root; // node index, or ~prim when the scene is a single primitive
    int32_t n_prims;
    RtCameraD cam;
    int32_t width, height, spp, max_depth;
    uint64_t seed_mix; // rt_mix64(seed)
    int32_t tiles_x, tiles_y;
    int32_t shard_index, shard_count, n_owned_tiles;
    double *out;            // packed owned tiles: [k][64][3]
    RtCounters *counters;   // may be null
};

#endif
