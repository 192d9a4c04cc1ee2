// rt_host.h -- host side of librt_mi355x: scene IR mirroring the reference's
// builder API, the flattener into SoA device records, and the BVH builder.
// Pure C++17, no HIP: compiled and unit-tested on CPU-only machines.
#ifndef RT_HOST_H
#define RT_HOST_H

#include "rt_types.h"

#include <cmath>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

namespace rt {

// ---- Mat4 (src/mat4.rs), column-major ----
void mat4_identity(double out[16]);
void mat4_translation(const double t[3], double out[16]);
void mat4_rotation(double radians, const double axis[3], double out[16]);
void mat4_multiplied(const double self[16], const double other[16], double out[16]);
double mat4_determinant(const double m[16]);
bool mat4_inversed(const double m[16], double out[16]);

struct Aabb {
    double lo[3], hi[3];
};

// ---- scene IR: what the rt_add_* calls record ----
struct TextureIR {
    uint32_t kind;
    double rgb[3];
    int a, b;
    int w, h;
    std::vector<uint8_t> texels;
};
struct MaterialIR {
    uint32_t kind;
    int tex;
    double param;
};
enum GeometryKind { GEO_SPHERE, GEO_RECTANGLE, GEO_CUBE, GEO_MEDIUM, GEO_BVH, GEO_TRANSFORMED };
struct GeometryIR {
    GeometryKind kind;
    double p[3];               // sphere: r | rectangle: w,h | cube: w,h,d | medium: density
    int boundary;              // medium: boundary geometry | transformed: inner geometry
    std::vector<int> children; // bvh: sprite ids (moved into the node, like the Vec BoundingVolumeHierarchyNode::new takes)
    double M[16];              // transformed: TransformedGeometry::new(geometry, M)
};
struct SpriteIR {
    int geometry, material;
    double M[16];
    bool owned; // listed in a GEO_BVH: not part of the world's own list any more
};

// host-side BVH node in binary64 (inspection, tests); RtNode is derived from it
struct HostNode {
    Aabb box[2];
    int32_t child[2]; // >= 0 inner node index, < 0 leaf: prim = ~child
};

struct FlatScene {
    std::vector<HostNode> host_nodes;
    std::vector<RtNode> nodes;
    std::vector<RtNodeH> nodes_half;      // the same tree with binary16 planes (empty: not representable, or no tree)
    std::vector<RtCubeGroup> cube_groups; // leaves that stand for the six faces of one cube (their records also sit behind the prims in prim_geo)
    std::vector<int> group_len;           // per leaf prim: 6 at the head of a cube group, 0 at its other five faces, 1 otherwise
    // prims [0, n_hoisted) are scene-filling and tested directly; [n_hoisted, n_leaf_prims)
    // are BVH leaves; the rest are the boundary prims of general media
    std::vector<RtPrimMeta> prim_meta;
    std::vector<RtPrimGeo> prim_geo;
    std::vector<RtPrimExtra> prim_extra;
    std::vector<Aabb> prim_bounds;
    int n_hoisted = 0;
    std::vector<RtXform> xforms;
    std::vector<RtXformBox> xform_boxes; // parallel to xforms
    // {boxes in reverse, xforms}: the memory image rtl::chain_boxes_admit expects around the transform array
    std::vector<double> xform_store;
    size_t xform_store_offset() const { return xforms.size() * sizeof(RtXformBox); } // bytes from the store's start to xforms[0]
    const RtXform *xforms_in_store() const {
        return xform_store.empty() ? nullptr : reinterpret_cast<const RtXform *>(reinterpret_cast<const unsigned char *>(xform_store.data()) + xform_store_offset());
    }
    std::vector<RtMaterial> materials;
    std::vector<RtTexture> textures;
    std::vector<uint8_t> image_blob;
    uint32_t root = RT_CUR_DONE; // reference (RT_REF_*) or RT_CUR_DONE if the BVH is empty
    bool wide = false;           // 32-bit references (more than 32767 prims or nodes)
    bool world_mid = true;       // every world-space sphere's centre and radius are <= 2^100 in magnitude (rtl::world_roots_rcp)
    int n_list = 0;              // > 0: `nodes` holds the box list of the n_list BVH leaves instead of the tree (small general scenes)
    // small general scenes (box list, or a tree of <= RT_RECLDS_TREE_MAX leaves): the records their kernels keep in LDS, packed {xforms, prim_geo, prim_meta, prim_extra, materials}, each array
    // at a 16-byte boundary; scene_blob_off[k] = byte offset of array k in the blob (RtLaunch::scene_blob)
    std::vector<unsigned char> scene_blob;
    uint32_t scene_blob_off[5] = {0, 0, 0, 0, 0};
    // (n_geo: prim_geo also holds the cube groups' records behind the prims)
    static size_t scene_blob_bytes(size_t n_xforms, size_t n_prims, size_t n_materials, size_t n_geo) {
        auto up = [](size_t b) { return (b + 15) & ~(size_t)15; };
        return up(n_xforms * sizeof(RtXform)) + up(n_geo * sizeof(RtPrimGeo)) + up(n_prims * sizeof(RtPrimMeta)) + up(n_prims * sizeof(RtPrimExtra)) +
               up(n_materials * sizeof(RtMaterial));
    }
    int n_leaf_prims = 0;
    int max_depth = 0;
    unsigned feature_mask = 0;
};

struct SceneIR {
    std::vector<TextureIR> textures;
    std::vector<MaterialIR> materials;
    std::vector<GeometryIR> geometries;
    std::vector<SpriteIR> sprites;
};

// Flatten + build.  Returns 0 or an RT_ERR_* code with `err` filled.
int flatten_scene(const SceneIR &ir, FlatScene *out, std::string *err);

// SAH BVH over prim_bounds[first..n); fills nodes, returns the root reference
// (>= 0 node index, < 0 leaf ~prim).  Depth <= RT_STACK_DEPTH.
int32_t build_bvh(const std::vector<Aabb> &bounds, int first, int n, std::vector<HostNode> *nodes, int *max_depth);
// binary32 culling box of a binary64 box: rounded outward, then padded
void cull_box(const Aabb &b, float lo[3], float hi[3]);
// binary32 -> binary16 bit pattern, rounded toward +inf (up) or -inf: the nearest grid value not inside the box.  |x| <= 60000.
inline uint16_t half_toward(float x, bool up) {
    auto to_float = [](uint16_t h) {
        const uint32_t sign = (uint32_t)(h & 0x8000u) << 16, e = (h >> 10) & 0x1Fu, m = h & 0x3FFu;
        uint32_t bits;
        if (e != 0u) {
            bits = sign | ((e + 112u) << 23) | (m << 13);
        } else {
            const float sub = (float)m * 0x1p-24f;
            std::memcpy(&bits, &sub, sizeof bits);
            bits |= sign;
        }
        float f;
        std::memcpy(&f, &bits, sizeof f);
        return f;
    };
    // truncate the magnitude to the grid (toward zero), then step away from zero when that is the wrong side
    uint32_t b;
    std::memcpy(&b, &x, sizeof b);
    const uint32_t sign = b >> 31;
    const float ax = std::fabs(x);
    uint16_t mag;
    if (ax < 0x1p-14f) {
        mag = (uint16_t)(ax * 0x1p24f); // subnormal grid: multiples of 2^-24 (truncation of an exact product)
    } else {
        uint32_t ab;
        std::memcpy(&ab, &ax, sizeof ab);
        mag = (uint16_t)((((ab >> 23) - 112u) << 10) | ((ab >> 13) & 0x3FFu));
    }
    uint16_t h = (uint16_t)((sign << 15) | mag);
    const float back = to_float(h);
    // the truncation is on the wrong side exactly when the wanted direction points away from zero: one step further out
    if (up ? back < x : back > x) h = (uint16_t)(h + 1u);
    return h;
}


// PerspectiveCamera::new (src/camera.rs:25-59)
void camera_perspective(RtCameraD *out, const double eye[3], const double center[3], const double up[3], double fov,
                        double aspect, double focus, double lens);

uint8_t tonemap_channel(double c);
// examples/main.rs:116-118: (c.sqrt() * 255.0).min(255.0) as u8  (Rust's float -> u8 cast saturates, NaN -> 0)
uint8_t png_channel(double c);
// 8-bit RGBA PNG file image (non-interlaced, filter 0, stored deflate blocks) of top-down RGBA rows
std::vector<uint8_t> encode_png_rgba8(const uint8_t *rgba, int width, int height);

} // namespace rt

#endif
