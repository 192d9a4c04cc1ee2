// rt_host.cpp -- host side: Mat4, camera, scene flattener, SAH BVH builder.
//
// The reference keeps every object behind trait objects with a Mat4Cached pair
// each (src/sprite.rs:11-15) and builds a random-axis median-split tree
// (src/optimize.rs:366-440).  Its traversal neither prunes nor orders
// (src/optimize.rs:469-498), so the nearest hit does not depend on the tree: the
// flattener is free to choose its own structure.  It emits
//   - one 64-byte record per sprite (translation-only sphere sprites -- all of
//     book-one -- collapse to world-space centre + radius, the arithmetic the
//     4x4 path performs for a translation matrix, rounding for rounding),
//   - one SAH-built BVH2 whose nodes carry both child boxes (one fetch per step),
//   - de-duplicated materials with solid colours inlined.
// Compile with -ffp-contract=off: matrices and bounds must round like the reference.

#include "rt_host.h"

#include "../../include/rt_mi355x.h"
#include "../../include/rt_rng.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <limits>
#include <numeric>

namespace rt {

namespace {
constexpr double kInf = std::numeric_limits<double>::infinity();
constexpr double kPi = 3.14159265358979323846264338327950288;
} // namespace

// ------------------------------------------------------------------ Mat4
void mat4_identity(double out[16]) { // src/mat4.rs:21-28
    for (int i = 0; i < 16; ++i) out[i] = 0.0;
    out[0] = out[5] = out[10] = out[15] = 1.0;
}
void mat4_translation(const double t[3], double out[16]) { // src/mat4.rs:36-47
    mat4_identity(out);
    out[12] = t[0];
    out[13] = t[1];
    out[14] = t[2];
}
void mat4_rotation(double radians, const double axis[3], double out[16]) { // src/mat4.rs:52-80
    const double x = axis[0], y = axis[1], z = axis[2];
    const double s = std::sin(radians), c = std::cos(radians), t = 1.0 - c;
    out[0] = x * x * t + c;
    out[1] = y * x * t + z * s;
    out[2] = z * x * t - y * s;
    out[3] = 0.0;
    out[4] = x * y * t - z * s;
    out[5] = y * y * t + c;
    out[6] = z * y * t + x * s;
    out[7] = 0.0;
    out[8] = x * z * t + y * s;
    out[9] = y * z * t - x * s;
    out[10] = z * z * t + c;
    out[11] = 0.0;
    out[12] = 0.0;
    out[13] = 0.0;
    out[14] = 0.0;
    out[15] = 1.0;
}
void mat4_multiplied(const double self[16], const double other[16], double out[16]) { // src/mat4.rs:85-143
    double r[16];
    for (int col = 0; col < 4; ++col) {
        const double b0 = other[col * 4], b1 = other[col * 4 + 1], b2 = other[col * 4 + 2], b3 = other[col * 4 + 3];
        for (int row = 0; row < 4; ++row)
            r[col * 4 + row] = b0 * self[row] + b1 * self[4 + row] + b2 * self[8 + row] + b3 * self[12 + row];
    }
    std::memcpy(out, r, sizeof r);
}
namespace {
struct Sub2 { // the twelve 2x2 minors shared by determinant and inverse (src/mat4.rs:166-177)
    double b[12];
};
Sub2 minors(const double a[16]) {
    Sub2 q;
    q.b[0] = a[0] * a[5] - a[1] * a[4];
    q.b[1] = a[0] * a[6] - a[2] * a[4];
    q.b[2] = a[0] * a[7] - a[3] * a[4];
    q.b[3] = a[1] * a[6] - a[2] * a[5];
    q.b[4] = a[1] * a[7] - a[3] * a[5];
    q.b[5] = a[2] * a[7] - a[3] * a[6];
    q.b[6] = a[8] * a[13] - a[9] * a[12];
    q.b[7] = a[8] * a[14] - a[10] * a[12];
    q.b[8] = a[8] * a[15] - a[11] * a[12];
    q.b[9] = a[9] * a[14] - a[10] * a[13];
    q.b[10] = a[9] * a[15] - a[11] * a[13];
    q.b[11] = a[10] * a[15] - a[11] * a[14];
    return q;
}
} // namespace
double mat4_determinant(const double a[16]) { // src/mat4.rs:146-181
    const Sub2 q = minors(a);
    const double *b = q.b;
    return b[0] * b[11] - b[1] * b[10] + b[2] * b[9] + b[3] * b[8] - b[4] * b[7] + b[5] * b[6];
}
bool mat4_inversed(const double a[16], double out[16]) { // src/mat4.rs:184-243
    const double det = mat4_determinant(a);
    if (det == 0.0) return false;
    const Sub2 q = minors(a);
    const double *b = q.b;
    double r[16];
    r[0] = (a[5] * b[11] - a[6] * b[10] + a[7] * b[9]) / det;
    r[1] = (a[2] * b[10] - a[1] * b[11] - a[3] * b[9]) / det;
    r[2] = (a[13] * b[5] - a[14] * b[4] + a[15] * b[3]) / det;
    r[3] = (a[10] * b[4] - a[9] * b[5] - a[11] * b[3]) / det;
    r[4] = (a[6] * b[8] - a[4] * b[11] - a[7] * b[7]) / det;
    r[5] = (a[0] * b[11] - a[2] * b[8] + a[3] * b[7]) / det;
    r[6] = (a[14] * b[2] - a[12] * b[5] - a[15] * b[1]) / det;
    r[7] = (a[8] * b[5] - a[10] * b[2] + a[11] * b[1]) / det;
    r[8] = (a[4] * b[10] - a[5] * b[8] + a[7] * b[6]) / det;
    r[9] = (a[1] * b[8] - a[0] * b[10] - a[3] * b[6]) / det;
    r[10] = (a[12] * b[4] - a[13] * b[2] + a[15] * b[0]) / det;
    r[11] = (a[9] * b[2] - a[8] * b[4] - a[11] * b[0]) / det;
    r[12] = (a[5] * b[7] - a[4] * b[9] - a[6] * b[6]) / det;
    r[13] = (a[0] * b[9] - a[1] * b[7] + a[2] * b[6]) / det;
    r[14] = (a[13] * b[1] - a[12] * b[3] - a[14] * b[0]) / det;
    r[15] = (a[8] * b[3] - a[9] * b[1] + a[10] * b[0]) / det;
    std::memcpy(out, r, sizeof r);
    return true;
}

// ---------------------------------------------------------------- camera
void camera_perspective(RtCameraD *out, const double eye[3], const double center[3], const double up_in[3], double fov,
                        double aspect, double focus, double lens) { // src/camera.rs:25-59
    auto len = [](const double v[3]) { return std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); };
    double up[3], w[3], u[3], v[3];
    const double ul = len(up_in);
    for (int i = 0; i < 3; ++i) up[i] = up_in[i] / ul;
    const double height = std::tan(fov / 2.0) * 2.0;
    const double width = aspect * height;
    double ec[3] = {eye[0] - center[0], eye[1] - center[1], eye[2] - center[2]};
    const double el = len(ec);
    for (int i = 0; i < 3; ++i) w[i] = ec[i] / el;
    // u = up x w, left un-normalised (quirk Q1); cross per src/vec3.rs:80-86
    u[0] = up[1] * w[2] - up[2] * w[1];
    u[1] = -(up[0] * w[2] - up[2] * w[0]);
    u[2] = up[0] * w[1] - up[1] * w[0];
    v[0] = w[1] * u[2] - w[2] * u[1];
    v[1] = -(w[0] * u[2] - w[2] * u[0]);
    v[2] = w[0] * u[1] - w[1] * u[0];
    for (int i = 0; i < 3; ++i) {
        out->eye[i] = eye[i];
        out->horizontal[i] = u[i] * width * focus;
        out->vertical[i] = v[i] * height * focus;
        out->lower_left[i] = eye[i] - out->horizontal[i] / 2.0 - out->vertical[i] / 2.0 - w[i] * focus;
    }
    out->lens_radius = lens;
}

uint8_t png_channel(double c) { // examples/main.rs:116-118
    // f64::min returns the non-NaN operand, so sqrt(negative) = NaN becomes 255.0 before the cast
    const double r = std::sqrt(c) * 255.0;
    const double v = std::isnan(r) ? 255.0 : (r < 255.0 ? r : 255.0);
    return v <= 0.0 ? (uint8_t)0 : (uint8_t)v; // `as u8`: truncation toward zero, saturating
}

namespace {
uint32_t crc32_of(const uint8_t *p, size_t n, uint32_t crc) {
    static uint32_t table[256];
    static bool ready = false;
    if (!ready) {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            table[i] = c;
        }
        ready = true;
    }
    crc = ~crc;
    for (size_t i = 0; i < n; ++i) crc = table[(crc ^ p[i]) & 0xFFu] ^ (crc >> 8);
    return ~crc;
}
void put_be32(std::vector<uint8_t> &v, uint32_t x) {
    v.push_back((uint8_t)(x >> 24));
    v.push_back((uint8_t)(x >> 16));
    v.push_back((uint8_t)(x >> 8));
    v.push_back((uint8_t)x);
}
void put_chunk(std::vector<uint8_t> &png, const char type[4], const std::vector<uint8_t> &data) {
    put_be32(png, (uint32_t)data.size());
    const size_t at = png.size();
    png.insert(png.end(), type, type + 4);
    png.insert(png.end(), data.begin(), data.end());
    put_be32(png, crc32_of(png.data() + at, png.size() - at, 0u));
}
} // namespace

std::vector<uint8_t> encode_png_rgba8(const uint8_t *rgba, int width, int height) {
    // scanlines: filter byte 0 + width * 4 bytes
    const size_t stride = (size_t)width * 4;
    std::vector<uint8_t> raw;
    raw.reserve((stride + 1) * (size_t)height);
    for (int y = 0; y < height; ++y) {
        raw.push_back(0);
        raw.insert(raw.end(), rgba + (size_t)y * stride, rgba + (size_t)(y + 1) * stride);
    }
    // zlib stream of stored (uncompressed) deflate blocks
    std::vector<uint8_t> z;
    z.push_back(0x78);
    z.push_back(0x01);
    size_t pos = 0;
    do {
        const size_t n = std::min<size_t>(65535, raw.size() - pos);
        z.push_back(pos + n == raw.size() ? 1 : 0);
        z.push_back((uint8_t)(n & 0xFF));
        z.push_back((uint8_t)(n >> 8));
        z.push_back((uint8_t)(~n & 0xFF));
        z.push_back((uint8_t)((~n >> 8) & 0xFF));
        z.insert(z.end(), raw.begin() + (long)pos, raw.begin() + (long)(pos + n));
        pos += n;
    } while (pos < raw.size());
    uint32_t a = 1, b = 0; // Adler-32
    for (uint8_t c : raw) {
        a = (a + c) % 65521u;
        b = (b + a) % 65521u;
    }
    put_be32(z, (b << 16) | a);

    std::vector<uint8_t> png = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1A, '\n'};
    std::vector<uint8_t> ihdr;
    put_be32(ihdr, (uint32_t)width);
    put_be32(ihdr, (uint32_t)height);
    ihdr.push_back(8); // bit depth
    ihdr.push_back(6); // RGBA
    ihdr.push_back(0);
    ihdr.push_back(0);
    ihdr.push_back(0);
    put_chunk(png, "IHDR", ihdr);
    put_chunk(png, "IDAT", z);
    put_chunk(png, "IEND", {});
    return png;
}

uint8_t tonemap_channel(double c) { // examples/book-one.rs:95-97 (quirk Q13)
    const double v = std::fmin(std::sqrt(c) * 255.0, 255.0);
    if (!(v > 0.0)) return 0;
    return (uint8_t)v;
}

// ------------------------------------------------------------- flattening
namespace {

void make_xform(const double M[16], const double Minv[16], RtXform *x) {
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 4; ++c) {
            x->m[r * 4 + c] = M[c * 4 + r];
            x->inv[r * 4 + c] = Minv[c * 4 + r];
        }
}

// x' = m0*x + m4*y + m8*z + m12*w, evaluated left to right (src/vec4.rs:78-91)
void xform_point(const double M[16], const double p[3], double out[3]) {
    for (int r = 0; r < 3; ++r) out[r] = M[r] * p[0] + M[4 + r] * p[1] + M[8 + r] * p[2] + M[12 + r] * 1.0;
}

Aabb merged(const Aabb &a, const Aabb &b) { // src/optimize.rs:44-57
    Aabb o;
    for (int i = 0; i < 3; ++i) {
        o.lo[i] = std::fmin(a.lo[i], b.lo[i]);
        o.hi[i] = std::fmax(a.hi[i], b.hi[i]);
    }
    return o;
}
Aabb sphere_bound(double r) { return Aabb{{-r, -r, -r}, {r, r, r}}; }                                      // src/optimize.rs:105-114
Aabb rect_bound(double w, double h) { return Aabb{{-w / 2.0, -h / 2.0, -1e-6}, {w / 2.0, h / 2.0, 1e-6}}; } // src/optimize.rs:116-126

// The culling boxes must contain every hit the f64 primitive test can report.
// Reference bounds are rounded to nearest; pad them by ~2^-40 relative so a
// grazing hit that only exists through rounding is never culled.
void pad(Aabb *b) {
    for (int i = 0; i < 3; ++i) {
        const double scale = std::fmax(std::fmax(std::fabs(b->lo[i]), std::fabs(b->hi[i])), b->hi[i] - b->lo[i]);
        const double e = scale * 0x1p-40;
        b->lo[i] -= e;
        b->hi[i] += e;
    }
}

bool is_pure_translation(const double M[16]) {
    static const int zero_idx[] = {1, 2, 3, 4, 6, 7, 8, 9, 11};
    for (int i : zero_idx)
        if (M[i] != 0.0) return false;
    return M[0] == 1.0 && M[5] == 1.0 && M[10] == 1.0 && M[15] == 1.0;
}

// Cube::new (src/geometry.rs:254-286): six TransformedGeometry<Rectangle>
struct CubeFace {
    double w, h;
    double M[16];
};
void cube_faces(double width, double height, double depth, CubeFace f[6]) {
    auto rad = [](double deg) { return deg * (kPi / 180.0); }; // f64::to_radians
    const double ex[3] = {1.0, 0.0, 0.0}, ey[3] = {0.0, 1.0, 0.0};
    double T[16], R[16];
    auto set = [&](int i, double w, double h, const double t[3], bool rot, double deg, const double *axis) {
        f[i].w = w;
        f[i].h = h;
        mat4_translation(t, T);
        if (rot) {
            mat4_rotation(rad(deg), axis, R);
            mat4_multiplied(T, R, f[i].M);
        } else {
            std::memcpy(f[i].M, T, sizeof T);
        }
    };
    const double t0[3] = {0.0, 0.0, depth / 2.0}, t1[3] = {-width / 2.0, 0.0, 0.0}, t2[3] = {0.0, 0.0, -depth / 2.0};
    const double t3[3] = {width / 2.0, 0.0, 0.0}, t4[3] = {0.0, height / 2.0, 0.0}, t5[3] = {0.0, -height / 2.0, 0.0};
    set(0, width, height, t0, false, 0.0, nullptr); // front
    set(1, depth, height, t1, true, -90.0, ey);     // left
    set(2, width, height, t2, true, 180.0, ey);     // back
    set(3, depth, height, t3, true, 90.0, ey);      // right
    set(4, width, depth, t4, true, -90.0, ex);      // top
    set(5, width, depth, t5, true, 90.0, ex);       // bottom
}

} // namespace

// binary32 culling box: round outward, then pad by 2^-21 of the coordinate
// magnitude / extent.  Together with the per-ray pad of the kernel (2^-21 |o|) this
// exceeds the worst-case displacement between the binary64 ray and its binary32
// image at the box (<= 2^-24 (|o| + t|d|) per axis, |o + t d| <= |box|) by > 2x.
void cull_box(const Aabb &b, float lo[3], float hi[3]) {
    for (int i = 0; i < 3; ++i) {
        float l = (float)b.lo[i], h = (float)b.hi[i];
        if ((double)l > b.lo[i]) l = std::nextafterf(l, -std::numeric_limits<float>::infinity());
        if ((double)h < b.hi[i]) h = std::nextafterf(h, std::numeric_limits<float>::infinity());
        if (std::isfinite(l) && std::isfinite(h)) {
            const float scale = std::fmax(std::fmax(std::fabs(l), std::fabs(h)), h - l);
            const float e = scale * 0x1p-21f + 1e-30f;
            l -= e;
            h += e;
        }
        lo[i] = l;
        hi[i] = h;
    }
}

namespace {

// one transform level above a leaf
struct Link {
    double M[16], Minv[16];
    bool translation; // M and M^-1 are pure translations by t and -t: only the offsets enter the arithmetic
    // the reference's Bound of the object that carries this transform, in the frame above it, when that object is a child of a
    // BoundingVolumeHierarchyNode (RtXform::box)
    bool has_box = false;
    Aabb box{};
};
bool make_link(const double M[16], Link *l) {
    l->has_box = false;
    std::memcpy(l->M, M, sizeof l->M);
    if (!mat4_inversed(M, l->Minv)) return false; // det == 0: unhittable (src/sprite.rs:131-134, src/geometry.rs:241-244)
    l->translation = is_pure_translation(M) && is_pure_translation(l->Minv) && l->Minv[12] == -M[12] && l->Minv[13] == -M[13] &&
                     l->Minv[14] == -M[14];
    return true;
}

// a sphere or rectangle under its chain, before it becomes a BVH leaf or a medium's boundary prim
struct Shape {
    uint32_t kind; // RT_PRIM_SPHERE_C / RT_PRIM_RECT_C / RT_PRIM_MEDIUM_C (a medium inside a medium's boundary)
    double a, b;   // r | w, h | density
    std::vector<Link> chain; // outermost first
    std::vector<Shape> inner; // RT_PRIM_MEDIUM_C: its own boundary, chains relative to this medium's frame
    uint32_t key = 0;         // RT_PRIM_MEDIUM_C: rt_medium_key_path of the sprites that lead to it (include/rt_rng.h)
};
struct Leaf {
    RtPrimMeta meta{};
    RtPrimGeo geo{};
    RtPrimExtra extra{};
    Aabb bound{};
    std::vector<Link> chain;
    std::vector<Shape> boundary; // RT_PRIM_MEDIUM_C
    bool unbounded = false;      // hits may lie anywhere on the ray, not only inside `bound` (a medium over an open boundary)
    int cube = -1, face = 0;     // face `face` of the cube-th GEO_CUBE expansion (all six faces emitted), else -1
};

// the shape's bound carried through every level of `outer` + `inner` (outermost first): the AABB of the 8 transformed
// corners.  Tighter than nesting the reference's per-level boxes (src/optimize.rs:128-241) and still a superset of every
// point the chain arithmetic can map a local hit to (affine images of a box are spanned by its corners).
Aabb chain_bound(const Aabb &local, const std::vector<Link> &outer, const std::vector<Link> &inner) {
    const double x0 = local.lo[0], y0 = local.lo[1], z0 = local.lo[2], x1 = local.hi[0], y1 = local.hi[1], z1 = local.hi[2];
    double c[8][3] = {{x0, y0, z0}, {x1, y0, z0}, {x0, y1, z0}, {x0, y0, z1}, {x1, y1, z0}, {x1, y0, z1}, {x0, y1, z1}, {x1, y1, z1}};
    Aabb o;
    for (int i = 0; i < 3; ++i) {
        o.lo[i] = kInf;
        o.hi[i] = -kInf;
    }
    for (auto &p : c) {
        double q[3] = {p[0], p[1], p[2]}, r[3];
        for (size_t k = inner.size(); k-- > 0;) {
            xform_point(inner[k].M, q, r);
            std::memcpy(q, r, sizeof q);
        }
        for (size_t k = outer.size(); k-- > 0;) {
            xform_point(outer[k].M, q, r);
            std::memcpy(q, r, sizeof q);
        }
        for (int i = 0; i < 3; ++i) {
            if (q[i] < o.lo[i]) o.lo[i] = q[i];
            if (q[i] > o.hi[i]) o.hi[i] = q[i];
        }
    }
    return o;
}
Aabb shape_local_bound(const Shape &s) { return s.kind == RT_PRIM_SPHERE_C ? sphere_bound(s.a) : rect_bound(s.a, s.b); }

// Bound for Sprite / TransformedGeometry (src/optimize.rs:128-241): the AABB of the inner bound's 8 corners under M
Aabb corners_bound(const Aabb &inner, const double M[16]) {
    Link l;
    std::memcpy(l.M, M, sizeof l.M);
    return chain_bound(inner, {l}, {});
}
void set_ref_box(Link *l, const Aabb &inner) {
    l->has_box = true;
    l->box = corners_bound(inner, l->M);
}

struct Flattener {
    const SceneIR &ir;
    FlatScene &fs;
    std::vector<Leaf> leaves;
    int n_cubes = 0;
    uint32_t medium_slots = 0;
    std::string error;
    int rc = RT_OK;

    bool fail(int code, const std::string &msg) {
        if (rc == RT_OK) {
            rc = code;
            error = msg;
        }
        return false;
    }

    // `Bound::bound` of geometry `gi` in its own frame, as the reference computes it (src/optimize.rs:98-241,502-516): sphere and
    // rectangle boxes, the 8-corner boxes of TransformedGeometry / Sprite, the union a BoundingVolumeHierarchyNode keeps as its
    // volume (min / max are exact, so the shape of the random tree does not enter), the boundary's for a ConstantMedium.
    // false: None (an empty node, only unhittable children)
    bool ref_bound(int gi, Aabb *out, int depth = 0) {
        if (gi < 0 || depth > 16) return false;
        const GeometryIR &g = ir.geometries[(size_t)gi];
        switch (g.kind) {
        case GEO_SPHERE:
            *out = sphere_bound(g.p[0]);
            return true;
        case GEO_RECTANGLE:
            *out = rect_bound(g.p[0], g.p[1]);
            return true;
        case GEO_CUBE: {
            CubeFace f[6];
            cube_faces(g.p[0], g.p[1], g.p[2], f);
            for (int i = 0; i < 6; ++i) {
                const Aabb b = corners_bound(rect_bound(f[i].w, f[i].h), f[i].M);
                *out = i ? merged(*out, b) : b;
            }
            return true;
        }
        case GEO_TRANSFORMED: {
            Aabb in;
            if (!ref_bound(g.boundary, &in, depth + 1)) return false;
            *out = corners_bound(in, g.M);
            return true;
        }
        case GEO_BVH: {
            bool any = false;
            for (int si : g.children) {
                const SpriteIR &sp = ir.sprites[(size_t)si];
                Aabb in;
                if (sp.geometry < 0 || !ref_bound(sp.geometry, &in, depth + 1)) continue;
                const Aabb b = corners_bound(in, sp.M);
                *out = any ? merged(*out, b) : b;
                any = true;
            }
            return any;
        }
        case GEO_MEDIUM:
            return ref_bound(g.boundary, out, depth + 1);
        }
        return false;
    }
    // a sprite (of the world's list or of a node geometry) is a child of a BoundingVolumeHierarchyNode: its box is tested
    void sprite_box(Link *l, int geometry) {
        Aabb in;
        if (ref_bound(geometry, &in)) set_ref_box(l, in);
    }

    // every sphere / rectangle below geometry `gi`, each with the transforms between it and `chain`'s end appended
    // path_hash: the sprites from the world's list down to here (keys of nested media); media: ConstantMedium levels entered so far
    bool collect_shapes(int gi, std::vector<Link> chain, std::vector<Shape> *out, int depth, uint64_t path_hash = 0, int media = 1) {
        if (depth > 16) return fail(RT_ERR_UNSUPPORTED, "geometry nesting deeper than 16 levels");
        const GeometryIR &g = ir.geometries[(size_t)gi];
        switch (g.kind) {
        case GEO_SPHERE:
            out->push_back(Shape{RT_PRIM_SPHERE_C, g.p[0], 0.0, chain, {}, 0u});
            return true;
        case GEO_RECTANGLE:
            out->push_back(Shape{RT_PRIM_RECT_C, g.p[0], g.p[1], chain, {}, 0u});
            return true;
        case GEO_CUBE: { // BoundingVolumeHierarchyNode::new(Cube::new(w, h, d)): six TransformedGeometry<Rectangle>
            CubeFace f[6];
            cube_faces(g.p[0], g.p[1], g.p[2], f);
            for (const CubeFace &face : f) {
                Link l;
                if (!make_link(face.M, &l)) continue;
                set_ref_box(&l, rect_bound(face.w, face.h)); // a child of the Cube's BoundingVolumeHierarchyNode
                std::vector<Link> c = chain;
                c.push_back(l);
                out->push_back(Shape{RT_PRIM_RECT_C, face.w, face.h, c, {}, 0u});
            }
            return true;
        }
        case GEO_TRANSFORMED: {
            Link l;
            if (!make_link(g.M, &l)) return true; // never hit
            chain.push_back(l);
            return collect_shapes(g.boundary, chain, out, depth + 1, path_hash, media);
        }
        case GEO_BVH:
            for (size_t k = 0; k < g.children.size(); ++k) {
                const SpriteIR &sp = ir.sprites[(size_t)g.children[k]];
                if (sp.geometry < 0) continue;
                Link l;
                if (!make_link(sp.M, &l)) continue;
                sprite_box(&l, sp.geometry);
                std::vector<Link> c = chain;
                c.push_back(l);
                if (!collect_shapes(sp.geometry, c, out, depth + 1, path_hash * RT_RNG_PATH_MUL + (uint64_t)k + 1ull, media)) return false;
            }
            return true;
        case GEO_MEDIUM: {
            // ConstantMedium<T: Hit> with a ConstantMedium inside T (src/volume.rs:18-44): a boundary prim that is itself a
            // medium, evaluated (with draws of its own) by each of the outer medium's two boundary.hit calls
            if (media >= RT_MAX_MEDIUM_NESTING)
                return fail(RT_ERR_UNSUPPORTED, "more than RT_MAX_MEDIUM_NESTING (3) ConstantMedium levels inside one another");
            if (chain.size() > RT_MAX_CHAIN_DEEP)
                return fail(RT_ERR_UNSUPPORTED, "more than RT_MAX_CHAIN_DEEP (15) transform levels inside a medium's boundary");
            Shape m{RT_PRIM_MEDIUM_C, g.p[0], 0.0, chain, {}, rt_medium_key_path(path_hash)};
            if (!collect_shapes(g.boundary, {}, &m.inner, depth + 1, path_hash, media + 1)) return false;
            for (const Shape &sh : m.inner)
                if (sh.chain.size() > RT_MAX_CHAIN_DEEP)
                    return fail(RT_ERR_UNSUPPORTED, "more than RT_MAX_CHAIN_DEEP (15) transform levels inside a medium's boundary");
            if (!m.inner.empty()) out->push_back(std::move(m)); // an empty boundary is never hit
            fs.feature_mask |= RT_FEAT_MEDIUM_NESTED;
            return true;
        }
        }
        return true;
    }

    // the AABB of everything a boundary can report a hit in, carried through `outer` (a nested medium hits inside its own boundary)
    Aabb boundary_bound(const std::vector<Shape> &shapes, const std::vector<Link> &outer, bool *any) {
        Aabb acc{};
        for (const Shape &sh : shapes) {
            Aabb sb;
            if (sh.kind == RT_PRIM_MEDIUM_C) {
                std::vector<Link> o = outer;
                o.insert(o.end(), sh.chain.begin(), sh.chain.end());
                bool sub = false;
                sb = boundary_bound(sh.inner, o, &sub);
                if (!sub) continue;
            } else {
                sb = chain_bound(shape_local_bound(sh), outer, sh.chain);
            }
            acc = *any ? merged(acc, sb) : sb;
            *any = true;
        }
        return acc;
    }

    // Does `gi`, seen from the frame of the ConstantMedium around it, enclose a volume whose FIRST hit from outside always
    // has normal . direction < 0?  Then a medium hit is never nearer than the ray's entry into the boundary's box and the
    // box may cull it.  True for a sphere, a cube (outward faces, src/geometry.rs:254-285) and unions of those moved by pure
    // translations.  A rectangle's normal is +z from either side (src/geometry.rs:176), and a general matrix carries
    // normals by M rather than M^-T (quirk Q5): with those, volume.rs:80-98 ("the origin is inside") fires for origins far
    // outside the box and returns t = distance < the box entry.  (Measure-zero leftovers: rays through the 1e-16-wide seams
    // of a cube's faces, spheres grazed with a discriminant that rounds to 0.)
    bool boundary_encloses(int gi, int depth) {
        if (depth > 16) return false;
        const GeometryIR &g = ir.geometries[(size_t)gi];
        switch (g.kind) {
        case GEO_SPHERE:
        case GEO_CUBE:
            return true;
        case GEO_TRANSFORMED: {
            Link l;
            if (!make_link(g.M, &l)) return true; // never hit
            return l.translation && boundary_encloses(g.boundary, depth + 1);
        }
        case GEO_BVH:
            for (int si : g.children) {
                const SpriteIR &sp = ir.sprites[(size_t)si];
                Link l;
                if (sp.geometry < 0 || !make_link(sp.M, &l)) continue;
                if (!l.translation || !boundary_encloses(sp.geometry, depth + 1)) return false;
            }
            return true;
        default:
            return false;
        }
    }

    void finish_meta(RtPrimMeta *m, const std::vector<Link> &chain) {
        uint32_t tmask = 0;
        for (size_t i = 0; i < chain.size() && i < (size_t)RT_MAX_CHAIN; ++i) // deeper levels always take the full 4x4 form
            if (chain[i].translation) tmask |= 1u << i;
        m->kind = (m->kind & 0xFFu) | ((uint32_t)chain.size() << RT_META_CHAIN_SHIFT) | (tmask << RT_META_TMASK_SHIFT);
    }

    // Emit the leaves below geometry `gi` reached through `chain`; `material` is the outermost sprite's (Sprite::hit
    // replaces whatever material the inner record carried, src/sprite.rs:119-127); path_hash / path_len identify the
    // sprites on the way down (keys of the media's random draws, include/rt_rng.h)
    bool emit(int gi, const std::vector<Link> &chain, uint32_t material, uint64_t path_hash, int path_len, uint32_t top_slot, int depth) {
        if (depth > 16) return fail(RT_ERR_UNSUPPORTED, "geometry nesting deeper than 16 levels");
        const GeometryIR &g = ir.geometries[(size_t)gi];
        switch (g.kind) {
        case GEO_SPHERE:
        case GEO_RECTANGLE:
        case GEO_CUBE: {
            std::vector<Shape> shapes;
            if (!collect_shapes(gi, chain, &shapes, depth)) return false;
            // the six faces of one Cube::new, in order (a face with a singular matrix is dropped by collect_shapes: no group then)
            const int cube = (g.kind == GEO_CUBE && shapes.size() == 6) ? n_cubes++ : -1;
            int face = 0;
            for (const Shape &sh : shapes) {
                if (sh.chain.size() > RT_MAX_CHAIN_DEEP)
                    return fail(RT_ERR_UNSUPPORTED, "more than RT_MAX_CHAIN_DEEP (15) transform levels above one primitive");
                // more than four levels (src/sprite.rs:87-93 nests without bound): the kernel family that walks chains of any
                // length, slower (the one for media over general boundaries)
                if (sh.chain.size() > RT_MAX_CHAIN) fs.feature_mask |= RT_FEAT_GENERAL | RT_FEAT_DEEP_CHAIN;
                Leaf lf;
                lf.meta.material = material;
                lf.chain = sh.chain;
                if (sh.kind == RT_PRIM_SPHERE_C && sh.chain.size() == 1 && sh.chain[0].translation) {
                    // all of book-one: world-space centre + radius, the arithmetic the 4x4 path performs for a translation
                    lf.meta.kind = RT_PRIM_SPHERE_T;
                    lf.geo.g[0] = sh.chain[0].M[12];
                    lf.geo.g[1] = sh.chain[0].M[13];
                    lf.geo.g[2] = sh.chain[0].M[14];
                    lf.geo.g[3] = sh.a;
                    fs.feature_mask |= RT_FEAT_SPHERE_T;
                } else {
                    lf.meta.kind = sh.kind;
                    lf.geo.g[0] = sh.a;
                    lf.geo.g[1] = sh.b;
                    fs.feature_mask |= RT_FEAT_GENERAL;
                }
                lf.bound = chain_bound(shape_local_bound(sh), sh.chain, {});
                pad(&lf.bound);
                lf.cube = cube;
                lf.face = face++;
                leaves.push_back(std::move(lf));
            }
            return true;
        }
        case GEO_TRANSFORMED: {
            Link l;
            if (!make_link(g.M, &l)) return true;
            std::vector<Link> c = chain;
            c.push_back(l);
            return emit(g.boundary, c, material, path_hash, path_len, top_slot, depth + 1);
        }
        case GEO_BVH:
            for (size_t k = 0; k < g.children.size(); ++k) {
                const SpriteIR &sp = ir.sprites[(size_t)g.children[k]];
                if (sp.geometry < 0) continue; // geometry None: never hit (src/sprite.rs:95,136)
                Link l;
                if (!make_link(sp.M, &l)) continue;
                sprite_box(&l, sp.geometry);
                std::vector<Link> c = chain;
                c.push_back(l);
                // the child's position in its node, not its creation index: independent of the order a front end records sprites in
                if (!emit(sp.geometry, c, material, path_hash * RT_RNG_PATH_MUL + (uint64_t)k + 1ull, path_len + 1, top_slot, depth + 1))
                    return false;
            }
            return true;
        case GEO_MEDIUM: {
            if (chain.size() > RT_MAX_CHAIN_DEEP)
                return fail(RT_ERR_UNSUPPORTED, "more than RT_MAX_CHAIN_DEEP (15) transform levels above one primitive");
            if (chain.size() > RT_MAX_CHAIN) fs.feature_mask |= RT_FEAT_GENERAL | RT_FEAT_DEEP_CHAIN;
            Leaf lf;
            lf.meta.material = material;
            lf.chain = chain;
            // a medium sprite of the world's own list keeps its creation-order slot; any other one is keyed by its path
            // (include/rt_rng.h): inside a node, behind a TransformedGeometry
            lf.meta.aux = (path_len == 1 && top_slot < RT_MEDIUM_SLOT_MAX) ? top_slot : rt_medium_key_path(path_hash);
            const GeometryIR &b = ir.geometries[(size_t)g.boundary];
            fs.feature_mask |= RT_FEAT_MEDIUM;
            // a medium whose material reads uv (src/volume.rs:64-66: the sums over both boundary hits) goes to the kernel
            // family that keeps uv inside the medium test
            if (material != RT_NO_MATERIAL && !fs.materials[material].solid) fs.feature_mask |= RT_FEAT_MEDIUM_GENERAL;
            if (b.kind == GEO_SPHERE && chain.size() == 1 && chain[0].translation) {
                lf.meta.kind = RT_PRIM_MEDIUM_T; // examples/main.rs:241-263
                lf.geo.g[0] = chain[0].M[12];
                lf.geo.g[1] = chain[0].M[13];
                lf.geo.g[2] = chain[0].M[14];
                lf.geo.g[3] = b.p[0];
                lf.extra.e[0] = g.p[0];
                lf.extra.e[1] = -1.0 / g.p[0]; // (-1.0 / self.density), src/volume.rs:62,84
                lf.bound = chain_bound(sphere_bound(b.p[0]), chain, {});
            } else {
                lf.meta.kind = RT_PRIM_MEDIUM_C;
                lf.geo.g[0] = g.p[0];
                if (!collect_shapes(g.boundary, {}, &lf.boundary, depth + 1, path_hash, 1)) return false;
                for (const Shape &sh : lf.boundary)
                    if (sh.chain.size() > RT_MAX_CHAIN_DEEP)
                        return fail(RT_ERR_UNSUPPORTED, "more than RT_MAX_CHAIN_DEEP (15) transform levels inside a medium's boundary");
                bool any = false;
                lf.bound = boundary_bound(lf.boundary, chain, &any);
                if (!any) return true; // empty boundary: never hit
                lf.unbounded = !boundary_encloses(g.boundary, depth + 1);
                fs.feature_mask |= RT_FEAT_GENERAL | RT_FEAT_MEDIUM_GENERAL;
            }
            pad(&lf.bound);
            leaves.push_back(std::move(lf));
            return true;
        }
        }
        return true;
    }
};

uint32_t push_chain(FlatScene &fs, const std::vector<Link> &chain) {
    const uint32_t first = (uint32_t)fs.xforms.size();
    for (const Link &l : chain) {
        RtXform x;
        make_xform(l.M, l.Minv, &x);
        RtXformBox b{{std::numeric_limits<double>::quiet_NaN(), 0.0, 0.0}, {0.0, 0.0, 0.0}}; // no reference box
        if (l.has_box)
            for (int i = 0; i < 3; ++i) {
                b.lo[i] = l.box.lo[i];
                b.hi[i] = l.box.hi[i];
            }
        fs.xforms.push_back(x);
        fs.xform_boxes.push_back(b);
    }
    return first;
}

} // namespace

int flatten_scene(const SceneIR &ir, FlatScene *out, std::string *err) {
    FlatScene fs;

    // textures
    for (const TextureIR &t : ir.textures) {
        RtTexture d{};
        d.kind = t.kind;
        d.a = (uint32_t)t.a;
        d.b = (uint32_t)t.b;
        d.w = (uint32_t)t.w;
        d.h = (uint32_t)t.h;
        for (int i = 0; i < 3; ++i) d.rgb[i] = t.rgb[i];
        if (t.kind == RT_TEX_IMAGE) {
            d.data = (uint32_t)fs.image_blob.size();
            fs.image_blob.insert(fs.image_blob.end(), t.texels.begin(), t.texels.end());
            while (fs.image_blob.size() % 16) fs.image_blob.push_back(0);
        }
        fs.textures.push_back(d);
    }
    // materials
    for (const MaterialIR &m : ir.materials) {
        RtMaterial d{};
        d.kind = m.kind;
        d.param = m.param;
        d.tex = 0;
        d.solid = 1;
        d.rgb[0] = d.rgb[1] = d.rgb[2] = 1.0; // Dielectric attenuation (src/material.rs:148)
        if (m.kind != RT_MAT_DIELECTRIC) {
            const TextureIR &t = ir.textures[(size_t)m.tex];
            d.tex = (uint32_t)m.tex;
            if (t.kind == RT_TEX_SOLID) {
                for (int i = 0; i < 3; ++i) d.rgb[i] = t.rgb[i];
            } else {
                d.solid = 0;
                fs.feature_mask |= RT_FEAT_TEXTURED;
            }
        }
        fs.materials.push_back(d);
    }

    // the world's own sprites -> leaves, in creation order (hoisted ones are moved to the front below); a sprite whose
    // geometry is a node of further sprites is expanded, every leaf keeping the transforms of all levels above it
    Flattener fl{ir, fs};
    uint64_t rank = 0; // position among the world's own sprites
    for (size_t si = 0; si < ir.sprites.size(); ++si) {
        const SpriteIR &s = ir.sprites[si];
        if (s.owned) continue;        // moved into a BoundingVolumeHierarchyNode geometry
        const uint64_t my_rank = rank++;
        // slots count the medium sprites of the world's own list in creation order; a sprite that reaches a medium only
        // through a TransformedGeometry has none (0x3FF) and its medium is keyed by its path
        uint32_t slot = 0x3FFu;
        if (s.geometry >= 0 && ir.geometries[(size_t)s.geometry].kind == GEO_MEDIUM) {
            slot = fl.medium_slots++;
            if (slot >= RT_MEDIUM_SLOT_MAX) {
                if (err) *err = "more than 1023 ConstantMedium sprites in the world's own list";
                return RT_ERR_UNSUPPORTED;
            }
        }
        if (s.geometry < 0) continue; // geometry None: never hit (src/sprite.rs:95,136)
        Link l;
        if (!make_link(s.M, &l)) continue; // det == 0: unhittable (src/sprite.rs:131-134)
        fl.sprite_box(&l, s.geometry);       // a child of the world's BoundingVolumeHierarchyNode
        const uint32_t material = s.material < 0 ? RT_NO_MATERIAL : (uint32_t)s.material;
        if (!fl.emit(s.geometry, {l}, material, my_rank + 1ull, 1, slot, 0)) {
            if (err) *err = fl.error;
            return fl.rc;
        }
    }
    std::vector<Leaf> &leaves = fl.leaves;
    if (leaves.empty()) {
        if (err) *err = "empty scene: BoundingVolumeHierarchyNode::new(vec![]) is None (src/optimize.rs:367-370)";
        return RT_ERR_EMPTY;
    }
    if (leaves.size() > (size_t)RT_REF_MAX_W) {
        if (err) *err = "more than 2^31 primitives";
        return RT_ERR_UNSUPPORTED;
    }

    // hoist scene-FILLING prims (book-one's sky and ground spheres, main.rs's fog): their boxes make every ancestor
    // an always-hit, so they are tested up front for each segment and give the traversal an early upper bound instead.
    // The measure is the box VOLUME: a wall of the Cornell box spans the scene too, but its box is flat and culls well.
    auto volume = [](const Aabb &b) { return (b.hi[0] - b.lo[0]) * (b.hi[1] - b.lo[1]) * (b.hi[2] - b.lo[2]); };
    std::vector<size_t> order(leaves.size());
    std::iota(order.begin(), order.end(), (size_t)0);
    std::vector<size_t> hoist;
    // media over an open boundary first, however many: no box bounds where they can be hit (Flattener::boundary_encloses)
    for (size_t i = 0; i < leaves.size(); ++i)
        if (leaves[i].unbounded) hoist.push_back(i);
    const size_t n_forced = hoist.size();
    {
        // peel off the largest box while it still fills >= 10 % of the box of what is left (sky first, then the ground
        // under it, ...): judged against the REMAINING prims, so a big ground inside a much bigger sky still qualifies
        std::vector<char> gone(leaves.size(), 0);
        for (size_t i : hoist) gone[i] = 1;
        while ((int)(hoist.size() - n_forced) < RT_MAX_HOISTED && hoist.size() + 1 < leaves.size()) {
            bool first = true;
            Aabb root{};
            size_t big = 0;
            double big_v = -1.0;
            for (size_t i = 0; i < leaves.size(); ++i) {
                if (gone[i]) continue;
                root = first ? leaves[i].bound : merged(root, leaves[i].bound);
                first = false;
                const double v = volume(leaves[i].bound);
                if (v > big_v) {
                    big_v = v;
                    big = i;
                }
            }
            if (!(volume(root) > 0.0) || !(big_v >= 0.1 * volume(root))) break;
            gone[big] = 1;
            hoist.push_back(big);
        }
        std::sort(hoist.begin(), hoist.end());
    }
    std::vector<size_t> final_order = hoist;
    for (size_t i : order)
        if (!std::binary_search(hoist.begin(), hoist.end(), i)) final_order.push_back(i);
    fs.n_hoisted = (int)hoist.size();
    for (size_t i : final_order) {
        Leaf &lf = leaves[i];
        if (lf.meta.kind != RT_PRIM_SPHERE_T && lf.meta.kind != RT_PRIM_MEDIUM_T) {
            lf.meta.xform = push_chain(fs, lf.chain);
            fl.finish_meta(&lf.meta, lf.chain);
        }
        if ((lf.meta.kind & 0xFFu) == RT_PRIM_RECT_C) {
            // the record's normal: Rectangle::hit's (0, 0, 1) (src/geometry.rs:176) carried up the chain as the kernel's
            // chain_up does it -- innermost level first, M (not M^-T, quirk Q5) row by row in Vec4::transformed's order
            // (src/vec4.rs:78-91) with w = 0, a pure-translation level leaving it alone.  A constant of the leaf.
            double n[3] = {0.0, 0.0, 1.0};
            for (size_t k = lf.chain.size(); k-- > 0;) {
                const Link &l = lf.chain[k];
                if (l.translation) continue;
                RtXform x;
                make_xform(l.M, l.Minv, &x);
                const double *m = x.m;
                const double a = m[0] * n[0] + m[1] * n[1] + m[2] * n[2] + m[3] * 0.0;
                const double b = m[4] * n[0] + m[5] * n[1] + m[6] * n[2] + m[7] * 0.0;
                const double c = m[8] * n[0] + m[9] * n[1] + m[10] * n[2] + m[11] * 0.0;
                n[0] = a, n[1] = b, n[2] = c;
            }
            lf.geo.g[2] = n[0];
            lf.geo.g[3] = n[1];
            lf.extra.e[0] = n[2];
        }
        if ((lf.meta.kind & 0xFFu) == RT_PRIM_SPHERE_T || (lf.meta.kind & 0xFFu) == RT_PRIM_MEDIUM_T)
            for (int c = 0; c < 4; ++c)
                if (!(std::fabs(lf.geo.g[c]) <= 0x1p100)) fs.world_mid = false; // (NaN included)
        fs.prim_meta.push_back(lf.meta);
        fs.prim_geo.push_back(lf.geo);
        fs.prim_extra.push_back(lf.extra);
        fs.prim_bounds.push_back(lf.bound);
    }
    fs.n_leaf_prims = (int)fs.prim_meta.size();

    // boundary prims of the general media (never BVH leaves): each medium's list is contiguous; a boundary prim that is itself
    // a medium points at its own list further on
    std::function<void(size_t, const std::vector<Shape> &)> emit_boundary = [&](size_t owner, const std::vector<Shape> &shapes) {
        const size_t first = fs.prim_meta.size();
        fs.prim_geo[owner].g[1] = (double)first;
        fs.prim_geo[owner].g[2] = (double)shapes.size();
        for (const Shape &sh : shapes) {
            RtPrimMeta c{};
            c.kind = sh.kind;
            c.material = RT_NO_MATERIAL;
            c.xform = push_chain(fs, sh.chain);
            c.aux = sh.key;
            fl.finish_meta(&c, sh.chain);
            RtPrimGeo cg{};
            cg.g[0] = sh.a;
            cg.g[1] = sh.b;
            fs.prim_meta.push_back(c);
            fs.prim_geo.push_back(cg);
            fs.prim_extra.push_back(RtPrimExtra{});
            fs.prim_bounds.push_back(Aabb{{0, 0, 0}, {0, 0, 0}});
        }
        for (size_t k = 0; k < shapes.size(); ++k)
            if (shapes[k].kind == RT_PRIM_MEDIUM_C) emit_boundary(first + k, shapes[k].inner);
    };
    for (size_t k = 0; k < final_order.size(); ++k) {
        const Leaf &lf = leaves[final_order[k]];
        if ((lf.meta.kind & 0xFFu) == RT_PRIM_MEDIUM_C) emit_boundary(k, lf.boundary);
    }

    // Cube groups (RT_META_GROUP_BIT, RtCubeGroup): the six faces of one Cube::new become ONE leaf of the tree when their culling boxes
    // are the six sides of one box -- a cube whose sprite chain keeps it axis-aligned: the 400 floor boxes of examples/main.rs:161-201
    // are 2400 of the cover's 3400 leaves.  The leaf step derives the faces' boxes from twelve planes (rtl::trav_leaf_step); each
    // derived box is checked here, plane by plane, to CONTAIN the face's own binary32 culling box.  Only for the kernel families whose
    // leaf step knows groups (rtl::CubeGroups: general prims, tree walk, 16-bit references, no media over general boundaries), never in a
    // scene small enough for the box list.  RT_NO_CUBE_GROUPS=1: every face a leaf of its own (A/B runs).
    std::vector<int> group_len((size_t)fs.n_leaf_prims, 1); // 6 at the head of a group, 0 at its other five prims
    {
        const char *no_groups = std::getenv("RT_NO_CUBE_GROUPS");
        const char *no_list_env = std::getenv("RT_NO_LIST");
        const int n_bvh = fs.n_leaf_prims - fs.n_hoisted;
        const bool list_scene = (fs.feature_mask & RT_FEAT_GENERAL) && !(fs.feature_mask & RT_FEAT_MEDIUM_NESTED) && n_bvh >= 2 && n_bvh <= RT_LIST_MAX &&
                                fs.n_leaf_prims <= (1 << RT_LIST_PRIM_BITS) && !(no_list_env && *no_list_env == '1');
        const bool family_ok = (fs.feature_mask & RT_FEAT_GENERAL) &&
                               !(fs.feature_mask & (RT_FEAT_MEDIUM_GENERAL | RT_FEAT_DEEP_CHAIN | RT_FEAT_MEDIUM_NESTED)) &&
                               (size_t)fs.n_leaf_prims <= RT_REF_MAX;
        if (family_ok && !list_scene && !(no_groups && *no_groups == '1')) {
            for (int k = fs.n_hoisted; k + 6 <= fs.n_leaf_prims; ++k) {
                const Leaf &h = leaves[final_order[(size_t)k]];
                if (h.cube < 0 || h.face != 0) continue;
                bool six = true;
                for (int f = 1; f < 6 && six; ++f) {
                    const Leaf &lf = leaves[final_order[(size_t)(k + f)]];
                    six = lf.cube == h.cube && lf.face == f;
                }
                if (!six) continue;
                // the faces' binary32 culling boxes, their union, and each face's slot: the axis on which it is thin, the side it is on
                float lo[6][3], hi[6][3], glo[3], ghi[3];
                for (int f = 0; f < 6; ++f) cull_box(fs.prim_bounds[(size_t)(k + f)], lo[f], hi[f]);
                bool finite = true;
                for (int a = 0; a < 3; ++a) {
                    glo[a] = lo[0][a];
                    ghi[a] = hi[0][a];
                    for (int f = 0; f < 6; ++f) {
                        glo[a] = std::fmin(glo[a], lo[f][a]);
                        ghi[a] = std::fmax(ghi[a], hi[f][a]);
                        finite = finite && std::isfinite(lo[f][a]) && std::isfinite(hi[f][a]);
                    }
                }
                if (!finite) continue;
                int face_of_slot[6] = {-1, -1, -1, -1, -1, -1};
                bool ok = true;
                for (int f = 0; f < 6 && ok; ++f) {
                    int axis = -1;
                    double thin = 0.02; // a face has to be thinner than 2 % of the box on its axis to be one of its sides
                    for (int a = 0; a < 3; ++a) {
                        const double ext = (double)ghi[a] - (double)glo[a];
                        const double rel = ext > 0.0 ? ((double)hi[f][a] - (double)lo[f][a]) / ext : 1.0;
                        if (rel < thin) {
                            thin = rel;
                            axis = a;
                        }
                    }
                    if (axis < 0) {
                        ok = false;
                        break;
                    }
                    const double mid = 0.5 * ((double)lo[f][axis] + (double)hi[f][axis]), gmid = 0.5 * ((double)glo[axis] + (double)ghi[axis]);
                    const int slot = 2 * axis + (mid < gmid ? 0 : 1);
                    if (face_of_slot[slot] >= 0) ok = false;
                    face_of_slot[slot] = f;
                }
                if (!ok) continue;
                RtCubeGroup cg{};
                for (int a = 0; a < 3; ++a) {
                    cg.outer_lo[a] = glo[a];
                    cg.outer_hi[a] = ghi[a];
                    cg.inner_lo[a] = hi[face_of_slot[2 * a]][a];     // where the low face's slab ends
                    cg.inner_hi[a] = lo[face_of_slot[2 * a + 1]][a]; // where the high face's slab begins
                }
                // every derived box contains the face's own culling box (outer planes: by the union; inner planes: by construction --
                // asserted all the same, and the slabs must not be inverted)
                for (int sl = 0; sl < 6 && ok; ++sl) {
                    const int f = face_of_slot[sl], a = sl / 2;
                    for (int b = 0; b < 3 && ok; ++b) {
                        const float dlo = (b == a && (sl & 1)) ? cg.inner_hi[a] : cg.outer_lo[b];
                        const float dhi = (b == a && !(sl & 1)) ? cg.inner_lo[a] : cg.outer_hi[b];
                        ok = dlo <= lo[f][b] && dhi >= hi[f][b] && dlo <= dhi;
                    }
                }
                if (!ok) continue;
                cg.first_prim = (uint32_t)k;
                for (int sl = 0; sl < 6; ++sl) cg.faces |= (uint32_t)face_of_slot[sl] << (3 * sl);
                fs.cube_groups.push_back(cg);
                group_len[(size_t)k] = 6;
                for (int f = 1; f < 6; ++f) group_len[(size_t)(k + f)] = 0;
                k += 5;
            }
        }
    }
    fs.group_len = group_len;
    // BVH over the non-hoisted leaves (a cube group is one leaf with the union of its faces' boxes)
    if (fs.n_hoisted < fs.n_leaf_prims) {
        std::vector<Aabb> item_bounds;
        std::vector<int> item_head;
        for (int k = fs.n_hoisted; k < fs.n_leaf_prims; ++k) {
            if (group_len[(size_t)k] == 0) continue;
            Aabb b = fs.prim_bounds[(size_t)k];
            for (int f = 1; f < group_len[(size_t)k]; ++f) b = merged(b, fs.prim_bounds[(size_t)(k + f)]);
            item_bounds.push_back(b);
            item_head.push_back(k);
        }
        int32_t r = build_bvh(item_bounds, 0, (int)item_bounds.size(), &fs.host_nodes, &fs.max_depth);
        // leaf references of the items -> of their (head) prims
        auto to_prim = [&](int32_t c) { return c >= 0 ? c : ~item_head[(size_t)(~c)]; };
        r = to_prim(r);
        for (HostNode &h : fs.host_nodes)
            for (int c = 0; c < 2; ++c) h.child[c] = to_prim(h.child[c]);
        // up to 32767 prims and nodes: 16-bit references (one LDS word per stack entry); beyond: 32-bit ones
        fs.wide = (size_t)fs.n_leaf_prims > RT_REF_MAX || fs.host_nodes.size() > RT_REF_MAX;
        if (fs.wide) fs.feature_mask |= RT_FEAT_WIDE;
        const bool wide = fs.wide;
        auto ref16 = [wide](int32_t c) {
            return c >= 0 ? (uint32_t)c : ((wide ? RT_REF_LEAF_W : RT_REF_LEAF) | (uint32_t)(~c));
        };
        fs.root = ref16(r);
        for (const HostNode &h : fs.host_nodes) {
            RtNode n{};
            for (int c = 0; c < 2; ++c) {
                float lo[3], hi[3];
                cull_box(h.box[c], lo, hi);
                n.lo_x[c] = lo[0];
                n.lo_y[c] = lo[1];
                n.lo_z[c] = lo[2];
                n.hi_x[c] = hi[0];
                n.hi_y[c] = hi[1];
                n.hi_z[c] = hi[2];
                n.child[c] = ref16(h.child[c]);
            }
            fs.nodes.push_back(n);
        }
        // the same nodes with binary16 planes (RtNodeH), for an LDS copy where the binary32 form does not fit: every plane rounded
        // outward once more, onto the binary16 grid.  Not built when a plane is not finite or too large for that grid.
        if (!fs.wide) {
            bool ok = true;
            std::vector<RtNodeH> half;
            for (const RtNode &n : fs.nodes) {
                RtNodeH h{};
                const float *lo[3] = {n.lo_x, n.lo_y, n.lo_z}, *hi[3] = {n.hi_x, n.hi_y, n.hi_z};
                uint16_t *hlo[3] = {h.lo_x, h.lo_y, h.lo_z}, *hhi[3] = {h.hi_x, h.hi_y, h.hi_z};
                for (int a = 0; a < 3 && ok; ++a)
                    for (int c = 0; c < 2 && ok; ++c) {
                        ok = std::isfinite(lo[a][c]) && std::isfinite(hi[a][c]) && std::fabs(lo[a][c]) <= 60000.0f && std::fabs(hi[a][c]) <= 60000.0f;
                        hlo[a][c] = half_toward(lo[a][c], false);
                        hhi[a][c] = half_toward(hi[a][c], true);
                    }
                h.child[0] = (uint16_t)n.child[0];
                h.child[1] = (uint16_t)n.child[1];
                half.push_back(h);
            }
            if (ok) fs.nodes_half = std::move(half);
        }
    } else {
        fs.root = RT_CUR_DONE; // every prim is hoisted (only possible in the 16-bit form)
        fs.max_depth = 0;
    }
    // the cube groups' records behind the prims in prim_geo (two slots each), the head prim's meta word and aux pointing at them
    for (const RtCubeGroup &cg : fs.cube_groups) {
        const size_t at = fs.prim_geo.size();
        fs.prim_geo.resize(at + 2);
        std::memcpy(&fs.prim_geo[at], &cg, sizeof cg);
        fs.prim_meta[cg.first_prim].kind |= RT_META_GROUP_BIT;
        fs.prim_meta[cg.first_prim].aux = (uint32_t)at;
    }
    // Small general scenes (the Cornell box: 18 leaves) are walked as a LIST: every lane of a wave tests all the leaves'
    // culling boxes in lock step (rtl::trav_list_step) instead of descending a five-level tree with half the wave idle, then
    // the binary64 tests run nearest box first exactly as after a tree walk.  Same boxes as the tree's leaf boxes (cull_box),
    // same binary64 tests, same tie rule: which boxes are looked at first never reaches a result.  The reference's own
    // counterpart is the linear scan of a `Vec` world (src/geometry.rs:76-116).  The tree is still built (inspection, tests);
    // the device gets the box list in the node array's place.  RT_NO_LIST=1 keeps the tree walk (A/B runs).
    const int n_bvh_leaves = fs.n_leaf_prims - fs.n_hoisted;
    const char *no_list = std::getenv("RT_NO_LIST");
    // (a scene with media inside media is always walked as a tree: its kernel family is compiled in that form only)
    if ((fs.feature_mask & RT_FEAT_GENERAL) && !(fs.feature_mask & RT_FEAT_MEDIUM_NESTED) && !fs.wide && n_bvh_leaves >= 2 && n_bvh_leaves <= RT_LIST_MAX &&
        fs.n_leaf_prims <= (1 << RT_LIST_PRIM_BITS) && // (a list kernel's stack entry has that many bits for the leaf's prim index)
        !(no_list && *no_list == '1')) {
        fs.n_list = n_bvh_leaves;
        std::vector<float> packed((size_t)n_bvh_leaves * RT_LIST_BOX_FLOATS, 0.0f);
        for (int i = 0; i < n_bvh_leaves; ++i) {
            float lo[3], hi[3];
            cull_box(fs.prim_bounds[(size_t)(fs.n_hoisted + i)], lo, hi);
            for (int a = 0; a < 3; ++a) {
                float *p = &packed[(size_t)i * RT_LIST_BOX_FLOATS + (size_t)a * 3];
                p[0] = lo[a];
                p[1] = hi[a];
                p[2] = lo[a];
            }
        }
        const size_t per = sizeof(RtNode) / sizeof(float);
        fs.nodes.assign((packed.size() + per - 1) / per, RtNode{});
        std::memcpy(fs.nodes.data(), packed.data(), packed.size() * sizeof(float));
        fs.root = 0u; // "at a node": the list step runs first (stack need: every box but the nearest may be pushed, n_list - 1)
    }
    // the material's kind rides in the prim's meta word (RtPrimMeta::kind bits 8-15)
    for (RtPrimMeta &m : fs.prim_meta) {
        const uint32_t mk = m.material == RT_NO_MATERIAL ? (uint32_t)RT_MAT_KIND_NONE : fs.materials[m.material].kind;
        m.kind = (m.kind & ~0xFF00u) | (mk << 8);
    }
    // the kernels address every scene array with a 32-bit byte offset built from a 24-bit record index (rtl::rec_at)
    const size_t most = std::max({fs.prim_meta.size(), fs.prim_geo.size(), fs.xforms.size(), fs.materials.size(), fs.textures.size(), fs.nodes.size()});
    if (most > (size_t)RT_MAX_RECORDS || fs.image_blob.size() >= ((size_t)1 << 32)) {
        if (err) *err = "more than 2^24 records in one scene array (prims, transforms, materials, textures, nodes) or 4 GiB of texels";
        return RT_ERR_UNSUPPORTED;
    }
    // what the device (and the CPU harness of the tests) reads: the boxes in reverse, then the transform records (RtXformBox)
    fs.xform_store.assign(fs.xforms.size() * (sizeof(RtXformBox) + sizeof(RtXform)) / sizeof(double), 0.0);
    for (size_t i = 0; i < fs.xforms.size(); ++i) {
        std::memcpy(fs.xform_store.data() + (fs.xforms.size() - 1 - i) * (sizeof(RtXformBox) / sizeof(double)), &fs.xform_boxes[i], sizeof(RtXformBox));
        std::memcpy(fs.xform_store.data() + fs.xforms.size() * (sizeof(RtXformBox) / sizeof(double)) + i * (sizeof(RtXform) / sizeof(double)), &fs.xforms[i],
                    sizeof(RtXform));
    }
    // the records the LIST kernels copy into LDS when they fit beside the stack and the queues (rt_api.cpp render_range decides per
    // launch), in the layout RtLaunch's offsets assume
    // ... and, since round 5, the kernels of a small TREE scene of the families that have that form (general prims, 16-bit references, no
    // media over general boundaries: rt_kernels.hip part 4): what the Cornell box gained from it (+ 9.5 %, round 4) is there for any scene
    // whose primitive tests read a transform level per chain level through the L1
    const bool small_tree = fs.n_list == 0 && (fs.feature_mask & RT_FEAT_GENERAL) && !fs.wide && fs.n_leaf_prims <= RT_RECLDS_TREE_MAX &&
                            !(fs.feature_mask & (RT_FEAT_MEDIUM_GENERAL | RT_FEAT_DEEP_CHAIN | RT_FEAT_MEDIUM_NESTED));
    if ((fs.n_list > 0 || small_tree) &&
        FlatScene::scene_blob_bytes(fs.xforms.size(), fs.prim_meta.size(), fs.materials.size(), fs.prim_geo.size()) <= (size_t)RT_LIST_SCENE_MAX) {
        auto put = [&](int k, const void *src, size_t bytes) {
            fs.scene_blob_off[k] = (uint32_t)fs.scene_blob.size();
            fs.scene_blob.resize(fs.scene_blob.size() + ((bytes + 15) & ~(size_t)15), 0);
            if (bytes) std::memcpy(fs.scene_blob.data() + fs.scene_blob_off[k], src, bytes);
        };
        put(0, fs.xforms.data(), fs.xforms.size() * sizeof(RtXform));
        put(1, fs.prim_geo.data(), fs.prim_geo.size() * sizeof(RtPrimGeo));
        put(2, fs.prim_meta.data(), fs.prim_meta.size() * sizeof(RtPrimMeta));
        put(3, fs.prim_extra.data(), fs.prim_extra.size() * sizeof(RtPrimExtra));
        put(4, fs.materials.data(), fs.materials.size() * sizeof(RtMaterial));
    }
    *out = std::move(fs);
    return RT_OK;
}

// ---------------------------------------------------------------- BVH build
namespace {

struct Builder {
    const std::vector<Aabb> &bounds;
    std::vector<HostNode> &nodes;
    int max_depth = 0;

    static double area(const Aabb &b) {
        const double dx = b.hi[0] - b.lo[0], dy = b.hi[1] - b.lo[1], dz = b.hi[2] - b.lo[2];
        return 2.0 * (dx * dy + dy * dz + dz * dx);
    }
    static int ceil_log2(size_t n) {
        int k = 0;
        while (((size_t)1 << k) < n) ++k;
        return k;
    }
    Aabb bound_of(const std::vector<int> &ids) const {
        Aabb b = bounds[(size_t)ids[0]];
        for (size_t i = 1; i < ids.size(); ++i) b = merged(b, bounds[(size_t)ids[i]]);
        return b;
    }

    // returns child reference (node index, or ~prim); `budget` = inner levels still allowed
    int32_t build(std::vector<int> ids, int depth, int budget, Aabb *box) {
        *box = bound_of(ids);
        if (ids.size() == 1) {
            max_depth = std::max(max_depth, depth);
            return ~ids[0];
        }
        const size_t n = ids.size();
        size_t best_split = n / 2;
        int best_axis = 0;
        double best_cost = kInf;
        std::vector<int> sorted[3];
        std::vector<double> right_area(n);
        for (int axis = 0; axis < 3; ++axis) {
            sorted[axis] = ids;
            std::stable_sort(sorted[axis].begin(), sorted[axis].end(), [&](int a, int b) {
                const double ca = bounds[(size_t)a].lo[axis] + bounds[(size_t)a].hi[axis];
                const double cb = bounds[(size_t)b].lo[axis] + bounds[(size_t)b].hi[axis];
                return ca < cb;
            });
            const std::vector<int> &s = sorted[axis];
            Aabb acc = bounds[(size_t)s[n - 1]];
            right_area[n - 1] = area(acc);
            for (size_t i = n - 1; i-- > 0;) {
                acc = merged(acc, bounds[(size_t)s[i]]);
                right_area[i] = area(acc);
            }
            acc = bounds[(size_t)s[0]];
            for (size_t i = 1; i < n; ++i) { // left = [0,i), right = [i,n)
                const double cost = area(acc) * (double)i + right_area[i] * (double)(n - i);
                if (cost < best_cost) {
                    best_cost = cost;
                    best_axis = axis;
                    best_split = i;
                }
                acc = merged(acc, bounds[(size_t)s[i]]);
            }
        }
        // depth guard: both halves must still fit the remaining stack budget
        const size_t big = std::max(best_split, n - best_split);
        if (!std::isfinite(best_cost) || (big > 1 && ceil_log2(big) > budget - 1)) best_split = n / 2;
        const std::vector<int> &s = sorted[best_axis];
        std::vector<int> left(s.begin(), s.begin() + (long)best_split), right(s.begin() + (long)best_split, s.end());

        const int32_t me = (int32_t)nodes.size();
        nodes.emplace_back();
        Aabb lb, rb;
        const int32_t c0 = build(std::move(left), depth + 1, budget - 1, &lb);
        const int32_t c1 = build(std::move(right), depth + 1, budget - 1, &rb);
        HostNode &nd = nodes[(size_t)me];
        nd.box[0] = lb;
        nd.box[1] = rb;
        nd.child[0] = c0;
        nd.child[1] = c1;
        return me;
    }
};

} // namespace

int32_t build_bvh(const std::vector<Aabb> &bounds, int first, int n, std::vector<HostNode> *nodes, int *max_depth) {
    nodes->clear();
    std::vector<int> ids;
    for (int i = first; i < n; ++i) ids.push_back(i);
    Builder b{bounds, *nodes};
    Aabb box;
    const int32_t root = b.build(std::move(ids), 0, RT_STACK_DEPTH - 1, &box);
    *max_depth = b.max_depth;
    return root; // a single prim yields a leaf reference and no nodes
}

} // namespace rt
