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

#include <algorithm>
#include <cmath>
#include <cstring>
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

// AABB of the eight transformed corners (src/optimize.rs:149-177 = :205-233)
Aabb transformed_bound(const Aabb &b, const double M[16]) {
    const double x0 = b.lo[0], y0 = b.lo[1], z0 = b.lo[2], x1 = b.hi[0], y1 = b.hi[1], z1 = b.hi[2];
    const double c[8][3] = {{x0, y0, z0}, {x1, y0, z0}, {x0, y1, z0}, {x0, y0, z1},
                            {x1, y1, z0}, {x1, y0, z1}, {x0, y1, z1}, {x1, y1, z1}};
    Aabb o;
    for (int i = 0; i < 3; ++i) {
        o.lo[i] = kInf;
        o.hi[i] = -kInf;
    }
    for (const auto &p : c) {
        double q[3];
        xform_point(M, p, q);
        for (int i = 0; i < 3; ++i) {
            if (q[i] < o.lo[i]) o.lo[i] = q[i];
            if (q[i] > o.hi[i]) o.hi[i] = q[i];
        }
    }
    return o;
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

    // sprites -> leaf prims (in creation order first; hoisted ones are moved to the front below)
    struct Leaf {
        RtPrimMeta meta;
        RtPrimGeo geo;
        RtPrimExtra extra;
        Aabb bound;
        bool is_group;
        CubeFace faces[6];
    };
    std::vector<Leaf> leaves;
    uint32_t medium_slots = 0;
    for (const SpriteIR &s : ir.sprites) {
        if (s.geometry < 0) continue; // geometry None: never hit (src/sprite.rs:95,136)
        const GeometryIR &g = ir.geometries[(size_t)s.geometry];
        uint32_t slot = 0;
        if (g.kind == GEO_MEDIUM) slot = (medium_slots++) & 0x3FFu; // slots count sprites in creation order
        double Minv[16];
        if (!mat4_inversed(s.M, Minv)) continue; // det == 0: unhittable (src/sprite.rs:131-134)
        Leaf lf{};
        RtPrimMeta &p = lf.meta;
        double *geo = lf.geo.g;
        p.material = s.material < 0 ? RT_NO_MATERIAL : (uint32_t)s.material;
        const bool trans = is_pure_translation(s.M);
        Aabb local{};
        switch (g.kind) {
        case GEO_SPHERE:
            local = sphere_bound(g.p[0]);
            if (trans) {
                p.kind = RT_PRIM_SPHERE_T;
                geo[0] = s.M[12];
                geo[1] = s.M[13];
                geo[2] = s.M[14];
                geo[3] = g.p[0];
                fs.feature_mask |= RT_FEAT_SPHERE_T;
            } else {
                p.kind = RT_PRIM_SPHERE_M;
                geo[0] = g.p[0];
                fs.feature_mask |= RT_FEAT_GENERAL;
            }
            break;
        case GEO_RECTANGLE:
            local = rect_bound(g.p[0], g.p[1]);
            p.kind = RT_PRIM_RECT_M;
            geo[0] = g.p[0];
            geo[1] = g.p[1];
            fs.feature_mask |= RT_FEAT_GENERAL;
            break;
        case GEO_CUBE: {
            lf.is_group = true;
            cube_faces(g.p[0], g.p[1], g.p[2], lf.faces);
            bool first = true;
            for (const CubeFace &f : lf.faces) {
                Aabb fb = transformed_bound(rect_bound(f.w, f.h), f.M);
                local = first ? fb : merged(local, fb);
                first = false;
            }
            p.kind = RT_PRIM_GROUP_M;
            geo[0] = 6.0;
            fs.feature_mask |= RT_FEAT_GENERAL;
            break;
        }
        case GEO_MEDIUM: {
            const GeometryIR &b = ir.geometries[(size_t)g.boundary];
            local = sphere_bound(b.p[0]);
            p.aux = slot;
            if (trans) {
                p.kind = RT_PRIM_MEDIUM_T;
                geo[0] = s.M[12];
                geo[1] = s.M[13];
                geo[2] = s.M[14];
                geo[3] = b.p[0];
                lf.extra.e[0] = g.p[0];
            } else {
                p.kind = RT_PRIM_MEDIUM_M;
                geo[0] = b.p[0];
                geo[1] = g.p[0];
            }
            fs.feature_mask |= RT_FEAT_MEDIUM;
            break;
        }
        }
        if (p.kind != RT_PRIM_SPHERE_T && p.kind != RT_PRIM_MEDIUM_T) {
            RtXform x;
            make_xform(s.M, Minv, &x);
            p.xform = (uint32_t)fs.xforms.size();
            fs.xforms.push_back(x);
        }
        lf.bound = transformed_bound(local, s.M);
        pad(&lf.bound);
        leaves.push_back(lf);
    }
    if (leaves.empty()) {
        if (err) *err = "empty scene: BoundingVolumeHierarchyNode::new(vec![]) is None (src/optimize.rs:367-370)";
        return RT_ERR_EMPTY;
    }
    if (leaves.size() > RT_REF_MAX) {
        if (err) *err = "more than 32767 sprites: outside the 16-bit node references of this build";
        return RT_ERR_UNSUPPORTED;
    }

    // hoist scene-spanning prims (book-one's sky and ground spheres, main.rs's fog): their
    // boxes make every ancestor an always-hit, so they are tested up front for each
    // segment and give the traversal an early upper bound instead.
    auto area = [](const Aabb &b) {
        const double dx = b.hi[0] - b.lo[0], dy = b.hi[1] - b.lo[1], dz = b.hi[2] - b.lo[2];
        return 2.0 * (dx * dy + dy * dz + dz * dx);
    };
    Aabb root = leaves[0].bound;
    for (const Leaf &l : leaves) root = merged(root, l.bound);
    std::vector<size_t> order(leaves.size());
    std::iota(order.begin(), order.end(), (size_t)0);
    std::vector<size_t> hoist;
    {
        std::vector<size_t> by_area = order;
        std::stable_sort(by_area.begin(), by_area.end(), [&](size_t a, size_t b) { return area(leaves[a].bound) > area(leaves[b].bound); });
        for (size_t i : by_area) {
            if ((int)hoist.size() >= RT_MAX_HOISTED) break;
            if (!(area(leaves[i].bound) >= 0.1 * area(root))) break;
            hoist.push_back(i);
        }
        std::sort(hoist.begin(), hoist.end());
    }
    std::vector<size_t> final_order = hoist;
    for (size_t i : order)
        if (!std::binary_search(hoist.begin(), hoist.end(), i)) final_order.push_back(i);
    fs.n_hoisted = (int)hoist.size();
    for (size_t i : final_order) {
        fs.prim_meta.push_back(leaves[i].meta);
        fs.prim_geo.push_back(leaves[i].geo);
        fs.prim_extra.push_back(leaves[i].extra);
        fs.prim_bounds.push_back(leaves[i].bound);
    }
    fs.n_leaf_prims = (int)fs.prim_meta.size();

    // group children (never BVH leaves): TransformedGeometry<Rectangle> records
    for (size_t k = 0; k < final_order.size(); ++k) {
        const Leaf &lf = leaves[final_order[k]];
        if (!lf.is_group) continue;
        fs.prim_meta[k].aux = (uint32_t)fs.prim_meta.size();
        int kept = 0;
        for (const CubeFace &f : lf.faces) {
            double Minv[16];
            if (!mat4_inversed(f.M, Minv)) continue;
            RtPrimMeta c{};
            c.kind = RT_PRIM_RECT_M;
            c.material = RT_NO_MATERIAL;
            RtPrimGeo cg{};
            cg.g[0] = f.w;
            cg.g[1] = f.h;
            RtXform x;
            make_xform(f.M, Minv, &x);
            c.xform = (uint32_t)fs.xforms.size();
            fs.xforms.push_back(x);
            fs.prim_meta.push_back(c);
            fs.prim_geo.push_back(cg);
            fs.prim_extra.push_back(RtPrimExtra{});
            fs.prim_bounds.push_back(Aabb{{0, 0, 0}, {0, 0, 0}});
            ++kept;
        }
        fs.prim_geo[k].g[0] = (double)kept;
    }

    // BVH over the non-hoisted leaves
    if (fs.n_hoisted < fs.n_leaf_prims) {
        const int32_t r = build_bvh(fs.prim_bounds, fs.n_hoisted, fs.n_leaf_prims, &fs.host_nodes, &fs.max_depth);
        if (fs.host_nodes.size() > RT_REF_MAX) {
            if (err) *err = "BVH has more than 32767 nodes: outside the 16-bit node references of this build";
            return RT_ERR_UNSUPPORTED;
        }
        auto ref16 = [](int32_t c) { return c >= 0 ? (uint32_t)c : (RT_REF_LEAF | (uint32_t)(~c)); };
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
    } else {
        fs.root = RT_CUR_DONE;
        fs.max_depth = 0;
    }
    // the material's kind rides in the prim's meta word (RtPrimMeta::kind bits 8-15)
    for (RtPrimMeta &m : fs.prim_meta) {
        const uint32_t mk = m.material == RT_NO_MATERIAL ? (uint32_t)RT_MAT_KIND_NONE : fs.materials[m.material].kind;
        m.kind = (m.kind & 0xFFu) | (mk << 8);
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
