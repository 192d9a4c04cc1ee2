// rt_oracle.cpp -- CPU oracle: plain C++17 / f64 restatement of the reference's
// per-pixel sampling loop.  TEST INFRASTRUCTURE ONLY (see rt_oracle.h).
//
// Build: g++ -O2 -std=c++17 -ffp-contract=off -fno-fast-math (oracle/Makefile).
// Every function cites the reference file:line it follows.  Structure follows
// the reference on purpose (trait objects -> virtual classes, recursive
// unpruned BVH, a 4x4 transform pair per sprite, uv on every sphere hit): in
// "reference" form it is also the CPU baseline that bench.py times.
//
// Parity unpinned against the Rust binary (cannot be built here; no golden
// vectors upstream); pinned by hand-derived KATs in tests/test_oracle_kat.py and,
// statistically, by the reference's one published render (cover.png): region
// means of the 8-bit picture at 800x800x1000, tests/test_cover_png.py.

#include "rt_oracle.h"

#include "../include/rt_rng.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <limits>
#include <memory>
#include <thread>
#include <vector>

namespace {

constexpr double kInf = std::numeric_limits<double>::infinity();
constexpr double kPi = 3.14159265358979323846264338327950288; // std::f64::consts::PI

// ---------------------------------------------------------------- src/vec3.rs
struct Vec3 {
    double x, y, z;
    double operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); } // vec3.rs:270-281
};
inline Vec3 v3(double x, double y, double z) { return Vec3{x, y, z}; }
inline Vec3 operator-(Vec3 a) { return v3(-a.x, -a.y, -a.z); }                      // vec3.rs:127-137
inline Vec3 operator+(Vec3 a, Vec3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); } // vec3.rs:139-149
inline Vec3 operator+(Vec3 a, double s) { return v3(a.x + s, a.y + s, a.z + s); }     // vec3.rs:153-163
inline Vec3 operator-(Vec3 a, Vec3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); } // vec3.rs:173-183
inline Vec3 operator-(Vec3 a, double s) { return v3(a.x - s, a.y - s, a.z - s); }     // vec3.rs:185-195
inline Vec3 operator*(Vec3 a, double s) { return v3(a.x * s, a.y * s, a.z * s); }     // vec3.rs:205-215
inline Vec3 operator*(double s, Vec3 a) { return v3(a.x * s, a.y * s, a.z * s); }     // vec3.rs:217-227
inline Vec3 operator*(Vec3 a, Vec3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); } // vec3.rs:229-239
inline Vec3 operator/(Vec3 a, double s) { return v3(a.x / s, a.y / s, a.z / s); }     // vec3.rs:249-259
inline double dot(Vec3 a, Vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }       // vec3.rs:76-78
inline Vec3 cross(Vec3 a, Vec3 b) {                                                   // vec3.rs:80-86
    return v3(a.y * b.z - a.z * b.y, -(a.x * b.z - a.z * b.x), a.x * b.y - a.y * b.x);
}
inline double length(Vec3 a) { return std::sqrt(a.x * a.x + a.y * a.y + a.z * a.z); } // vec3.rs:88-90
inline Vec3 normalized(Vec3 a) { return a / length(a); }                              // vec3.rs:96-98
inline Vec3 reflected(Vec3 v, Vec3 n) { return v - n * dot(v, n) * 2.0; }             // vec3.rs:100-102
// vec3.rs:113-124 -- note: the result is built from the UN-normalised self
inline bool refracted(Vec3 self, Vec3 normal, double ratio, Vec3 *out) {
    Vec3 uv = normalized(self);
    double dt = dot(uv, normal);
    double discriminant = 1.0 - ratio * ratio * (1.0 - dt * dt);
    if (discriminant > 0.0) {
        *out = ratio * (self - normal * dt) - normal * std::sqrt(discriminant);
        return true;
    }
    return false;
}

// ------------------------------------------------------- src/vec4.rs, mat4.rs
struct Mat4 {
    double a[16]; // column-major, mat4.rs:5-17
};
inline Mat4 mat4_zero() {
    Mat4 m;
    for (double &v : m.a) v = 0.0;
    return m;
}
inline Mat4 mat4_identity() { // mat4.rs:21-28
    Mat4 m = mat4_zero();
    m.a[0] = m.a[5] = m.a[10] = m.a[15] = 1.0;
    return m;
}
inline Mat4 mat4_translation(Vec3 t) { // mat4.rs:36-47
    Mat4 m = mat4_identity();
    m.a[12] = t.x;
    m.a[13] = t.y;
    m.a[14] = t.z;
    return m;
}
inline Mat4 mat4_rotation(double radians, Vec3 axis) { // mat4.rs:52-80
    double x = axis.x, y = axis.y, z = axis.z;
    double s = std::sin(radians), c = std::cos(radians), t = 1.0 - c;
    Mat4 m;
    double a[16] = {x * x * t + c,     y * x * t + z * s, z * x * t - y * s, 0.0,
                    x * y * t - z * s, y * y * t + c,     z * y * t + x * s, 0.0,
                    x * z * t + y * s, y * z * t - x * s, z * z * t + c,     0.0,
                    0.0,               0.0,               0.0,               1.0};
    std::memcpy(m.a, a, sizeof a);
    return m;
}
inline Mat4 mat4_multiplied(const Mat4 &self, const Mat4 &other) { // mat4.rs:85-143
    const double *A = self.a;
    Mat4 r;
    for (int col = 0; col < 4; ++col) {
        double b0 = other.a[col * 4 + 0], b1 = other.a[col * 4 + 1], b2 = other.a[col * 4 + 2], b3 = other.a[col * 4 + 3];
        r.a[col * 4 + 0] = b0 * A[0] + b1 * A[4] + b2 * A[8] + b3 * A[12];
        r.a[col * 4 + 1] = b0 * A[1] + b1 * A[5] + b2 * A[9] + b3 * A[13];
        r.a[col * 4 + 2] = b0 * A[2] + b1 * A[6] + b2 * A[10] + b3 * A[14];
        r.a[col * 4 + 3] = b0 * A[3] + b1 * A[7] + b2 * A[11] + b3 * A[15];
    }
    return r;
}
struct Minors {
    double b00, b01, b02, b03, b04, b05, b06, b07, b08, b09, b10, b11;
};
inline Minors mat4_minors(const Mat4 &m) { // mat4.rs:166-177 / 210-221
    const double *a = m.a;
    double a00 = a[0], a01 = a[1], a02 = a[2], a03 = a[3], a10 = a[4], a11 = a[5], a12 = a[6], a13 = a[7];
    double a20 = a[8], a21 = a[9], a22 = a[10], a23 = a[11], a30 = a[12], a31 = a[13], a32 = a[14], a33 = a[15];
    Minors q;
    q.b00 = a00 * a11 - a01 * a10;
    q.b01 = a00 * a12 - a02 * a10;
    q.b02 = a00 * a13 - a03 * a10;
    q.b03 = a01 * a12 - a02 * a11;
    q.b04 = a01 * a13 - a03 * a11;
    q.b05 = a02 * a13 - a03 * a12;
    q.b06 = a20 * a31 - a21 * a30;
    q.b07 = a20 * a32 - a22 * a30;
    q.b08 = a20 * a33 - a23 * a30;
    q.b09 = a21 * a32 - a22 * a31;
    q.b10 = a21 * a33 - a23 * a31;
    q.b11 = a22 * a33 - a23 * a32;
    return q;
}
inline double mat4_determinant(const Mat4 &m) { // mat4.rs:146-181
    Minors q = mat4_minors(m);
    return q.b00 * q.b11 - q.b01 * q.b10 + q.b02 * q.b09 + q.b03 * q.b08 - q.b04 * q.b07 + q.b05 * q.b06;
}
inline bool mat4_inversed(const Mat4 &m, Mat4 *out) { // mat4.rs:184-243
    double det = mat4_determinant(m);
    if (det == 0.0) return false;
    const double *a = m.a;
    double a00 = a[0], a01 = a[1], a02 = a[2], a03 = a[3], a10 = a[4], a11 = a[5], a12 = a[6], a13 = a[7];
    double a20 = a[8], a21 = a[9], a22 = a[10], a23 = a[11], a30 = a[12], a31 = a[13], a32 = a[14], a33 = a[15];
    Minors q = mat4_minors(m);
    double r[16] = {(a11 * q.b11 - a12 * q.b10 + a13 * q.b09) / det, (a02 * q.b10 - a01 * q.b11 - a03 * q.b09) / det,
                    (a31 * q.b05 - a32 * q.b04 + a33 * q.b03) / det, (a22 * q.b04 - a21 * q.b05 - a23 * q.b03) / det,
                    (a12 * q.b08 - a10 * q.b11 - a13 * q.b07) / det, (a00 * q.b11 - a02 * q.b08 + a03 * q.b07) / det,
                    (a32 * q.b02 - a30 * q.b05 - a33 * q.b01) / det, (a20 * q.b05 - a22 * q.b02 + a23 * q.b01) / det,
                    (a10 * q.b10 - a11 * q.b08 + a13 * q.b06) / det, (a01 * q.b08 - a00 * q.b10 - a03 * q.b06) / det,
                    (a30 * q.b04 - a31 * q.b02 + a33 * q.b00) / det, (a21 * q.b02 - a20 * q.b04 - a23 * q.b00) / det,
                    (a11 * q.b07 - a10 * q.b09 - a12 * q.b06) / det, (a00 * q.b09 - a01 * q.b07 + a02 * q.b06) / det,
                    (a31 * q.b01 - a30 * q.b03 - a32 * q.b00) / det, (a20 * q.b03 - a21 * q.b01 + a22 * q.b00) / det};
    std::memcpy(out->a, r, sizeof r);
    return true;
}
struct Mat4Cached { // mat4.rs:413-451
    Mat4 origin, inversed;
    double determinant;
    explicit Mat4Cached(const Mat4 &m) : origin(m), inversed(mat4_zero()), determinant(mat4_determinant(m)) {
        Mat4 inv;
        if (mat4_inversed(m, &inv)) inversed = inv;
    }
    const Mat4 *inv() const { return determinant == 0.0 ? nullptr : &inversed; } // mat4.rs:440-446
};
struct Vec4 {
    double x, y, z, w;
};
inline Vec4 transformed(Vec4 v, const Mat4 &t) { // vec4.rs:78-91
    const double *m = t.a;
    return Vec4{m[0] * v.x + m[4] * v.y + m[8] * v.z + m[12] * v.w, m[1] * v.x + m[5] * v.y + m[9] * v.z + m[13] * v.w,
                m[2] * v.x + m[6] * v.y + m[10] * v.z + m[14] * v.w, m[3] * v.x + m[7] * v.y + m[11] * v.z + m[15] * v.w};
}
inline Vec4 xyz1(Vec3 v) { return Vec4{v.x, v.y, v.z, 1.0}; } // vec3.rs:67-69
inline Vec4 xyz0(Vec3 v) { return Vec4{v.x, v.y, v.z, 0.0}; } // vec3.rs:72-74
inline Vec3 xyz(Vec4 v) { return v3(v.x, v.y, v.z); }         // vec4.rs:97-103

// ------------------------------------------------------------------ src/ray.rs
struct Ray { // ray.rs:7-33
    Vec3 origin, direction;
    Vec3 at(double t) const { return origin + direction * t; }
};
struct HitRecord { // ray.rs:36-83; Default at ray.rs:89-99
    double t = kInf;
    Vec3 intersection{0, 0, 0};
    Vec3 normal{0, 0, 0};
    int material = -1; // Option<&dyn Material>
    double u = 0.0, v = 0.0;
};

// per-sample context: the injected generator in place of thread_rng()
struct Ctx {
    rt_rng rng;
    uint32_t segment = 0;     // index of the current path segment (keyed medium draws)
    uint32_t medium_slot = 0; // slot of the sprite whose geometry is being tested
    uint64_t path_hash = 0;   // ranks / child indices of the sprites entered so far, folded (include/rt_rng.h, rt_medium_key_path)
    int path_len = 0;         // sprites entered so far: 1 = a sprite of the world's own list
    int medium_depth = 0;     // ConstantMedium::hit calls in progress (a medium inside another medium's boundary)
    uint32_t outer_key = 0, outer_pass = 0; // of the innermost one: its key and which of its two boundary.hit calls is running
    orc_counters cnt{0, 0, 0, 0, 0};
    double next53() { // rand::random::<f64>()
        ++cnt.rng_draws;
        return rt_rng_unit53(&rng);
    }
    double range01() { // thread_rng().gen_range(0.0, 1.0)
        ++cnt.rng_draws;
        return rt_rng_range01(&rng);
    }
    double range11() { // gen_range(-1.0, 1.0)
        ++cnt.rng_draws;
        return rt_rng_range11(&rng);
    }
    // key of the ConstantMedium::hit that starts now (include/rt_rng.h): the creation-order slot for a medium sprite of the
    // world's own list, else a key from the path of sprites above it; inside another medium's boundary, derived from the outer
    // evaluation's key and pass as well
    uint32_t medium_key() const {
        const bool own_slot = path_len == 1 && medium_slot < RT_MEDIUM_SLOT_MAX;
        const uint32_t key = own_slot ? medium_slot : rt_medium_key_path(path_hash);
        return medium_depth == 0 ? key : rt_medium_key_inner(outer_key, outer_pass, rt_medium_key_path(path_hash));
    }
    double keyed01(uint32_t key) { // ConstantMedium's gen_range(0.0, 1.0)
        ++cnt.rng_draws;
        return rt_u64_to_range01(rt_rng_medium_draw(rng.base, segment, key));
    }
};

// ----------------------------------------------------------------- src/util.rs
Vec3 randomInUnitSphere(Ctx &c) { // util.rs:6-15
    Vec3 p = v3(1.0, 1.0, 1.0);
    while (dot(p, p) >= 1.0) {
        double a = c.next53();
        double b = c.next53();
        double d = c.next53();
        p = v3(a, b, d) * 2.0 - v3(1.0, 1.0, 1.0);
    }
    return p;
}
Vec3 randomInUnitDisk(Ctx &c) { // util.rs:27-42
    for (;;) {
        double a = c.range11();
        double b = c.range11();
        Vec3 p = v3(a, b, 0.0);
        if (length(p) >= 1.0) continue;
        return p;
    }
}

// ------------------------------------------------- src/material.rs : textures
struct Scene;
struct Texture { // material.rs:196-198
    virtual ~Texture() {}
    virtual Vec3 value(const Scene &s, double u, double v, Vec3 p) const = 0;
};
struct SolidColor : Texture { // material.rs:200-215
    Vec3 color;
    explicit SolidColor(Vec3 c) : color(c) {}
    Vec3 value(const Scene &, double, double, Vec3) const override { return color; }
};
struct CheckerTexture : Texture { // material.rs:217-245
    int black, white;
    CheckerTexture(int b, int w) : black(b), white(w) {}
    Vec3 value(const Scene &s, double u, double v, Vec3 p) const override;
};
// ImageTexture<F> holds an arbitrary closure (material.rs:247-265); the only
// closure in the reference is the nearest-texel lookup of examples/main.rs:267-280.
struct ImageTexture : Texture {
    std::vector<uint8_t> rgb;
    int w, h;
    ImageTexture(const uint8_t *p, int w_, int h_) : rgb(p, p + (size_t)w_ * h_ * 3), w(w_), h(h_) {}
    static uint32_t as_u32(double x) { // Rust `as u32`: saturating, NaN -> 0
        if (!(x > 0.0)) return 0u;
        if (x >= 4294967295.0) return 4294967295u;
        return (uint32_t)x;
    }
    Vec3 value(const Scene &, double u, double v, Vec3) const override {
        uint32_t px = as_u32(u * (double)w);
        uint32_t py = as_u32((1.0 - v) * (double)h);
        // get_pixel() panics in the reference when px == w (u == 1.0 on the seam);
        // clamped here -- deliberate, pixel-neutral deviation.
        if (px >= (uint32_t)w) px = (uint32_t)w - 1;
        if (py >= (uint32_t)h) py = (uint32_t)h - 1;
        const uint8_t *t = &rgb[((size_t)py * w + px) * 3];
        return v3((double)t[0] / 255.0, (double)t[1] / 255.0, (double)t[2] / 255.0);
    }
};

// ------------------------------------------------ src/material.rs : materials
struct Material { // material.rs:13-21
    virtual ~Material() {}
    virtual bool scatter(const Scene &s, Ctx &c, const Ray &rayIn, const HitRecord &rec, Ray *scattered,
                         Vec3 *attenuation) const = 0;
    virtual Vec3 emitted(const Scene &, double, double, Vec3) const { return v3(0.0, 0.0, 0.0); }
};

// -------------------------------------------------------------- Hit / Bound
struct AABB { // optimize.rs:21-83
    Vec3 min, max;
    AABB merged(const AABB &o) const { // optimize.rs:44-57 (f64::min / f64::max ignore a NaN operand)
        return AABB{v3(std::fmin(min.x, o.min.x), std::fmin(min.y, o.min.y), std::fmin(min.z, o.min.z)),
                    v3(std::fmax(max.x, o.max.x), std::fmax(max.y, o.max.y), std::fmax(max.z, o.max.z))};
    }
    bool hit(const Ray &ray) const { // optimize.rs:61-82
        double tmin = 0.0;
        double tmax = kInf;
        for (int i = 0; i < 3; ++i) {
            double inv = 1.0 / ray.direction[i];
            double t0 = (min[i] - ray.origin[i]) * inv;
            double t1 = (max[i] - ray.origin[i]) * inv;
            if (inv < 0.0) std::swap(t0, t1);
            tmin = t0 > tmin ? t0 : tmin;
            tmax = t1 < tmax ? t1 : tmax;
            if (tmax <= tmin) return false;
        }
        return true;
    }
};

// `trait Hit` (ray.rs:85-87) + `trait Bound<AABB>` (optimize.rs:88-95) in one base.
struct Object {
    virtual ~Object() {}
    virtual bool hit(const Scene &s, Ctx &c, const Ray &ray, HitRecord *rec) const = 0;
    virtual bool bound(const Scene &s, AABB *out) const = 0;
};

struct Scene {
    std::vector<std::unique_ptr<Texture>> textures;
    std::vector<std::unique_ptr<Material>> materials;
    std::vector<std::unique_ptr<Object>> geometries;
    std::vector<std::unique_ptr<Object>> objects;
    std::unique_ptr<Object> world;
    int world_nodes = 0;
    int medium_slots = 0;
    uint64_t n_sprites = 0;
#ifdef ORC_WITH_HYPOTHESES // the probe library only (oracle/_build/librt_oracle_hyp.so); the parity anchor is compiled without
    double hyp_param = 0.0;
    unsigned hypothesis = 0; // ORC_HYP_*: earlier forms of the reference's code the cover.png probes try
#endif
    // camera (camera.rs:10-21)
    Vec3 eye{0, 0, 0}, lowerLeft{0, 0, 0}, horizontal{0, 0, 0}, vertical{0, 0, 0};
    double lensRadius = 0.0;
};

Vec3 CheckerTexture::value(const Scene &s, double u, double v, Vec3 p) const { // material.rs:235-245
    double sine = std::sin(2.0 * kPi * 10.0 * u) * std::sin(2.0 * kPi * 10.0 * v);
    if (sine > 0.0) return s.textures[black]->value(s, u, v, p);
    return s.textures[white]->value(s, u, v, p);
}

struct Lambertian : Material { // material.rs:24-70
    int albedo;
    explicit Lambertian(int t) : albedo(t) {}
    bool scatter(const Scene &s, Ctx &c, const Ray &, const HitRecord &rec, Ray *sc, Vec3 *att) const override {
        *sc = Ray{rec.intersection, normalized(rec.normal + randomInUnitSphere(c))};
        *att = s.textures[albedo]->value(s, rec.u, rec.v, rec.intersection);
        return true;
    }
};
struct Metal : Material { // material.rs:73-119
    int albedo;
    double fuzziness;
    Metal(int t, double f) : albedo(t), fuzziness(f) {}
    bool scatter(const Scene &s, Ctx &c, const Ray &rayIn, const HitRecord &rec, Ray *sc, Vec3 *att) const override {
        if (dot(rayIn.direction, rec.normal) < 0.0) {
            Vec3 refl = reflected(normalized(rayIn.direction), rec.normal);
            Vec3 dir = fuzziness == 0.0 ? refl : normalized(refl + fuzziness * randomInUnitSphere(c));
            *sc = Ray{rec.intersection, dir};
            *att = s.textures[albedo]->value(s, rec.u, rec.v, rec.intersection);
            return true;
        }
        return false;
    }
};
inline double schlickReflectionProbability(double theta, double n1, double n2) { // material.rs:140-143
    double r0 = std::pow((n1 - n2) / (n1 + n2), 2.0);
    return r0 + (1.0 - r0) * std::pow(1.0 - std::cos(theta), 5.0);
}
struct Dielectric : Material { // material.rs:122-193
    double refractive;
    explicit Dielectric(double r) : refractive(r) {}
    bool scatter(const Scene &s, Ctx &c, const Ray &rayIn, const HitRecord &rec, Ray *sc, Vec3 *att) const override {
        *att = v3(1.0, 1.0, 1.0);
        double ratio;
        Vec3 normal = rec.normal;
        if (dot(rayIn.direction, rec.normal) < 0.0) {
            ratio = 1.0 / refractive;
        } else {
            ratio = refractive;
            normal = -normal;
        }
        Vec3 refr;
        if (refracted(rayIn.direction, normal, ratio, &refr)) {
            double theta = std::acos(-dot(rayIn.direction, normal));
            double u = c.range01(); // drawn before the probability is evaluated (operand order)
            double pr = schlickReflectionProbability(theta, ratio, 1.0);
#ifdef ORC_WITH_HYPOTHESES
            if (s.hypothesis) { // cover.png probes only
                const bool inside = ratio > 1.0;
                if ((s.hypothesis & ORC_HYP_NO_FRESNEL) || ((s.hypothesis & ORC_HYP_NO_INSIDE_FRESNEL) && inside)) pr = 0.0;
                if ((s.hypothesis & ORC_HYP_SCHLICK_OUTSIDE_ANGLE) && inside) {
                    Vec3 rn = normalized(refr);
                    pr = schlickReflectionProbability(std::acos(dot(rn, rec.normal)), ratio, 1.0);
                }
            }
#endif
            if (u < pr) {
                *sc = Ray{rec.intersection, reflected(rayIn.direction, rec.normal)};
            } else {
                *sc = Ray{rec.intersection, normalized(refr)};
            }
        } else {
            *sc = Ray{rec.intersection, reflected(rayIn.direction, normal)};
        }
        return true;
    }
};
struct DiffuseLight : Material { // material.rs:274-298
    int emission;
    explicit DiffuseLight(int t) : emission(t) {}
    bool scatter(const Scene &, Ctx &, const Ray &, const HitRecord &, Ray *, Vec3 *) const override { return false; }
    Vec3 emitted(const Scene &s, double u, double v, Vec3 p) const override { return s.textures[emission]->value(s, u, v, p); }
};
struct Isotropic : Material { // material.rs:302-326
    int albedo;
    explicit Isotropic(int t) : albedo(t) {}
    bool scatter(const Scene &s, Ctx &c, const Ray &rayIn, const HitRecord &rec, Ray *sc, Vec3 *att) const override {
        Vec3 p = randomInUnitSphere(c);
#ifdef ORC_WITH_HYPOTHESES
        if (s.hypothesis & ORC_HYP_ISOTROPIC_FORWARD) p = p + rayIn.direction * s.hyp_param; // probe: a forward-biased lobe
        if (s.hypothesis & ORC_HYP_ISOTROPIC_UNNORMALIZED) {                                 // probe: the book's listing
            *sc = Ray{rec.intersection, p};
            *att = s.textures[albedo]->value(s, rec.u, rec.v, rec.intersection);
            return true;
        }
#endif
        *sc = Ray{rec.intersection, normalized(p)};
        *att = s.textures[albedo]->value(s, rec.u, rec.v, rec.intersection);
        return true;
    }
};

// ------------------------------------------------------------ src/geometry.rs
struct Sphere : Object { // geometry.rs:12-74
    double radius;
    explicit Sphere(double r) : radius(r) {}
    static void unitSphereUv(Vec3 p, double *u, double *v) { // geometry.rs:35-39
        *u = 0.5 + std::atan2(p.x, p.z) / (2.0 * kPi);
        *v = 1.0 - std::acos(p.y) / kPi;
    }
    bool hit(const Scene &, Ctx &c, const Ray &ray, HitRecord *rec) const override { // geometry.rs:43-73
        ++c.cnt.prim_tests;
        Vec3 center = v3(0.0, 0.0, 0.0);
        Vec3 oc = ray.origin - center;
        double a = dot(ray.direction, ray.direction);
        double b = dot(oc, ray.direction) * 2.0;
        double cc = dot(oc, oc) - radius * radius;
        double discriminant = b * b - 4.0 * a * cc;
        if (discriminant < 0.0) return false;
        double t1 = (-b - std::sqrt(discriminant)) / (2.0 * a);
        double t2 = (-b + std::sqrt(discriminant)) / (2.0 * a);
        if (!(t1 < t2)) std::swap(t1, t2);
        const double eps = 1e-6; // (10.0 as f64).powf(-6.0); literal per SURVEY.md Appendix B
        double t;
        if (t1 > eps)
            t = t1;
        else if (t2 > eps)
            t = t2;
        else
            return false;
        Vec3 intersection = ray.at(t);
        Vec3 normal = normalized((intersection - center) / radius);
        double u, v;
        unitSphereUv((intersection - center) / radius, &u, &v);
        rec->t = t;
        rec->intersection = intersection;
        rec->normal = normal;
        rec->material = -1;
        rec->u = u;
        rec->v = v;
        return true;
    }
    bool bound(const Scene &, AABB *out) const override { // optimize.rs:105-114
        Vec3 center = v3(0.0, 0.0, 0.0);
        *out = AABB{center - v3(radius, radius, radius), center + v3(radius, radius, radius)};
        return true;
    }
};

struct Rectangle : Object { // geometry.rs:127-181
    double width, height;
    Rectangle(double w, double h) : width(w), height(h) {}
    bool hit(const Scene &, Ctx &c, const Ray &ray, HitRecord *rec) const override { // geometry.rs:153-180
        ++c.cnt.prim_tests;
        double z = 0.0;
        double a0 = -width / 2.0, a1 = -height / 2.0, b0 = width / 2.0, b1 = height / 2.0;
        double t = (z - ray.origin.z) / ray.direction.z;
        if (std::isinf(t) || std::isnan(t) || t < 1e-6) return false;
        double x = ray.origin.x + ray.direction.x * t;
        double y = ray.origin.y + ray.direction.y * t;
        if (x < a0 || x > b0 || y < a1 || y > b1) return false;
        rec->u = (x - a0) / (b0 - a0);
        rec->v = (y - a1) / (b1 - a1);
        rec->t = t;
        rec->intersection = ray.at(t);
        rec->normal = v3(0.0, 0.0, 1.0);
        rec->material = -1;
        return true;
    }
    bool bound(const Scene &, AABB *out) const override { // optimize.rs:116-126
        double z = 0.0;
        *out = AABB{v3(-width / 2.0, -height / 2.0, z - 1e-6), v3(width / 2.0, height / 2.0, z + 1e-6)};
        return true;
    }
};

// AABB of the 8 transformed corners: optimize.rs:149-177 (Sprite) == :205-233 (TransformedGeometry)
bool transformed_bound(const AABB &b, const Mat4 &m, AABB *out) {
    double x0 = b.min[0], y0 = b.min[1], z0 = b.min[2], x1 = b.max[0], y1 = b.max[1], z1 = b.max[2];
    double mn[3] = {kInf, kInf, kInf}, mx[3] = {-kInf, -kInf, -kInf};
    Vec3 pts[8] = {v3(x0, y0, z0), v3(x1, y0, z0), v3(x0, y1, z0), v3(x0, y0, z1),
                   v3(x1, y1, z0), v3(x1, y0, z1), v3(x0, y1, z1), v3(x1, y1, z1)};
    for (const Vec3 &p : pts) {
        Vec3 q = xyz(transformed(xyz1(p), m));
        for (int i = 0; i < 3; ++i) {
            if (q[i] < mn[i]) mn[i] = q[i];
            if (q[i] > mx[i]) mx[i] = q[i];
        }
    }
    *out = AABB{v3(mn[0], mn[1], mn[2]), v3(mx[0], mx[1], mx[2])};
    return true;
}

// inverse-transform the ray, hit locally, forward-transform point and normal
// (geometry.rs:214-245 and sprite.rs:94-138 are the same code)
template <class LocalHit>
bool transformed_hit(const Mat4Cached &transform, const Ray &ray, HitRecord *rec, LocalHit local) {
    const Mat4 *inversed = transform.inv();
    if (!inversed) return false; // det == 0
    Vec3 origin = xyz(transformed(xyz1(ray.origin), *inversed));
    Vec3 direction = xyz(transformed(xyz0(ray.direction), *inversed)); // NOT renormalised (Q5)
    Ray lray{origin, direction};
    HitRecord lrec;
    if (!local(lray, &lrec)) return false;
    rec->t = lrec.t;                                                           // t shared between spaces
    rec->intersection = xyz(transformed(xyz1(lrec.intersection), transform.origin));
    rec->normal = xyz(transformed(xyz0(lrec.normal), transform.origin));       // M, not M^-T (Q5)
    rec->u = lrec.u;
    rec->v = lrec.v;
    rec->material = -1;
    return true;
}

struct TransformedGeometry : Object { // geometry.rs:185-246
    std::unique_ptr<Object> geometry;
    Mat4Cached transform;
    TransformedGeometry(std::unique_ptr<Object> g, const Mat4 &m) : geometry(std::move(g)), transform(m) {}
    bool hit(const Scene &s, Ctx &c, const Ray &ray, HitRecord *rec) const override {
        return transformed_hit(transform, ray, rec, [&](const Ray &r, HitRecord *lr) { return geometry->hit(s, c, r, lr); });
    }
    bool bound(const Scene &s, AABB *out) const override { // optimize.rs:188-241
        AABB b;
        if (!geometry->bound(s, &b)) return false;
        return transformed_bound(b, transform.origin, out);
    }
};

// --------------------------------------------------------------- src/volume.rs
struct ConstantMedium : Object { // volume.rs:18-101
    int boundary; // Arc<T>
    double density;
    ConstantMedium(int b, double d) : boundary(b), density(d) {}
    bool hit(const Scene &s, Ctx &c, const Ray &ray, HitRecord *rec) const override { // volume.rs:46-100
        const Object &bnd = *s.geometries[boundary];
        // the boundary may hold media of its own (T: Hit is generic, volume.rs:18-44): they key their draws from this evaluation
        const uint32_t key = c.medium_key();
        struct Scope {
            Ctx &c;
            int depth;
            uint32_t key, pass;
            Scope(Ctx &cc, uint32_t k) : c(cc), depth(cc.medium_depth), key(cc.outer_key), pass(cc.outer_pass) {
                ++c.medium_depth;
                c.outer_key = k;
                c.outer_pass = 0;
            }
            ~Scope() {
                c.medium_depth = depth;
                c.outer_key = key;
                c.outer_pass = pass;
            }
        };
        HitRecord record1;
        {
            Scope scope(c, key);
            if (!bnd.hit(s, c, ray, &record1)) return false;
        }
        if (dot(record1.normal, ray.direction) < 0.0) {
            // entering: restart just inside
            Ray ray2{record1.intersection + ray.direction * 1e-6, ray.direction};
            HitRecord record2;
            {
                Scope scope(c, key);
                c.outer_pass = 1;
                if (!bnd.hit(s, c, ray2, &record2)) return false;
            }
            double distanceInsideGeometry = record2.t;
            double distance = (-1.0 / density) * std::log(c.keyed01(key));
            if (distance > distanceInsideGeometry) return false;
            rec->u = record1.u + record2.u;
            rec->v = record1.v + record2.v;
            rec->t = record1.t + distance;
            rec->intersection = ray2.at(record1.t + distance); // Q9: on the RESTARTED ray
            rec->normal = (record1.normal + record2.normal) / 2.0;
            rec->material = -1;
            return true;
        }
        // origin inside the boundary
#ifdef ORC_WITH_HYPOTHESES
        if (s.hypothesis & ORC_HYP_INSIDE_NONE) return false; // the book's listing as the reference's comment reads it (volume.rs:44-45)
#endif
        double distanceInsideGeometry = record1.t;
        double distance = (-1.0 / density) * std::log(c.keyed01(key));
        if (distance > distanceInsideGeometry) return false;
        rec->u = record1.u;
        rec->v = record1.v;
        rec->t = distance; // volume.rs:90 "written wrong originally"
        rec->intersection = ray.at(distance);
#ifdef ORC_WITH_HYPOTHESES
        if (s.hypothesis & ORC_HYP_INSIDE_T_ADDS_T1) rec->t = record1.t + distance;
        if (s.hypothesis & ORC_HYP_INSIDE_POINT_ADDS_T1) rec->intersection = ray.at(record1.t + distance);
#endif
        rec->normal = record1.normal;
        rec->material = -1;
        return true;
    }
    bool bound(const Scene &s, AABB *out) const override { return s.geometries[boundary]->bound(s, out); } // optimize.rs:507-516
};

// --------------------------------------------------------------- src/sprite.rs
struct GeometryRef : Object { // Arc<T> to a shared geometry
    int geometry;
    explicit GeometryRef(int g) : geometry(g) {}
    bool hit(const Scene &s, Ctx &c, const Ray &r, HitRecord *rec) const override { return s.geometries[geometry]->hit(s, c, r, rec); }
    bool bound(const Scene &s, AABB *out) const override { return s.geometries[geometry]->bound(s, out); }
};

struct Sprite : Object { // sprite.rs:11-139
    int geometry, material; // Option<Arc<T>>, Option<Arc<U>>
    Mat4Cached transform;
    uint32_t medium_slot = 0x3FFu;
    uint64_t sid = 0;      // creation index among the sprites
    uint64_t place = 0;    // position among the world's own sprites, or in the list of the node that owns it (path keys, rt_rng.h)
    bool owned = false;    // moved into a node used as a geometry
    Sprite(int g, int m, const Mat4 &t) : geometry(g), material(m), transform(t) {}
    bool hit(const Scene &s, Ctx &c, const Ray &ray, HitRecord *rec) const override { // sprite.rs:94-138
        if (geometry < 0) return false;
        const uint32_t saved = c.medium_slot;
        const uint64_t saved_hash = c.path_hash;
        c.medium_slot = medium_slot;
        c.path_hash = c.path_len == 0 ? place + 1ull : c.path_hash * RT_RNG_PATH_MUL + place + 1ull;
        ++c.path_len;
        bool ok = transformed_hit(transform, ray, rec,
                                  [&](const Ray &r, HitRecord *lr) { return s.geometries[geometry]->hit(s, c, r, lr); });
        --c.path_len;
        c.path_hash = saved_hash;
        c.medium_slot = saved;
        if (ok) rec->material = material;
        return ok;
    }
    bool bound(const Scene &s, AABB *out) const override { // optimize.rs:128-185
        if (geometry < 0) return false;
        AABB b;
        if (!s.geometries[geometry]->bound(s, &b)) return false;
        return transformed_bound(b, transform.origin, out);
    }
};

// ------------------------------------------------------------- src/optimize.rs
// a child of a BVH node / element of a Vec: either owned or a reference to scene.objects[i]
struct Child {
    const Object *obj = nullptr;
    std::unique_ptr<Object> owned;
};

struct BVHNode : Object { // optimize.rs:339-498
    AABB volume;
    const Object *left = nullptr, *right = nullptr;
    std::unique_ptr<Object> left_owned, right_owned;
    int nodes = 1; // this node + nested nodes created by new()

    // BoundingVolumeHierarchyNode::new (optimize.rs:366-440).  `objs` is moved in.
    static std::unique_ptr<BVHNode> make(const Scene &s, std::vector<const Object *> objs, rt_rng *axis_rng) {
        if (objs.empty()) return nullptr;
        uint32_t axis = rt_rng_gen_below(axis_rng, 3); // generator.gen_range(0, 3)
        auto key = [&](const Object *o) {
            AABB b;
            o->bound(s, &b); // .unwrap()
            return b.min[(int)axis];
        };
        auto compare = [&](const Object *v, const Object *w) { return key(v) < key(w); }; // optimize.rs:442-461
        std::unique_ptr<BVHNode> node(new BVHNode);
        if (objs.size() == 1) {
            node->left = objs[0];
        } else if (objs.size() == 2) {
            if (compare(objs[0], objs[1])) {
                node->left = objs[0];
                node->right = objs[1];
            } else {
                node->left = objs[1];
                node->right = objs[0];
            }
        } else {
            // sort_by with Less/Greater only (never Equal); key order is what matters
            std::stable_sort(objs.begin(), objs.end(), compare);
            size_t middle = objs.size() / 2;
            std::vector<const Object *> rightHalf(objs.begin() + (long)middle, objs.end()); // split_off(middle)
            objs.resize(middle);
            auto r = make(s, std::move(rightHalf), axis_rng); // right first (optimize.rs:404-408)
            auto l = make(s, std::move(objs), axis_rng);
            if (r) {
                node->nodes += r->nodes;
                node->right = r.get();
                node->right_owned = std::move(r);
            }
            if (l) {
                node->nodes += l->nodes;
                node->left = l.get();
                node->left_owned = std::move(l);
            }
        }
        AABB lv, rv;
        bool hl = node->left && node->left->bound(s, &lv);
        bool hr = node->right && node->right->bound(s, &rv);
        if (hl && hr)
            node->volume = lv.merged(rv);
        else if (hl)
            node->volume = lv;
        else if (hr)
            node->volume = rv;
        else
            return nullptr;
        return node;
    }

    bool hit(const Scene &s, Ctx &c, const Ray &ray, HitRecord *rec) const override { // optimize.rs:469-498
        ++c.cnt.aabb_tests;
        if (!volume.hit(ray)) return false;
        HitRecord record; // t = inf
        if (left) {
            HitRecord lr;
            if (left->hit(s, c, ray, &lr)) {
                if (lr.t < record.t) record = lr;
            }
        }
        if (right) {
            HitRecord rr;
            if (right->hit(s, c, ray, &rr)) {
                if (rr.t < record.t) record = rr;
            }
        }
        if (std::isinf(record.t)) return false;
        *rec = record;
        return true;
    }
    bool bound(const Scene &, AABB *out) const override { // optimize.rs:502-506
        *out = volume;
        return true;
    }
};

struct ObjectList : Object { // impl Hit for Vec<Arc<dyn Hit>> (geometry.rs:98-116, optimize.rs:316-334)
    std::vector<const Object *> items;
    bool hit(const Scene &s, Ctx &c, const Ray &ray, HitRecord *rec) const override {
        bool have = false;
        for (const Object *o : items) {
            HitRecord r;
            if (o->hit(s, c, ray, &r)) {
                if (!have) {
                    *rec = r;
                    have = true;
                } else if (r.t < rec->t) {
                    *rec = r;
                }
            }
        }
        return have;
    }
    bool bound(const Scene &s, AABB *out) const override { // optimize.rs:275-292
        bool have = false;
        for (const Object *o : items) {
            AABB b;
            if (o->bound(s, &b)) {
                *out = have ? out->merged(b) : b;
                have = true;
            }
        }
        return have;
    }
};

// Cube::new (geometry.rs:251-287) wrapped in a BVH as the examples do
std::unique_ptr<Object> make_cube_bvh(const Scene &s, double width, double height, double depth, uint64_t seed) {
    auto rad = [](double deg) { return deg * (kPi / 180.0); }; // f64::to_radians
    auto tg = [](double w, double h, const Mat4 &m) {
        return std::unique_ptr<Object>(new TransformedGeometry(std::unique_ptr<Object>(new Rectangle(w, h)), m));
    };
    std::vector<std::unique_ptr<Object>> faces;
    faces.push_back(tg(width, height, mat4_translation(v3(0.0, 0.0, depth / 2.0)))); // front
    faces.push_back(tg(depth, height,
                       mat4_multiplied(mat4_translation(v3(-width / 2.0, 0.0, 0.0)), mat4_rotation(rad(-90.0), v3(0, 1, 0))))); // left
    faces.push_back(tg(width, height,
                       mat4_multiplied(mat4_translation(v3(0.0, 0.0, -depth / 2.0)), mat4_rotation(rad(180.0), v3(0, 1, 0))))); // back
    faces.push_back(tg(depth, height,
                       mat4_multiplied(mat4_translation(v3(width / 2.0, 0.0, 0.0)), mat4_rotation(rad(90.0), v3(0, 1, 0))))); // right
    faces.push_back(tg(width, depth,
                       mat4_multiplied(mat4_translation(v3(0.0, height / 2.0, 0.0)), mat4_rotation(rad(-90.0), v3(1, 0, 0))))); // top
    faces.push_back(tg(width, depth,
                       mat4_multiplied(mat4_translation(v3(0.0, -height / 2.0, 0.0)), mat4_rotation(rad(90.0), v3(1, 0, 0))))); // bottom
    struct CubeBvh : Object {
        std::vector<std::unique_ptr<Object>> faces;
        std::unique_ptr<BVHNode> root;
        bool hit(const Scene &s, Ctx &c, const Ray &r, HitRecord *rec) const override { return root->hit(s, c, r, rec); }
        bool bound(const Scene &s, AABB *out) const override { return root->bound(s, out); }
    };
    std::unique_ptr<CubeBvh> cb(new CubeBvh);
    cb->faces = std::move(faces);
    std::vector<const Object *> ptrs;
    for (auto &f : cb->faces) ptrs.push_back(f.get());
    rt_rng g;
    rt_rng_init(&g, seed, RT_RNG_SCENE_STREAM);
    cb->root = BVHNode::make(s, std::move(ptrs), &g);
    return cb;
}

// --------------------------------------------------------------- src/camera.rs
void camera_new(Scene &s, Vec3 eye, Vec3 center, Vec3 up, double fov, double aspect, double focusDistance,
                double lensRadius) { // camera.rs:25-59
    up = normalized(up);
    double height = std::tan(fov / 2.0) * 2.0;
    double width = aspect * height;
    Vec3 w = normalized(eye - center);
    Vec3 u = cross(up, w); // NOT normalised (Q1)
    Vec3 v = cross(w, u);
    Vec3 horizontal = u * width * focusDistance;
    Vec3 vertical = v * height * focusDistance;
    Vec3 lowerLeft = eye - horizontal / 2.0 - vertical / 2.0 - w * focusDistance;
    s.eye = eye;
    s.lensRadius = lensRadius;
    s.lowerLeft = lowerLeft;
    s.horizontal = horizontal;
    s.vertical = vertical;
}
Ray camera_ray(const Scene &s, Ctx &c, double u, double v) { // camera.rs:91-106
    if (s.lensRadius == 0.0) {
        return Ray{s.eye, normalized(s.lowerLeft + s.horizontal * u + s.vertical * v - s.eye)};
    }
    Vec3 rd = s.lensRadius * randomInUnitDisk(c);
    double offset = rd.x * u + rd.y * v; // scalar, from the SCREEN coordinates (Q2)
    return Ray{s.eye + offset, normalized(s.lowerLeft + s.horizontal * u + s.vertical * v - s.eye - offset)};
}

// --------------------------------------------------------------- src/render.rs
Vec3 color(const Scene &s, Ctx &c, const Ray &ray, int maxDepth) { // render.rs:5-29 (recursive form)
    if (maxDepth == 0) return v3(0.0, 0.0, 0.0);
    ++c.cnt.segments;
    HitRecord record;
    if (s.world->hit(s, c, ray, &record)) {
        if (record.material >= 0) {
            const Material &m = *s.materials[record.material];
            Ray scattered;
            Vec3 attenuation;
            if (m.scatter(s, c, ray, record, &scattered, &attenuation)) {
                ++c.segment;
                return attenuation * color(s, c, scattered, maxDepth - 1) + m.emitted(s, record.u, record.v, record.intersection);
            }
            return m.emitted(s, record.u, record.v, record.intersection);
        }
        return v3(0.0, 0.0, 0.0);
    }
    return v3(0.0, 0.0, 0.0);
}
// Iterative equivalent L = sum_k (prod_{j<k} att_j) * e_k; differs from
// the nested form only in rounding (a few ulp).  This is the evaluation order the HIP kernel uses.
Vec3 color_iterative(const Scene &s, Ctx &c, Ray ray, int maxDepth) {
    Vec3 L = v3(0.0, 0.0, 0.0);
    Vec3 T = v3(1.0, 1.0, 1.0);
    for (int k = 0; k < maxDepth; ++k) {
        ++c.cnt.segments;
        HitRecord record;
        if (!s.world->hit(s, c, ray, &record)) break;
        if (record.material < 0) break;
        const Material &m = *s.materials[record.material];
        Ray scattered;
        Vec3 attenuation;
        bool sc = m.scatter(s, c, ray, record, &scattered, &attenuation);
        L = L + T * m.emitted(s, record.u, record.v, record.intersection);
        if (!sc) break;
        T = T * attenuation;
        ray = scattered;
        ++c.segment;
    }
    return L;
}

// one sample of one pixel: examples/book-one.rs:69-75
Vec3 sample_pixel(const Scene &s, int W, int H, int spp, int maxDepth, uint64_t seed, int x, int y, int si, unsigned flags,
                  orc_counters *cnt) {
    Ctx c;
    uint64_t pixel = (uint64_t)y * (uint64_t)W + (uint64_t)x;
    rt_rng_init(&c.rng, seed, pixel * (uint64_t)spp + (uint64_t)si);
    double u = ((double)x + c.range01()) / (double)W;
    double v = ((double)y + c.range01()) / (double)H;
    Ray ray = camera_ray(s, c, u, v);
    Vec3 col = (flags & ORC_FLAG_ITERATIVE) ? color_iterative(s, c, ray, maxDepth) : color(s, c, ray, maxDepth);
    if (cnt) {
        cnt->samples += 1;
        cnt->segments += c.cnt.segments;
        cnt->aabb_tests += c.cnt.aabb_tests;
        cnt->prim_tests += c.cnt.prim_tests;
        cnt->rng_draws += c.cnt.rng_draws;
    }
    return col;
}

uint8_t tonemap_channel(double c) { // examples/book-one.rs:95-97
    double v = std::fmin(std::sqrt(c) * 255.0, 255.0); // f64::min: NaN -> 255
    if (!(v > 0.0)) return 0;                          // `as usize` saturates; -0.0 -> 0
    return (uint8_t)v;                                 // truncation
}

} // namespace

struct orc_scene {
    Scene s;
};

extern "C" {

orc_scene *orc_scene_new(void) { return new orc_scene; }
void orc_scene_free(orc_scene *p) { delete p; }

#ifdef ORC_WITH_HYPOTHESES
void orc_set_hypothesis(orc_scene *p, unsigned flags) { p->s.hypothesis = flags; }
void orc_set_hypothesis_param(orc_scene *p, double v) { p->s.hyp_param = v; }
#endif
int orc_tex_solid(orc_scene *p, double r, double g, double b) {
    p->s.textures.emplace_back(new SolidColor(v3(r, g, b)));
    return (int)p->s.textures.size() - 1;
}
int orc_tex_checker(orc_scene *p, int black, int white) {
    p->s.textures.emplace_back(new CheckerTexture(black, white));
    return (int)p->s.textures.size() - 1;
}
int orc_tex_image_rgb8(orc_scene *p, const uint8_t *rgb, int w, int h) {
    p->s.textures.emplace_back(new ImageTexture(rgb, w, h));
    return (int)p->s.textures.size() - 1;
}
int orc_mat_lambertian(orc_scene *p, int t) {
    p->s.materials.emplace_back(new Lambertian(t));
    return (int)p->s.materials.size() - 1;
}
int orc_mat_metal(orc_scene *p, int t, double f) {
    p->s.materials.emplace_back(new Metal(t, f));
    return (int)p->s.materials.size() - 1;
}
int orc_mat_dielectric(orc_scene *p, double r) {
    p->s.materials.emplace_back(new Dielectric(r));
    return (int)p->s.materials.size() - 1;
}
int orc_mat_diffuse_light(orc_scene *p, int t) {
    p->s.materials.emplace_back(new DiffuseLight(t));
    return (int)p->s.materials.size() - 1;
}
int orc_mat_isotropic(orc_scene *p, int t) {
    p->s.materials.emplace_back(new Isotropic(t));
    return (int)p->s.materials.size() - 1;
}
int orc_geom_sphere(orc_scene *p, double r) {
    p->s.geometries.emplace_back(new Sphere(r));
    return (int)p->s.geometries.size() - 1;
}
int orc_geom_rectangle(orc_scene *p, double w, double h) {
    p->s.geometries.emplace_back(new Rectangle(w, h));
    return (int)p->s.geometries.size() - 1;
}
int orc_geom_cube_bvh(orc_scene *p, double w, double h, double d, uint64_t seed) {
    p->s.geometries.push_back(make_cube_bvh(p->s, w, h, d, seed));
    return (int)p->s.geometries.size() - 1;
}
int orc_geom_constant_medium(orc_scene *p, int boundary, double density) {
    p->s.geometries.emplace_back(new ConstantMedium(boundary, density));
    return (int)p->s.geometries.size() - 1;
}
int orc_sprite(orc_scene *p, int geometry, int material, const double *M) {
    Mat4 m = mat4_identity();
    if (M) std::memcpy(m.a, M, sizeof m.a);
    Sprite *sp = new Sprite(geometry, material, m);
    sp->sid = p->s.n_sprites++;
    // (medium_slot is assigned when the world is built: rank_world_sprites)
    p->s.objects.emplace_back(sp);
    return (int)p->s.objects.size() - 1;
}
static std::vector<const Object *> gather(orc_scene *p, const int *objects, int n) {
    std::vector<const Object *> v;
    for (int i = 0; i < n; ++i) v.push_back(p->s.objects[objects[i]].get());
    return v;
}
// positions of the world's own sprites among themselves, in creation order (the product's world list)
// ... and the creation-order slots of the medium sprites among them (keys of their draws, include/rt_rng.h)
static void rank_world_sprites(orc_scene *p) {
    uint64_t r = 0;
    int slots = 0;
    for (auto &o : p->s.objects)
        if (Sprite *sp = dynamic_cast<Sprite *>(o.get())) {
            sp->medium_slot = 0x3FFu;
            if (sp->owned) continue;
            sp->place = r++;
            if (sp->geometry >= 0 && dynamic_cast<ConstantMedium *>(p->s.geometries[sp->geometry].get())) {
                const int slot = slots++;
                if (slot < (int)RT_MEDIUM_SLOT_MAX) sp->medium_slot = (uint32_t)slot; // beyond: a path key (the product refuses such scenes)
            }
        }
}
int orc_geom_bvh(orc_scene *p, const int *objects, int n, uint64_t seed) {
    for (int i = 0; i < n; ++i)
        if (Sprite *sp = dynamic_cast<Sprite *>(p->s.objects[objects[i]].get())) {
            sp->owned = true;
            sp->place = (uint64_t)i;
        }
    rt_rng g;
    rt_rng_init(&g, seed, RT_RNG_SCENE_STREAM);
    auto node = BVHNode::make(p->s, gather(p, objects, n), &g);
    if (!node) return -1;
    p->s.geometries.push_back(std::move(node));
    return (int)p->s.geometries.size() - 1;
}
int orc_geom_transformed(orc_scene *p, int geometry, const double *M) {
    Mat4 m = mat4_identity();
    if (M) std::memcpy(m.a, M, sizeof m.a);
    p->s.geometries.emplace_back(new TransformedGeometry(std::unique_ptr<Object>(new GeometryRef(geometry)), m));
    return (int)p->s.geometries.size() - 1;
}
int orc_object_bvh(orc_scene *p, const int *objects, int n, uint64_t seed) {
    rt_rng g;
    rt_rng_init(&g, seed, RT_RNG_SCENE_STREAM);
    auto node = BVHNode::make(p->s, gather(p, objects, n), &g);
    if (!node) return -1;
    p->s.objects.push_back(std::move(node));
    return (int)p->s.objects.size() - 1;
}
int orc_world_bvh(orc_scene *p, const int *objects, int n, uint64_t seed) {
    rank_world_sprites(p);
    rt_rng g;
    rt_rng_init(&g, seed, RT_RNG_SCENE_STREAM);
    auto node = BVHNode::make(p->s, gather(p, objects, n), &g);
    if (!node) return -1;
    p->s.world_nodes = node->nodes;
    p->s.world = std::move(node);
    return 0;
}
int orc_world_list(orc_scene *p, const int *objects, int n) {
    rank_world_sprites(p);
    ObjectList *l = new ObjectList;
    l->items = gather(p, objects, n);
    p->s.world.reset(l);
    p->s.world_nodes = 0;
    return 0;
}
void orc_camera_perspective(orc_scene *p, const double eye[3], const double center[3], const double up[3], double fov,
                            double aspect, double focus, double lens) {
    camera_new(p->s, v3(eye[0], eye[1], eye[2]), v3(center[0], center[1], center[2]), v3(up[0], up[1], up[2]), fov, aspect, focus,
               lens);
}

int orc_render(orc_scene *p, int W, int H, int spp, int max_depth, uint64_t seed, int x0, int y0, int x1, int y1, unsigned flags,
               int nthreads, double *out, orc_counters *counters) {
    if (!p->s.world) return -1;
    if (nthreads < 1) nthreads = 1;
    std::vector<orc_counters> per(nthreads, orc_counters{0, 0, 0, 0, 0});
    const Scene &s = p->s;
    auto work = [&](int i) { // examples/book-one.rs:56-81
        for (int y = H - 1; y >= 0; --y) {
            if (y % nthreads != i) continue;
            if (y < y0 || y >= y1) continue;
            for (int x = x0; x < x1; ++x) {
                Vec3 pixel = v3(0.0, 0.0, 0.0);
                for (int si = 0; si < spp; ++si) pixel = pixel + sample_pixel(s, W, H, spp, max_depth, seed, x, y, si, flags, &per[i]);
                pixel = pixel / (double)spp;
                double *o = out + ((size_t)y * W + x) * 3;
                o[0] = pixel.x;
                o[1] = pixel.y;
                o[2] = pixel.z;
            }
        }
    };
    if (nthreads == 1) {
        work(0);
    } else {
        std::vector<std::thread> th;
        for (int i = 0; i < nthreads; ++i) th.emplace_back(work, i);
        for (auto &t : th) t.join();
    }
    if (counters) {
        *counters = orc_counters{0, 0, 0, 0, 0};
        for (auto &c : per) {
            counters->samples += c.samples;
            counters->segments += c.segments;
            counters->aabb_tests += c.aabb_tests;
            counters->prim_tests += c.prim_tests;
            counters->rng_draws += c.rng_draws;
        }
    }
    return 0;
}

int orc_render_pixel_samples(orc_scene *p, int W, int H, int spp, int max_depth, uint64_t seed, int x, int y, unsigned flags,
                             double *out) {
    if (!p->s.world) return -1;
    for (int si = 0; si < spp; ++si) {
        Vec3 c = sample_pixel(p->s, W, H, spp, max_depth, seed, x, y, si, flags, nullptr);
        out[si * 3 + 0] = c.x;
        out[si * 3 + 1] = c.y;
        out[si * 3 + 2] = c.z;
    }
    return 0;
}

void orc_tonemap_rgb8(const double *rgb, int n_pixels, uint8_t *out) {
    for (size_t i = 0; i < (size_t)n_pixels * 3; ++i) out[i] = tonemap_channel(rgb[i]);
}

int orc_write_ppm_p3(const char *path, const double *rgb, int W, int H) { // examples/book-one.rs:28-30,90-100
    FILE *f = std::fopen(path, "w");
    if (!f) return -1;
    std::fprintf(f, "P3\n%d %d\n255\n", W, H);
    for (int y = H - 1; y >= 0; --y)
        for (int x = 0; x < W; ++x) {
            const double *px = rgb + ((size_t)y * W + x) * 3;
            std::fprintf(f, "%u %u %u\n", (unsigned)tonemap_channel(px[0]), (unsigned)tonemap_channel(px[1]),
                         (unsigned)tonemap_channel(px[2]));
        }
    std::fclose(f);
    return 0;
}

// ------------------------------------------------------------------ KAT probes
static void put_rec(const HitRecord &r, double out[9]) {
    out[0] = r.t;
    out[1] = r.intersection.x;
    out[2] = r.intersection.y;
    out[3] = r.intersection.z;
    out[4] = r.normal.x;
    out[5] = r.normal.y;
    out[6] = r.normal.z;
    out[7] = r.u;
    out[8] = r.v;
}
int orc_kat_sphere_hit(double radius, const double o[3], const double d[3], double out[9]) {
    Scene s;
    Ctx c;
    rt_rng_init(&c.rng, 0, 0);
    HitRecord r;
    Sphere sp(radius);
    if (!sp.hit(s, c, Ray{v3(o[0], o[1], o[2]), v3(d[0], d[1], d[2])}, &r)) return 0;
    put_rec(r, out);
    return 1;
}
int orc_kat_rectangle_hit(double w, double h, const double o[3], const double d[3], double out[9]) {
    Scene s;
    Ctx c;
    rt_rng_init(&c.rng, 0, 0);
    HitRecord r;
    Rectangle rc(w, h);
    if (!rc.hit(s, c, Ray{v3(o[0], o[1], o[2]), v3(d[0], d[1], d[2])}, &r)) return 0;
    put_rec(r, out);
    return 1;
}
int orc_kat_aabb_hit(const double mn[3], const double mx[3], const double o[3], const double d[3]) {
    AABB b{v3(mn[0], mn[1], mn[2]), v3(mx[0], mx[1], mx[2])};
    return b.hit(Ray{v3(o[0], o[1], o[2]), v3(d[0], d[1], d[2])}) ? 1 : 0;
}
void orc_kat_reflect(const double v[3], const double n[3], double out[3]) {
    Vec3 r = reflected(v3(v[0], v[1], v[2]), v3(n[0], n[1], n[2]));
    out[0] = r.x;
    out[1] = r.y;
    out[2] = r.z;
}
int orc_kat_refract(const double v[3], const double n[3], double ratio, double out[3]) {
    Vec3 r;
    if (!refracted(v3(v[0], v[1], v[2]), v3(n[0], n[1], n[2]), ratio, &r)) return 0;
    out[0] = r.x;
    out[1] = r.y;
    out[2] = r.z;
    return 1;
}
double orc_kat_schlick(double theta, double n1, double n2) { return schlickReflectionProbability(theta, n1, n2); }
void orc_kat_mat4_translation(const double t[3], double out[16]) {
    Mat4 m = mat4_translation(v3(t[0], t[1], t[2]));
    std::memcpy(out, m.a, sizeof m.a);
}
void orc_kat_mat4_rotation(double radians, const double axis[3], double out[16]) {
    Mat4 m = mat4_rotation(radians, v3(axis[0], axis[1], axis[2]));
    std::memcpy(out, m.a, sizeof m.a);
}
void orc_kat_mat4_multiplied(const double a[16], const double b[16], double out[16]) {
    Mat4 A, B;
    std::memcpy(A.a, a, sizeof A.a);
    std::memcpy(B.a, b, sizeof B.a);
    Mat4 m = mat4_multiplied(A, B);
    std::memcpy(out, m.a, sizeof m.a);
}
double orc_kat_mat4_determinant(const double a[16]) {
    Mat4 A;
    std::memcpy(A.a, a, sizeof A.a);
    return mat4_determinant(A);
}
int orc_kat_mat4_inversed(const double a[16], double out[16]) {
    Mat4 A, inv;
    std::memcpy(A.a, a, sizeof A.a);
    if (!mat4_inversed(A, &inv)) return 0;
    std::memcpy(out, inv.a, sizeof inv.a);
    return 1;
}
void orc_kat_vec4_transformed(const double v[4], const double m[16], double out[4]) {
    Mat4 M;
    std::memcpy(M.a, m, sizeof M.a);
    Vec4 r = transformed(Vec4{v[0], v[1], v[2], v[3]}, M);
    out[0] = r.x;
    out[1] = r.y;
    out[2] = r.z;
    out[3] = r.w;
}
void orc_kat_camera_frame(orc_scene *p, double out[9]) {
    const Scene &s = p->s;
    double v[9] = {s.lowerLeft.x, s.lowerLeft.y, s.lowerLeft.z, s.horizontal.x, s.horizontal.y,
                   s.horizontal.z, s.vertical.x,  s.vertical.y,  s.vertical.z};
    std::memcpy(out, v, sizeof v);
}
void orc_kat_camera_ray(orc_scene *p, double u, double v, uint64_t seed, uint64_t stream, double out[6]) {
    Ctx c;
    rt_rng_init(&c.rng, seed, stream);
    Ray r = camera_ray(p->s, c, u, v);
    double o[6] = {r.origin.x, r.origin.y, r.origin.z, r.direction.x, r.direction.y, r.direction.z};
    std::memcpy(out, o, sizeof o);
}
int orc_kat_world_hit(orc_scene *p, const double o[3], const double d[3], uint64_t seed, uint64_t stream, double out[10]) {
    if (!p->s.world) return 0;
    Ctx c;
    rt_rng_init(&c.rng, seed, stream);
    HitRecord r;
    if (!p->s.world->hit(p->s, c, Ray{v3(o[0], o[1], o[2]), v3(d[0], d[1], d[2])}, &r)) return 0;
    put_rec(r, out);
    out[9] = (double)r.material;
    return 1;
}
int orc_kat_world_node_count(orc_scene *p) { return p->s.world_nodes; }
void orc_kat_rng_u64(uint64_t seed, uint64_t stream, int n, uint64_t *out) {
    rt_rng g;
    rt_rng_init(&g, seed, stream);
    for (int i = 0; i < n; ++i) out[i] = rt_rng_next(&g);
}
void orc_kat_xoroshiro(uint64_t s0, uint64_t s1, int n, uint64_t *out) {
    for (int i = 0; i < n; ++i) out[i] = rt_xoroshiro_next(&s0, &s1);
}
void orc_kat_random_in_unit_sphere(uint64_t seed, uint64_t stream, double out[3]) {
    Ctx c;
    rt_rng_init(&c.rng, seed, stream);
    Vec3 p = randomInUnitSphere(c);
    out[0] = p.x;
    out[1] = p.y;
    out[2] = p.z;
}
void orc_kat_random_in_unit_disk(uint64_t seed, uint64_t stream, double out[3]) {
    Ctx c;
    rt_rng_init(&c.rng, seed, stream);
    Vec3 p = randomInUnitDisk(c);
    out[0] = p.x;
    out[1] = p.y;
    out[2] = p.z;
}
void orc_kat_texture_value(orc_scene *p, int tex, double u, double v, double out[3]) {
    Vec3 c = p->s.textures[tex]->value(p->s, u, v, v3(0, 0, 0));
    out[0] = c.x;
    out[1] = c.y;
    out[2] = c.z;
}

} // extern "C"
