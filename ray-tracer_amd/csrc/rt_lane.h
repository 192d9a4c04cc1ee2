// rt_lane.h -- the per-lane program of the MI355X path tracer: everything one
// SIMD lane computes for one path segment (camera ray, BVH traversal, primitive
// intersection, material scatter), written once as header-only RT_HD functions.
//
// rt_kernels.hip instantiates it inside the HIP kernels (the product).  The same
// header can be compiled for the host by tests/ to diff the lane program against
// the oracle on machines without a GPU; librt_mi355x.so never contains or calls a
// host instantiation -- the product path is the HIP kernel or an error.
//
// All arithmetic is binary64 in the reference's operation order (each function
// cites file:line); compile with -ffp-contract=off for bit-comparable results.
#ifndef RT_LANE_H
#define RT_LANE_H

#include "../../include/rt_rng.h"
#include "rt_libm.h"
#include "rt_types.h"

#include <math.h>

namespace rtl {

#define RTL_INF (__builtin_huge_val())
// The seed of the shared-reciprocal divisions below.  Device: v_rcp_f64 (relative error <= 2^-23 by the ISA manual's bound).
// Host emulation (RT_EMULATE_DEVICE_MATH, include/rt_rng.h): the exact reciprocal spoilt to that bound in a chosen direction
// -- low 29 mantissa bits cleared, set, or either by a hash of the operand (rtl_emul_rcp_mode 0 / 1 / 2) -- so that the CPU suite
// shows the refinement reaching the correctly rounded quotient from ANY seed of the documented accuracy, not from one
// implementation's.
#if defined(__HIP_DEVICE_COMPILE__)
#define RTL_RCP64(x) __builtin_amdgcn_rcp(x)
#define RTL_OUT_OF_LINE static __device__ __attribute__((noinline))
#elif defined(RT_EMULATE_DEVICE_MATH)
extern int rtl_emul_rcp_mode;
extern unsigned long long rtl_emul_rcp_calls;
static inline double rtl_emul_rcp64(double x) {
    double r = 1.0 / x;
    uint64_t b;
    memcpy(&b, &r, sizeof b);
    const bool up = rtl_emul_rcp_mode == 1 || (rtl_emul_rcp_mode == 2 && ((b * 0x9E3779B97F4A7C15ull) >> 63));
    b = up ? (b | 0x1FFFFFFFull) : (b & ~0x1FFFFFFFull);
    memcpy(&r, &b, sizeof b);
    ++rtl_emul_rcp_calls;
    return r;
}
static inline float rtl_emul_rcp32(float x) { // v_rcp_f32's 1 ulp: the correctly rounded reciprocal moved one float down or up
    float r = 1.0f / x;
    if (!(r == r) || r == 0.0f || r - r != 0.0f) return r; // NaN, zero, infinity as they are
    uint32_t b;
    memcpy(&b, &r, sizeof b);
    const bool up = rtl_emul_rcp_mode == 1 || (rtl_emul_rcp_mode == 2 && ((b * 0x9E3779B9u) >> 31));
    b = up ? b + 1u : b - 1u; // in magnitude
    memcpy(&r, &b, sizeof b);
    return r;
}
#define RTL_RCP64(x) rtl::rtl_emul_rcp64(x)
#define RTL_OUT_OF_LINE static inline
#endif
// high word of a binary64 (sign, exponent, 20 mantissa bits)
#if defined(__HIP_DEVICE_COMPILE__)
#define RT_HI32(x) ((uint32_t)__double2hiint(x))
#else
static inline uint32_t rt_hi32(double x) {
    uint64_t b;
    memcpy(&b, &x, sizeof b);
    return (uint32_t)(b >> 32);
}
#define RT_HI32(x) rtl::rt_hi32(x)
#endif
RT_HD uint32_t umin(uint32_t a, uint32_t b) { return a < b ? a : b; }
RT_HD uint32_t umax(uint32_t a, uint32_t b) { return a > b ? a : b; }
#define RTL_PI 3.14159265358979323846264338327950288
#define RTL_EPS 1e-6 /* src/geometry.rs:57,62,159 */

struct V3 {
    double x, y, z;
};
RT_HD V3 mk(double x, double y, double z) {
    V3 r;
    r.x = x;
    r.y = y;
    r.z = z;
    return r;
}
RT_HD V3 operator+(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
RT_HD V3 operator-(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
RT_HD V3 operator-(V3 a) { return mk(-a.x, -a.y, -a.z); }
RT_HD V3 operator*(V3 a, double s) { return mk(a.x * s, a.y * s, a.z * s); }
RT_HD V3 operator*(V3 a, V3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
// Vec3 / f64 (src/vec3.rs:221-231): three quotients by ONE denominator.  The compiler's binary64 division is div_scale x 2,
// v_rcp_f64, two Newton steps on the reciprocal (4 fma), q = a * r, e = fma(-b, q, a), div_fmas(e, r, q), div_fixup -- and
// for operands in the middle of the exponent range div_scale scales nothing, div_fmas is a plain fma and div_fixup returns
// its argument.  The refined reciprocal then depends on the denominator alone: it is computed once and each component costs
// mul + fma + fma with exactly the instructions and operands the full expansion would use (same bits by construction; the
// quotients stay correctly rounded, tests/test_gpu_parity.py::test_device_sqrt_div_correctly_rounded and every bit-exact
// image test).  Operands outside [2^-500, 2^256] in magnitude (zero, denormal, huge) take the ordinary division.
#if defined(RT_DEVICE_MATH) && !defined(RT_PLAIN_DIV3)
// (out of line: inlined at every division site, the ordinary divisions cost registers the hot path needs)
RTL_OUT_OF_LINE V3 div3_ordinary(V3 a, double s) { return mk(a.x / s, a.y / s, a.z / s); }
RT_HD V3 operator/(V3 a, double s) {
    // every operand's biased exponent in [523, 1279]: the three numerators through their min / max
    const uint32_t ex = RT_DOUBLE_HI(a.x) & 0x7FF00000u, ey = RT_DOUBLE_HI(a.y) & 0x7FF00000u,
                   ez = RT_DOUBLE_HI(a.z) & 0x7FF00000u, es = RT_DOUBLE_HI(s) & 0x7FF00000u;
    const uint32_t lo = umin(umin(ex, ey), umin(ez, es)), hi = umax(umax(ex, ey), umax(ez, es));
    if (lo >= 0x20B00000u && hi <= 0x4FF00000u) {
        const double ns = -s;
        const double r0 = RTL_RCP64(s);
        const double f0 = fma(ns, r0, 1.0);
        const double r1 = fma(r0, f0, r0);
        const double f1 = fma(ns, r1, 1.0);
        const double r = fma(r1, f1, r1);
        const double qx = a.x * r, qy = a.y * r, qz = a.z * r;
        return mk(fma(fma(ns, qx, a.x), r, qx), fma(fma(ns, qy, a.y), r, qy), fma(fma(ns, qz, a.z), r, qz));
    }
    return div3_ordinary(a, s);
}
#else
RT_HD V3 operator/(V3 a, double s) { return mk(a.x / s, a.y / s, a.z / s); }
#endif
RT_HD V3 adds(V3 a, double s) { return mk(a.x + s, a.y + s, a.z + s); } // Add<f64>, src/vec3.rs:153-163
RT_HD V3 subs(V3 a, double s) { return mk(a.x - s, a.y - s, a.z - s); } // Sub<f64>, src/vec3.rs:185-195
RT_HD double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }          // src/vec3.rs:76-78
RT_HD double length(V3 a) { return sqrt(a.x * a.x + a.y * a.y + a.z * a.z); }       // src/vec3.rs:88-90
RT_HD V3 normalized(V3 a) { return a / length(a); }                                 // src/vec3.rs:96-98
RT_HD V3 reflected(V3 v, V3 n) { return v - n * dot(v, n) * 2.0; }                  // src/vec3.rs:100-102
RT_HD V3 ld3(const double *p) { return mk(p[0], p[1], p[2]); }

// Vec4::transformed for a point / a vector, rows 0..2 (src/vec4.rs:78-91)
RT_HD V3 xf_point(const double *m, V3 p) {
    return mk(m[0] * p.x + m[1] * p.y + m[2] * p.z + m[3] * 1.0, m[4] * p.x + m[5] * p.y + m[6] * p.z + m[7] * 1.0,
              m[8] * p.x + m[9] * p.y + m[10] * p.z + m[11] * 1.0);
}
RT_HD V3 xf_vector(const double *m, V3 v) {
    return mk(m[0] * v.x + m[1] * v.y + m[2] * v.z + m[3] * 0.0, m[4] * v.x + m[5] * v.y + m[6] * v.z + m[7] * 0.0,
              m[8] * v.x + m[9] * v.y + m[10] * v.z + m[11] * 0.0);
}

// Record i of a scene array as base + a 32-BIT byte offset: the scene arrays are kernel arguments (uniform, in SGPRs), so the
// load becomes `global_load ..., v_offset, s[base:base+1]` -- one full-rate 24-bit multiply (or shift) per record instead of the
// 64-bit index arithmetic of base[i] (v_mad_u64_u32 / v_lshl_add_u64 pairs: quarter-rate, and a VGPR pair per address).  The
// host limits every array to 2^24 records and 4 GiB (rt_host.cpp, RT_MAX_RECORDS).
#if defined(__HIP_DEVICE_COMPILE__)
RT_HD uint32_t mul24(uint32_t a, uint32_t b) { return __umul24(a, b); }
#else
RT_HD uint32_t mul24(uint32_t a, uint32_t b) { return a * b; }
#endif
template <class T>
RT_HD uint32_t rec_off(uint32_t i) { // byte offset of record i: a shift for the power-of-two records, one v_mul_u32_u24 for the others
    return (sizeof(T) & (sizeof(T) - 1)) == 0 ? i * (uint32_t)sizeof(T) : mul24(i, (uint32_t)sizeof(T));
}
// LDS = true (kernels of the box-LIST mode, GENERAL == 2): the scene's records live in the workgroup's LDS (copied there at kernel
// entry, rt_kernels.hip) and `base` is not a pointer but the array's byte offset in the LDS, which the host put into the launch
// descriptor's pointer field (rt_api.cpp fill_launch): a small general scene reads a transform level (96-192 bytes) per primitive
// test, and those loads -- L1 hits all of them -- kept the texture-address path as busy as the VALUs (Cornell box 138 -> 126 ms).
#if defined(__HIP_DEVICE_COMPILE__)
extern __shared__ __attribute__((aligned(16))) unsigned char rt_lds[];
#endif
template <class T> struct RecLdsBit { static constexpr unsigned value = 0u; };
template <> struct RecLdsBit<RtXform> { static constexpr unsigned value = 1u; };
template <> struct RecLdsBit<RtPrimGeo> { static constexpr unsigned value = 2u; };
template <> struct RecLdsBit<RtPrimMeta> { static constexpr unsigned value = 4u; };
template <> struct RecLdsBit<RtPrimExtra> { static constexpr unsigned value = 8u; };
template <> struct RecLdsBit<RtMaterial> { static constexpr unsigned value = 16u; };
template <bool LDS = false, class T>
RT_HD const T &rec_at(const T *base, uint32_t i) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (LDS && (RecLdsBit<T>::value & (unsigned)RT_LIST_LDS_ARRAYS) != 0u) {
        // (the offsets are multiples of 16 -- rt_host.cpp packs the arrays that way -- and saying so here is what lets the compiler
        // read a record with ds_read_b128 instead of pairs of 64-bit reads)
        const uint32_t at = ((uint32_t)reinterpret_cast<uintptr_t>(base) & ~15u) + rec_off<T>(i);
        return *reinterpret_cast<const T *>(__builtin_assume_aligned(rt_lds + at, 16));
    }
#endif
    return *reinterpret_cast<const T *>(reinterpret_cast<const unsigned char *>(base) + rec_off<T>(i));
}

struct Rec { // HitRecord, src/ray.rs:36-43 (material lives on the prim)
    double t;
    V3 p, n;
    double u, v;
};

// ---- per-sample random numbers: include/rt_rng.h ----
struct Rng {
    uint64_t base;   // stream key (keyed medium draws)
    uint64_t s0, s1; // xoroshiro128+ state
    unsigned long long draws;
};
RT_HD double rng_unit53(Rng &g) { // rand::random::<f64>()
    ++g.draws;
    return rt_u64_to_unit53(rt_xoroshiro_next(&g.s0, &g.s1));
}
RT_HD double rng_pm1(Rng &g) { // rand::random::<f64>() * 2.0 - 1.0 (exact, see rt_u64_to_pm1)
    ++g.draws;
    return rt_u64_to_pm1(rt_xoroshiro_next(&g.s0, &g.s1));
}
RT_HD double rng_range01(Rng &g) { // gen_range(0.0, 1.0)
    ++g.draws;
    return rt_u64_to_range01(rt_xoroshiro_next(&g.s0, &g.s1));
}
RT_HD double rng_range11(Rng &g) { // gen_range(-1.0, 1.0)
    ++g.draws;
    return rt_u64_to_range11(rt_xoroshiro_next(&g.s0, &g.s1));
}
// The same sampler with a bound on the iterations of THIS call: when the bound is hit the draws made so far stay
// consumed and *ok is false; calling again simply continues the reference's loop (the rejected candidates are
// never looked at again), so splitting the loop over several calls cannot change the accepted point or the stream.
// A wave runs the loop until its slowest lane accepts (~5.5 iterations for 40 lanes, 1.9 on average per lane): the
// kernel bounds it and lets the unlucky lanes finish in the next shade block.  max_iter <= 0: unbounded.
RT_HD V3 random_in_unit_sphere_bounded(Rng &g, int max_iter, bool *ok) {
    V3 p = mk(1.0, 1.0, 1.0);
    int it = 0;
    *ok = true;
    while (dot(p, p) >= 1.0) {
        if (max_iter > 0 && it == max_iter) {
            *ok = false;
            break;
        }
        ++it;
        const double a = rng_pm1(g), b = rng_pm1(g), c = rng_pm1(g); // random::<f64>() * 2.0 - 1.0, three draws in order
        p = mk(a, b, c);
    }
    return p;
}
RT_HD V3 random_in_unit_sphere(Rng &g) { // src/util.rs:6-15
    V3 p = mk(1.0, 1.0, 1.0);
    while (dot(p, p) >= 1.0) {
        const double a = rng_pm1(g), b = rng_pm1(g), c = rng_pm1(g); // random::<f64>() * 2.0 - 1.0, three draws in order
        p = mk(a, b, c);
    }
    return p;
}
RT_HD V3 random_in_unit_disk(Rng &g) { // src/util.rs:27-42
    for (;;) {
        double a = rng_range11(g);
        double b = rng_range11(g);
        V3 p = mk(a, b, 0.0);
        // `p.length() >= 1.0`: sqrt is monotone and exact at 1, so sqrt(s) >= 1 <=> s >= 1 for every
        // binary64 s (checked in tests/test_oracle_kat.py); the square root itself is never needed
        if (a * a + b * b >= 1.0) continue;
        return p;
    }
}

// ---- PerspectiveCamera::ray (src/camera.rs:91-106) ----
template <bool LENS>
RT_HD void camera_ray(const RtCameraD &c, double u, double v, Rng &g, V3 *o, V3 *d) {
    V3 eye = ld3(c.eye), ll = ld3(c.lower_left), hor = ld3(c.horizontal), ver = ld3(c.vertical);
    if (!LENS) {
        *o = eye;
        *d = normalized(ll + hor * u + ver * v - eye);
    } else {
        V3 rd = random_in_unit_disk(g) * c.lens_radius;
        double offset = rd.x * u + rd.y * v; // scalar built from the screen coordinates (quirk Q2)
        *o = adds(eye, offset);
        *d = normalized(subs(ll + hor * u + ver * v - eye, offset));
    }
}

// The two roots (-b -+ sqrt(disc)) / (2 a) of Sphere::hit share their denominator: one refined reciprocal as in Vec3 / f64
// above, when 2a lies in [2^-100, 2^100] and neither numerator exceeds 2^256.  A numerator below 2^-500 (zero, denormal,
// cancellation dust) may then come out with other low bits than the ordinary division's, but both are below 2^-300 in
// magnitude and a root only ever reaches `> 1e-6` tests before it is used: the same roots are rejected either way.
struct Roots {
    double t1, t2;
};
#if defined(RT_DEVICE_MATH) && !defined(RT_PLAIN_DIV3)
RTL_OUT_OF_LINE Roots roots_ordinary(double n1, double n2, double den) {
    Roots r;
    r.t1 = n1 / den;
    r.t2 = n2 / den;
    return r;
}
RT_HD Roots sphere_roots(double n1, double n2, double den) {
    const uint32_t e1 = RT_DOUBLE_HI(n1) & 0x7FF00000u, e2 = RT_DOUBLE_HI(n2) & 0x7FF00000u,
                   ed = RT_DOUBLE_HI(den) & 0x7FF00000u;
    if (ed - 0x39B00000u <= 0x0C800000u && umax(e1, e2) <= 0x4FF00000u) { // biased exponent of den in [923, 1123], of n1, n2 <= 1279
        const double nd = -den;
        const double r0 = RTL_RCP64(den);
        const double f0 = fma(nd, r0, 1.0);
        const double r1 = fma(r0, f0, r0);
        const double f1 = fma(nd, r1, 1.0);
        const double r = fma(r1, f1, r1);
        const double q1 = n1 * r, q2 = n2 * r;
        Roots out;
        out.t1 = fma(fma(nd, q1, n1), r, q1);
        out.t2 = fma(fma(nd, q2, n2), r, q2);
        return out;
    }
    return roots_ordinary(n1, n2, den);
}
#else
RT_HD Roots sphere_roots(double n1, double n2, double den) {
    Roots r;
    r.t1 = n1 / den;
    r.t2 = n2 / den;
    return r;
}
#endif

// Every sphere a segment tests in WORLD space (translation-only sprites: RT_PRIM_SPHERE_T / RT_PRIM_MEDIUM_T, all of book-one)
// divides by the same 2 d.d, so the refined reciprocal is taken once per segment (trav_begin) and each test's two roots
// cost mul + fma + fma each.  Valid -- not NaN -- only when the conditions of sphere_roots hold for every such test of the
// segment: 2 d.d in [2^-100, 2^100], |o| <= 2^100 per component, and (RtLaunch::world_mid, checked at commit) every world-space
// sphere's centre and radius <= 2^100: then |b| <= 2^155, disc <= 2^311, both numerators <= 2^157 < 2^256.
#define RTL_NAN (__builtin_nan(""))
RT_HD double world_roots_rcp(const RtLaunch &L, V3 o, double a) {
#if defined(RT_DEVICE_MATH) && !defined(RT_PLAIN_DIV3)
    const double den = 2.0 * a;
    const uint32_t ed = RT_DOUBLE_HI(den) & 0x7FF00000u;
    const uint32_t eo = umax(umax(RT_DOUBLE_HI(o.x) & 0x7FF00000u, RT_DOUBLE_HI(o.y) & 0x7FF00000u),
                            RT_DOUBLE_HI(o.z) & 0x7FF00000u);
    if (L.world_mid && ed - 0x39B00000u <= 0x0C800000u && eo <= 0x46300000u) { // den: biased exponent in [923, 1123]; o: <= 1123
        const double nd = -den;
        const double r0 = RTL_RCP64(den);
        const double f0 = fma(nd, r0, 1.0);
        const double r1 = fma(r0, f0, r0);
        const double f1 = fma(nd, r1, 1.0);
        return fma(r1, f1, r1);
    }
#endif
    return RTL_NAN;
}

// ---- Sphere::hit in the sphere's own frame (src/geometry.rs:43-73) ----
// oc = local ray origin (centre at 0), a = d.d.  Returns the parametric hit only.
// r2a: world_roots_rcp of the segment for a world-space sphere, NaN otherwise
RT_HD bool sphere_t(V3 oc, V3 d, double a, double radius, double *t_out, double r2a = RTL_NAN) {
    double b = dot(oc, d) * 2.0;
    double c = dot(oc, oc) - radius * radius;
    double disc = b * b - 4.0 * a * c;
    if (disc < 0.0) return false;
    double sq = sqrt(disc);
    Roots rt;
#if defined(RT_DEVICE_MATH) && !defined(RT_PLAIN_DIV3)
    if (r2a == r2a) {
        const double n1 = -b - sq, n2 = -b + sq, nd = -(2.0 * a);
        const double q1 = n1 * r2a, q2 = n2 * r2a;
        rt.t1 = fma(fma(nd, q1, n1), r2a, q1);
        rt.t2 = fma(fma(nd, q2, n2), r2a, q2);
    } else
#endif
        rt = sphere_roots(-b - sq, -b + sq, 2.0 * a);
    const double t1 = rt.t1, t2 = rt.t2;
    // The reference orders the roots here (`if t1 < t2 { (t1, t2) } else { (t2, t1) }`).  That never changes the outcome:
    // sq >= 0 (or -0, NaN) and 2a >= 0 (a sum of squares), so -b - sq <= -b + sq and t1 <= t2 by the monotonicity of
    // rounding; the swap fires only for equal roots or when one is NaN, and then "the first root above 1e-6, else the
    // second" picks the same value from either order (a NaN is never > 1e-6, zeros of either sign neither).
    double t;
    if (t1 > RTL_EPS)
        t = t1;
    else if (t2 > RTL_EPS)
        t = t2;
    else
        return false;
    *t_out = t;
    return true;
}
// ---- out-of-line transcendental helpers ----
// atan2 / acos / sin / log are the HOST libm's functions restated (rt_libm.h: glibc 2.35's algorithms, its roundings and
// its tables), because that is what the reference calls (`f64::atan2 / acos / sin / ln`) and the device's own library is an
// ulp away from it on 3-27 % of arguments.  Each expands to binary64 polynomials with a dozen 64-bit constants.  Inlined
// into the persistent kernel loop, the compiler hoists every one of those constants out of the loop and keeps them in
// registers for the kernel's whole life (+70 VGPRs, the difference between 2 and 3 waves per SIMD).  As real functions
// they cost a call on paths that are rare (textured hits) or already expensive (a medium test).
#if defined(__HIP_DEVICE_COMPILE__)
#define RT_COLD __device__ __attribute__((noinline))
#else
#define RT_COLD static inline
#endif
// (tests/lane_emul.cpp defines RTL_LOG ... before this header to record the arguments a path passes to them)
#if !defined(RTL_LOG)
#define RTL_LOG(x) rtm::log(x)
#define RTL_SIN(x) rtm::sin(x)
#define RTL_ATAN2(y, x) rtm::atan2(y, x)
#define RTL_ACOS(x) rtm::acos(x)
#endif
struct UV {
    double u, v;
};
RT_COLD UV sphere_uv_cold(double qx, double qy, double qz) { // unitSphereUv, src/geometry.rs:35-39
    UV r;
    r.u = 0.5 + RTL_ATAN2(qx, qz) / (2.0 * RTL_PI);
    r.v = 1.0 - RTL_ACOS(qy) / RTL_PI;
    return r;
}
#if defined(__HIP_DEVICE_COMPILE__) && defined(RT_TU_PART) && RT_TU_PART >= 2
// the kernel families with media / textures (compilations 2 and 3 of rt_kernels.hip hold nothing else): glibc's log table sits at
// the front of the workgroup's LDS (rt_lds.h RT_LDS_LOG_TABLE_BYTES, copied there at kernel entry)
extern __shared__ __attribute__((aligned(16))) unsigned char rt_lds[];
RT_COLD double log_cold(double x) {
    return rtm::log_from(x, (const __attribute__((address_space(3))) double *)reinterpret_cast<const double *>(rt_lds));
}
#else
RT_COLD double log_cold(double x) { return RTL_LOG(x); }
#endif
RT_COLD double checker_sine_cold(double u, double v) { // src/material.rs:238
    return RTL_SIN(2.0 * RTL_PI * 10.0 * u) * RTL_SIN(2.0 * RTL_PI * 10.0 * v);
}
RT_HD void sphere_uv(V3 q, double *u, double *v) {
    const UV r = sphere_uv_cold(q.x, q.y, q.z);
    *u = r.u;
    *v = r.v;
}
// full local record for a known t (src/geometry.rs:66-71).  uv: the reference evaluates atan2 / acos on every hit; they
// only ever reach Texture::value, so they are computed when the hit material's texture reads them (checker / image)
RT_HD void sphere_finish(V3 oc, V3 d, double radius, double t, bool uv, Rec *r) {
    V3 p = oc + d * t;
    V3 q = p / radius;
    r->t = t;
    r->p = p;
    r->n = normalized(q);
    r->u = 0.0;
    r->v = 0.0;
    if (uv) sphere_uv(q, &r->u, &r->v);
}

// ---- Rectangle::hit, local frame (src/geometry.rs:153-180) ----
RT_HD bool rect_hit(V3 o, V3 d, double width, double height, Rec *r) {
    double a0 = -width / 2.0, a1 = -height / 2.0, b0 = width / 2.0, b1 = height / 2.0;
    double t = (0.0 - o.z) / d.z;
    if (isinf(t) || isnan(t) || t < RTL_EPS) return false;
    double x = o.x + d.x * t;
    double y = o.y + d.y * t;
    if (x < a0 || x > b0 || y < a1 || y > b1) return false;
    r->u = (x - a0) / (b0 - a0);
    r->v = (y - a1) / (b1 - a1);
    r->t = t;
    r->p = o + d * t;
    r->n = mk(0.0, 0.0, 1.0);
    return true;
}

// forward half of Sprite::hit / TransformedGeometry::hit (src/sprite.rs:108-126)
RT_HD void to_world(const RtXform &x, Rec *r) {
    r->p = xf_point(x.m, r->p);
    r->n = xf_vector(x.m, r->n); // M, not M^-T, not renormalised (quirk Q5)
}

// ---- ConstantMedium<Sphere>::hit, boundary frame (src/volume.rs:46-100) ----
// neg_inv_density = (-1.0 / density), the reference's own first factor (volume.rs:62,84), divided once on the host.
// RECORD = false (traversal: only t is used): the reference decides "entering or leaving" by the sign of
// record1.normal . direction with normal = normalized(p / r) -- six divisions and a square root for one sign.  That dot
// product equals c * sum_i p_i d_i (1 + theta_i) with c = 1 / (r |p / r|) > 0 and |theta_i| < 8 * 2^-53 (one rounding each
// in p_i / r, the length, q_i / length, the product, two sums), so whenever |fl(sum p_i d_i)| > 1e-12 * sum |p_i d_i| (and
// nothing is near the ends of the exponent range) its sign is the sign of the plain sum; only the remaining, grazing
// cases evaluate the reference's expression.  The record (RECORD = true) always does.
template <bool RECORD>
RT_HD bool medium_hit(V3 oc, V3 d, double radius, double neg_inv_density, uint64_t rng_base, uint32_t segment, uint32_t slot,
                      unsigned long long *draws, bool uv, Rec *r) {
    double a = dot(d, d);
    double t1;
    if (!sphere_t(oc, d, a, radius, &t1)) return false;
    Rec r1;
    bool entering = false;
    bool sure = false;
    if (!RECORD) {
        r1.t = t1;
        r1.p = oc + d * t1; // sphere_finish's p
        const double x = r1.p.x * d.x, y = r1.p.y * d.y, z = r1.p.z * d.z;
        const double s = x + y + z, mag = fabs(x) + fabs(y) + fabs(z);
        sure = radius > 1e-100 && radius < 1e100 && mag > 1e-150 && mag < 1e150 && fabs(s) > 1e-12 * mag;
        entering = s < 0.0;
    }
    if (!sure) {
        sphere_finish(oc, d, radius, t1, uv, &r1);
        entering = dot(r1.n, d) < 0.0;
    }
    if (entering) {
        V3 o2 = r1.p + d * 1e-6; // restarted ray
        double t2;
        if (!sphere_t(o2, d, a, radius, &t2)) return false;
        Rec r2;
        r2.t = t2;
        if (RECORD) sphere_finish(o2, d, radius, t2, uv, &r2);
        double inside = r2.t;
        ++*draws;
        double distance = neg_inv_density * log_cold(rt_u64_to_range01(rt_rng_keyed_from_base(rng_base, segment, slot)));
        if (distance > inside) return false;
        r->t = r1.t + distance;
        if (RECORD) {
            r->u = r1.u + r2.u;
            r->v = r1.v + r2.v;
            r->p = o2 + d * (r1.t + distance); // on the restarted ray (quirk Q9)
            r->n = (r1.n + r2.n) / 2.0;
        }
        return true;
    }
    double inside = r1.t;
    ++*draws;
    double distance = neg_inv_density * log_cold(rt_u64_to_range01(rt_rng_keyed_from_base(rng_base, segment, slot)));
    if (distance > inside) return false;
    r->t = distance;
    if (RECORD) {
        r->u = r1.u;
        r->v = r1.v;
        r->p = oc + d * distance;
        r->n = r1.n;
    }
    return true;
}

struct SegCtx { // what a primitive test may need besides the ray
    double r2a; // world_roots_rcp of this segment
    uint64_t rng_base;
    uint32_t segment;
    unsigned long long draws;
    unsigned long long prims_tested;
};

// ---- transform chains: Sprite::hit / TransformedGeometry::hit level by level (src/sprite.rs:101-126, src/geometry.rs:220-236)
// A level that is a pure translation by t (bit RT_META_TMASK_SHIFT + i of the meta word) contributes only its offsets:
// M^-1 (o,1) = o + inv_t, M^-1 (d,0) = d, M (p,1) = p + t, M (n,0) = n -- the values the 4x4 products give for it.
// (constant trip counts: the compiler unrolls them, and a level the wave does not have is one skipped branch; a
// data-dependent loop here made it keep the hit record in scratch memory)
template <bool DEEP = false, bool LDSREC = false>
RT_HD void chain_down(const RtLaunch &L, uint32_t first, uint32_t len, uint32_t tmask, V3 *o, V3 *d) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (uint32_t i = 0; i < (uint32_t)RT_MAX_CHAIN; ++i) {
        if (i >= len) continue;
        const RtXform &X = rec_at<LDSREC>(L.xforms, first + i);
        if ((tmask >> i) & 1u) {
            *o = mk(o->x + X.inv[3], o->y + X.inv[7], o->z + X.inv[11]);
        } else {
            const V3 lo = xf_point(X.inv, *o);
            *d = xf_vector(X.inv, *d); // not renormalised (quirk Q5)
            *o = lo;
        }
    }
    // levels beyond the four unrolled ones (only the kernel family for general media / deep chains is compiled with them):
    // always the full 4x4 products, the reference's own form (src/sprite.rs:101-106)
    if (DEEP)
        for (uint32_t i = (uint32_t)RT_MAX_CHAIN; i < len; ++i) {
            const RtXform &X = rec_at<LDSREC>(L.xforms, first + i);
            const V3 lo = xf_point(X.inv, *o);
            *d = xf_vector(X.inv, *d);
            *o = lo;
        }
}
template <bool DEEP = false, bool LDSREC = false>
RT_HD void chain_up(const RtLaunch &L, uint32_t first, uint32_t len, uint32_t tmask, Rec *r) {
    if (DEEP)
        for (uint32_t i = len; i > (uint32_t)RT_MAX_CHAIN; --i) to_world(rec_at<LDSREC>(L.xforms, first + i - 1u), r);
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (uint32_t k = 0; k < (uint32_t)RT_MAX_CHAIN; ++k) {
        const uint32_t i = (uint32_t)RT_MAX_CHAIN - 1u - k;
        if (i >= len) continue;
        const RtXform &X = rec_at<LDSREC>(L.xforms, first + i);
        if ((tmask >> i) & 1u)
            r->p = mk(r->p.x + X.m[3], r->p.y + X.m[7], r->p.z + X.m[11]);
        else
            to_world(X, r);
    }
}

template <bool DEEP = false, bool LDSREC = false>
RT_HD void chain_up_point(const RtLaunch &L, uint32_t first, uint32_t len, uint32_t tmask, V3 *p) {
    if (DEEP)
        for (uint32_t i = len; i > (uint32_t)RT_MAX_CHAIN; --i) *p = xf_point(rec_at<LDSREC>(L.xforms, first + i - 1u).m, *p);
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (uint32_t k = 0; k < (uint32_t)RT_MAX_CHAIN; ++k) {
        const uint32_t i = (uint32_t)RT_MAX_CHAIN - 1u - k;
        if (i >= len) continue;
        const RtXform &X = rec_at<LDSREC>(L.xforms, first + i);
        if ((tmask >> i) & 1u)
            *p = mk(p->x + X.m[3], p->y + X.m[7], p->z + X.m[11]);
        else
            *p = xf_point(X.m, *p);
    }
}

// ---- the reference's own boxes, for the one class of ray on which they are NOT result-neutral ----
// Everywhere else the binary32 culling boxes stand in for AxisAlignedBoundingBox::hit because a box never rejects what the
// primitive inside it accepts (F7): the primitive's extent lies inside its box, and a ray that crosses a box plane does so with a
// slab interval far wider than any rounding.  That premise fails for a ray that runs IN a box's boundary plane -- a direction
// component that is exactly zero (1 / 0 = inf; (plane - o) * inf is NaN when o sits on the plane, and the axis drops out of the
// reference's selects, but +-inf when o is one rounding error beyond it: both ends of the slab the same infinity, the box and
// everything in it dropped, src/optimize.rs:61-82,469-498) or so small (1e-17 of the others) that the ray moves less than an
// ulp of its own coordinates across the whole scene: whether such a ray is "in" a box whose plane it runs along is decided by
// the last bit of o, while the primitive's own test reaches the same plane through a different chain of roundings and compares
// inclusively (src/geometry.rs:153-180).  Seen on rays that leave a cube's face ALONG the face, and then run along its edges,
// after the degenerate refraction of scenes scaled up 1e8-fold (profiles/r04_boundary_plane_probe.json).  So a hit found on a
// segment whose direction is that close to an axis plane is put through the binary64 boxes the reference keeps for the objects
// above the primitive -- each object's OWN box (RtXformBox), i.e. the answer of every tree in which that object sits in a node
// of its own; where the reference's random trees disagree with one another (a box shared with a sibling can be larger) this is
// one of their answers.  For any other ray the boxes admit what the primitive accepts, so the gate only decides who pays.
// near_axis: some component is zero or below 2^-30 of the largest (compared through the exponent fields; the rays in
// question have components of exactly 0 or ~1e-17 of the others, a random direction qualifies once in ~1e9)
// NOT covered (docs/parity.md section 4): the gate looks at the direction in the WORLD frame (the medium's frame inside a medium).
// A ray that is axis-parallel only in the local frame of a rotated object -- its world direction has no small component, its
// direction after M^-1 does -- still goes through the binary32 culling boxes above that object; the reference's boxes of the
// levels BELOW the rotation see it as an in-plane ray.  No sweep has produced one (it needs a direction that cancels to ~1e-17
// through a general matrix); it would show as a pixel on which the oracle's own trees disagree, like the nine of round 4.
// Cost of the class where it does fire (trav_begin): one binary64 test per leaf prim and segment in ONE lane -- n_prims x the
// path's remaining segments; nothing in an ordinary scene, and in a wide scene (41 000 prims) an edge-running path of 100
// segments is 4 M tests = a few ms of one wave, bounded by max_depth.
RT_HD bool near_axis(V3 d) {
    const uint32_t ex = RT_HI32(d.x) & 0x7FFFFFFFu, ey = RT_HI32(d.y) & 0x7FFFFFFFu, ez = RT_HI32(d.z) & 0x7FFFFFFFu;
    return umin(umin(ex, ey), ez) + (30u << 20) <= umax(umax(ex, ey), ez);
}
// AxisAlignedBoundingBox::hit (src/optimize.rs:61-82): t in [0, inf), the reference's selects and their NaN behaviour
RT_HD bool ref_slab(double lo, double hi, double o, double d, double *tmin, double *tmax) { // one axis of the loop
    const double inv = 1.0 / d;
    double t0 = (lo - o) * inv;
    double t1 = (hi - o) * inv;
    if (inv < 0.0) {
        const double t = t0;
        t0 = t1;
        t1 = t;
    }
    *tmin = t0 > *tmin ? t0 : *tmin;
    *tmax = t1 < *tmax ? t1 : *tmax;
    return !(*tmax <= *tmin);
}
RT_HD bool ref_box_hit(const double *lo, const double *hi, V3 o, V3 d) {
    double tmin = 0.0, tmax = RTL_INF;
    return ref_slab(lo[0], hi[0], o.x, d.x, &tmin, &tmax) && ref_slab(lo[1], hi[1], o.y, d.y, &tmin, &tmax) &&
           ref_slab(lo[2], hi[2], o.z, d.z, &tmin, &tmax);
}
// the boxes of a chain's levels, each against the ray in the frame above the level (chain_down's arithmetic level by level)
// (out of line, and WITHOUT the launch descriptor: a by-value kernel argument whose address reaches a real call is copied to
// scratch memory at kernel entry and read from there ever after -- measured: cornell 138 -> 184 ms)
template <bool DEEP>
RT_COLD bool chain_boxes_admit(const RtXform *xforms, uint32_t first, uint32_t len, uint32_t tmask, V3 o, V3 d) {
    for (uint32_t i = 0; i < len; ++i) {
        if (!DEEP && i >= (uint32_t)RT_MAX_CHAIN) break;
        const RtXform &X = rec_at(xforms, first + i);
        const RtXformBox &B = reinterpret_cast<const RtXformBox *>(xforms)[-1 - (int32_t)(first + i)]; // rt_types.h
        if (B.lo[0] == B.lo[0] && !ref_box_hit(B.lo, B.hi, o, d)) return false;
        if (i < (uint32_t)RT_MAX_CHAIN && ((tmask >> i) & 1u)) {
            o = mk(o.x + X.inv[3], o.y + X.inv[7], o.z + X.inv[11]);
        } else {
            const V3 lo = xf_point(X.inv, o);
            d = xf_vector(X.inv, d);
            o = lo;
        }
    }
    return true;
}

// sphere / rectangle in its own frame.  RECORD = false: only r->t is meaningful
template <bool RECORD>
RT_HD bool shape_hit(uint32_t kind, const RtPrimGeo &G, V3 lo, V3 ld, bool uv, Rec *r) {
    if (kind == RT_PRIM_SPHERE_C) {
        const double la = dot(ld, ld);
        double t;
        if (!sphere_t(lo, ld, la, G.g[0], &t)) return false;
        r->t = t;
        if (RECORD) sphere_finish(lo, ld, G.g[0], t, uv, r);
        return true;
    }
    return rect_hit(lo, ld, G.g[0], G.g[1], r); // RT_PRIM_RECT_C
}

// ---- ConstantMedium<T: Hit>::hit over any boundary (src/volume.rs:40-100): the boundary's spheres / rectangles / media
// prims [first, first + count), each under its own chain below the medium; boundary.hit = their nearest hit,
// strict <, the earlier one wins ties (src/geometry.rs:76-116, src/optimize.rs:469-498).
// A boundary prim may itself be a ConstantMedium (T: Hit is generic, src/volume.rs:18-44): it is evaluated -- with a draw of
// its own, keyed by this evaluation's key and pass (include/rt_rng.h) -- by each of the two boundary.hit calls.  NEST = levels
// of media that may still follow below this one (the host bounds the nesting to RT_MAX_MEDIUM_NESTING).
template <int NEST, bool LDSREC>
RT_HD bool medium_general_hit(const RtLaunch &L, const RtPrimGeo &G, V3 o, V3 d, uint64_t rng_base, uint32_t segment, uint32_t key,
                              unsigned long long *draws, unsigned long long *tests, bool uv, Rec *r);
// a medium inside a medium's boundary: a real call (rare, and inlining three copies of the evaluation into each other triples the
// slow kernel family's code)
template <int NEST, bool LDSREC>
RT_COLD bool medium_nested_call(const RtLaunch &L, const RtPrimGeo &G, V3 o, V3 d, uint64_t rng_base, uint32_t segment, uint32_t key,
                                unsigned long long *draws, unsigned long long *tests, bool uv, Rec *r) {
    return medium_general_hit<NEST, LDSREC>(L, G, o, d, rng_base, segment, key, draws, tests, uv, r);
}
template <int NEST, bool LDSREC>
RT_HD bool medium_general_hit(const RtLaunch &L, const RtPrimGeo &G, V3 o, V3 d, uint64_t rng_base, uint32_t segment, uint32_t key,
                              unsigned long long *draws, unsigned long long *tests, bool uv, Rec *r) {
    const double density = G.g[0];
    const uint32_t first = (uint32_t)G.g[1], count = (uint32_t)G.g[2];
    Rec r1, r2;
    V3 ro = o;
    const bool zd = near_axis(d);
    for (int pass = 0; pass < 2; ++pass) { // boundary.hit(ray), then boundary.hit(restarted ray)
        Rec best;
        bool have = false;
        for (uint32_t k = 0; k < count; ++k) {
            const uint32_t ci = first + k;
            const RtPrimMeta &CM = rec_at<LDSREC>(L.prim_meta, ci);
            const uint32_t kw = CM.kind;
            const uint32_t cf = CM.xform, cl = (kw >> RT_META_CHAIN_SHIFT) & 0xFu, cm = (kw >> RT_META_TMASK_SHIFT) & 0xFu;
            V3 co = ro, cd = d;
            chain_down<true, LDSREC>(L, cf, cl, cm, &co, &cd);
            Rec cr;
            ++*tests;
            bool hit;
            if ((kw & 0xFFu) == RT_PRIM_MEDIUM_C) {
                if constexpr (NEST > 0)
                    hit = medium_nested_call<NEST - 1, LDSREC>(L, rec_at<LDSREC>(L.prim_geo, ci), co, cd, rng_base, segment,
                                                       rt_medium_key_inner(key, (uint32_t)pass, CM.aux), draws, tests, uv, &cr);
                else
                    hit = false; // (the host refuses deeper nesting)
            } else {
                hit = shape_hit<true>(kw & 0xFFu, rec_at<LDSREC>(L.prim_geo, ci), co, cd, uv, &cr);
            }
            // (the boundary's own nodes -- a Cube, a node of sprites -- keep boxes too)
            if (hit && zd && !chain_boxes_admit<true>(L.xforms_global, cf, cl, cm, ro, d)) hit = false;
            if (hit && (!have || cr.t < best.t)) {
                chain_up<true, LDSREC>(L, cf, cl, cm, &cr);
                best = cr;
                have = true;
            }
        }
        if (!have) return false;
        if (pass == 0) {
            r1 = best;
            if (!(dot(r1.n, d) < 0.0)) break;  // the origin is inside the boundary
            ro = r1.p + d * 1e-6;               // restarted ray
        } else {
            r2 = best;
        }
    }
    if (dot(r1.n, d) < 0.0) {
        const double inside = r2.t;
        ++*draws;
        const double distance = (-1.0 / density) * log_cold(rt_u64_to_range01(rt_rng_medium_draw(rng_base, segment, key)));
        if (distance > inside) return false;
        r->u = r1.u + r2.u;
        r->v = r1.v + r2.v;
        r->t = r1.t + distance;
        r->p = ro + d * (r1.t + distance); // on the restarted ray (quirk Q9)
        r->n = (r1.n + r2.n) / 2.0;
        return true;
    }
    const double inside = r1.t;
    ++*draws;
    const double distance = (-1.0 / density) * log_cold(rt_u64_to_range01(rt_rng_medium_draw(rng_base, segment, key)));
    if (distance > inside) return false;
    r->u = r1.u;
    r->v = r1.v;
    r->t = distance;
    r->p = o + d * distance;
    r->n = r1.n;
    return true;
}

// Intersect primitive `pi` with the world ray; on a hit fill the world-space record.
// RECORD = false: only r->t is meaningful (traversal); true: full record (shading), uv when `uv`.
// MEDIUM: 0 no media in the scene, 1 only ConstantMedium<Sphere> sprites under a pure translation (RT_PRIM_MEDIUM_T,
// the reference's own scenes), 2 media over any boundary (RT_PRIM_MEDIUM_C) as well, 3 media inside the boundary of media too.
template <int GENERAL, int MEDIUM, bool RECORD>
RT_HD bool prim_hit(const RtLaunch &L, uint32_t pi, V3 o, V3 d, double a, SegCtx &sc, Rec *r, bool uv_wanted) {
    // uv inside the record is only ever asked for by prim_uv's medium case (kernel family MEDIUM = 2)
    const bool uv = MEDIUM >= 2 && uv_wanted;
    // (the spheres-only family keeps the plain indexing: measured 0.9 % faster there, 2.5 % slower on the book-two cover)
    const RtPrimGeo &G = (GENERAL || MEDIUM) ? rec_at<GENERAL == 2>(L.prim_geo, pi) : L.prim_geo[pi];
    uint32_t kw = (uint32_t)RT_PRIM_SPHERE_T;
    if (GENERAL || MEDIUM) kw = rec_at<GENERAL == 2>(L.prim_meta, pi).kind; // bits 8-15 carry the material's kind, 16-23 the chain
    const uint32_t kind = kw & 0xFFu;
    ++sc.prims_tested;
    if (kind == RT_PRIM_SPHERE_T) {
        // Sprite::hit with a translation matrix: M^-1 (o,1) = o - c, M^-1 (d,0) = d,
        // M (p,1) = p + c, M (n,0) = n  (src/sprite.rs:101-126, src/vec4.rs:78-91)
        V3 c = mk(G.g[0], G.g[1], G.g[2]);
        V3 oc = o - c;
        double t;
        if (!sphere_t(oc, d, a, G.g[3], &t, sc.r2a)) return false;
        r->t = t;
        if (RECORD) {
            sphere_finish(oc, d, G.g[3], t, uv, r);
            r->p = r->p + c;
        }
        return true;
    }
    if (GENERAL || MEDIUM) {
        const RtPrimMeta &P = rec_at<GENERAL == 2>(L.prim_meta, pi);
        if (MEDIUM && kind == RT_PRIM_MEDIUM_T) {
            V3 c = mk(G.g[0], G.g[1], G.g[2]);
            if (!medium_hit<RECORD>(o - c, d, G.g[3], rec_at<GENERAL == 2>(L.prim_extra, pi).e[1], sc.rng_base, sc.segment, P.aux, &sc.draws, uv, r)) return false;
            if (RECORD) r->p = r->p + c;
            return true;
        }
        const uint32_t first = P.xform, len = (kw >> RT_META_CHAIN_SHIFT) & 0xFu, tmask = (kw >> RT_META_TMASK_SHIFT) & 0xFu;
        V3 lo = o, ld = d;
        chain_down<(MEDIUM >= 2), GENERAL == 2>(L, first, len, tmask, &lo, &ld);
        bool ok;
        if (MEDIUM >= 2 && kind == RT_PRIM_MEDIUM_C)
            ok = medium_general_hit<(MEDIUM >= 3 ? RT_MAX_MEDIUM_NESTING - 1 : 0), GENERAL == 2>(L, G, lo, ld, sc.rng_base, sc.segment, P.aux, &sc.draws, &sc.prims_tested, uv, r);
        else
            ok = shape_hit<RECORD>(kind, G, lo, ld, uv, r);
        if (!ok) return false;
        if (RECORD) chain_up<(MEDIUM >= 2), GENERAL == 2>(L, first, len, tmask, r);
        return true;
    }
    return false;
}

// A hit of primitive `pi` on a segment whose direction is (all but) parallel to an axis plane: would the reference's boxes
// above the primitive have let the ray through (see ref_box_hit)?  Hardly ever called; the work is out of line.
// a sprite of the world's list under a pure translation (centre c): the 8 corners (+-r, +-r, +-r) through M are +-r + c
RT_HD bool sphere_box_admits(double cx, double cy, double cz, double radius, V3 o, V3 d) {
    const double r = fabs(radius);
    double tmin = 0.0, tmax = RTL_INF;
    return ref_slab(-r + cx, r + cx, o.x, d.x, &tmin, &tmax) && ref_slab(-r + cy, r + cy, o.y, d.y, &tmin, &tmax) &&
           ref_slab(-r + cz, r + cz, o.z, d.z, &tmin, &tmax);
}
RT_COLD bool sphere_box_admits_cold(double cx, double cy, double cz, double radius, V3 o, V3 d) {
    return sphere_box_admits(cx, cy, cz, radius, o, d);
}
template <int GENERAL, int MEDIUM>
RT_HD bool own_boxes_admit(const RtLaunch &L, uint32_t pi, V3 o, V3 d) {
    uint32_t kw = (uint32_t)RT_PRIM_SPHERE_T;
    if (GENERAL || MEDIUM) kw = rec_at<GENERAL == 2>(L.prim_meta, pi).kind;
    const uint32_t kind = kw & 0xFFu;
    if (kind == RT_PRIM_SPHERE_T || kind == RT_PRIM_MEDIUM_T) {
        const RtPrimGeo &G = (GENERAL || MEDIUM) ? rec_at<GENERAL == 2>(L.prim_geo, pi) : L.prim_geo[pi];
        // (the spheres-only family has no other real call with arguments on the stack: inline there, it runs without scratch memory)
        return (GENERAL || MEDIUM) ? sphere_box_admits_cold(G.g[0], G.g[1], G.g[2], G.g[3], o, d) : sphere_box_admits(G.g[0], G.g[1], G.g[2], G.g[3], o, d);
    }
    if (GENERAL || MEDIUM)
        return chain_boxes_admit<(MEDIUM >= 2)>(L.xforms_global, rec_at<GENERAL == 2>(L.prim_meta, pi).xform, (kw >> RT_META_CHAIN_SHIFT) & 0xFu,
                                                (kw >> RT_META_TMASK_SHIFT) & 0xFu, o, d);
    return true;
}

// The hit record of primitive `pi` at the parameter t the traversal found (shading).  The same arithmetic as the full test
// -- which would only compute the same t again from the same operands (two divisions and a square root for a sphere, a
// division for a rectangle) and re-check bounds that are known to hold.  A rectangle's world normal is a constant of the
// leaf: the chain applied to (0, 0, 1), evaluated once at commit by the same expressions (rt_host.cpp, leaf_normal) and kept
// in geo.g[2], geo.g[3], extra.e[0].  Media go through the full test (keyed draw, both boundary hits).
template <int GENERAL, int MEDIUM>
RT_HD void prim_record(const RtLaunch &L, uint32_t pi, V3 o, V3 d, double t, SegCtx &sc, Rec *r) {
    const RtPrimGeo &G = rec_at<GENERAL == 2>(L.prim_geo, pi);
    uint32_t kw = (uint32_t)RT_PRIM_SPHERE_T;
    if (GENERAL || MEDIUM) kw = rec_at<GENERAL == 2>(L.prim_meta, pi).kind;
    const uint32_t kind = kw & 0xFFu;
    if (kind == RT_PRIM_SPHERE_T) {
        const V3 c = mk(G.g[0], G.g[1], G.g[2]);
        sphere_finish(o - c, d, G.g[3], t, false, r);
        r->p = r->p + c;
        return;
    }
    if (GENERAL || MEDIUM) {
        if (kind == RT_PRIM_MEDIUM_T || kind == RT_PRIM_MEDIUM_C) {
            prim_hit<GENERAL, MEDIUM, true>(L, pi, o, d, dot(d, d), sc, r, false);
            return;
        }
        const uint32_t first = rec_at<GENERAL == 2>(L.prim_meta, pi).xform, len = (kw >> RT_META_CHAIN_SHIFT) & 0xFu, tmask = (kw >> RT_META_TMASK_SHIFT) & 0xFu;
        V3 lo = o, ld = d;
        chain_down<(MEDIUM >= 2), GENERAL == 2>(L, first, len, tmask, &lo, &ld);
        if (kind == RT_PRIM_SPHERE_C) {
            sphere_finish(lo, ld, G.g[0], t, false, r);
            chain_up<(MEDIUM >= 2), GENERAL == 2>(L, first, len, tmask, r);
        } else { // RT_PRIM_RECT_C: rect_hit's t, p (src/geometry.rs:160,176), the chain for the point, the leaf's normal
            r->t = t;
            r->u = 0.0;
            r->v = 0.0;
            r->p = lo + ld * t;
            chain_up_point<(MEDIUM >= 2), GENERAL == 2>(L, first, len, tmask, &r->p);
            r->n = mk(G.g[2], G.g[3], rec_at<GENERAL == 2>(L.prim_extra, pi).e[0]);
        }
    }
}

// ---------------------------------------------------------------- traversal
// The reference's BVH walk (src/optimize.rs:469-498) visits every box the ray
// touches in [0, inf), never prunes, never orders; its result is the nearest
// primitive hit, whatever the tree.  Here the boxes are binary32 CULLING volumes
// (rounded outward + padded, rt_host.cpp cull_box) tested against a binary32 image
// of the ray with a per-ray pad, in [0, best_t]: they decide only which binary64
// primitive tests are skipped, and a skipped test is one that could not win.
// Every hit/miss/nearest decision is made by prim_hit in binary64.
struct Trav {
    uint32_t cur;        // node / leaf reference, kDone or kDead in the Stack policy's reference form (Stack::Ref)
    int32_t sp;          // stack pointer (entries live in the Stack policy object)
    double best_t;
    uint32_t best_prim;
    float best32;        // best_t rounded up to binary32
    double r2a;          // world_roots_rcp of the segment
    float idx, idy, idz; // 1 / d
    float nx, ny, nz;    // -(o/d + pad): entry planes,  t = plane * id + n
    float fx, fy, fz;    // -(o/d - pad): exit planes
    uint32_t ox, oy, oz; // byte offset inside an RtNode of the (child0, child1) pair of the plane the ray ENTERS through on
                         // each axis (lo_* for a positive direction, hi_* for a negative one); the exit pair is at
                         // ox ^ 24, oy ^ 40, oz ^ 56
};

RT_HD float up32(double t) { // >= t in binary32, with room for the ~3e-7 relative error of a computed tnear
    float f = (float)t;
    return f * 1.000001f;
}
RT_HD float rcp32(float x) { // 1/x to 1 ulp: v_rcp_f32 on the device (culling only; covered by the slack below)
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rcpf(x);
#elif defined(RT_EMULATE_DEVICE_MATH)
    return rtl_emul_rcp32(x);
#else
    return 1.0f / x;
#endif
}
RT_HD float bits_f32(uint32_t u) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __uint_as_float(u);
#else
    float f;
    memcpy(&f, &u, sizeof f);
    return f;
#endif
}
RT_HD uint32_t f32_bits(float f) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __float_as_uint(f);
#else
    uint32_t u;
    memcpy(&u, &f, sizeof u);
    return u;
#endif
}

// next node from the stack, skipping entries that start behind the best hit
template <class Stack>
RT_HD void trav_pop(Trav &tv, Stack &st) {
    for (;;) {
        if (tv.sp == 0) {
            tv.cur = Stack::Ref::kDone;
            return;
        }
        float tnear;
        uint32_t ref;
        st.pop(tv.sp, &tnear, &ref);
        if (tnear > tv.best32) continue; // the stored tnear is a lower bound (16-bit form: rounded DOWN): conservative
        tv.cur = ref;
        return;
    }
}

// The reference never renormalises a direction that went through a matrix (quirk Q5) and never looks at its components: a path
// that keeps scattering under a non-rigid matrix can carry |d| = 1e-38, and one that bounces between two parallel mirrors drives
// ONE component towards zero, 3-fold per bounce, while |d| stays put -- ordinary binary64 rays for the reference's binary64
// slabs.  In the binary32 culling boxes 1 / d_i of such a component is finite but o_i / d_i and plane / d_i overflow, separately,
// to infinities of either sign; their difference came out +inf where the slab holds the whole ray, and the box -- with the
// mirror, or the medium, in it -- was culled (two scenes of the 60 000- and 50 000-scene sweeps of round 3).  A reciprocal
// beyond 2^60 is therefore made INFINITE: with it every product of that axis is inf or NaN and the axis drops out of the slab
// test (the rule the exact zeros of a direction have always followed, see below) -- conservative, and when all three
// components are that small the segment simply walks without culling.  Nothing changes for any other segment.
// binary32 constants of a segment's ray (what a traversal needs besides o, d and the best hit so far)
// Returns true for a segment that runs (all but) in an axis plane -- some |d_i| below 2^-29 of the largest, zero included: the
// one kind of ray that does not use the culling structure at all (trav_begin, ref_box_hit).  The test rides on the rare branch
// the reciprocals needed anyway: three binary32 operations per segment.
// does the Stack policy's node array hold RtNodeH records (binary16 planes, 32 bytes)?  (a policy without the member: no)
template <class S, class = void>
struct StackHalfNodes {
    static constexpr bool value = false;
};
template <class S>
struct StackHalfNodes<S, decltype((void)S::kHalfNodes)> {
    static constexpr bool value = S::kHalfNodes;
};
template <bool HALFNODES = false>
RT_HD bool trav_ray_constants(const RtLaunch &L, V3 o, V3 d, Trav &tv) {
    const float ox = (float)o.x, oy = (float)o.y, oz = (float)o.z;
    tv.idx = rcp32((float)d.x);
    tv.idy = rcp32((float)d.y);
    tv.idz = rcp32((float)d.z);
    bool near_plane = false;
    const float big = fmaxf(fmaxf(fabsf(tv.idx), fabsf(tv.idy)), fabsf(tv.idz)), small = fminf(fminf(fabsf(tv.idx), fabsf(tv.idy)), fabsf(tv.idz));
    if (__builtin_expect(!(big <= fminf(small * 0x1p29f, 0x1p60f)), 0)) { // (hardly ever; a direction of three NaNs comes here too)
        // a direction with a NaN component hits nothing on either way (every primitive test compares false); it takes the ordinary
        // walk, whose slabs drop a NaN axis, instead of the scan over all prims below (41 000 tests per segment in a wide scene)
        near_plane = !(big <= small * 0x1p29f) && d.x == d.x && d.y == d.y && d.z == d.z;
        if (fabsf(tv.idx) > 0x1p60f) tv.idx = copysignf(__builtin_huge_valf(), tv.idx);
        if (fabsf(tv.idy) > 0x1p60f) tv.idy = copysignf(__builtin_huge_valf(), tv.idy);
        if (fabsf(tv.idz) > 0x1p60f) tv.idz = copysignf(__builtin_huge_valf(), tv.idz);
    }
    // the binary32 ray is displaced from the binary64 one by <= 2^-24 (|o| + t|d|) per axis;
    // the box pad covers the |o + t d| share, this pad the |o| share (>2x margin each)
    const float e = fmaxf(fmaxf(fabsf(ox), fabsf(oy)), fabsf(oz)) * 0x1p-21f + 1e-30f;
    const float qx = ox * tv.idx, qy = oy * tv.idy, qz = oz * tv.idz;
    const float ex = e * fabsf(tv.idx), ey = e * fabsf(tv.idy), ez = e * fabsf(tv.idz);
    tv.nx = -(qx + ex);
    tv.ny = -(qy + ey);
    tv.nz = -(qz + ez);
    tv.fx = -(qx - ex);
    tv.fy = -(qy - ey);
    tv.fz = -(qz - ez);
    // sign bit of 1/d picks the entry plane of each axis (an infinite or NaN reciprocal only ever removes constraints)
    // (list mode, trav_list_step: byte offset inside a list box of the {entry, exit} plane pair of each axis)
    // (RtNodeH: lo_x at byte 0, lo_y 4, lo_z 8, the hi_* planes 12 bytes on; the exit pair is at ox ^ 12, oy ^ 20, oz ^ 28)
    const uint32_t flip = HALFNODES ? 12u : (L.n_list ? 4u : 24u), ay = HALFNODES ? 4u : (L.n_list ? 12u : 8u), az = HALFNODES ? 8u : (L.n_list ? 24u : 16u);
    tv.ox = (f32_bits(tv.idx) >> 31) * flip;
    tv.oy = (f32_bits(tv.idy) >> 31) * flip + ay;
    tv.oz = (f32_bits(tv.idz) >> 31) * flip + az;
    return near_plane;
}

// start a segment: binary32 ray constants, hoisted prims, root
// number of leaf prims: only the careful scan below reads it.  On the device it is re-read from the kernel-argument segment where
// it is used (the launch descriptor is the kernel's first and only argument): a field read through the by-value argument is
// loaded in the kernel's prologue and kept in an SGPR for the whole persistent loop, and one more live SGPR there is a spill to a
// VGPR lane -- the 121st VGPR of the spheres-only kernel, which then no longer leaves room for the sums of the previous render
// beside it (rt_kernels.hip kernarg_now, reduce_kernel)
#if defined(__HIP_DEVICE_COMPILE__)
RT_HD int32_t launch_n_prims(const RtLaunch &) {
    const __attribute__((address_space(4))) RtLaunch *p = (const __attribute__((address_space(4))) RtLaunch *)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return p->n_prims;
}
#else
RT_HD int32_t launch_n_prims(const RtLaunch &L) { return L.n_prims; }
#endif

template <int GENERAL, int MEDIUM, class Stack>
RT_HD void trav_begin(const RtLaunch &L, V3 o, V3 d, SegCtx &sc, Trav &tv, Stack &st) {
    const bool near_plane = trav_ray_constants<StackHalfNodes<Stack>::value>(L, o, d, tv);
    tv.best_t = RTL_INF;
    tv.best_prim = 0xFFFFFFFFu;
    const double a = dot(d, d);
    tv.r2a = world_roots_rcp(L, o, a);
    sc.r2a = tv.r2a;
    tv.sp = 0;
#if !defined(RT_NO_CAREFUL)
    // A segment that runs (all but) IN an axis plane is the one kind of ray for which the reference's binary64 boxes are not
    // result-neutral (see ref_box_hit): it does not walk the culling structure at all.  Every leaf prim is tested in turn and a
    // hit counts only if the reference's own boxes above that prim admit the ray; the lane is DONE at once.  About one
    // segment in 1e9 of an ordinary scene; nearly all of an edge-running path in a scene scaled up 1e8-fold.
    // (Inline: as a real call -- on the launch descriptor or on a private copy of it -- the descriptor, a by-value kernel
    // argument, is copied to scratch memory and read from there ever after: Cornell box 138 -> 184 / 201 ms.)
    if (__builtin_expect(near_plane, 0)) {
        const int32_t n = launch_n_prims(L);
        for (int32_t pi = 0; pi < n; ++pi) {
            Rec r;
            if (prim_hit<GENERAL, MEDIUM, false>(L, (uint32_t)pi, o, d, a, sc, &r, false) && r.t < tv.best_t &&
                own_boxes_admit<GENERAL, MEDIUM>(L, (uint32_t)pi, o, d)) {
                tv.best_t = r.t; // ascending prim id, strict <: ties keep the lower id
                tv.best_prim = (uint32_t)pi;
            }
        }
        tv.best32 = up32(tv.best_t);
        tv.cur = Stack::Ref::kDone;
        return;
    }
#endif
    for (int32_t pi = 0; pi < L.n_hoisted; ++pi) {
        Rec r;
        if (prim_hit<GENERAL, MEDIUM, false>(L, (uint32_t)pi, o, d, a, sc, &r, false)) {
            if (r.t < tv.best_t) { // ascending prim id: ties keep the lower id
                tv.best_t = r.t;
                tv.best_prim = (uint32_t)pi;
            }
        }
    }
    tv.best32 = up32(tv.best_t);
    tv.cur = L.root;
}

// one inner-node step (tv.cur is an inner node reference).  `nodes` is the node array (LDS copy or global).
// Slab test of BOTH children: per axis the ray's entry plane and exit plane are read as (child0, child1) pairs through
// the per-ray offsets tv.ox/oy/oz -- 12 fused multiply-adds and two max3 / min3 per node instead of testing all four
// plane combinations and sorting them with min / max (v_min_f32 / v_max_f32 issue at half the rate of v_fma_f32 on
// gfx950, profiles/r02_valu_issue.json).  Branch-free up to the push/pop.
struct F2 {
    float a, b;
};
RT_HD F2 ld_pair(const RtNode *base, uint32_t byte_off) { // base: a node (LDS copy) or the node ARRAY (global: one 32-bit offset per load)
    const float *p = reinterpret_cast<const float *>(reinterpret_cast<const unsigned char *>(base) + byte_off);
    F2 r;
    r.a = p[0];
    r.b = p[1];
    return r;
}
// GLOBAL: the node array is in global memory: every load is `array base (SGPRs) + one 32-bit offset` (seven v_add_u32 instead
// of seven 64-bit v_lshl_add_u64 and six VGPR pairs of offsets: -2 % on the book-two cover); the LDS copy keeps node pointer +
// plane offset (one v_lshl_add_u32 for the node, one v_add_u32 per plane)
// a (child0, child1) pair of binary16 planes of an RtNodeH, widened (exactly) to binary32: on the device the widening happens
// inside the multiply-add that consumes it (v_fma_mix_f32 reads either half of the word as a binary16 operand)
#if defined(__HIP_DEVICE_COMPILE__)
struct H2 {
    _Float16 a, b;
};
RT_HD H2 ld_pair_h(const unsigned char *node, uint32_t byte_off) { return *reinterpret_cast<const H2 *>(node + byte_off); }
#else
RT_HD float half_bits_to_float(uint32_t h) { // IEEE binary16 -> binary32, exact
    const uint32_t sign = (h & 0x8000u) << 16, e = (h >> 10) & 0x1Fu, m = h & 0x3FFu;
    if (e == 0x1Fu) return bits_f32(sign | 0x7F800000u | (m << 13));
    if (e != 0u) return bits_f32(sign | ((e + 112u) << 23) | (m << 13));
    const float sub = (float)m * 0x1p-24f; // subnormal (or zero): m * 2^-24, exact
    return sign ? -sub : sub;
}
typedef F2 H2;
RT_HD H2 ld_pair_h(const unsigned char *node, uint32_t byte_off) {
    const uint16_t *p = reinterpret_cast<const uint16_t *>(node + byte_off);
    H2 r;
    r.a = half_bits_to_float(p[0]);
    r.b = half_bits_to_float(p[1]);
    return r;
}
#endif
template <bool GLOBAL = false, class Stack>
RT_HD void trav_node_step(const RtNode *nodes, Trav &tv, Stack &st) {
    if constexpr (StackHalfNodes<Stack>::value) {
        // RtNodeH (the LDS copy only): the same slab test on the planes' binary16 images, which contain the binary32 boxes
        const unsigned char *N = reinterpret_cast<const unsigned char *>(nodes) + tv.cur * (uint32_t)sizeof(RtNodeH);
        const H2 px = ld_pair_h(N, tv.ox), py = ld_pair_h(N, tv.oy), pz = ld_pair_h(N, tv.oz);
        const H2 qx = ld_pair_h(N, tv.ox ^ 12u), qy = ld_pair_h(N, tv.oy ^ 20u), qz = ld_pair_h(N, tv.oz ^ 28u);
        const uint32_t cw = *reinterpret_cast<const uint32_t *>(N + 24u);
        const uint32_t c0 = cw & 0xFFFFu, c1 = cw >> 16;
        const float tmin0 = fmaxf(fmaxf(fmaf((float)px.a, tv.idx, tv.nx), fmaf((float)py.a, tv.idy, tv.ny)), fmaxf(fmaf((float)pz.a, tv.idz, tv.nz), 0.0f));
        const float tmin1 = fmaxf(fmaxf(fmaf((float)px.b, tv.idx, tv.nx), fmaf((float)py.b, tv.idy, tv.ny)), fmaxf(fmaf((float)pz.b, tv.idz, tv.nz), 0.0f));
        const float tmax0 = fminf(fminf(fmaf((float)qx.a, tv.idx, tv.fx), fmaf((float)qy.a, tv.idy, tv.fy)), fminf(fmaf((float)qz.a, tv.idz, tv.fz), tv.best32));
        const float tmax1 = fminf(fminf(fmaf((float)qx.b, tv.idx, tv.fx), fmaf((float)qy.b, tv.idy, tv.fy)), fminf(fmaf((float)qz.b, tv.idz, tv.fz), tv.best32));
        const bool h0 = tmin0 <= tmax0 * 1.000002f, h1 = tmin1 <= tmax1 * 1.000002f;
        const bool one_first = tmin1 < tmin0;
        const bool both = h0 && h1;
        const uint32_t first = both ? (one_first ? c1 : c0) : (h0 ? c0 : c1);
        const uint32_t second = one_first ? c0 : c1;
        const float second_t = one_first ? tmin0 : tmin1;
        if (both) st.push(tv.sp, second_t, second);
        if (h0 || h1)
            tv.cur = first;
        else
            trav_pop(tv, st);
        return;
    }
    const uint32_t nb = GLOBAL ? rec_off<RtNode>(tv.cur) : 0u;
    const RtNode *N = GLOBAL ? nodes : &nodes[tv.cur];
    const F2 px = ld_pair(N, nb + tv.ox), py = ld_pair(N, nb + tv.oy), pz = ld_pair(N, nb + tv.oz);                   // entry planes
    const F2 qx = ld_pair(N, nb + (tv.ox ^ 24u)), qy = ld_pair(N, nb + (tv.oy ^ 40u)), qz = ld_pair(N, nb + (tv.oz ^ 56u)); // exit planes
    const uint32_t *cp = reinterpret_cast<const uint32_t *>(reinterpret_cast<const unsigned char *>(N) + (nb + 48u)); // RtNode::child
    const uint32_t c0 = cp[0], c1 = cp[1];
    // fminf / fmaxf ignore a NaN operand (0 * inf planes), like the reference's selects: such an axis drops out
    const float tmin0 = fmaxf(fmaxf(fmaf(px.a, tv.idx, tv.nx), fmaf(py.a, tv.idy, tv.ny)), fmaxf(fmaf(pz.a, tv.idz, tv.nz), 0.0f));
    const float tmin1 = fmaxf(fmaxf(fmaf(px.b, tv.idx, tv.nx), fmaf(py.b, tv.idy, tv.ny)), fmaxf(fmaf(pz.b, tv.idz, tv.nz), 0.0f));
    const float tmax0 = fminf(fminf(fmaf(qx.a, tv.idx, tv.fx), fmaf(qy.a, tv.idy, tv.fy)), fminf(fmaf(qz.a, tv.idz, tv.fz), tv.best32));
    const float tmax1 = fminf(fminf(fmaf(qx.b, tv.idx, tv.fx), fmaf(qy.b, tv.idy, tv.fy)), fminf(fmaf(qz.b, tv.idz, tv.fz), tv.best32));
    // slack for the rounding of the slab arithmetic itself
    const bool h0 = tmin0 <= tmax0 * 1.000002f, h1 = tmin1 <= tmax1 * 1.000002f;
    const bool one_first = tmin1 < tmin0;
    const bool both = h0 && h1;
    const uint32_t first = both ? (one_first ? c1 : c0) : (h0 ? c0 : c1);
    const uint32_t second = one_first ? c0 : c1;
    const float second_t = one_first ? tmin0 : tmin1;
    if (both) st.push(tv.sp, second_t, second);
    if (h0 || h1)
        tv.cur = first;
    else
        trav_pop(tv, st);
}

// List mode (small general scenes, RtLaunch::n_list > 0): ONE step tests the culling boxes of all n leaves -- every lane of
// the wave the same box at the same time, no divergence, no stack traffic for misses -- keeps the nearest box in registers
// and pushes the others; the leaf steps that follow test the nearest first and pop the rest against the shrinking best hit,
// exactly as after a tree walk.  Box b is leaf prim first_prim + b; its record is {lo, hi, lo} per axis, so the lane's
// entry plane of an axis sits at byte tv.o? and its exit plane 4 bytes on (two adjacent words: one ds_read2_b32 per axis).
template <class Stack>
RT_HD void trav_list_step(const float *boxes, uint32_t n, uint32_t first_prim, Trav &tv, Stack &st) {
    typedef typename Stack::Ref Ref;
    const unsigned char *base = reinterpret_cast<const unsigned char *>(boxes);
    const float *bx = reinterpret_cast<const float *>(base + tv.ox), *by = reinterpret_cast<const float *>(base + tv.oy),
                *bz = reinterpret_cast<const float *>(base + tv.oz);
    float near_t = 0.0f;
    uint32_t near_ref = Ref::kDone;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 6
#endif
    for (uint32_t b = 0; b < n; ++b) {
        const float tmin = fmaxf(fmaxf(fmaf(bx[0], tv.idx, tv.nx), fmaf(by[0], tv.idy, tv.ny)), fmaxf(fmaf(bz[0], tv.idz, tv.nz), 0.0f));
        const float tmax = fminf(fminf(fmaf(bx[1], tv.idx, tv.fx), fmaf(by[1], tv.idy, tv.fy)), fminf(fmaf(bz[1], tv.idz, tv.fz), tv.best32));
        if (tmin <= tmax * 1.000002f) { // the same test, slack included, as a child box of trav_node_step
            const uint32_t ref = Ref::kLeaf | (first_prim + b);
            if (near_ref == Ref::kDone) {
                near_t = tmin;
                near_ref = ref;
            } else {
                const bool nearer = tmin < near_t;
                st.push(tv.sp, nearer ? near_t : tmin, nearer ? near_ref : ref);
                near_t = nearer ? tmin : near_t;
                near_ref = nearer ? ref : near_ref;
            }
        }
        bx += RT_LIST_BOX_FLOATS;
        by += RT_LIST_BOX_FLOATS;
        bz += RT_LIST_BOX_FLOATS;
    }
    tv.cur = near_ref; // kDone when no box is hit (nothing was pushed then)
}

// The kernel families whose tree leaves may be cube groups (RT_META_GROUP_BIT; rt_host.cpp forms groups for these families only):
// general prims with the tree walk, without and with sphere media / textures.  (Box-LIST scenes and 32-bit references: never.)
template <class S, class = void>
struct StackCubeGroups { // a Stack policy says so itself (kCubeGroups): the 16-bit tree walks do, the list walk and the wide one do not
    static constexpr bool value = false;
};
template <class S>
struct StackCubeGroups<S, decltype((void)S::kCubeGroups)> {
    static constexpr bool value = S::kCubeGroups;
};
template <int GENERAL, int MEDIUM, class Stack>
struct CubeGroups {
    static constexpr bool value = GENERAL >= 1 && MEDIUM <= 1 && StackCubeGroups<Stack>::value; // (GENERAL == 2: the scene's records in LDS)
};

// one leaf step (tv.cur is a leaf reference): binary64 primitive test, then pop.
// A leaf that is the head of a CUBE GROUP stands for six rectangle prims: the culling boxes of the six faces are derived from the
// group's twelve planes (RtCubeGroup) with the slab arithmetic of trav_node_step -- entry faces share the box's entry time, exit
// faces its exit time -- and the faces whose box the ray crosses in [0, best] are tested nearest first, each lane its own next
// candidate in the same pass of the loop; an ordinary leaf is the one-candidate case of the same loop (ONE inlined prim_hit).
template <int GENERAL, int MEDIUM, class Stack>
RT_HD void trav_leaf_step(const RtLaunch &L, V3 o, V3 d, SegCtx &sc, Trav &tv, Stack &st) {
    const uint32_t head = tv.cur & Stack::Ref::kMask;
    const double a = dot(d, d);
    uint32_t cand = 1u;             // bit s: the prim of slot s is still to be tested (an ordinary leaf: slot 0, the prim itself)
    float tn_exit_x = 0.0f, tn_exit_y = 0.0f, tn_exit_z = 0.0f, tn_entry = 0.0f; // where the ray enters each candidate's box
    uint32_t faces = 0u; // three bits per candidate slot (0-2: entry face of x y z, 3-5: exit face): the face's prim is head + that
    if constexpr (CubeGroups<GENERAL, MEDIUM, Stack>::value) {
        const RtPrimMeta &HM = rec_at<GENERAL == 2>(L.prim_meta, head);
        if (HM.kind & RT_META_GROUP_BIT) {
            const RtCubeGroup &C = reinterpret_cast<const RtCubeGroup &>(rec_at<GENERAL == 2>(L.prim_geo, HM.aux));
            const bool sx = (f32_bits(tv.idx) >> 31) != 0u, sy = (f32_bits(tv.idy) >> 31) != 0u, sz = (f32_bits(tv.idz) >> 31) != 0u;
            // per axis: te the ray reaches the entry face's slab, ti leaves it; txi reaches the exit face's slab, tx leaves it
            const float te_x = fmaf(sx ? C.outer_hi[0] : C.outer_lo[0], tv.idx, tv.nx), ti_x = fmaf(sx ? C.inner_hi[0] : C.inner_lo[0], tv.idx, tv.fx);
            const float txi_x = fmaf(sx ? C.inner_lo[0] : C.inner_hi[0], tv.idx, tv.nx), tx_x = fmaf(sx ? C.outer_lo[0] : C.outer_hi[0], tv.idx, tv.fx);
            const float te_y = fmaf(sy ? C.outer_hi[1] : C.outer_lo[1], tv.idy, tv.ny), ti_y = fmaf(sy ? C.inner_hi[1] : C.inner_lo[1], tv.idy, tv.fy);
            const float txi_y = fmaf(sy ? C.inner_lo[1] : C.inner_hi[1], tv.idy, tv.ny), tx_y = fmaf(sy ? C.outer_lo[1] : C.outer_hi[1], tv.idy, tv.fy);
            const float te_z = fmaf(sz ? C.outer_hi[2] : C.outer_lo[2], tv.idz, tv.nz), ti_z = fmaf(sz ? C.inner_hi[2] : C.inner_lo[2], tv.idz, tv.fz);
            const float txi_z = fmaf(sz ? C.inner_lo[2] : C.inner_hi[2], tv.idz, tv.nz), tx_z = fmaf(sz ? C.outer_lo[2] : C.outer_hi[2], tv.idz, tv.fz);
            // (fmaxf / fminf ignore a NaN operand, 0 * inf: such an axis drops out, as in trav_node_step)
            const float t_in = fmaxf(fmaxf(te_x, te_y), fmaxf(te_z, 0.0f));             // every entry face's box begins here
            const float t_out = fminf(fminf(tx_x, tx_y), fminf(tx_z, tv.best32));        // every exit face's box ends here
            const float k = 1.000002f;                                                   // the slack of trav_node_step's comparison
            // entry face of axis a: [te_a, ti_a] on its own axis, the box's interval on the other two
            const bool e_x = t_in <= fminf(fminf(ti_x, tx_y), fminf(tx_z, tv.best32)) * k;
            const bool e_y = t_in <= fminf(fminf(tx_x, ti_y), fminf(tx_z, tv.best32)) * k;
            const bool e_z = t_in <= fminf(fminf(tx_x, tx_y), fminf(ti_z, tv.best32)) * k;
            // exit face of axis a: [txi_a, tx_a] on its own axis
            tn_exit_x = fmaxf(fmaxf(txi_x, te_y), fmaxf(te_z, 0.0f));
            tn_exit_y = fmaxf(fmaxf(te_x, txi_y), fmaxf(te_z, 0.0f));
            tn_exit_z = fmaxf(fmaxf(te_x, te_y), fmaxf(txi_z, 0.0f));
            const bool x_x = tn_exit_x <= t_out * k, x_y = tn_exit_y <= t_out * k, x_z = tn_exit_z <= t_out * k;
            tn_entry = t_in;
            cand = (e_x ? 1u : 0u) | (e_y ? 2u : 0u) | (e_z ? 4u : 0u) | (x_x ? 8u : 0u) | (x_y ? 16u : 0u) | (x_z ? 32u : 0u);
            // the prim of each slot: entry side of an axis = its LOW side for a positive direction (slot 2a), its high side else
            const uint32_t gf = C.faces; // by the box's slots 2 * axis + side
            const uint32_t x_lo = gf & 7u, x_hi = (gf >> 3) & 7u, y_lo = (gf >> 6) & 7u, y_hi = (gf >> 9) & 7u, z_lo = (gf >> 12) & 7u, z_hi = (gf >> 15) & 7u;
            faces = (sx ? x_hi : x_lo) | ((sy ? y_hi : y_lo) << 3) | ((sz ? z_hi : z_lo) << 6) | ((sx ? x_lo : x_hi) << 9) | ((sy ? y_lo : y_hi) << 12) |
                    ((sz ? z_lo : z_hi) << 15);
        }
    }
    while (cand != 0u) {
        uint32_t pi = head;
        if constexpr (CubeGroups<GENERAL, MEDIUM, Stack>::value) {
            // the next candidate of THIS lane: entry faces first (they begin at t_in <= every exit face's entry)
            const uint32_t slot = (uint32_t)__builtin_ctz(cand);
            cand &= cand - 1u;
            const float tn = slot < 3u ? tn_entry : (slot == 3u ? tn_exit_x : (slot == 4u ? tn_exit_y : tn_exit_z));
            if (tn > tv.best32) continue; // behind the best hit so far: the rule of trav_pop
            pi = head + ((faces >> (3u * slot)) & 7u);
        } else {
            cand = 0u;
        }
        Rec r;
        if (prim_hit<GENERAL, MEDIUM, false>(L, pi, o, d, a, sc, &r, false)) {
            // nearest t; exact ties go to the lower prim id, whatever the visiting order
            if (r.t < tv.best_t || (r.t == tv.best_t && pi < tv.best_prim)) {
                tv.best_t = r.t;
                tv.best_prim = pi;
                tv.best32 = up32(r.t);
            }
        }
    }
    trav_pop(tv, st);
}

// ---- Texture::value (src/material.rs:211-215,235-245; examples/main.rs:267-280) ----
RT_HD uint32_t as_u32(double x) { // Rust `as u32`: saturating, NaN -> 0
    if (!(x > 0.0)) return 0u;
    if (x >= 4294967295.0) return 4294967295u;
    return (uint32_t)x;
}
RT_HD V3 texture_value(const RtLaunch &L, uint32_t tex, double u, double v) {
    for (int guard = 0; guard < 64; ++guard) {
        const RtTexture &T = rec_at(L.textures, tex);
        if (T.kind == RT_TEX_SOLID) return ld3(T.rgb);
        if (T.kind == RT_TEX_CHECKER) {
            double sine = checker_sine_cold(u, v);
            tex = sine > 0.0 ? T.a : T.b;
            continue;
        }
        uint32_t px = as_u32(u * (double)T.w);
        uint32_t py = as_u32((1.0 - v) * (double)T.h);
        if (px >= T.w) px = T.w - 1; // the reference would panic here (u == 1.0); clamp, see DESIGN.md
        if (py >= T.h) py = T.h - 1;
        const uint8_t *t = L.image_blob + (uint32_t)(T.data + (py * T.w + px) * 3u); // the blob is < 4 GiB (rt_host.cpp)
        return mk((double)t[0] / 255.0, (double)t[1] / 255.0, (double)t[2] / 255.0);
    }
    return mk(0.0, 0.0, 0.0);
}

// Dielectric::schlickReflectionProbability(theta = acos(c), n1, n2) (src/material.rs:140-143)
// evaluated from c directly: powf(2) -> q*q, cos(acos(c)) -> c, powf(5) by multiplication.  Each substitution moves the
// probability by a few ulp at most, and the value is only ever compared with a uniform draw: in 20 M random (c, index, draw)
// triples the decision never differs from the reference's own evaluation with the host libm's acos / cos / pow
// (tests/test_libm_cpu.py::test_schlick_by_multiplication_decides_like_the_reference); deviation (ii) of docs/parity.md.
// (The reference's evaluation, restated bit for bit, exists -- rtm::acos / cos / pow -- and a draw within 1e-9 of the threshold
// could be handed to it.  Tried in round 4: as a real call its 54 VGPRs push the spheres-only kernel from 120 to 128, inline its
// constants are hoisted into the persistent loop; either costs the headline 1 % for a decision that changes once in 1e15 draws.)
// acos(c) is NaN for |c| > 1, which makes the reference refract: kept.
RT_HD bool schlick_reflects(double c, double n1, double n2, double u) {
    if (!(c >= -1.0 && c <= 1.0)) return false; // u < NaN
    double q = (n1 - n2) / (n1 + n2);
    double r0 = q * q;
    double y = 1.0 - c;
    double y2 = y * y;
    double y5 = y2 * y2 * y;
    return u < r0 + (1.0 - r0) * y5;
}

// Material::scatter + emitted for the hit material (src/material.rs:61-69,99-118,
// 147-192,291-297,318-325).  Returns true if the path continues with (o, d).
// The per-material branches only build the un-normalised direction; the rejection
// sampler and the final normalisation are shared, so a wave with mixed materials
// executes each of them once.  Per lane the operations (and the random draws) are
// exactly those of its own material's scatter().
// `tex` = the value of the material's texture at the hit (albedo / emission)
RT_HD bool shade(const RtMaterial &M, V3 tex, const Rec &rec, V3 d_in, Rng &g, V3 *o_out, V3 *d_out, V3 *att, V3 *emit,
                 int ball_iters = 0, bool *pending = nullptr) {
    *emit = mk(0.0, 0.0, 0.0);
    *o_out = rec.p;
    *att = tex;
    const uint32_t kind = M.kind;
    if (kind == RT_MAT_DIFFUSE_LIGHT) {
        *emit = tex;
        return false;
    }
    const double dn = dot(d_in, rec.n);
    if (kind == RT_MAT_METAL && !(dn < 0.0)) return false; // hit from behind: absorbed (quirk Q6)

    // d.normalized() for the specular materials
    V3 uvn = d_in;
    if (kind == RT_MAT_METAL || kind == RT_MAT_DIELECTRIC) uvn = normalized(d_in);
    // randomInUnitSphere() for the diffuse ones and fuzzy metal
    V3 ball = mk(0.0, 0.0, 0.0);
    // (the sampler is the first consumer of the stream in these scatter()s, so a bounded call can be resumed)
    bool ball_ok = true;
    if (kind == RT_MAT_LAMBERTIAN || kind == RT_MAT_ISOTROPIC || (kind == RT_MAT_METAL && M.param != 0.0))
        ball = random_in_unit_sphere_bounded(g, ball_iters, &ball_ok);
    if (!ball_ok) {
        *pending = true; // nothing but the stream has changed: shade this hit again later
        return true;
    }

    V3 v = ball; // Isotropic: randomInUnitSphere().normalized()
    bool norm = true;
    if (kind == RT_MAT_LAMBERTIAN) {
        v = rec.n + ball;
    } else if (kind == RT_MAT_METAL) {
        V3 refl = reflected(uvn, rec.n);
        if (M.param == 0.0) {
            v = refl;
            norm = false;
        } else {
            v = refl + ball * M.param;
        }
    } else if (kind == RT_MAT_DIELECTRIC) {
        *att = mk(1.0, 1.0, 1.0);
        double ratio;
        V3 normal = rec.n;
        if (dn < 0.0) {
            ratio = 1.0 / M.param;
        } else {
            ratio = M.param;
            normal = -normal;
        }
        // Vec3::refracted, src/vec3.rs:113-124 (built from the UN-normalised direction)
        double dt = dot(uvn, normal);
        double disc = 1.0 - ratio * ratio * (1.0 - dt * dt);
        if (disc > 0.0) {
            V3 refr = (d_in - normal * dt) * ratio - normal * sqrt(disc);
            double u = rng_range01(g);
            if (schlick_reflects(-dot(d_in, normal), ratio, 1.0, u)) {
                v = reflected(d_in, rec.n); // about the un-flipped normal (quirk Q7)
                norm = false;
            } else {
                v = refr;
            }
        } else {
            v = reflected(d_in, normal); // total internal reflection: no draw
            norm = false;
        }
    }
    V3 vn = normalized(v);
    // component-wise selects: a struct-valued ?: is lowered to a round trip through scratch memory
    d_out->x = norm ? vn.x : v.x;
    d_out->y = norm ? vn.y : v.y;
    d_out->z = norm ? vn.z : v.z;
    return true;
}

// uv of the hit on primitive `pi` at parameter t, on its own: the record the shading needs is built WITHOUT uv, and a hit
// whose texture reads them (checker / image) gets them here first, then its Texture::value -- the atan2 / acos / sin
// evaluations then run while only the path state is live, not the record and the scatter temporaries as well (the
// kernel with media and textures drops from 225 to below 168 VGPRs: 3 waves per SIMD instead of 2).  Same arithmetic
// as the record: p = o' + d' t in the primitive's frame, unitSphereUv(p / r) (src/geometry.rs:66-70) or the
// rectangle's ((x + w/2) / w, (y + h/2) / h) (src/geometry.rs:170-174); a medium's uv are the sums over its boundary
// hits (src/volume.rs:64-66) and come out of the full medium test.
template <int GENERAL, int MEDIUM>
RT_HD void prim_uv(const RtLaunch &L, uint32_t pi, V3 o, V3 d, double t, SegCtx &sc, double *u, double *v) {
    const RtPrimGeo &G = rec_at<GENERAL == 2>(L.prim_geo, pi);
    *u = 0.0;
    *v = 0.0;
    uint32_t kw = (uint32_t)RT_PRIM_SPHERE_T;
    if (GENERAL || MEDIUM) kw = rec_at<GENERAL == 2>(L.prim_meta, pi).kind;
    const uint32_t kind = kw & 0xFFu;
    if (kind == RT_PRIM_SPHERE_T) {
        const V3 oc = o - mk(G.g[0], G.g[1], G.g[2]);
        sphere_uv((oc + d * t) / G.g[3], u, v);
        return;
    }
    if (GENERAL || MEDIUM) {
        if (kind == RT_PRIM_MEDIUM_T || kind == RT_PRIM_MEDIUM_C) {
            if (MEDIUM >= 2) { // scenes with a textured medium are served by the general-media kernel
                Rec r;
                if (prim_hit<GENERAL, MEDIUM, true>(L, pi, o, d, dot(d, d), sc, &r, true)) {
                    *u = r.u;
                    *v = r.v;
                }
            }
            return;
        }
        const RtPrimMeta &P = rec_at<GENERAL == 2>(L.prim_meta, pi);
        V3 lo = o, ld = d;
        chain_down<(MEDIUM >= 2), GENERAL == 2>(L, P.xform, (kw >> RT_META_CHAIN_SHIFT) & 0xFu, (kw >> RT_META_TMASK_SHIFT) & 0xFu, &lo, &ld);
        if (kind == RT_PRIM_SPHERE_C) {
            sphere_uv((lo + ld * t) / G.g[0], u, v);
        } else {
            Rec r;
            if (rect_hit(lo, ld, G.g[0], G.g[1], &r)) {
                *u = r.u;
                *v = r.v;
            }
        }
    }
}

// ---------------------------------------------------------------- path state
struct PathState {
    // Only DiffuseLight emits and it never scatters (src/material.rs:17-20,291-297), so the
    // iterative sum L = sum_k T_k * e_k has at most one non-zero term, the last one:
    // the sample's radiance is T * emitted at the terminal hit, else 0.
    V3 o, d, T;
    Rng g;
    int32_t k; // segments traced so far
};

// examples/book-one.rs:69-73: stream, jitter, camera ray
// LaunchT: RtLaunch, or the kernel's RtSampleSetup (the same fields, re-read from the kernel-argument segment per refill)
template <bool LENS, class LaunchT>
RT_HD void start_sample(const LaunchT &L, uint32_t x, uint32_t y, uint32_t s, PathState *ps) {
    const uint64_t pixel = (uint64_t)y * (uint64_t)L.width + (uint64_t)x;
    const uint64_t stream = pixel * (uint64_t)L.spp + (uint64_t)s;
    ps->g.base = L.seed_mix + (stream << RT_RNG_STREAM_SHIFT) * RT_RNG_GAMMA;
    rt_rng_seed_state(ps->g.base, &ps->g.s0, &ps->g.s1);
    ps->g.draws = 0;
    double u = ((double)x + rng_range01(ps->g)) / (double)L.width;
    double v = ((double)y + rng_range01(ps->g)) / (double)L.height;
    camera_ray<LENS>(L.cam, u, v, ps->g, &ps->o, &ps->d);
    ps->T = mk(1.0, 1.0, 1.0);
    ps->k = 0;
}

template <int GENERAL, int MEDIUM, bool TEXTURED, class Stack>
RT_HD void begin_segment(const RtLaunch &L, PathState *ps, Trav &tv, Stack &st, unsigned long long *prims_tested) {
    SegCtx sc;
    sc.r2a = RTL_NAN; // set by trav_begin
    sc.rng_base = ps->g.base;
    sc.segment = (uint32_t)ps->k;
    sc.draws = 0;
    sc.prims_tested = 0;
    trav_begin<GENERAL, MEDIUM>(L, ps->o, ps->d, sc, tv, st);
    ps->g.draws += sc.draws;
    *prims_tested += sc.prims_tested;
}

template <int GENERAL, int MEDIUM, bool TEXTURED, class Stack>
RT_HD void leaf_step(const RtLaunch &L, PathState *ps, Trav &tv, Stack &st, unsigned long long *prims_tested) {
    SegCtx sc;
    sc.r2a = tv.r2a;
    sc.rng_base = ps->g.base;
    sc.segment = (uint32_t)ps->k;
    sc.draws = 0;
    sc.prims_tested = 0;
    trav_leaf_step<GENERAL, MEDIUM>(L, ps->o, ps->d, sc, tv, st);
    ps->g.draws += sc.draws;
    *prims_tested += sc.prims_tested;
}

// The traversal of this segment is finished (tv.cur == RT_CUR_DONE): shade it.
// render::color, src/render.rs:5-29, in its iterative form
// L = sum_k (prod_{j<k} att_j) * e_k.  Returns true when the sample is finished, with its
// radiance in *radiance.
// ball_iters > 0 bounds the rejection sampler of this call; when the bound is hit *pending is set, nothing but the
// stream has advanced and the same hit has to be finished again (the kernel does so in its next shade block).
template <int GENERAL, int MEDIUM, bool TEXTURED>
RT_HD bool finish_segment(const RtLaunch &L, PathState *ps, const Trav &tv, V3 *radiance, int ball_iters = 0, bool *pending = nullptr) {
    *radiance = mk(0.0, 0.0, 0.0);
    if (tv.best_prim == 0xFFFFFFFFu) return true; // background is black (src/render.rs:21-28)
    const uint32_t prim = tv.best_prim;
    const uint32_t mat = (GENERAL || MEDIUM) ? rec_at<GENERAL == 2>(L.prim_meta, prim).material : L.prim_meta[prim].material;
    if (mat == RT_NO_MATERIAL) return true; // src/render.rs:18-20
    const RtMaterial &M = (GENERAL || MEDIUM) ? rec_at<GENERAL == 2>(L.materials, mat) : L.materials[mat];
    SegCtx sc;
    sc.r2a = RTL_NAN; // the record is built at the known t: no roots
    sc.rng_base = ps->g.base;
    sc.segment = (uint32_t)ps->k;
    sc.draws = 0;
    sc.prims_tested = 0;
    // Texture::value first (src/material.rs:196-198); uv feed nothing else, so they are evaluated only when the
    // material's texture is not a solid colour -- the reference computes atan2 / acos on every sphere hit
    V3 tex = ld3(M.rgb);
    if (TEXTURED && !M.solid) {
        double u, v;
        prim_uv<GENERAL, MEDIUM>(L, prim, ps->o, ps->d, tv.best_t, sc, &u, &v);
        tex = texture_value(L, M.tex, u, v);
    }
    Rec rec;
    if (GENERAL || MEDIUM) {
        // rebuild the record of the winner (same arithmetic, same t, same keyed draws)
        prim_record<GENERAL, MEDIUM>(L, prim, ps->o, ps->d, tv.best_t, sc, &rec);
    } else {
        const RtPrimGeo &G = L.prim_geo[prim];
        V3 c = mk(G.g[0], G.g[1], G.g[2]);
        sphere_finish(ps->o - c, ps->d, G.g[3], tv.best_t, false, &rec);
        rec.p = rec.p + c;
    }
    V3 o2, d2, att, emit;
    bool pend = false;
    const bool cont = shade(M, tex, rec, ps->d, ps->g, &o2, &d2, &att, &emit, ball_iters, &pend);
    if (pend) {
        *pending = true;
        return false;
    }
    if (M.kind == RT_MAT_DIFFUSE_LIGHT) *radiance = ps->T * emit;
    if (!cont) return true;
    ps->T = ps->T * att;
    ps->o = o2;
    ps->d = d2;
    ps->k += 1;
    return ps->k >= L.max_depth; // color(.., 0) returns black (src/render.rs:6-8)
}

} // namespace rtl

#endif
