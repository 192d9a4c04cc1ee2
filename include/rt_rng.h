/*
 * rt_rng.h -- the ONE definition of the injected random-number contract.
 *
 * Shared verbatim by the CPU oracle (oracle/), the host scene builders and the
 * HIP kernels (ray-tracer_amd/csrc).  Nothing else in the build may define a
 * generator of its own.
 *
 * Why it exists: the reference draws every random number from the `rand` crate's
 * OS-seeded, unseedable `thread_rng()` (reference src/util.rs:2-4,10,28-34,
 * src/material.rs:6-7,164, src/volume.rs:58-60,80-82, examples/book-one.rs:70-72)
 * with `rand = "*"` and no lockfile (Cargo.toml:9-13), so the reference can never
 * reproduce its own image.  "Identical RNG seeds" therefore means: oracle and
 * kernel consume THIS generator, in the program order documented below.
 *
 * Generator: one xoroshiro128+ stream (Blackman & Vigna 2018; a=24, b=16, c=37) per
 * sample, seeded -- as its authors recommend -- by SplitMix64 (Steele, Lea, Flood 2014):
 *       base   = mix64(seed) + (stream << 24) * GAMMA          (injective in `stream`)
 *       s0, s1 = mix64(base + GAMMA), mix64(base + 2*GAMMA)    (SplitMix64 outputs 1, 2 of `base`)
 *       draw   : r = s0 + s1;  s1 ^= s0;  s0 = rotl(s0,24) ^ s1 ^ (s1 << 16);  s1 = rotl(s1,37)
 *   Only the top 53 / 52 bits of a draw are used (the weak low bits of the + scrambler
 *   never are).  xoroshiro needs no multiplies: 64-bit multiplies are quarter-rate on the
 *   CDNA4 VALU and the rejection samplers draw ~6 numbers per diffuse bounce.
 *   "Keyed" draws (ConstantMedium) stay counter-based: mix64(base + n * GAMMA), n >= 2^23.
 *
 * Float conversions restate rand 0.7.x (the newest series that still has the
 * two-argument `gen_range(low, high)` the reference calls):
 *   random::<f64>()         -> (x >> 11) * 2^-53                 [Standard, 53 bit]
 *   gen_range(lo, hi)       -> v12 * (hi-lo) + (lo - (hi-lo)),   v12 = 1.mantissa(x >> 12)
 *                              retried while the result is >= hi  [UniformFloat::sample_single]
 *
 * Stream id of a sample: pixel_index * spp + s, pixel_index = y * W + x (y up).
 * Draw order inside a sample (SURVEY.md section 8(a) "RNG order"):
 *   U1 (x jitter), U2 (y jitter)                    examples/book-one.rs:71-72
 *   lens disk trials, 2 draws each, if lens != 0    src/util.rs:30-41
 *   per segment: the hit material's draws           src/material.rs:61-69,99-118,147-192,318-325
 *     Lambertian / Isotropic / fuzzy Metal: ball trials, 3 draws each (src/util.rs:6-15)
 *     Dielectric: 1 draw iff refraction is possible
 * ConstantMedium free-flight draws are NOT taken from the sequential counter
 * (in the reference they depend on the random tree's visiting order,
 * src/volume.rs:58-63 + src/optimize.rs:473-488, i.e. they are unpinned); they
 * are keyed by (stream, segment index, medium slot) -> rt_rng_keyed().
 */
#ifndef RT_RNG_H
#define RT_RNG_H

#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define RT_HD __host__ __device__ inline __attribute__((always_inline))
#else
#define RT_HD static inline
#endif

/* RT_DEVICE_MATH selects the bit-level forms the kernels compute with (funnel-shift rotations, numbers assembled from their
 * bits; rt_lane.h adds the shared-reciprocal divisions).  It is on for device code.  A HOST build may define
 * RT_EMULATE_DEVICE_MATH to compile the very same forms with portable stand-ins for the intrinsics, so that the CPU suite
 * executes them too (tests/lane_emul.cpp is built both ways; tests/test_lane_parity_cpu.py compares both with the oracle). */
#if defined(__HIP_DEVICE_COMPILE__)
#define RT_DEVICE_MATH 1
#define RT_ALIGNBIT(a, b, s) __builtin_amdgcn_alignbit((a), (b), (s))
#define RT_HILO_TO_DOUBLE(hi, lo) __hiloint2double((int)(hi), (int)(lo))
#define RT_DOUBLE_HI(x) ((uint32_t)__double2hiint(x))
#elif defined(RT_EMULATE_DEVICE_MATH)
#define RT_DEVICE_MATH 1
static inline uint32_t rt_emul_alignbit(uint32_t a, uint32_t b, uint32_t s) { /* v_alignbit_b32: low word of {a, b} >> s[4:0] */
    return (uint32_t)(((((uint64_t)a) << 32) | (uint64_t)b) >> (s & 31u));
}
static inline double rt_emul_hilo_to_double(uint32_t hi, uint32_t lo) {
    const uint64_t b = ((uint64_t)hi << 32) | (uint64_t)lo;
    double d;
    memcpy(&d, &b, sizeof d);
    return d;
}
static inline uint32_t rt_emul_double_hi(double x) {
    uint64_t b;
    memcpy(&b, &x, sizeof b);
    return (uint32_t)(b >> 32);
}
#define RT_ALIGNBIT(a, b, s) rt_emul_alignbit((a), (b), (s))
#define RT_HILO_TO_DOUBLE(hi, lo) rt_emul_hilo_to_double((uint32_t)(hi), (uint32_t)(lo))
#define RT_DOUBLE_HI(x) rt_emul_double_hi(x)
#endif

#define RT_RNG_GAMMA 0x9E3779B97F4A7C15ull
#define RT_RNG_STREAM_SHIFT 24
#define RT_RNG_KEYED_BASE (1ull << 23)
/* stream id reserved for host-side scene construction (never a sample id) */
#define RT_RNG_SCENE_STREAM ((1ull << 40) - 1ull)

typedef struct rt_rng {
    uint64_t base;   /* mix64(seed) + (stream << 24) * GAMMA: key of the stream (keyed draws) */
    uint64_t s0, s1; /* xoroshiro128+ state */
} rt_rng;

RT_HD uint64_t rt_mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

RT_HD uint64_t rt_rng_base(uint64_t seed, uint64_t stream) {
    return rt_mix64(seed) + (stream << RT_RNG_STREAM_SHIFT) * RT_RNG_GAMMA;
}

#if defined(RT_DEVICE_MATH)
/* two funnel shifts (v_alignbit_b32: the low word of {a, b} >> s) instead of 64-bit shifts and an or */
RT_HD uint64_t rt_rotl64(uint64_t x, int k) {
    uint32_t hi = (uint32_t)(x >> 32), lo = (uint32_t)x;
    if (k >= 32) { /* constant at every call site */
        const uint32_t t = hi;
        hi = lo;
        lo = t;
        k -= 32;
    }
    if (k == 0) return ((uint64_t)hi << 32) | lo;
    const uint32_t nh = RT_ALIGNBIT(hi, lo, (uint32_t)(32 - k)), nl = RT_ALIGNBIT(lo, hi, (uint32_t)(32 - k));
    return ((uint64_t)nh << 32) | nl;
}
#else
RT_HD uint64_t rt_rotl64(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
#endif

/* seed the xoroshiro128+ state from the stream key */
RT_HD void rt_rng_seed_state(uint64_t base, uint64_t *s0, uint64_t *s1) {
    *s0 = rt_mix64(base + RT_RNG_GAMMA);
    *s1 = rt_mix64(base + 2ull * RT_RNG_GAMMA);
    if ((*s0 | *s1) == 0ull) *s0 = RT_RNG_GAMMA; /* the all-zero state is the one fixed point */
}

RT_HD void rt_rng_init(rt_rng *g, uint64_t seed, uint64_t stream) {
    g->base = rt_rng_base(seed, stream);
    rt_rng_seed_state(g->base, &g->s0, &g->s1);
}

/* one xoroshiro128+ step on explicit state words */
RT_HD uint64_t rt_xoroshiro_next(uint64_t *s0p, uint64_t *s1p) {
    const uint64_t s0 = *s0p;
    uint64_t s1 = *s1p;
    const uint64_t r = s0 + s1;
    s1 ^= s0;
    *s0p = rt_rotl64(s0, 24) ^ s1 ^ (s1 << 16);
    *s1p = rt_rotl64(s1, 37);
    return r;
}

/* next sequential 64-bit draw */
RT_HD uint64_t rt_rng_next(rt_rng *g) { return rt_xoroshiro_next(&g->s0, &g->s1); }

/* keyed draw: independent of how many sequential draws were made */
RT_HD uint64_t rt_rng_keyed_from_base(uint64_t base, uint32_t segment, uint32_t slot) {
    uint64_t n = RT_RNG_KEYED_BASE + ((uint64_t)(segment & 0xFFFu) << 10) + (uint64_t)(slot & 0x3FFu);
    return rt_mix64(base + n * RT_RNG_GAMMA);
}

/* ---- keys of the free-flight draws of ConstantMedium::hit (src/volume.rs:58-60, 80-82) ----
 * The reference draws from the thread's sequential generator in the order its (random) tree visits the media: unpinned
 * (quirk Q10).  Here every evaluation of a medium on a segment owns a keyed draw, independent of traversal order:
 *   - a medium sprite of the world's own list (Sprite<ConstantMedium<..>> directly, the reference's own scenes): key = its
 *     creation-order slot among such sprites, 0 .. RT_MEDIUM_SLOT_MAX - 1, drawn by rt_rng_keyed_from_base (round-1 streams);
 *   - any other medium (inside a node used as a geometry, behind a TransformedGeometry): a 32-bit key >= 2^31 from the path
 *     of sprites that leads to it,  h = rank(top) + 1;  h = h * RT_RNG_PATH_MUL + index(child in its node) + 1  per level
 *     (rank = position among the world's own sprites, index = position in the node's list: independent of the order a front
 *     end happens to create sprites in), so every instance draws its own numbers;
 *   - a medium inside the BOUNDARY of another medium (ConstantMedium<T: Hit> with T containing a ConstantMedium,
 *     src/volume.rs:18-44) is evaluated up to twice per evaluation of the outer one (boundary.hit of the ray, then of the
 *     restarted ray): key = rt_medium_key_inner(outer key, pass 0 / 1, its own path key).
 * Wide keys draw from the upper quarter of the stream's keyed window (rt_rng_keyed_wide). */
#define RT_RNG_PATH_MUL 1000003ull
#define RT_MEDIUM_SLOT_MAX 0x3FFu /* slots 0 .. 0x3FE; 0x3FF marks "no slot" */
#define RT_MEDIUM_KEY_WIDE 0x400u /* keys from here on are path keys */
RT_HD uint32_t rt_medium_key_path(uint64_t path_hash) { return (uint32_t)rt_mix64(path_hash) | 0x80000000u; }
RT_HD uint32_t rt_medium_key_inner(uint32_t outer, uint32_t pass, uint32_t own) {
    return (uint32_t)rt_mix64((((uint64_t)outer << 32) | (uint64_t)own) * RT_RNG_PATH_MUL + (uint64_t)pass + 1ull) | 0x80000000u;
}
RT_HD uint64_t rt_rng_keyed_wide(uint64_t base, uint32_t segment, uint32_t key) {
    const uint64_t n = RT_RNG_KEYED_BASE + (1ull << 22) + (rt_mix64(((uint64_t)segment << 32) | (uint64_t)key) & 0x3FFFFFull);
    return rt_mix64(base + n * RT_RNG_GAMMA);
}
RT_HD uint64_t rt_rng_medium_draw(uint64_t base, uint32_t segment, uint32_t key) {
    return key < RT_MEDIUM_KEY_WIDE ? rt_rng_keyed_from_base(base, segment, key) : rt_rng_keyed_wide(base, segment, key);
}

RT_HD double rt_bits_to_double(uint64_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __longlong_as_double((long long)b);
#else
    double d;
    memcpy(&d, &b, sizeof d);
    return d;
#endif
}

/* rand 0.7 Standard for f64: 53 random bits, [0,1) */
RT_HD double rt_u64_to_unit53(uint64_t x) {
    return (double)(x >> 11) * (1.0 / 9007199254740992.0);
}

/* `random::<f64>() * 2.0 - 1.0`, a coordinate of randomInUnitSphere's candidate point (src/util.rs:8-12), exactly: with
 * k = x >> 11 < 2^53 the reference's value is fl(fl(fl(k * 2^-53) * 2) - 1) = k * 2^-52 - 1 with NO rounding anywhere (a 53-bit
 * integer scaled by a power of two; doubling; and k * 2^-52 - 1 is a multiple of 2^-52 below 1 in magnitude).  The device
 * builds the same number from bits: 1.m = 1 + (k mod 2^52) * 2^-52, minus 1 when bit 52 of k is set, minus 2 when it is not --
 * both differences are exact (Sterbenz) -- instead of a 64-bit integer conversion, two multiplications and a subtraction. */
RT_HD double rt_u64_to_pm1(uint64_t x) {
#if defined(RT_DEVICE_MATH)
    const uint32_t hi = (uint32_t)(x >> 32), lo = (uint32_t)x;
    const uint32_t mhi = 0x3FF00000u | ((hi >> 11) & 0xFFFFFu), mlo = (hi << 21) | (lo >> 11);
    const double one_m = RT_HILO_TO_DOUBLE(mhi, mlo);
    return one_m - (((int32_t)hi < 0) ? 1.0 : 2.0);
#else
    return rt_u64_to_unit53(x) * 2.0 - 1.0;
#endif
}

/* 1.mantissa in [1,2), 52 random bits (rand 0.7 into_float_with_exponent(0)) */
RT_HD double rt_u64_to_v12(uint64_t x) {
    return rt_bits_to_double((x >> 12) | 0x3FF0000000000000ull);
}

/* gen_range(0.0, 1.0): scale = 1, offset = -1; always < 1 */
RT_HD double rt_u64_to_range01(uint64_t x) {
    return rt_u64_to_v12(x) - 1.0;
}

/* gen_range(-1.0, 1.0): scale = 2, offset = -3; exact, always < 1 */
RT_HD double rt_u64_to_range11(uint64_t x) {
    return rt_u64_to_v12(x) * 2.0 - 3.0;
}

RT_HD double rt_rng_unit53(rt_rng *g) { return rt_u64_to_unit53(rt_rng_next(g)); }
RT_HD double rt_rng_range01(rt_rng *g) { return rt_u64_to_range01(rt_rng_next(g)); }
RT_HD double rt_rng_range11(rt_rng *g) { return rt_u64_to_range11(rt_rng_next(g)); }

#if !defined(__HIP_DEVICE_COMPILE__)
/* general gen_range(low, high) with the rejection step of
 * UniformFloat::sample_single (host-side scene construction only) */
static inline double rt_rng_gen_range(rt_rng *g, double low, double high) {
    double scale = high - low;
    double offset = low - scale;
    for (;;) {
        double v12 = rt_u64_to_v12(rt_rng_next(g));
        volatile double prod = v12 * scale; /* no fused multiply-add */
        double res = prod + offset;
        if (res < high) return res;
    }
}
/* gen_range(0, 3) on integers (BVH split axis, src/optimize.rs:374-375): the
 * oracle only needs *a* seeded axis stream; widening multiply, top bits. */
static inline uint32_t rt_rng_gen_below(rt_rng *g, uint32_t n) {
    uint64_t x = rt_rng_next(g) >> 32;
    return (uint32_t)((x * (uint64_t)n) >> 32);
}
#endif

#endif /* RT_RNG_H */
