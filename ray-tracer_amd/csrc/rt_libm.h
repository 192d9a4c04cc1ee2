// rt_libm.h -- binary64 log / sin / cos / acos / atan2 / pow that return the bits of the HOST's libm.
//
// The reference takes its transcendentals from the platform libm (`f64::ln`, `sin`, `acos`, `atan2`:
// src/volume.rs:59-60,81-82, src/geometry.rs:35-39, src/material.rs:238) -- on the machines this library runs beside,
// glibc 2.35 for x86-64, whose ifunc resolvers pick the FMA builds of e_log.c, s_sin.c, e_asin.c and e_atan2.c on every
// CPU with FMA + AVX2.  The device's own math library differs from those by an ulp on 3-27 % of arguments, which was the
// only source of GPU != oracle pixels left.  The four functions below restate glibc's algorithms operation
// for operation -- every rounding, including WHICH products the FMA builds fuse into their sums (that is not in glibc's
// source: the compiler chose it; it was read off the installed library) -- so that they return the same bits.  Lookup
// tables: rt_libm_tables.h (data of that library, tools/libm/extract_glibc_tables.py).
//
// Compile with -ffp-contract=off: a fusion the text does not spell out changes a result.  Every `fma_` below is a fused
// operation of the original; every separate `*` followed by `+` is a separate rounding of the original.
// tests/test_libm_cpu.py compares the host compilation of this header with glibc itself on tens of millions of arguments per
// function (0 differing results), tests/test_gpu_parity.py the device compilation with the same.
//
// Not reproduced: errno, floating-point exception flags, and the sign / payload of a NaN result (x86 and gfx950 produce
// different default NaNs); a NaN comes back wherever glibc returns one.
// ---------------------------------------------------------------------------------------------------------------------------
// LICENCE OF THIS FILE.  This file is a derived work of the GNU C Library 2.35 (sysdeps/ieee754/dbl-64/: e_log.c, e_pow.c,
// e_exp_data.c / e_pow_log_data.c / e_log_data.c, s_sin.c, dosincos.c, sincostab.c, e_asin.c, e_atan2.c, s_atan.c and their
// headers), whose algorithms it restates and whose tables it carries:
//   Copyright (C) 1996-2022 Free Software Foundation, Inc.
//   e_log.c, e_pow.c and their data: derived from the ARM Optimized Routines, Copyright (c) 2018, Arm Limited (contributed to
//     glibc under the LGPL; upstream dual-licensed MIT OR Apache-2.0 WITH LLVM-exception).
//   s_sin.c, e_asin.c, e_atan2.c, s_atan.c and their tables: IBM Accurate Mathematical Library, written by International
//     Business Machines Corp., Copyright (C) 2001-2022 Free Software Foundation, Inc.
// The GNU C Library is free software; you can redistribute it and/or modify it under the terms of the GNU Lesser General Public
// License as published by the Free Software Foundation; either version 2.1 of the License, or (at your option) any later
// version.  It is distributed in the hope that it will be useful, but WITHOUT ANY WARRANTY; without even the implied warranty
// of MERCHANTABILITY or FITNESS FOR A PARTICULAR PURPOSE.  See the GNU Lesser General Public License for more details
// (LICENSES/LGPL-2.1.txt at the root of this repository; NOTICE there says what that means for librt_mi355x.so).
// ---------------------------------------------------------------------------------------------------------------------------
#ifndef RT_LIBM_H
#define RT_LIBM_H

#include <stdint.h>
#include <string.h>

#include "rt_libm_tables.h"

#if defined(__HIPCC__)
#define RTM_FN __host__ __device__ static inline
#else
#define RTM_FN static inline
#endif

namespace rtm {

RTM_FN uint64_t bits(double x) {
    uint64_t u;
    memcpy(&u, &x, sizeof u);
    return u;
}
RTM_FN double from_bits(uint64_t u) {
    double x;
    memcpy(&x, &u, sizeof x);
    return x;
}
RTM_FN double fma_(double a, double b, double c) { return __builtin_fma(a, b, c); }
RTM_FN double abs_(double x) { return from_bits(bits(x) & 0x7FFFFFFFFFFFFFFFull); }
RTM_FN double with_sign_of(double mag, double s) { // |mag| with s's sign bit
    return from_bits((bits(mag) & 0x7FFFFFFFFFFFFFFFull) | (bits(s) & 0x8000000000000000ull));
}
RTM_FN uint32_t hi32(double x) { return (uint32_t)(bits(x) >> 32); }
RTM_FN uint32_t lo32(double x) { return (uint32_t)bits(x); }
RTM_FN double nan_() { return from_bits(0x7FF8000000000000ull); }

// ---------------------------------------------------------------------------------------------------------------------
// log: glibc 2.35 sysdeps/ieee754/dbl-64/e_log.c (the table-driven algorithm of ARM's optimized routines), __log_fma
// ---------------------------------------------------------------------------------------------------------------------
// Tab: where the {invc, logc} table is read from -- rtm_log_tab, or a copy of it (the kernels with media keep one in LDS)
template <class Tab>
RTM_FN double log_from(double x, Tab tab) {
    const double Ln2hi = 0x1.62e42fefa3800p-1, Ln2lo = 0x1.ef35793c76730p-45;
    const double A0 = -0x1.0000000000001p-1, A1 = 0x1.555555551305bp-2, A2 = -0x1.fffffffeb4590p-3,
                 A3 = 0x1.999b324f10111p-3, A4 = -0x1.55575e506c89fp-3;
    uint64_t ix = bits(x);
    const uint32_t top = (uint32_t)(ix >> 48);
    if (ix - 0x3FEE000000000000ull < 0x0003090000000000ull) { // 1 - 2^-4 <= x < 1 + 0x1.09p-4
        const double B0 = -0x1p-1, B1 = 0x1.5555555555577p-2, B2 = -0x1.ffffffffffdcbp-3, B3 = 0x1.999999995dd0cp-3,
                     B4 = -0x1.55555556745a7p-3, B5 = 0x1.24924a344de30p-3, B6 = -0x1.fffffa4423d65p-4,
                     B7 = 0x1.c7184282ad6cap-4, B8 = -0x1.999eb43b068ffp-4, B9 = 0x1.78182f7afd085p-4,
                     B10 = -0x1.5521375d145cdp-4;
        if (ix == 0x3FF0000000000000ull) return 0.0;
        const double r = x - 1.0;
        const double r2 = r * r;
        const double r3 = r * r2;
        const double p1 = fma_(r2, B3, fma_(r, B2, B1));
        const double p4 = fma_(r2, B6, fma_(r, B5, B4));
        double p7 = fma_(r2, B9, fma_(r, B8, B7));
        p7 = fma_(r3, B10, p7);
        double p = fma_(p7, r3, p4);
        p = fma_(p, r3, p1);
        const double rw = fma_(r, 0x1p27, r);     // r + w, w = r * 2^27
        const double rhi = fma_(-0x1p27, r, rw);  // (r + w) - w
        const double rlo = r - rhi;
        const double sq = rhi * rhi;
        const double hi = fma_(sq, B0, r);           // r + rhi * rhi * B0
        double lo = fma_(sq, B0, r - hi);            // r - hi + w
        lo = fma_(B0 * rlo, r + rhi, lo);            // lo += B0 * rlo * (rhi + r)
        const double y = fma_(p, r3, lo);
        return hi + y;
    }
    if (top - 0x0010u >= 0x7FF0u - 0x0010u) { // x < 2^-1022, infinite or NaN
        if (ix * 2 == 0) return -__builtin_huge_val();
        if (ix == 0x7FF0000000000000ull) return x;
        if ((top & 0x8000u) || (top & 0x7FF0u) == 0x7FF0u) return nan_();
        ix = bits(x * 0x1p52); // subnormal: normalise
        ix -= 52ull << 52;
    }
    const uint64_t tmp = ix - 0x3FE6000000000000ull;
    const uint32_t i = (uint32_t)(tmp >> 45) & 127u;
    const int32_t k = (int32_t)((int64_t)tmp >> 52);
    const uint64_t iz = ix - (tmp & 0xFFF0000000000000ull);
    const double invc = tab[2 * i], logc = tab[2 * i + 1];
    const double z = from_bits(iz);
    const double r = fma_(z, invc, -1.0);
    const double kd = (double)k;
    const double w = fma_(kd, Ln2hi, logc);
    const double hi = w + r;
    double lo = (w - hi) + r;
    lo = fma_(kd, Ln2lo, lo);
    const double r2 = r * r;
    const double rr2 = r * r2;
    const double q12 = fma_(r, A2, A1);
    const double q34 = fma_(r, A4, A3);
    const double lo2 = fma_(r2, A0, lo);
    const double q = fma_(q34, r2, q12);
    const double y = fma_(rr2, q, lo2);
    return y + hi;
}
RTM_FN double log(double x) { return log_from(x, rtm_log_tab); }

// ---------------------------------------------------------------------------------------------------------------------
// sin: glibc 2.35 sysdeps/ieee754/dbl-64/s_sin.c (IBM Accurate Mathematical Library), __sin_fma
// ---------------------------------------------------------------------------------------------------------------------
namespace sincos_ {
static constexpr double big = 0x1.8p+45, hp0 = 0x1.921fb54442d18p+0, hp1 = 0x1.1a62633145c07p-54;
static constexpr double sn3 = -0x1.5555555555515p-3, sn5 = 0x1.11110e829872fp-7;
static constexpr double cs4 = -0x1.5555555555535p-5, cs6 = 0x1.6c16bedd9e239p-10; // cs2 rounds to 0.5
static constexpr double s1 = -0x1.5555555555555p-3, s2 = 0x1.1111111110ecep-7, s3 = -0x1.a01a019db08b8p-13,
                        s4 = 0x1.71de27b9a7ed9p-19, s5 = -0x1.addffc2fcdf59p-26;
} // namespace sincos_

// TAYLOR_SIN (xx = a * a, a, da)
RTM_FN double taylor_sin(double a, double da) {
    using namespace sincos_;
    const double xx = a * a;
    double p = fma_(xx, s5, s4);
    p = fma_(xx, p, s3);
    p = fma_(xx, p, s2);
    p = fma_(xx, p, s1);
    const double t = fma_(p, a, -(0.5 * da)); // POLYNOMIAL (xx) * a - 0.5 * da
    return a + fma_(xx, t, da);
}
// sin (x + dx), |x| < 0.855469 (table of sin / cos at k / 128 + a short series around it)
RTM_FN double do_sin(double x, double dx) {
    using namespace sincos_;
    const double xold = x;
    if (abs_(x) < 0.126) return taylor_sin(x, dx);
    if (x <= 0.0) dx = -dx;
    const double u = big + abs_(x);
    x = abs_(x) - (u - big);
    const uint32_t k = lo32(u) * 4u;
    const double xx = x * x;
    const double s = x + fma_(x * xx, fma_(xx, sn5, sn3), dx);
    const double c = fma_(x, dx, xx * fma_(xx, fma_(xx, cs6, cs4), 0.5));
    const double sn = rtm_sincostab[k], ssn = rtm_sincostab[k + 1], cs = rtm_sincostab[k + 2], ccs = rtm_sincostab[k + 3];
    double cor = fma_(s, ccs, ssn);
    cor = fma_(-c, sn, cor);
    cor = fma_(s, cs, cor);
    return with_sign_of(sn + cor, xold);
}
// cos (x + dx)
RTM_FN double do_cos(double x, double dx) {
    using namespace sincos_;
    if (x < 0.0) dx = -dx;
    const double u = big + abs_(x);
    x = (abs_(x) - (u - big)) + dx;
    const uint32_t k = lo32(u) * 4u;
    const double xx = x * x;
    const double s = fma_(x * xx, fma_(xx, sn5, sn3), x);
    const double c = xx * fma_(xx, fma_(xx, cs6, cs4), 0.5);
    const double sn = rtm_sincostab[k], ssn = rtm_sincostab[k + 1], cs = rtm_sincostab[k + 2], ccs = rtm_sincostab[k + 3];
    double cor = fma_(-s, ssn, ccs);
    cor = fma_(-c, cs, cor);
    cor = fma_(-s, sn, cor);
    return cs + cor;
}
// reduce_sincos: x = n * pi/2 + (a + da), 2.426265 < |x| < 105414350
RTM_FN uint32_t reduce_sincos(double x, double *a, double *da) {
    const double toint = 0x1.8p+52, hpinv = 0x1.45f306dc9c883p-1;
    const double mp1 = 0x1.921fb58000000p+0, mp2 = -0x1.dde973c000000p-27, pp3 = -0x1.cb3b398000000p-55,
                 pp4 = -0x1.d747f23e32ed7p-83;
    const double t = fma_(x, hpinv, toint);
    const double xn = t - toint;
    double y = fma_(-xn, mp1, x);
    y = fma_(-xn, mp2, y);
    const double t2 = fma_(-xn, pp3, y);
    double db = fma_(-pp3, xn, y - t2);
    const double b = fma_(-xn, pp4, t2);
    db = db + fma_(-xn, pp4, t2 - b);
    *a = b;
    *da = db;
    return lo32(t) & 3u;
}
RTM_FN double do_sincos(double a, double da, uint32_t n) {
    const double r = (n & 1u) ? do_cos(a, da) : do_sin(a, da);
    return (n & 2u) ? -r : r;
}

// __branred (sysdeps/ieee754/dbl-64/branred.c): x = n * pi/2 + (a + aa) for 105414350 <= |x| < 2^1024, by multiplying the
// two halves of x with the 24-bit digits of 2 / pi that can still matter.  Plain binary64, nothing fused (the one build of
// this file is not an FMA build).
RTM_FN void branred_half(double x1, double *sum_out, double *b_out, double *bb_out) {
    const double tm24 = 0x1p-24, big = 0x1.8p+52, big1 = 0x1.8p+54;
    int32_t k = (int32_t)((hi32(x1) >> 20) & 2047u);
    k = (k - 450) / 24;
    if (k < 0) k = 0;
    double gor = from_bits((uint64_t)(0x63F00000u - (uint32_t)((k * 24) << 20)) << 32); // 2^576 / 2^(24 k)
    double r[6];
    for (int i = 0; i < 6; ++i) {
        r[i] = x1 * rtm_toverp[k + i] * gor;
        gor *= tm24;
    }
    double sum = 0.0;
    for (int i = 0; i < 3; ++i) {
        const double s = (r[i] + big) - big; // round to integer
        sum += s;
        r[i] -= s;
    }
    double t = 0.0;
    for (int i = 0; i < 6; ++i) t += r[5 - i];
    double bb = (((((r[0] - t) + r[1]) + r[2]) + r[3]) + r[4]) + r[5];
    double s = (t + big) - big;
    sum += s;
    t -= s;
    const double b = t + bb;
    bb = (t - b) + bb;
    s = (sum + big1) - big1;
    sum -= s;
    *sum_out = sum;
    *b_out = b;
    *bb_out = bb;
}
RTM_FN uint32_t branred(double x, double *a, double *aa) {
    const double split = 0x1.0000002p+27, hp0 = 0x1.921fb54442d18p+0, hp1 = 0x1.1a62633145c07p-54,
                 mp1 = 0x1.921fb58000000p+0, mp2 = -0x1.dde9740000000p-27;
    x *= 0x1p-600;
    double t = x * split;
    const double x1 = t - (t - x);
    const double x2 = x - x1;
    double sum1, b1, bb1, sum2, b2, bb2;
    branred_half(x1, &sum1, &b1, &bb1);
    branred_half(x2, &sum2, &b2, &bb2);
    double sum = sum1 + sum2;
    double b = b1 + b2;
    double bb = abs_(b1) > abs_(b2) ? (b1 - b) + b2 : (b2 - b) + b1;
    if (b > 0.5) {
        b -= 1.0;
        sum += 1.0;
    } else if (b < -0.5) {
        b += 1.0;
        sum -= 1.0;
    }
    double s = b + (bb + bb1 + bb2);
    t = ((b - s) + bb) + (bb1 + bb2);
    b = s * split;
    const double t1 = b - (b - s);
    const double t2 = s - t1;
    b = s * hp0;
    bb = (((t1 * mp1 - b) + t1 * mp2) + t2 * mp1) + (t2 * mp2 + s * hp1 + t * hp0);
    s = b + bb;
    t = (b - s) + bb;
    *a = s;
    *aa = t;
    return (uint32_t)(int32_t)sum & 3u;
}
RTM_FN double sin(double x) {
    using namespace sincos_;
    const uint32_t k = hi32(x) & 0x7FFFFFFFu;
    if (k < 0x3E500000u) return x;                  // |x| < 2^-26
    if (k < 0x3FEB6000u) return do_sin(x, 0.0);     // |x| < 0.855469
    if (k < 0x400368FDu) {                          // |x| < 2.426265
        const double t = hp0 - abs_(x);
        return with_sign_of(do_cos(t, hp1), x);
    }
    if (k < 0x7FF00000u) { // finite
        double a, da;
        const uint32_t n = k < 0x419921FBu ? reduce_sincos(x, &a, &da) : branred(x, &a, &da); // |x| < 105414350 ?
        return do_sincos(a, da, n);
    }
    return nan_(); // infinite or NaN
}

// cos: the same file (__cos_fma)
RTM_FN double cos(double x) {
    using namespace sincos_;
    const uint32_t k = hi32(x) & 0x7FFFFFFFu;
    if (k < 0x3E400000u) return 1.0;            // |x| < 2^-27
    if (k < 0x3FEB6000u) return do_cos(x, 0.0); // |x| < 0.855469
    if (k < 0x400368FDu) {                      // |x| < 2.426265: sin (pi/2 - |x|), the difference to twice the precision
        const double y = hp0 - abs_(x);
        const double a = y + hp1;
        const double da = (y - a) + hp1;
        return do_sin(a, da);
    }
    if (k < 0x7FF00000u) {
        double a, da;
        const uint32_t n = k < 0x419921FBu ? reduce_sincos(x, &a, &da) : branred(x, &a, &da);
        return do_sincos(a, da, n + 1u);
    }
    return nan_();
}

// ---------------------------------------------------------------------------------------------------------------------
// acos: glibc 2.35 sysdeps/ieee754/dbl-64/e_asin.c (__ieee754_acos, IBM Accurate Mathematical Library), FMA build
// ---------------------------------------------------------------------------------------------------------------------
// |x| in one of the table intervals: a degree-D expansion around T[n] (coefficients T[n+1 .. n+D+1], value T[n+D+2])
RTM_FN double acos_interval(double x, bool pos, uint32_t n, uint32_t D) {
    const double hp0 = 0x1.921fb54442d18p+0, hp1 = 0x1.1a62633145c07p-54;
    const double *T = rtm_asincos + n;
    const double xx = (pos ? x : -x) - T[0];
    double p = T[D];
    for (uint32_t j = D - 1; j >= 2; --j) p = fma_(xx, p, T[j]);
    p = fma_(xx * xx, p, T[D + 1]);
    const double t = fma_(xx, T[1], p);
    if (pos) return (hp1 - t) + (hp0 - T[D + 2]);
    return (t + hp1) + (T[D + 2] + hp0);
}
RTM_FN double acos(double x) {
    const double hp0 = 0x1.921fb54442d18p+0, hp1 = 0x1.1a62633145c07p-54;
    const double f1 = 0x1.55555555554f9p-3, f2 = 0x1.333333336127dp-4, f3 = 0x1.6db6dae42c0e4p-5,
                 f4 = 0x1.f1c7e04f4ad99p-6, f5 = 0x1.6e442c822d419p-6, f6 = 0x1.292d80f453c72p-6;
    const int32_t m = (int32_t)hi32(x);
    const uint32_t k = (uint32_t)m & 0x7FFFFFFFu;
    const bool pos = m > 0;
    if (k < 0x3C880000u) return hp0; // |x| < 2^-55
    if (k < 0x3FC00000u) {           // |x| < 0.125
        const double x2 = x * x;
        double p = fma_(x2, f6, f5);
        p = fma_(x2, p, f4);
        p = fma_(x2, p, f3);
        p = fma_(x2, p, f2);
        p = fma_(x2, p, f1);
        const double r = hp0 - x;
        const double c0 = ((hp0 - r) - x) + hp1;
        const double cor = fma_(-p, x * x2, c0);
        return r + cor;
    }
    if (k < 0x3FD00000u) return acos_interval(x, pos, 11u * ((k >> 15) & 0x1Fu), 6);
    if (k < 0x3FE00000u) return acos_interval(x, pos, 11u * ((k >> 14) & 0x3Fu) + 352u, 6);
    if (k < 0x3FE80000u) return acos_interval(x, pos, 12u * ((k >> 13) & 0x7Fu) + 1056u, 7);
    if (k < 0x3FED8000u) return acos_interval(x, pos, 13u * ((k >> 13) & 0x7Fu) + 992u, 8);
    if (k < 0x3FEE8000u) return acos_interval(x, pos, 14u * ((k >> 13) & 0x7Fu) + 884u, 9);
    if (k < 0x3FEF0000u) return acos_interval(x, pos, 15u * ((k >> 13) & 0x7Fu) + 768u, 10);
    if (k < 0x3FF00000u) { // 0.96875 <= |x| < 1: acos = 2 asin (sqrt ((1 - |x|) / 2)), the root by Newton from a table seed
        const double rt0 = 0x1.fffffffecc1ddp-1, rt1 = 0x1.fffffff757304p-2, rt2 = 0x1.800496769c91ap-2,
                     rt3 = 0x1.4006318d1dab9p-2, t27 = 0x1p27;
        const double z = (pos ? 1.0 - x : x + 1.0) * 0.5;
        const uint64_t zb = bits(z);
        // inroot[(k & 0x001fffff) >> 14] * powtwo[511 - (k >> 21)], k = the high word of z; powtwo[j] = 2^j, j = 3 .. 27 here
        const int32_t j = 511 - (int32_t)(zb >> 53);
        double t = rtm_inroot[(uint32_t)(zb >> 46) & 127u] * from_bits((uint64_t)(1023 + j) << 52);
        const double r = fma_(-(t * t), z, 1.0);
        double q = fma_(r, rt3, rt2);
        q = fma_(r, q, rt1);
        q = fma_(r, q, rt0);
        t = q * t;
        const double c = z * t;
        const double h = fma_(-c, t * 0.5, 1.5); // 1.5 - 0.5 * t * c
        const double cw = fma_(c, t27, c);
        const double y = fma_(-t27, c, cw);       // (t27 * c + c) - t27 * c
        const double den = fma_(h, c, y);         // c * (1.5 - 0.5 t c) + y
        const double cc = fma_(-y, y, z) / den;   // (z - y * y) / (t + y)
        double p = fma_(z, f6, f5);
        p = fma_(z, p, f4);
        p = fma_(z, p, f3);
        p = fma_(z, p, f2);
        p = fma_(z, p, f1);
        p = p * z;
        const double e = p * (y + cc);
        if (m < 0) {
            const double v = ((hp1 - cc) - e) + (hp0 - y);
            return v + v;
        }
        const double v = (cc + e) + y;
        return v + v;
    }
    if (k == 0x3FF00000u && lo32(x) == 0u) return pos ? 0.0 : 0x1.921fb54442d18p+1; // acos (+-1)
    if (k > 0x7FF00000u || (k == 0x7FF00000u && lo32(x) != 0u)) return x + x;       // NaN in
    return nan_();                                                                  // |x| > 1
}

// ---------------------------------------------------------------------------------------------------------------------
// atan2: glibc 2.35 sysdeps/ieee754/dbl-64/e_atan2.c (__ieee754_atan2, IBM Accurate Mathematical Library), FMA build
// ---------------------------------------------------------------------------------------------------------------------
RTM_FN double atan2_series(double u) { // u^3 (d3 + v (d5 + ... v d13)) / u^3, v = u * u
    const double d3 = -0x1.5555555555555p-2, d5 = 0x1.99999999997fdp-3, d7 = -0x1.24924923f7603p-3,
                 d9 = 0x1.c71c6e5129a3bp-4, d11 = -0x1.7458022b13c25p-4, d13 = 0x1.375f08b31cbcep-4;
    const double v = u * u;
    double p = fma_(v, d13, d11);
    p = fma_(v, p, d9);
    p = fma_(v, p, d7);
    p = fma_(v, p, d5);
    return fma_(v, p, d3);
}
RTM_FN int32_t atan2_index(double u) { // i of the table interval around u >= 1/16
    const double two52 = 0x1p52;
    return (int32_t)(fma_(u, 256.0, two52) - two52) - 16;
}
RTM_FN double atan2_table_poly(const double *c, double v) { // c2 + v (c3 + v (c4 + v (c5 + v c6)))
    double p = fma_(v, c[6], c[5]);
    p = fma_(v, p, c[4]);
    p = fma_(v, p, c[3]);
    return fma_(v, p, c[2]);
}
RTM_FN double atan2(double y, double x) {
    const double hpi = 0x1.921fb54442d18p+0, hpi1 = 0x1.1a62633145c07p-54, opi = 0x1.921fb54442d18p+1,
                 opi1 = 0x1.1a62633145c07p-53, qpi = 0x1.921fb54442d18p-1, tqpi = 0x1.2d97c7f3321d2p+1;
    const double twom500 = 0x1p-500, two500 = 0x1p+500, inv16 = 0x1p-4;
    const uint32_t ux = hi32(x), dx = lo32(x), uy = hi32(y), dy = lo32(y);
    if ((ux & 0x7FF00000u) == 0x7FF00000u && ((ux & 0x000FFFFFu) | dx) != 0u) return x + y;
    if ((uy & 0x7FF00000u) == 0x7FF00000u && ((uy & 0x000FFFFFu) | dy) != 0u) return y + y;
    if (uy == 0u && dy == 0u) return (ux & 0x80000000u) ? opi : 0.0;     // y = +0
    if (uy == 0x80000000u && dy == 0u) return (ux & 0x80000000u) ? -opi : -0.0; // y = -0
    if (x == 0.0) return (uy & 0x80000000u) ? -hpi : hpi;
    if (ux == 0x7FF00000u && dx == 0u) { // x = +inf
        if (uy == 0x7FF00000u && dy == 0u) return qpi;
        if (uy == 0xFFF00000u && dy == 0u) return -qpi;
        return (uy & 0x80000000u) ? -0.0 : 0.0;
    }
    if (ux == 0xFFF00000u && dx == 0u) { // x = -inf
        if (uy == 0x7FF00000u && dy == 0u) return tqpi;
        if (uy == 0xFFF00000u && dy == 0u) return -tqpi;
        return (uy & 0x80000000u) ? -opi : opi;
    }
    if (uy == 0x7FF00000u && dy == 0u) return hpi;
    if (uy == 0xFFF00000u && dy == 0u) return -hpi;

    double ax = x < 0.0 ? -x : x, ay = y < 0.0 ? -y : y;
    const int32_t de = (int32_t)(uy & 0x7FF00000u) - (int32_t)(ux & 0x7FF00000u);
    if (de >= 59768832) return y > 0.0 ? hpi : -hpi; // |y / x| beyond 2^57
    if (de <= -59768832) {
        if (x > 0.0) return with_sign_of(ay / ax, y);
        return y > 0.0 ? opi : -opi;
    }
    if (ax < twom500 || ay < twom500) {
        ax *= two500;
        ay *= two500;
    }
    if (ax > two500 || ay > two500) {
        ax *= twom500;
        ay *= twom500;
    }
    // u + du = min / max to twice the precision
    double u, du;
    const bool y_small = ay < ax;
    if (y_small) {
        u = ay / ax;
        const double v = ax * u;
        const double vv = fma_(ax, u, -v);
        du = ((ay - v) - vv) / ax;
    } else {
        u = ax / ay;
        const double v = ay * u;
        const double vv = fma_(ay, u, -v);
        du = ((ax - v) - vv) / ay;
    }
    double z;
    if (x > 0.0) {
        if (y_small) { // (i) atan (ay / ax)
            if (u < inv16) {
                const double v = u * u;
                const double zz = fma_(u * v, atan2_series(u), du);
                z = u + zz;
            } else {
                const double *c = rtm_atan2_cij + 7 * atan2_index(u);
                const double t3 = u - c[0];
                const double v = du + t3;
                const double dv = abs_(t3) > abs_(du) ? (t3 - v) + du : (du - v) + t3;
                double p = fma_(v, c[6], c[5]);
                p = fma_(v, p, c[4]);
                p = fma_(v, p, c[3]);
                double zz = (v * v) * p;
                zz = fma_(dv, c[2], zz);
                zz = fma_(v, c[2], zz);
                z = zz + c[1];
            }
        } else { // (ii) pi/2 - atan (ax / ay)
            if (u < inv16) {
                const double v = u * u;
                const double zz = (u * v) * atan2_series(u);
                const double t2 = hpi - u;
                const double cor = hpi > abs_(u) ? (hpi - t2) - u : hpi - (u + t2);
                z = ((((cor + hpi1) - du) - zz) + t2);
            } else {
                const double *c = rtm_atan2_cij + 7 * atan2_index(u);
                const double v = (u - c[0]) + du;
                const double zz = fma_(-v, atan2_table_poly(c, v), hpi1);
                z = (hpi - c[1]) + zz;
            }
        }
    } else {
        if (!y_small && ax < ay) { // (iii) pi/2 + atan (ax / ay)
            if (u < inv16) {
                const double v = u * u;
                const double zz = (v * u) * atan2_series(u);
                const double t2 = u + hpi;
                const double cor = hpi > abs_(u) ? (hpi - t2) + u : (u - t2) + hpi;
                z = ((((cor + hpi1) + du) + zz) + t2);
            } else {
                const double *c = rtm_atan2_cij + 7 * atan2_index(u);
                const double v = (u - c[0]) + du;
                const double zz = fma_(v, atan2_table_poly(c, v), hpi1);
                z = (hpi + c[1]) + zz;
            }
        } else { // (iv) pi - atan (ay / ax)
            if (u < inv16) {
                const double v = u * u;
                const double zz = (v * u) * atan2_series(u);
                const double t2 = opi - u;
                const double cor = opi > abs_(u) ? (opi - t2) - u : opi - (t2 + u);
                z = ((((cor + opi1) - du) - zz) + t2);
            } else {
                const double *c = rtm_atan2_cij + 7 * atan2_index(u);
                const double v = (u - c[0]) + du;
                const double zz = fma_(-v, atan2_table_poly(c, v), opi1);
                z = (opi - c[1]) + zz;
            }
        }
    }
    return with_sign_of(z, y);
}

// ---------------------------------------------------------------------------------------------------------------------
// pow: glibc 2.35 sysdeps/ieee754/dbl-64/e_pow.c (ARM's optimized routines: log to 68 bits, then exp), __pow_fma
// ---------------------------------------------------------------------------------------------------------------------
namespace pow_ {
// 0 if y is not an integer, 1 if odd, 2 if even
RTM_FN int checkint(uint64_t iy) {
    const int e = (int)((iy >> 52) & 0x7FF);
    if (e < 0x3FF) return 0;
    if (e > 0x3FF + 52) return 2;
    if (iy & ((1ull << (0x3FF + 52 - e)) - 1)) return 0;
    if (iy & (1ull << (0x3FF + 52 - e))) return 1;
    return 2;
}
RTM_FN bool zeroinfnan(uint64_t i) { return 2 * i - 1 >= 2 * 0x7FF0000000000000ull - 1; }
// log (x) as hi + *tail, x > 0 given by its bits (a subnormal already scaled)
RTM_FN double log_inline(uint64_t ix, double *tail) {
    const double Ln2hi = 0x1.62e42fefa3800p-1, Ln2lo = 0x1.ef35793c76730p-45;
    const double A0 = -0x1p-1, A1 = -0x1.5555555555560p-1, A2 = 0x1.0000000000006p-1, A3 = 0x1.999999959554ep-1,
                 A4 = -0x1.555555529a47ap-1, A5 = -0x1.2495b9b4845e9p+0, A6 = 0x1.0002b8b263fc3p+0;
    const uint64_t tmp = ix - 0x3FE6955500000000ull;
    const uint32_t i = (uint32_t)(tmp >> 45) & 127u;
    const int32_t k = (int32_t)((int64_t)tmp >> 52);
    const uint64_t iz = ix - (tmp & 0xFFF0000000000000ull);
    const double z = from_bits(iz), kd = (double)k;
    const double invc = rtm_pow_log_tab[4 * i], logc = rtm_pow_log_tab[4 * i + 2], logctail = rtm_pow_log_tab[4 * i + 3];
    const double r = fma_(z, invc, -1.0);
    const double t1 = fma_(kd, Ln2hi, logc);
    const double t2 = r + t1;
    const double lo1 = fma_(kd, Ln2lo, logctail);
    const double lo2 = (t1 - t2) + r;
    const double ar = r * A0;
    const double ar2 = r * ar;
    const double ar3 = r * ar2;
    const double hi = t2 + ar2;
    const double lo3 = fma_(ar, r, -ar2);
    const double lo4 = (t2 - hi) + ar2;
    const double p12 = fma_(r, A2, A1), p34 = fma_(r, A4, A3), p56 = fma_(r, A6, A5);
    const double inner = fma_(ar2, fma_(p56, ar2, p34), p12);
    double lo = lo1 + lo2;
    lo = lo + lo3;
    lo = lo + lo4;
    lo = fma_(ar3, inner, lo);
    const double y = hi + lo;
    *tail = (hi - y) + lo;
    return y;
}
// exp (x + xtail) with the sign of sign_bias; the results that overflow or fall into the subnormal range
RTM_FN double exp_special(double tmp, uint64_t sbits, uint64_t ki) {
    if ((ki & 0x80000000ull) == 0) { // k > 0: the exponent of scale might have overflowed by <= 460
        sbits -= 1009ull << 52;
        const double scale = from_bits(sbits);
        return 0x1p1009 * fma_(scale, tmp, scale);
    }
    sbits += 1022ull << 52; // k < 0: special care in the subnormal range
    const double scale = from_bits(sbits);
    const double st = scale * tmp;
    double y = scale + st;
    if (abs_(y) < 1.0) {
        const double one = y < 0.0 ? -1.0 : 1.0;
        double lo = (scale - y) + st;
        const double hi = y + one;
        lo = ((one - hi) + y) + lo;
        y = (lo + hi) - one;
        if (y == 0.0) y = from_bits(sbits & 0x8000000000000000ull);
    }
    return 0x1p-1022 * y;
}
RTM_FN double exp_inline(double x, double xtail, uint32_t sign_bias) {
    const double InvLn2N = 0x1.71547652b82fep+7, Shift = 0x1.8p+52, NegLn2hiN = -0x1.62e42fefa0000p-8, NegLn2loN = -0x1.cf79abc9e3b3ap-47;
    const double C2 = 0x1.ffffffffffdbdp-2, C3 = 0x1.555555555543cp-3, C4 = 0x1.55555cf172b91p-5, C5 = 0x1.1111167a4d017p-7;
    uint32_t abstop = (uint32_t)(bits(x) >> 52) & 0x7FFu;
    if (abstop - 0x3C9u >= 0x3Fu) {
        if (abstop - 0x3C9u >= 0x80000000u) { // tiny: 1 + x
            const double one = 1.0 + x;
            return sign_bias ? -one : one;
        }
        if (abstop >= 0x409u) { // |x| >= 1024: the result underflows to (signed) zero or overflows to (signed) infinity
            if (bits(x) >> 63) return sign_bias ? -0.0 : 0.0;
            return sign_bias ? -__builtin_huge_val() : __builtin_huge_val();
        }
        abstop = 0u; // large x is special-cased below
    }
    double kd = fma_(x, InvLn2N, Shift);
    const uint64_t ki = bits(kd);
    kd -= Shift;
    double r = fma_(kd, NegLn2hiN, x);
    r = fma_(kd, NegLn2loN, r);
    r = xtail + r;
    const uint32_t idx = 2u * ((uint32_t)ki & 127u);
    const uint64_t top = (ki + sign_bias) << 45;
    const double tail = rtm_exp_tab[idx];
    const uint64_t sbits = bits(rtm_exp_tab[idx + 1]) + top;
    const double r2 = r * r;
    double tmp = fma_(fma_(r, C3, C2), r2, r + tail);
    tmp = fma_(fma_(r, C5, C4), r2 * r2, tmp);
    if (abstop == 0u) return exp_special(tmp, sbits, ki);
    const double scale = from_bits(sbits);
    return fma_(tmp, scale, scale);
}
} // namespace pow_

RTM_FN double pow(double x, double y) {
    using namespace pow_;
    uint32_t sign_bias = 0;
    uint64_t ix = bits(x);
    const uint64_t iy = bits(y);
    uint32_t topx = (uint32_t)(ix >> 52);
    const uint32_t topy = (uint32_t)(iy >> 52);
    if (topx - 0x001u >= 0x7FFu - 0x001u || (topy & 0x7FFu) - 0x3BEu >= 0x43Eu - 0x3BEu) {
        // x < 2^-1022, infinite or NaN; or |y| < 2^-65, >= 2^63 or NaN
        if (zeroinfnan(iy)) {
            if (2 * iy == 0) return 1.0; // (a signalling NaN x would give NaN: not told apart here)
            if (ix == 0x3FF0000000000000ull) return 1.0;
            if (2 * ix > 2 * 0x7FF0000000000000ull || 2 * iy > 2 * 0x7FF0000000000000ull) return x + y;
            if (2 * ix == 2 * 0x3FF0000000000000ull) return 1.0;
            if ((2 * ix < 2 * 0x3FF0000000000000ull) == !(iy >> 63)) return 0.0; // |x| < 1 and y = inf, or |x| > 1 and y = -inf
            return y * y;
        }
        if (zeroinfnan(ix)) {
            double x2 = x * x;
            if ((ix >> 63) && checkint(iy) == 1) {
                x2 = -x2;
                sign_bias = 1;
            }
            if (2 * ix == 0 && (iy >> 63)) return sign_bias ? -__builtin_huge_val() : __builtin_huge_val();
            return (iy >> 63) ? 1.0 / x2 : x2;
        }
        if (ix >> 63) { // finite x < 0
            const int yint = checkint(iy);
            if (yint == 0) return nan_();
            if (yint == 1) sign_bias = 0x800u << 7;
            ix &= 0x7FFFFFFFFFFFFFFFull;
            topx &= 0x7FFu;
        }
        if ((topy & 0x7FFu) - 0x3BEu >= 0x43Eu - 0x3BEu) {
            if (ix == 0x3FF0000000000000ull) return 1.0;
            if ((topy & 0x7FFu) < 0x3BEu) return ix > 0x3FF0000000000000ull ? 1.0 + y : 1.0 - y; // |y| < 2^-65
            return (ix > 0x3FF0000000000000ull) == (topy < 0x800u) ? __builtin_huge_val() : 0.0; // overflow / underflow
        }
        if (topx == 0) { // subnormal x: scale it
            ix = bits(x * 0x1p52);
            ix &= 0x7FFFFFFFFFFFFFFFull;
            ix -= 52ull << 52;
        }
    }
    double lo;
    const double hi = log_inline(ix, &lo);
    const double ehi = y * hi;
    const double elo = fma_(y, lo, fma_(hi, y, -ehi));
    return exp_inline(ehi, elo, sign_bias);
}

} // namespace rtm
#endif
