// libm_emul.cpp -- TEST-ONLY: ray-tracer_amd/csrc/rt_libm.h compiled for the host next to the host's own libm, so that
// the restatement can be compared with glibc bit for bit without a GPU (tests/test_libm_cpu.py).  Not part of the product
// library.
#include <math.h>
#include <stdint.h>
#include <string.h>

#include "../ray-tracer_amd/csrc/rt_libm.h"

namespace {
typedef double (*fn1)(double);
typedef double (*fn2)(double, double);
// through volatile pointers: the compiler must call the library, not fold or substitute a builtin
volatile fn1 g_log = ::log, g_sin = ::sin, g_acos = ::acos, g_cos = ::cos;
volatile fn2 g_atan2 = ::atan2, g_pow = ::pow;

inline uint64_t bits_of(double x) {
    uint64_t u;
    memcpy(&u, &x, 8);
    return u;
}
inline bool same(double a, double b) { // same bits, or both NaN (sign / payload of a NaN are not claimed)
    return bits_of(a) == bits_of(b) || (a != a && b != b);
}
} // namespace

extern "C" {
// which: 0 log, 1 sin, 2 acos, 3 atan2 (y = a, x = b), 4 cos, 5 pow (a, b)
void libm_emul_eval(int which, const double *a, const double *b, int64_t n, double *mine, double *host) {
    for (int64_t i = 0; i < n; ++i) {
        switch (which) {
        case 0: mine[i] = rtm::log(a[i]); host[i] = g_log(a[i]); break;
        case 1: mine[i] = rtm::sin(a[i]); host[i] = g_sin(a[i]); break;
        case 2: mine[i] = rtm::acos(a[i]); host[i] = g_acos(a[i]); break;
        case 4: mine[i] = rtm::cos(a[i]); host[i] = g_cos(a[i]); break;
        case 5: mine[i] = rtm::pow(a[i], b[i]); host[i] = g_pow(a[i], b[i]); break;
        default: mine[i] = rtm::atan2(a[i], b[i]); host[i] = g_atan2(a[i], b[i]); break;
        }
    }
}
// number of arguments on which the restatement and the host differ; the first one is reported
int64_t libm_emul_count_diffs(int which, const double *a, const double *b, int64_t n, int64_t *first) {
    int64_t bad = 0;
    for (int64_t i = 0; i < n; ++i) {
        double m, h;
        switch (which) {
        case 0: m = rtm::log(a[i]); h = g_log(a[i]); break;
        case 1: m = rtm::sin(a[i]); h = g_sin(a[i]); break;
        case 2: m = rtm::acos(a[i]); h = g_acos(a[i]); break;
        case 4: m = rtm::cos(a[i]); h = g_cos(a[i]); break;
        case 5: m = rtm::pow(a[i], b[i]); h = g_pow(a[i], b[i]); break;
        default: m = rtm::atan2(a[i], b[i]); h = g_atan2(a[i], b[i]); break;
        }
        if (!same(m, h)) {
            if (!bad) *first = i;
            ++bad;
        }
    }
    return bad;
}
}
