// lane_emul.cpp -- TEST HARNESS ONLY (built into tests/_build/liblane_emul.so).
// Compiles the device lane program (ray-tracer_amd/csrc/rt_lane.h) for the HOST
// and runs it pixel by pixel over the committed flat scene, so the flattener,
// the BVH and the lane arithmetic can be diffed against the oracle on a machine
// without a GPU.  It is not part of librt_mi355x.so and nothing in the product
// calls it; GPU parity is established separately by the `-m gpu` tests.
//
// Built twice (tests/Makefile): liblane_emul.so takes rt_lane.h's host forms (plain divisions, 64-bit shifts); with
// -DRT_EMULATE_DEVICE_MATH liblane_emul_devmath.so takes the DEVICE forms -- the shared-reciprocal divisions, the funnel-shift
// rotations, the numbers assembled from bits -- with portable stand-ins for the three intrinsics (include/rt_rng.h, rt_lane.h:
// RTL_RCP64 is a reciprocal deliberately spoilt to v_rcp_f64's documented error bound).  Both have to match the oracle bit for bit.
// The lane program's libm calls (log in media, sin in the checker texture, atan2 / acos in a sphere's uv) go through recording
// wrappers -- the only place where the device's results may legitimately differ from the host's (its libm is not correctly
// rounded, the host's mostly is): tests/sweeps/libm_attribution.py replays a sample's recorded arguments through the device's functions
// (tools/microbench/libm_probe.hip) to show which call made a GPU pixel differ.  Macros, so that rt_lane.h stays as it is.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <math.h>
#include <vector>
#include "../ray-tracer_amd/csrc/rt_libm.h" // the kernels' log / sin / atan2 / acos (the host libm's, restated)
namespace lane_trace {
struct Call {
    int sample, fn; // fn: 0 log, 1 sin, 2 atan2, 3 acos
    double a, b, r;
};
static std::vector<Call> *sink = nullptr;
static std::vector<double> *seg_sink = nullptr; // lane_emul_trace_segments: {sample, o, d, t, prim} per segment
static bool seg_wide = false;                   // lane_emul_dump_segments: + {stream key bits, segment number}
static int cur_sample = 0;
static inline double rec(int fn, double a, double b, double r) {
    if (sink) sink->push_back(Call{cur_sample, fn, a, b, r});
    return r;
}
static int perturb_log = 0; // lane_emul_set_log_perturbation: move log's result by one ulp for about 1 argument in N (a stand-in for a libm that is not correctly rounded)
static inline double t_log(double x) {
    double r = rtm::log(x);
    if (perturb_log > 0) {
        uint64_t b;
        memcpy(&b, &x, sizeof b);
        b *= 0x9E3779B97F4A7C15ull;
        if ((b >> 20) % (uint64_t)perturb_log == 0u) r = nextafter(r, (b >> 63) ? 1.0 : -100.0);
    }
    return rec(0, x, 0.0, r);
}
static inline double t_sin(double x) { return rec(1, x, 0.0, rtm::sin(x)); }
static inline double t_atan2(double y, double x) { return rec(2, y, x, rtm::atan2(y, x)); }
static inline double t_acos(double x) { return rec(3, x, 0.0, rtm::acos(x)); }
} // namespace lane_trace
#define RTL_LOG(x) lane_trace::t_log(x)
#define RTL_SIN(x) lane_trace::t_sin(x)
#define RTL_ATAN2(y, x) lane_trace::t_atan2(y, x)
#define RTL_ACOS(x) lane_trace::t_acos(x)
#include "../ray-tracer_amd/csrc/rt_lane.h"
#include "../ray-tracer_amd/csrc/rt_lds.h"
#include "../ray-tracer_amd/csrc/rt_scene_priv.h"
#include "../include/rt_mi355x.h"

#include <cstring>

#if defined(RT_EMULATE_DEVICE_MATH)
namespace rtl {
int rtl_emul_rcp_mode = 0;
unsigned long long rtl_emul_rcp_calls = 0;
} // namespace rtl
#endif

namespace {
struct Probe { // lane_emul_world_hit
    bool on = false;
    rtl::V3 o, d;
    uint64_t base = 0;
    double t = 0.0;
    uint32_t prim = 0;
} g_probe;
// the LDS stacks of rt_kernels.hip in host memory: the 16-bit form truncates tnear exactly as the device does
// HALF: the node array holds RtNodeH records (the device's LDS copy with binary16 planes)
template <class R, bool HALF = false>
struct ArrayStack {
    typedef R Ref;
    static constexpr bool kHalfNodes = HALF;
    static constexpr bool kCubeGroups = R::kLeaf == RT_REF_LEAF; // as the device's 16-bit tree walks (rt_kernels.hip LdsStack)
    float t[RT_STACK_DEPTH];
    uint32_t r[RT_STACK_DEPTH];
    int high_water = 0;
    void push(int32_t &sp, float tnear, uint32_t ref) {
        t[sp] = R::kLeaf == RT_REF_LEAF ? rtl::bits_f32(rtl::f32_bits(tnear) & 0xFFFF0000u) : tnear;
        r[sp++] = ref;
        if (sp > high_water) high_water = sp;
    }
    void pop(int32_t &sp, float *tnear, uint32_t *ref) {
        --sp;
        *tnear = t[sp];
        *ref = r[sp];
    }
};

// One lane's state machine run to completion: the wave-vote loop of render_kernel only
// decides WHEN a lane's next step runs, never what it computes.
template <bool G, int M, bool T, bool LENS, class R = RtRef16, bool HALF = false>
void run(const RtLaunch &L, int x0, int y0, int x1, int y1, double *out, double *samples_out, int sx, int sy,
         unsigned long long *cnt, int *stack_high) {
    ArrayStack<R, HALF> st;
    if (g_probe.on) { // lane_emul_world_hit: one given ray through begin_segment and the walk, nothing shaded
        rtl::PathState ps;
        rtl::Trav tv;
        ps.o = g_probe.o;
        ps.d = g_probe.d;
        ps.T = rtl::mk(1.0, 1.0, 1.0);
        ps.k = 0;
        ps.g.base = g_probe.base;
        rt_rng_seed_state(ps.g.base, &ps.g.s0, &ps.g.s1);
        ps.g.draws = 0;
        rtl::begin_segment<G, M, T>(L, &ps, tv, st, &cnt[3]);
        while (tv.cur != R::kDone) {
            if (tv.cur < R::kLeaf) {
                if (L.n_list)
                    rtl::trav_list_step(reinterpret_cast<const float *>(L.nodes), (uint32_t)L.n_list, (uint32_t)L.n_hoisted, tv, st);
                else
                    rtl::trav_node_step(L.nodes, tv, st);
                cnt[2]++;
            } else {
                rtl::leaf_step<G, M, T>(L, &ps, tv, st, &cnt[3]);
            }
        }
        g_probe.t = tv.best_t;
        g_probe.prim = tv.best_prim;
        *stack_high = st.high_water;
        return;
    }
    for (int y = y0; y < y1; ++y)
        for (int x = x0; x < x1; ++x) {
            rtl::V3 acc = rtl::mk(0, 0, 0);
            for (int s = 0; s < L.spp; ++s) {
                lane_trace::cur_sample = s;
                rtl::PathState ps;
                rtl::Trav tv;
                rtl::V3 rad = rtl::mk(0, 0, 0);
                rtl::start_sample<LENS>(L, (uint32_t)x, (uint32_t)y, (uint32_t)s, &ps);
                for (;;) {
                    cnt[1]++;
                    rtl::begin_segment<G, M, T>(L, &ps, tv, st, &cnt[3]);
                    while (tv.cur != R::kDone) {
                        if (tv.cur < R::kLeaf) {
                            if (L.n_list) {
                                cnt[2] += (unsigned long long)L.n_list;
                                rtl::trav_list_step(reinterpret_cast<const float *>(L.nodes), (uint32_t)L.n_list, (uint32_t)L.n_hoisted, tv, st);
                            } else {
                                cnt[2]++;
                                rtl::trav_node_step(L.nodes, tv, st);
                            }
                        } else {
                            rtl::leaf_step<G, M, T>(L, &ps, tv, st, &cnt[3]);
                        }
                    }
                    if (lane_trace::seg_sink) {
                        const double row[9] = {(double)s, ps.o.x, ps.o.y, ps.o.z, ps.d.x, ps.d.y, ps.d.z, tv.best_t, (double)tv.best_prim};
                        lane_trace::seg_sink->insert(lane_trace::seg_sink->end(), row, row + 9);
                        if (lane_trace::seg_wide) { // + the stream key (raw bits) and the segment number: what a keyed medium draw needs
                            double key;
                            std::memcpy(&key, &ps.g.base, sizeof key);
                            const double more[2] = {key, (double)ps.k};
                            lane_trace::seg_sink->insert(lane_trace::seg_sink->end(), more, more + 2);
                        }
                    }
                    if (rtl::finish_segment<G, M, T>(L, &ps, tv, &rad)) break;
                }
                cnt[0]++;
                cnt[4] += ps.g.draws;
                acc = acc + rad;
                if (samples_out && x == sx && y == sy) {
                    samples_out[s * 3 + 0] = rad.x;
                    samples_out[s * 3 + 1] = rad.y;
                    samples_out[s * 3 + 2] = rad.z;
                }
            }
            double *o = out + ((size_t)y * L.width + x) * 3;
            o[0] = acc.x / (double)L.spp;
            o[1] = acc.y / (double)L.spp;
            o[2] = acc.z / (double)L.spp;
        }
    *stack_high = st.high_water;
}
} // namespace

// 1: scenes whose tree exists with binary16 planes (FlatScene::nodes_half) are walked through it, as the device does when it keeps
// that form in LDS (rt_api.cpp render_range)
static int g_half_nodes = 0;
extern "C" void lane_emul_half_nodes(int on) { g_half_nodes = on; }

// The binary16 tree (FlatScene::nodes_half) against the binary32 one: every plane's binary16 image lies OUTSIDE the binary32 plane
// it stands for (lower planes <=, upper planes >=), by at most one binary16 step, and the children are the same.
// Returns the number of violations, -1 when the scene has no binary16 tree; *n_out = nodes checked
extern "C" int lane_emul_half_tree_check(rt_scene *s, int *n_out) {
    const rt::FlatScene &f = s->flat;
    *n_out = (int)f.nodes_half.size();
    if (f.nodes_half.empty()) return -1;
    if (f.nodes_half.size() != f.nodes.size()) return 1 << 30;
    int bad = 0;
    for (size_t i = 0; i < f.nodes.size(); ++i) {
        const RtNode &n = f.nodes[i];
        const RtNodeH &h = f.nodes_half[i];
        const float *lo[3] = {n.lo_x, n.lo_y, n.lo_z}, *hi[3] = {n.hi_x, n.hi_y, n.hi_z};
        const uint16_t *hlo[3] = {h.lo_x, h.lo_y, h.lo_z}, *hhi[3] = {h.hi_x, h.hi_y, h.hi_z};
        for (int a = 0; a < 3; ++a)
            for (int c = 0; c < 2; ++c) {
                const float l = rtl::half_bits_to_float(hlo[a][c]), u = rtl::half_bits_to_float(hhi[a][c]);
                // one binary16 step at the plane's magnitude: 2^-10 relative (2^-24 absolute in the subnormal range)
                const float step_l = std::fmax(std::fabs(lo[a][c]) * 0x1p-10f, 0x1p-24f), step_u = std::fmax(std::fabs(hi[a][c]) * 0x1p-10f, 0x1p-24f);
                if (!(l <= lo[a][c]) || !(lo[a][c] - l <= step_l)) ++bad;
                if (!(u >= hi[a][c]) || !(u - hi[a][c] <= step_u)) ++bad;
            }
        if (h.child[0] != n.child[0] || h.child[1] != n.child[1]) ++bad;
    }
    return bad;
}
// rt::half_toward on its own: out[0] = the binary16 value next to x on the given side, as a float
extern "C" float lane_emul_half_toward(float x, int up) { return rtl::half_bits_to_float(rt::half_toward(x, up != 0)); }

// counters: samples, segments, nodes_visited, prims_tested, rng_draws
extern "C" int lane_emul_render(rt_scene *s, const rt_camera *cam, int W, int H, int spp, int max_depth, uint64_t seed, int x0,
                                int y0, int x1, int y1, double *out, double *samples_out, int sx, int sy,
                                unsigned long long counters[5], int *stack_high) {
    if (!s || !s->committed) return -1;
    RtLaunch L;
    std::memset(&L, 0, sizeof L);
    L.nodes = s->flat.nodes.data();
    L.prim_meta = s->flat.prim_meta.data();
    L.prim_geo = s->flat.prim_geo.data();
    L.prim_extra = s->flat.prim_extra.data();
    L.xforms = s->flat.xforms_in_store();
    L.xforms_global = L.xforms;
    L.materials = s->flat.materials.data();
    L.textures = s->flat.textures.data();
    L.image_blob = s->flat.image_blob.data();
    L.root = s->flat.root;
    L.n_list = s->flat.n_list;
    L.n_nodes = (int)s->flat.nodes.size();
    L.n_hoisted = s->flat.n_hoisted;
    L.world_mid = s->flat.world_mid ? 1 : 0;
    L.n_prims = s->flat.n_leaf_prims;
    for (int i = 0; i < 3; ++i) {
        L.cam.eye[i] = cam->eye[i];
        L.cam.lower_left[i] = cam->lower_left[i];
        L.cam.horizontal[i] = cam->horizontal[i];
        L.cam.vertical[i] = cam->vertical[i];
    }
    L.cam.lens_radius = cam->lens_radius;
    L.width = W;
    L.height = H;
    L.spp = spp;
    L.max_depth = max_depth;
    L.seed_mix = rt_mix64(seed);
    unsigned long long cnt[5] = {0, 0, 0, 0, 0};
    int hw = 0;
    const bool general = (s->flat.feature_mask & (RT_FEAT_GENERAL | RT_FEAT_MEDIUM | RT_FEAT_TEXTURED | RT_FEAT_WIDE)) != 0;
    const bool lens = cam->lens_radius != 0.0;
    if (max_depth <= 0) {
        for (int y = y0; y < y1; ++y)
            for (int x = x0; x < x1; ++x) {
                double *o = out + ((size_t)y * W + x) * 3;
                o[0] = o[1] = o[2] = 0.0;
            }
    } else if (!general) {
        if (lens)
            run<false, 0, false, true>(L, x0, y0, x1, y1, out, samples_out, sx, sy, cnt, &hw);
        else
            run<false, 0, false, false>(L, x0, y0, x1, y1, out, samples_out, sx, sy, cnt, &hw);
    } else if (!s->flat.cube_groups.empty() || (g_half_nodes && !s->flat.nodes_half.empty() && !s->flat.n_list && !s->flat.wide &&
                                                !(s->flat.feature_mask & (RT_FEAT_MEDIUM_GENERAL | RT_FEAT_DEEP_CHAIN | RT_FEAT_MEDIUM_NESTED)))) {
        // a scene with cube groups is served by the lane program for sphere media only (MEDIUM = 1: rtl::CubeGroups; the host forms
        // groups for those kernel families alone); with lane_emul_half_nodes(1) the walk reads the binary16 tree (RtNodeH)
        const bool half = g_half_nodes && !s->flat.nodes_half.empty();
        if (half) L.nodes = reinterpret_cast<const RtNode *>(s->flat.nodes_half.data());
        if (half) {
            if (lens)
                run<true, 1, true, true, RtRef16, true>(L, x0, y0, x1, y1, out, samples_out, sx, sy, cnt, &hw);
            else
                run<true, 1, true, false, RtRef16, true>(L, x0, y0, x1, y1, out, samples_out, sx, sy, cnt, &hw);
        } else {
            if (lens)
                run<true, 1, true, true>(L, x0, y0, x1, y1, out, samples_out, sx, sy, cnt, &hw);
            else
                run<true, 1, true, false>(L, x0, y0, x1, y1, out, samples_out, sx, sy, cnt, &hw);
        }
    } else if (!s->flat.wide) {
        if (lens)
            run<true, 3, true, true>(L, x0, y0, x1, y1, out, samples_out, sx, sy, cnt, &hw);
        else
            run<true, 3, true, false>(L, x0, y0, x1, y1, out, samples_out, sx, sy, cnt, &hw);
    } else {
        if (lens)
            run<true, 3, true, true, RtRef32>(L, x0, y0, x1, y1, out, samples_out, sx, sy, cnt, &hw);
        else
            run<true, 3, true, false, RtRef32>(L, x0, y0, x1, y1, out, samples_out, sx, sy, cnt, &hw);
    }
    if (counters) std::memcpy(counters, cnt, sizeof cnt);
    if (stack_high) *stack_high = hw;
    return 0;
}

// The resumable, bounded form of the unit-ball sampler (rtl::random_in_unit_sphere_bounded, used by kernels built
// with RT_BALL_ITERS > 0) against the plain loop: same accepted point, same generator state, same number of draws,
// however the iterations are split over calls.  Returns the number of bounded calls it took.
extern "C" int lane_emul_ball_check(uint64_t seed, uint64_t stream, int max_iter, double ref_point[3], double bounded_point[3],
                                    uint64_t ref_state[3], uint64_t bounded_state[3]) {
    rtl::Rng a, b;
    a.base = b.base = rt_mix64(seed) + (stream << RT_RNG_STREAM_SHIFT) * RT_RNG_GAMMA;
    rt_rng_seed_state(a.base, &a.s0, &a.s1);
    rt_rng_seed_state(b.base, &b.s0, &b.s1);
    a.draws = b.draws = 0;
    const rtl::V3 p = rtl::random_in_unit_sphere(a);
    rtl::V3 q;
    int calls = 0;
    for (;;) {
        bool ok = false;
        q = rtl::random_in_unit_sphere_bounded(b, max_iter, &ok);
        ++calls;
        if (ok) break;
        if (calls > 1000) return -1;
    }
    ref_point[0] = p.x, ref_point[1] = p.y, ref_point[2] = p.z;
    bounded_point[0] = q.x, bounded_point[1] = q.y, bounded_point[2] = q.z;
    ref_state[0] = a.s0, ref_state[1] = a.s1, ref_state[2] = a.draws;
    bounded_state[0] = b.s0, bounded_state[1] = b.s1, bounded_state[2] = b.draws;
    return calls;
}

// ConstantMedium<Sphere>::hit in its traversal form (sign of normal . direction from the plain sum p . d, rt_lane.h
// medium_hit<false>) against the record form (the reference's normalized(p / r) . d): same hit / miss, same t.
// out = {hit_traversal, t_traversal, hit_record, t_record}
extern "C" void lane_emul_medium_forms(const double oc[3], const double d[3], double radius, double density, uint64_t base,
                                       uint32_t segment, uint32_t slot, double out[4]) {
    const rtl::V3 o = rtl::mk(oc[0], oc[1], oc[2]), dir = rtl::mk(d[0], d[1], d[2]);
    unsigned long long draws = 0;
    rtl::Rec a, b;
    a.t = b.t = 0.0;
    const bool ha = rtl::medium_hit<false>(o, dir, radius, -1.0 / density, base, segment, slot, &draws, false, &a);
    const bool hb = rtl::medium_hit<true>(o, dir, radius, -1.0 / density, base, segment, slot, &draws, false, &b);
    out[0] = ha ? 1.0 : 0.0;
    out[1] = ha ? a.t : 0.0;
    out[2] = hb ? 1.0 : 0.0;
    out[3] = hb ? b.t : 0.0;
}

// the shared LDS layout (ray-tracer_amd/csrc/rt_lds.h) for host-side sweeps: out = {stack_off, node_off, job_off, swap_off,
// swap_class_bytes, total, aligned, cap that fits, effective cap}
extern "C" void lane_emul_lds_layout(unsigned stack_entries, unsigned block, unsigned entry_bytes, unsigned node_bytes, unsigned groups_per_cu,
                                     unsigned front_bytes, unsigned *out) {
    const uint32_t cap = rt_swap_cap_that_fits(stack_entries, block, entry_bytes, node_bytes, groups_per_cu, front_bytes);
    const uint32_t eff = rt_swap_cap_effective(block, cap);
    const RtLdsLayout l = rt_lds_layout(stack_entries, block, entry_bytes, node_bytes, eff, front_bytes);
    out[0] = l.stack_off;
    out[1] = l.node_off;
    out[2] = l.job_off;
    out[3] = l.swap_off;
    out[4] = l.swap_class_bytes;
    out[5] = l.total;
    out[6] = rt_lds_layout_aligned(l) ? 1u : 0u;
    out[7] = cap;
    out[8] = eff;
}

// ---- which arithmetic this build carries, and its pieces one at a time ----
// The records a box-LIST kernel copies into LDS (FlatScene::scene_blob, rtl::rec_at<true>): 0 when the blob holds every array byte
// for byte at a 16-byte offset, -1 when the scene has none (neither a list scene nor a small tree, or more than RT_LIST_SCENE_MAX bytes), else the
// number of the first array that is wrong.  *bytes_out = the blob's size.
extern "C" int lane_emul_scene_blob_check(rt_scene *s, unsigned *bytes_out, int *n_list_out) {
    const rt::FlatScene &f = s->flat;
    *bytes_out = (unsigned)f.scene_blob.size();
    *n_list_out = f.n_list;
    if (f.scene_blob.empty()) return -1;
    struct Arr { const void *p; size_t bytes; } arr[5] = {{f.xforms.data(), f.xforms.size() * sizeof(RtXform)},
                                                            {f.prim_geo.data(), f.prim_geo.size() * sizeof(RtPrimGeo)},
                                                            {f.prim_meta.data(), f.prim_meta.size() * sizeof(RtPrimMeta)},
                                                            {f.prim_extra.data(), f.prim_extra.size() * sizeof(RtPrimExtra)},
                                                            {f.materials.data(), f.materials.size() * sizeof(RtMaterial)}};
    size_t end = 0;
    for (int k = 0; k < 5; ++k) {
        const size_t off = f.scene_blob_off[k];
        if (off % 16 != 0 || off < end || off + arr[k].bytes > f.scene_blob.size()) return k + 1;
        if (arr[k].bytes && std::memcmp(f.scene_blob.data() + off, arr[k].p, arr[k].bytes) != 0) return k + 1;
        end = off + arr[k].bytes;
    }
    if (f.scene_blob.size() % 16 != 0 || f.scene_blob.size() > (size_t)RT_LIST_SCENE_MAX) return 6;
    // (prim_geo also holds two slots per cube group behind the prims)
    if (f.prim_geo.size() != f.prim_meta.size() + 2 * f.cube_groups.size() || f.prim_extra.size() != f.prim_meta.size()) return 7;
    return 0;
}

extern "C" int lane_emul_device_math(void) {
#if defined(RT_EMULATE_DEVICE_MATH)
    return 1;
#else
    return 0;
#endif
}
// direction in which the emulated v_rcp_f64 errs: 0 down, 1 up, 2 either (by a hash of the value); no effect on the host-form build
extern "C" void lane_emul_set_rcp_mode(int mode) {
#if defined(RT_EMULATE_DEVICE_MATH)
    rtl::rtl_emul_rcp_mode = mode;
#else
    (void)mode;
#endif
}
extern "C" unsigned long long lane_emul_rcp_calls(void) { // shared-reciprocal divisions taken so far (0 for the host-form build)
#if defined(RT_EMULATE_DEVICE_MATH)
    return rtl::rtl_emul_rcp_calls;
#else
    return 0ull;
#endif
}
// n triples a / s through rtl::operator/ (Vec3 / f64)
extern "C" void lane_emul_div3(long n, const double *a, const double *s, double *out) {
    for (long i = 0; i < n; ++i) {
        const rtl::V3 q = rtl::mk(a[3 * i], a[3 * i + 1], a[3 * i + 2]) / s[i];
        out[3 * i] = q.x, out[3 * i + 1] = q.y, out[3 * i + 2] = q.z;
    }
}
// n root pairs (n1 / den, n2 / den) through rtl::sphere_roots
extern "C" void lane_emul_sphere_roots(long n, const double *n1, const double *n2, const double *den, double *out) {
    for (long i = 0; i < n; ++i) {
        const rtl::Roots r = rtl::sphere_roots(n1[i], n2[i], den[i]);
        out[2 * i] = r.t1, out[2 * i + 1] = r.t2;
    }
}
// Sphere::hit's parametric result with the segment's shared reciprocal (world_roots_rcp), as a world-space sphere is tested
// during traversal: out = t, or NaN for a miss
extern "C" void lane_emul_sphere_t_world(long n, const double *oc, const double *d, const double *radius, double *out) {
    RtLaunch L{};
    L.world_mid = 1;
    for (long i = 0; i < n; ++i) {
        const rtl::V3 o = rtl::mk(oc[3 * i], oc[3 * i + 1], oc[3 * i + 2]), dir = rtl::mk(d[3 * i], d[3 * i + 1], d[3 * i + 2]);
        const double a = rtl::dot(dir, dir);
        double t = 0.0;
        out[i] = rtl::sphere_t(o, dir, a, radius[i], &t, rtl::world_roots_rcp(L, o, a)) ? t : __builtin_nan("");
    }
}
extern "C" void lane_emul_rng_forms(long n, const uint64_t *x, uint64_t *rot24, uint64_t *rot37, uint64_t *rot16, double *pm1) {
    for (long i = 0; i < n; ++i) {
        rot24[i] = rt_rotl64(x[i], 24), rot37[i] = rt_rotl64(x[i], 37), rot16[i] = rt_rotl64(x[i], 16);
        pm1[i] = rt_u64_to_pm1(x[i]);
    }
}

// The libm calls of one pixel's samples, in program order: renders the pixel alone with the recording on.
// out: max_calls x {sample, fn, a, b, host result}; returns the number of calls (may exceed max_calls: then only the first are stored)
extern "C" long lane_emul_trace_pixel(rt_scene *s, const rt_camera *cam, int W, int H, int spp, int max_depth, uint64_t seed, int x, int y,
                                      double *out, long max_calls) {
    std::vector<lane_trace::Call> calls;
    std::vector<double> img((size_t)W * (size_t)H * 3);
    unsigned long long cnt[5];
    int hw = 0;
    lane_trace::sink = &calls;
    const int rc = lane_emul_render(s, cam, W, H, spp, max_depth, seed, x, y, x + 1, y + 1, img.data(), nullptr, -1, -1, cnt, &hw);
    lane_trace::sink = nullptr;
    if (rc != 0) return -1;
    for (long i = 0; i < (long)calls.size() && i < max_calls; ++i) {
        out[5 * i + 0] = (double)calls[(size_t)i].sample;
        out[5 * i + 1] = (double)calls[(size_t)i].fn;
        out[5 * i + 2] = calls[(size_t)i].a;
        out[5 * i + 3] = calls[(size_t)i].b;
        out[5 * i + 4] = calls[(size_t)i].r;
    }
    return (long)calls.size();
}

// The segments of one pixel's samples as the lane program walks them: rows of {sample, o[3], d[3], t of the nearest hit (inf: none),
// prim}; returns the number of segments (may exceed max_rows: then only the first are stored)
extern "C" long lane_emul_trace_segments(rt_scene *s, const rt_camera *cam, int W, int H, int spp, int max_depth, uint64_t seed, int x, int y,
                                         double *out, long max_rows) {
    std::vector<double> rows;
    std::vector<double> img((size_t)W * (size_t)H * 3);
    unsigned long long cnt[5];
    int hw = 0;
    lane_trace::seg_sink = &rows;
    const int rc = lane_emul_render(s, cam, W, H, spp, max_depth, seed, x, y, x + 1, y + 1, img.data(), nullptr, -1, -1, cnt, &hw);
    lane_trace::seg_sink = nullptr;
    if (rc != 0) return -1;
    const long n = (long)(rows.size() / 9);
    for (long i = 0; i < n && i < max_rows; ++i) std::memcpy(out + 9 * i, rows.data() + 9 * (size_t)i, 9 * sizeof(double));
    return n;
}

// Every segment of a REGION's samples: rows of 11 doubles {sample, o[3], d[3], t, prim, stream key (raw 64 bits), segment number};
// returns the number of rows (may exceed max_rows: then only the first are stored).  tools/wavefront_probe.py
extern "C" long lane_emul_dump_segments(rt_scene *s, const rt_camera *cam, int W, int H, int spp, int max_depth, uint64_t seed, int x0, int y0, int x1,
                                        int y1, double *out, long max_rows) {
    std::vector<double> rows;
    std::vector<double> img((size_t)W * (size_t)H * 3);
    unsigned long long cnt[5];
    int hw = 0;
    lane_trace::seg_sink = &rows;
    lane_trace::seg_wide = true;
    const int rc = lane_emul_render(s, cam, W, H, spp, max_depth, seed, x0, y0, x1, y1, img.data(), nullptr, -1, -1, cnt, &hw);
    lane_trace::seg_sink = nullptr;
    lane_trace::seg_wide = false;
    if (rc != 0) return -1;
    const long n = (long)(rows.size() / 11);
    for (long i = 0; i < n && i < max_rows; ++i) std::memcpy(out + 11 * i, rows.data() + 11 * (size_t)i, 11 * sizeof(double));
    return n;
}

// sensitivity probe: log's result off by one ulp for about one argument in n (0: never) -- what a device libm does to an image
extern "C" void lane_emul_set_log_perturbation(int one_in_n) { lane_trace::perturb_log = one_in_n; }

// World::hit for one given ray as the lane program finds it (hoisted prims, binary32 culling, binary64 tests; keyed medium draws
// of stream (seed, stream), segment 0): out = {t, prim}; returns 1 for a hit.  The oracle's counterpart is orc_kat_world_hit.
extern "C" int lane_emul_world_hit(rt_scene *s, const rt_camera *cam, const double o[3], const double d[3], uint64_t seed, uint64_t stream,
                                   double out[2]) {
    g_probe.on = true;
    g_probe.o = rtl::mk(o[0], o[1], o[2]);
    g_probe.d = rtl::mk(d[0], d[1], d[2]);
    g_probe.base = rt_rng_base(seed, stream);
    double img[3] = {0, 0, 0};
    unsigned long long cnt[5];
    int hw = 0;
    const int rc = lane_emul_render(s, cam, 1, 1, 1, 1, seed, 0, 0, 1, 1, img, nullptr, -1, -1, cnt, &hw);
    g_probe.on = false;
    if (rc != 0) return -1;
    out[0] = g_probe.t;
    out[1] = (double)g_probe.prim;
    return g_probe.prim != 0xFFFFFFFFu ? 1 : 0;
}
