// rt_probe.hip -- EXPERIMENT ONLY, not part of librt_mi355x.so: `make -C ray-tracer_amd/csrc probe` links it into
// ../lib/librt_mi355x_travprobe.so, tools/wavefront_probe.py drives it.
//
// Question: what would the two traversal-side stages of a WAVEFRONT organisation of the general-media family (rays and hits in
// global memory, docs/experiments.md) cost on the book-two cover scene, measured rather than priced?
//   begin_kernel    -- one begun ray per lane, everything full: binary32 ray constants + the hoisted prims (the fog, a
//                      ConstantMedium with its logarithm), result {best_t, best_prim} to global memory;
//   traverse_kernel -- persistent waves, the megakernel's voted node / leaf blocks, but a lane whose walk is finished writes its
//                      hit (12 bytes) and takes the NEXT ray of a global queue at once (wave-aggregated atomic) instead of waiting
//                      for a shade quorum: the stage a wavefront design runs "always full".
// Input: the segments of real paths of the scene (o, d, stream key, segment number) recorded by the CPU lane program
// (tests/lane_emul.cpp), so the ray population is the megakernel's own; output checked against the t / prim recorded there.
#define RT_TU_PART 4
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <vector>

#include "rt_lane.h"
#include "rt_lds.h"
#include "rt_scene_priv.h"
#include "rt_types.h"

extern __shared__ __attribute__((aligned(16))) unsigned char rt_lds[];

namespace {
constexpr int kBlock = 256;
struct ProbeRay {
    double o[3], d[3];
    uint64_t base; // stream key of the sample (keyed medium draws)
    uint32_t k;    // segment number
    uint32_t pad;
};
struct ProbeBegun {
    double best_t;
    uint32_t best_prim;
    uint32_t done; // the careful scan settled the segment (rtl::trav_begin)
};
struct ProbeHit {
    double t;
    uint32_t prim;
    uint32_t pad;
};
struct LdsStack16 {
    typedef RtRef16 Ref;
    uint32_t *base;
    __device__ __forceinline__ void set(unsigned char *lds) { base = reinterpret_cast<uint32_t *>(lds) + threadIdx.x; }
    __device__ __forceinline__ void push(int32_t &sp, float tnear, uint32_t ref) {
        base[sp * kBlock] = (__float_as_uint(tnear) & 0xFFFF0000u) | ref;
        ++sp;
    }
    __device__ __forceinline__ void pop(int32_t &sp, float *tnear, uint32_t *ref) {
        --sp;
        const uint32_t e = base[sp * kBlock];
        *tnear = __uint_as_float(e & 0xFFFF0000u);
        *ref = e & 0xFFFFu;
    }
};
// ---- a 4-wide tree (round 4, second question): the cover family waits for the L2 on every node step (docs/experiments.md 1.5); a
// node with four children halves the number of DEPENDENT fetches per ray.  Built here from the scene's binary tree (the larger inner
// child of a node is replaced by its own two children until there are four or none is left), same binary32 culling boxes.
struct RtNode4 { // 128 bytes: per axis the four children's lower planes, then their upper planes; then the references
    float lo_x[4], hi_x[4], lo_y[4], hi_y[4], lo_z[4], hi_z[4];
    uint32_t child[4]; // RtRef16 references; kNoChild4 for an empty slot
    uint32_t pad[4];
};
constexpr uint32_t kNoChild4 = 0xFFFFFFFFu;
template <class Stack>
__device__ __forceinline__ void node4_step(const RtNode4 *nodes4, rtl::Trav &tv, Stack &st) {
    const unsigned char *N = reinterpret_cast<const unsigned char *>(nodes4) + tv.cur * 128u;
    // the entry plane of an axis is the lower one exactly when the binary tree's cursor offsets say so (rtl::trav_ray_constants)
    const uint32_t sx = tv.ox == 0u ? 0u : 16u, sy = tv.oy == 8u ? 0u : 16u, sz = tv.oz == 16u ? 0u : 16u;
    const float4 ex = *reinterpret_cast<const float4 *>(N + sx), qx = *reinterpret_cast<const float4 *>(N + (sx ^ 16u));
    const float4 ey = *reinterpret_cast<const float4 *>(N + 32u + sy), qy = *reinterpret_cast<const float4 *>(N + 32u + (sy ^ 16u));
    const float4 ez = *reinterpret_cast<const float4 *>(N + 64u + sz), qz = *reinterpret_cast<const float4 *>(N + 64u + (sz ^ 16u));
    const uint4 ch = *reinterpret_cast<const uint4 *>(N + 96u);
    const float pex[4] = {ex.x, ex.y, ex.z, ex.w}, pey[4] = {ey.x, ey.y, ey.z, ey.w}, pez[4] = {ez.x, ez.y, ez.z, ez.w};
    const float pqx[4] = {qx.x, qx.y, qx.z, qx.w}, pqy[4] = {qy.x, qy.y, qy.z, qy.w}, pqz[4] = {qz.x, qz.y, qz.z, qz.w};
    const uint32_t c[4] = {ch.x, ch.y, ch.z, ch.w};
    float tmin[4];
    bool hit[4];
    uint32_t near_c = kNoChild4;
    float near_t = 0.0f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        tmin[i] = fmaxf(fmaxf(fmaf(pex[i], tv.idx, tv.nx), fmaf(pey[i], tv.idy, tv.ny)), fmaxf(fmaf(pez[i], tv.idz, tv.nz), 0.0f));
        const float tmax = fminf(fminf(fmaf(pqx[i], tv.idx, tv.fx), fmaf(pqy[i], tv.idy, tv.fy)), fminf(fmaf(pqz[i], tv.idz, tv.fz), tv.best32));
        hit[i] = c[i] != kNoChild4 && tmin[i] <= tmax * 1.000002f;
        if (hit[i] && (near_c == kNoChild4 || tmin[i] < near_t)) {
            near_c = (uint32_t)i;
            near_t = tmin[i];
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (hit[i] && (uint32_t)i != near_c) st.push(tv.sp, tmin[i], c[i]);
    if (near_c != kNoChild4)
        tv.cur = near_c == 0u ? c[0] : (near_c == 1u ? c[1] : (near_c == 2u ? c[2] : c[3]));
    else
        rtl::trav_pop(tv, st);
}

__device__ __forceinline__ void log_table_to_lds() { // rt_lds.h RT_LDS_LOG_TABLE_BYTES: rtl::log_cold reads it at the front
    for (uint32_t i = threadIdx.x; i < RT_LDS_LOG_TABLE_BYTES / 8u; i += (uint32_t)kBlock) reinterpret_cast<double *>(rt_lds)[i] = rtm_log_tab[i];
    __syncthreads();
}

// stage 1: begin every segment (general + sphere media + textures family: <true, 1>)
// (`total` >= n: the recorded rays are gone through again and again, ray j = rays[j % n], until the launch is long enough for a
// steady-state figure -- a few rays per lane measure the launch's start and tail, not the stage)
__global__ __launch_bounds__(kBlock, 4) void begin_kernel(const RtLaunch L, const ProbeRay *rays, uint32_t n, uint32_t total, ProbeBegun *out) {
    log_table_to_lds();
    LdsStack16 st;
    st.set(rt_lds + RT_LDS_LOG_TABLE_BYTES);
    for (uint32_t j = blockIdx.x * kBlock + threadIdx.x; j < total; j += gridDim.x * kBlock) {
        const uint32_t i = j % n;
        const ProbeRay r = rays[i];
        rtl::PathState ps;
        ps.o = rtl::mk(r.o[0], r.o[1], r.o[2]);
        ps.d = rtl::mk(r.d[0], r.d[1], r.d[2]);
        ps.k = (int32_t)r.k;
        ps.g.base = r.base;
        ps.g.draws = 0;
        rtl::Trav tv;
        unsigned long long prims = 0;
        rtl::begin_segment<true, 1, true>(L, &ps, tv, st, &prims);
        ProbeBegun b;
        b.best_t = tv.best_t;
        b.best_prim = tv.best_prim;
        b.done = tv.cur == RtRef16::kDone ? 1u : 0u;
        out[i] = b;
    }
}

// stage 2: persistent traversal with immediate refill from the global ray queue
// stats (per launch, lane 0 of each wave): [0] node-block executions, [1] lanes at a node in them, [2] leaf-block executions,
// [3] lanes at a leaf, [4] fetch executions, [5] lanes fetched
template <bool WIDE4>
__global__ __launch_bounds__(kBlock, 4) void traverse_kernel(const RtLaunch L, const ProbeRay *rays, const ProbeBegun *begun, uint32_t n, uint32_t total,
                                                             unsigned int *counter, ProbeHit *out, int fetch_min, int vote_leaf, int node_keep,
                                                             unsigned long long *stats, const RtNode4 *nodes4, uint32_t root4) {
    typedef RtRef16 Ref;
    log_table_to_lds();
    LdsStack16 st;
    st.set(rt_lds + RT_LDS_LOG_TABLE_BYTES);
    rtl::PathState ps;
    rtl::Trav tv;
    tv.cur = Ref::kDead; // no ray yet
    tv.sp = 0;
    tv.best_prim = 0xFFFFFFFFu;
    tv.best_t = 0.0;
    uint32_t idx = 0;
    uint32_t chunk_next = 0, chunk_end = 0; // wave-uniform
    constexpr int kChunk = 4096;
    bool has = false, drained = false;
    unsigned long long s_nw = 0, s_nl = 0, s_lw = 0, s_ll = 0, s_fw = 0, s_fl = 0, prims = 0;
    const bool lane0 = (threadIdx.x & 63u) == 0u;
    for (uint32_t trip = 0; trip < 0x4000000u; ++trip) { // (bounded: an experiment must not hang a box)
        const bool is_node = has && tv.cur < Ref::kLeaf;
        const bool is_leaf = has && tv.cur >= Ref::kLeaf && tv.cur < Ref::kDone;
        const bool is_free = !has || tv.cur == Ref::kDone;
        const unsigned long long mN = __ballot(is_node), mL = __ballot(is_leaf), mF = __ballot(is_free && !(drained && !has));
        const int nN = __popcll(mN), nL = __popcll(mL), nF = __popcll(mF);
        if (nN == 0 && nL == 0 && nF == 0) break; // every lane is retired
        if (nF > 0 && (nF >= fetch_min || (nN == 0 && nL == 0))) {
            // ---- fetch: finished lanes write their hit and take the next rays ----
            if (has && tv.cur == Ref::kDone) {
                ProbeHit h;
                h.t = tv.best_t;
                h.prim = tv.best_prim;
                h.pad = 0u;
                out[idx] = h;
                has = false;
            }
            // ray indices come in chunks of kChunk per wave (one global atomic per chunk, like the megakernel's jobs: one atomic per
            // fetch on a single counter was what the first form of this probe measured -- its time went with the number of fetches)
            const unsigned long long m = __ballot(!has && !drained);
            const uint32_t want = (uint32_t)__popcll(m);
            uint32_t first = 0;
            if (want != 0u) {
                if (chunk_next + want > chunk_end) { // (wave-uniform) the rest of the old chunk is dropped into the next one's count
                    uint32_t c = 0;
                    const int leader = __ffsll((long long)m) - 1;
                    if ((int)(threadIdx.x & 63u) == leader) c = atomicAdd(counter, (unsigned)kChunk);
                    c = (uint32_t)__builtin_amdgcn_readlane((int)c, leader);
                    // the unused tail of the previous chunk: those rays are simply not traversed by anybody (the figure is per ray
                    // actually traversed; `done_rays` counts them)
                    chunk_next = c;
                    chunk_end = c + (uint32_t)kChunk;
                }
                first = chunk_next;
                chunk_next += want;
            }
            if (!has && !drained) {
                const uint32_t mine = first + (uint32_t)__popcll(m & ((1ull << (threadIdx.x & 63u)) - 1ull));
                if (mine < total) {
                    idx = mine % n;
                    const ProbeRay r = rays[idx];
                    const ProbeBegun b = begun[idx];
                    ps.o = rtl::mk(r.o[0], r.o[1], r.o[2]);
                    ps.d = rtl::mk(r.d[0], r.d[1], r.d[2]);
                    ps.k = (int32_t)r.k;
                    ps.g.base = r.base;
                    ps.g.draws = 0;
                    rtl::trav_ray_constants(L, ps.o, ps.d, tv); // (binary32 constants again; the hoisted prims are begin_kernel's)
                    tv.best_t = b.best_t;
                    tv.best_prim = b.best_prim;
                    tv.best32 = rtl::up32(tv.best_t);
                    tv.r2a = rtl::world_roots_rcp(L, ps.o, rtl::dot(ps.d, ps.d));
                    tv.sp = 0;
                    tv.cur = b.done ? (uint32_t)Ref::kDone : (WIDE4 ? root4 : L.root);
                    has = true;
                } else {
                    drained = true;
                    tv.cur = Ref::kDead;
                }
            }
            if (lane0) {
                ++s_fw;
                s_fl += want;
            }
            (void)first;
        } else if (nL >= vote_leaf || nN == 0) {
            if (is_leaf) rtl::leaf_step<true, 1, true>(L, &ps, tv, st, &prims);
            if (lane0) {
                ++s_lw;
                s_ll += (unsigned long long)nL;
            }
        } else {
            for (;;) {
                const bool at_node = has && tv.cur < Ref::kLeaf;
                const int cnt = __popcll(__ballot(at_node));
                if (cnt == 0) break;
                if (at_node) {
                    if (WIDE4)
                        node4_step(nodes4, tv, st);
                    else
                        rtl::trav_node_step<true>(L.nodes, tv, st);
                }
                if (lane0) {
                    ++s_nw;
                    s_nl += (unsigned long long)cnt;
                }
                if (cnt < node_keep) break;
            }
        }
    }
    if (lane0 && stats) {
        atomicAdd(&stats[0], s_nw);
        atomicAdd(&stats[1], s_nl);
        atomicAdd(&stats[2], s_lw);
        atomicAdd(&stats[3], s_ll);
        atomicAdd(&stats[4], s_fw);
        atomicAdd(&stats[5], s_fl);
    }
}
} // namespace

#define PROBE_TRY(x)                                                                      \
    do {                                                                                  \
        hipError_t e_ = (x);                                                              \
        if (e_ != hipSuccess) {                                                           \
            std::fprintf(stderr, "rt_probe: %s -> %s\n", #x, hipGetErrorString(e_));      \
            return -1;                                                                    \
        }                                                                                 \
    } while (0)

// rays: n records of 80 bytes {o[3], d[3], base, k, pad}; hits_out: n x {t, prim, pad}; ms_out: {begin_kernel, traverse_kernel} (best of
// `repeats`); stats_out: 6 counters of the last traverse launch.  The scene must be committed on a device with a tree (no list mode)
// and belong to the general + sphere media family.
// the scene's binary tree collapsed into 4-wide nodes; returns the root's index and the deepest stack a walk can need
static double area4(const rt::Aabb &b) {
    const double dx = b.hi[0] - b.lo[0], dy = b.hi[1] - b.lo[1], dz = b.hi[2] - b.lo[2];
    return dx * dy + dy * dz + dz * dx;
}
static uint32_t go4(const rt::FlatScene &f, int32_t node, std::vector<RtNode4> *out, int depth, int *deepest) { // (depth <= the binary tree's: 24)
    struct Slot {
        rt::Aabb box;
        int32_t child; // >= 0 inner node of the binary tree, < 0 leaf (~prim)
    };
    std::vector<Slot> slots;
    for (int c = 0; c < 2; ++c) slots.push_back(Slot{f.host_nodes[(size_t)node].box[c], f.host_nodes[(size_t)node].child[c]});
    while (slots.size() < 4) {
        int pick = -1;
        for (size_t i = 0; i < slots.size(); ++i)
            if (slots[i].child >= 0 && (pick < 0 || area4(slots[i].box) > area4(slots[(size_t)pick].box))) pick = (int)i;
        if (pick < 0) break;
        const rt::HostNode h = f.host_nodes[(size_t)slots[(size_t)pick].child];
        slots[(size_t)pick] = Slot{h.box[0], h.child[0]};
        slots.push_back(Slot{h.box[1], h.child[1]});
    }
    const uint32_t me = (uint32_t)out->size();
    out->emplace_back();
    *deepest = std::max(*deepest, depth + 1);
    RtNode4 nd;
    std::memset(&nd, 0, sizeof nd);
    for (int i = 0; i < 4; ++i) nd.child[i] = kNoChild4;
    for (size_t i = 0; i < slots.size(); ++i) {
        float lo[3], hi[3];
        rt::cull_box(slots[i].box, lo, hi);
        nd.lo_x[i] = lo[0], nd.lo_y[i] = lo[1], nd.lo_z[i] = lo[2];
        nd.hi_x[i] = hi[0], nd.hi_y[i] = hi[1], nd.hi_z[i] = hi[2];
        nd.child[i] = slots[i].child >= 0 ? go4(f, slots[i].child, out, depth + 1, deepest) : (RT_REF_LEAF | (uint32_t)(~slots[i].child));
    }
    (*out)[me] = nd;
    return me;
}

extern "C" int rt_probe_traverse(rt_scene *s, const void *rays_host, int n, int replicate, int fetch_min, int vote_leaf, int node_keep, int repeats, double *ms_out,
                                 void *hits_out, unsigned long long *stats_out, int wide4) {
    if (!s || !s->committed || s->device < 0 || n <= 0 || replicate < 1 || (long long)n * replicate > 0xFFFFFFF0ll) return -1;
    const uint32_t total = (uint32_t)n * (uint32_t)replicate;
    if (s->flat.n_list != 0 || s->flat.wide) return -2;
    PROBE_TRY(hipSetDevice(s->device));
    RtLaunch L;
    std::memset(&L, 0, sizeof L);
    L.nodes = (const RtNode *)s->d_nodes;
    L.prim_meta = (const RtPrimMeta *)s->d_prim_meta;
    L.prim_geo = (const RtPrimGeo *)s->d_prim_geo;
    L.prim_extra = (const RtPrimExtra *)s->d_prim_extra;
    L.xforms = s->d_xforms ? (const RtXform *)((const unsigned char *)s->d_xforms + s->flat.xform_store_offset()) : nullptr;
    L.xforms_global = L.xforms;
    L.materials = (const RtMaterial *)s->d_materials;
    L.textures = (const RtTexture *)s->d_textures;
    L.image_blob = (const uint8_t *)s->d_blob;
    L.n_nodes = (int)s->flat.nodes.size();
    L.n_list = 0;
    L.stack_entries = std::min(RT_STACK_DEPTH, s->flat.max_depth + 1);
    L.root = s->flat.root;
    L.n_hoisted = s->flat.n_hoisted;
    L.world_mid = s->flat.world_mid ? 1 : 0;
    L.n_prims = s->flat.n_leaf_prims;
    L.max_depth = 100;
    std::vector<RtNode4> nodes4;
    int stack4 = 0;
    uint32_t root4 = 0;
    RtNode4 *d_nodes4 = nullptr;
    if (wide4) {
        if (s->flat.root >= RT_REF_LEAF) return -4; // (a single leaf: no tree)
        root4 = go4(s->flat, (int32_t)s->flat.root, &nodes4, 0, &stack4);
        stack4 = 3 * stack4 + 1;
        if (nodes4.size() > RT_REF_MAX) return -4;
        PROBE_TRY(hipMalloc((void **)&d_nodes4, nodes4.size() * sizeof(RtNode4)));
        PROBE_TRY(hipMemcpy(d_nodes4, nodes4.data(), nodes4.size() * sizeof(RtNode4), hipMemcpyHostToDevice));
        L.stack_entries = stack4;
        ms_out[3] = (double)nodes4.size();
        ms_out[4] = (double)stack4;
    }
    const unsigned lds = RT_LDS_LOG_TABLE_BYTES + (unsigned)L.stack_entries * kBlock * 4u;
    ProbeRay *d_rays = nullptr;
    ProbeBegun *d_begun = nullptr;
    ProbeHit *d_hits = nullptr;
    unsigned int *d_counter = nullptr;
    unsigned long long *d_stats = nullptr;
    PROBE_TRY(hipMalloc((void **)&d_rays, (size_t)n * sizeof(ProbeRay)));
    PROBE_TRY(hipMalloc((void **)&d_begun, (size_t)n * sizeof(ProbeBegun)));
    PROBE_TRY(hipMalloc((void **)&d_hits, (size_t)n * sizeof(ProbeHit)));
    PROBE_TRY(hipMalloc((void **)&d_counter, 256));
    PROBE_TRY(hipMalloc((void **)&d_stats, 64));
    PROBE_TRY(hipMemcpy(d_rays, rays_host, (size_t)n * sizeof(ProbeRay), hipMemcpyHostToDevice));
    int dev = 0;
    hipDeviceProp_t prop;
    PROBE_TRY(hipGetDevice(&dev));
    PROBE_TRY(hipGetDeviceProperties(&prop, dev));
    int per_cu = 0;
    if (wide4)
        PROBE_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)traverse_kernel<true>, kBlock, lds));
    else
        PROBE_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)traverse_kernel<false>, kBlock, lds));
    if (per_cu < 1) return -3;
    const int blocks = per_cu * prop.multiProcessorCount;
    hipEvent_t e0, e1;
    PROBE_TRY(hipEventCreate(&e0));
    PROBE_TRY(hipEventCreate(&e1));
    double best_b = 1e30, best_t = 1e30;
    for (int rep = 0; rep < repeats; ++rep) {
        PROBE_TRY(hipEventRecord(e0, nullptr));
        hipLaunchKernelGGL(begin_kernel, dim3((unsigned)std::min(blocks * 4, (n + kBlock - 1) / kBlock)), dim3(kBlock), lds, nullptr, L, d_rays, (uint32_t)n, total, d_begun);
        PROBE_TRY(hipEventRecord(e1, nullptr));
        PROBE_TRY(hipEventSynchronize(e1));
        float ms = 0.0f;
        PROBE_TRY(hipEventElapsedTime(&ms, e0, e1));
        best_b = std::min(best_b, (double)ms);
        PROBE_TRY(hipMemset(d_counter, 0, 256));
        PROBE_TRY(hipMemset(d_stats, 0, 64));
        PROBE_TRY(hipEventRecord(e0, nullptr));
        if (wide4)
            hipLaunchKernelGGL(traverse_kernel<true>, dim3((unsigned)blocks), dim3(kBlock), lds, nullptr, L, d_rays, d_begun, (uint32_t)n, total, d_counter, d_hits,
                               fetch_min, vote_leaf, node_keep, d_stats, d_nodes4, root4);
        else
            hipLaunchKernelGGL(traverse_kernel<false>, dim3((unsigned)blocks), dim3(kBlock), lds, nullptr, L, d_rays, d_begun, (uint32_t)n, total, d_counter, d_hits,
                               fetch_min, vote_leaf, node_keep, d_stats, d_nodes4, root4);
        PROBE_TRY(hipEventRecord(e1, nullptr));
        PROBE_TRY(hipEventSynchronize(e1));
        PROBE_TRY(hipEventElapsedTime(&ms, e0, e1));
        best_t = std::min(best_t, (double)ms);
    }
    PROBE_TRY(hipGetLastError());
    ms_out[0] = best_b;
    ms_out[1] = best_t;
    ms_out[2] = (double)blocks;
    PROBE_TRY(hipMemcpy(hits_out, d_hits, (size_t)n * sizeof(ProbeHit), hipMemcpyDeviceToHost));
    PROBE_TRY(hipMemcpy(stats_out, d_stats, 6 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    (void)hipFree(d_rays);
    (void)hipFree(d_begun);
    (void)hipFree(d_hits);
    (void)hipFree(d_counter);
    (void)hipFree(d_stats);
    if (d_nodes4) (void)hipFree(d_nodes4);
    return 0;
}
