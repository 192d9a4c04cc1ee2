// rt_kernels.hip -- HIP kernels for gfx950 (MI355X).  The only compute path of
// librt_mi355x.so: there is no CPU fallback.
//
// render_kernel -- persistent wavefronts, wave-vote scheduling.
//   Every lane owns one camera path at a time and is in one of three states:
//   NODE (at an inner BVH node), LEAF (at a primitive), DONE (traversal finished,
//   waiting to be shaded).  Each trip round the loop the wave ballots the states
//   and runs exactly ONE of three blocks for the lanes in that state:
//     node block  -- binary32 slab tests of both children (culling only), ordered
//                    descent, LDS stack push/pop with tnear culling;
//     leaf block  -- the binary64 primitive test (the reference's arithmetic);
//     shade block -- binary64 hit record + Material::scatter, then the next
//                    segment; a finished path stores its radiance.  Finished lanes
//                    of one wave hold a mix of materials, so the wave first swaps
//                    paths with the other waves of its workgroup through per-class
//                    queues in LDS ("swap at shade", below): it scatters one
//                    material class at a time, or starts new samples for the whole
//                    wave at once (jobs = 8x8-pixel tile x job_spp samples, drawn
//                    from one device-wide atomic counter).
//   The expensive blocks only run when enough lanes have queued up for them (or
//   nothing else can run), so each block executes at high SIMD occupancy instead
//   of every lane dragging the other 63 through its own branch.  Lanes never idle
//   at sample, pixel or tile boundaries; a wave retires only when the job queue
//   is empty.  Scheduling never changes a result: each path consumes its own
//   random stream (include/rt_rng.h) and every sample is an independent value.
//   The traversal stack lives in LDS, [depth][thread], conflict-free ds_read/write_b32.
//   Six kernel families are instantiated from this one template (VGPRs / scratch bytes per lane / waves per SIMD of the timed
//   builds, csrc/_obj/resource_usage.txt, tools/kernel_resources.py): spheres only (book-one: 120 / 0 / 4, 512-thread groups, two per
//   CU); lean general -- matrices, rectangles, cubes, node geometries -- (Cornell box, walked as a box list with its transform / prim /
//   material records in LDS and half-word stack entries: 128 / 64 / 4, 256-thread groups, four per CU; small TREE scenes keep their
//   records in LDS too); general + sphere media + textures (book-two cover: 128 / 80 / 4, ONE 1024-thread group per CU since round 5,
//   with a set of class queues per eight waves; the cover's tree -- 1406 nodes over cube groups, rtl::trav_leaf_step -- lives in LDS
//   with binary16 planes, the HALF instantiation); media over general boundaries and chains of 5-15 transform levels (168 / 384 / 3);
//   media inside the boundary of media (168 / 2160 / 3, its own compilation); and each of the general ones with 32-bit node references
//   (> 32 767 prims or nodes).  The scratch bytes of the first three belong to real calls on paths that hardly ever run
//   (transcendentals, the reference's boxes for rays in an axis plane: rt_lane.h).
//
// reduce_kernel -- pixel = (((s_0 + s_1) + s_2) + ...) / spp in sample order, the
//   rounding of `pixel += color(...)` in examples/book-one.rs:69-76.  Per-sample
//   radiance takes one 32-byte record per sample (15.4 GB for 1200x800x500; the
//   MI355X has 288 GB) written once and read once, noise next to the traversal.
//
// No MFMA anywhere: the workload has no dense contraction.

#include <hip/hip_runtime.h>

#include "rt_lane.h"
#include "rt_lds.h"
#include "rt_types.h"

// The library is built from FOUR compilations of this file (Makefile; parts 3 and 4: the nested-media family, the small tree scenes with
// their records in LDS): part 1 holds the spheres-only and the lean general
// families plus the small kernels and every extern "C" entry; part 2 holds the families with media / textures, which sit
// at their register limit and are compiled with the ordinary Vec3 division (-DRT_PLAIN_DIV3: the shared-reciprocal form of
// rt_lane.h costs them 32-48 bytes of scratch per lane, +1 % on the book-two cover, where it gains 2 % elsewhere).
// RT_TU_PART 0 = everything in one object (not used by the Makefile).
#ifndef RT_TU_PART
#define RT_TU_PART 0
#endif

#ifndef RT_BLOCK
#define RT_BLOCK 512 /* threads per workgroup: 8 waves share one LDS copy of the node array */
#endif
#ifndef RT_BLOCK_GENERAL
#define RT_BLOCK_GENERAL 256 /* kernel with media over general boundaries: 3 waves per SIMD as three groups per CU */
#endif
#ifndef RT_BLOCK_MEDIUM
#define RT_BLOCK_MEDIUM 1024 /* general prims + sphere media + textures (the book-two cover): ONE workgroup of sixteen waves per CU (round 5; four 256-thread
                                groups until then).  One copy of the node array serves the whole CU -- with binary16 planes the cover's tree of 1406
                                nodes over cube groups fits beside the stack -- and the class queues keep their 64 entries: the cover with its nodes
                                still in global memory 98.6 -> 95.8 ms at 300 spp (profiles/r05_logs/cover_lds_probe0.log) */
#endif
// the general kernel without medium / texture code (Cornell box) needs 157 VGPRs = 3 waves per SIMD: three workgroups
// of 4 waves (one per SIMD) per CU.  6-wave groups (384 threads) do NOT work: the second group no longer fits the
// SIMDs the first one loaded unevenly (1028 vs 1701 Msamples/s on the Cornell box)
#ifndef RT_BLOCK_LEAN
#define RT_BLOCK_LEAN 256
#endif

// vote thresholds (lanes).  A block runs when at least this many lanes wait for
// it, or when no cheaper block has work.
#ifndef RT_VOTE_SHADE
#define RT_VOTE_SHADE 56 /* (48 until round 4) book-one kernel: 40: 65.3 ms, 48: 63.5, 52: 63.4, 56: 62.95, 58: 63.5, 60: 64.0, 64: 68.2
                            (profiles/r04_experiments/ab_votes_book*.log; node-keep 12 / 16 and leaf vote 12 / 24: nothing) */
#endif
#ifndef RT_VOTE_LEAF
#define RT_VOTE_LEAF 16
#endif
#ifndef RT_NODE_KEEP
#define RT_NODE_KEEP 8
#endif
// the general kernel (matrix sprites, cubes, media, textures) has much heavier leaf tests
#ifndef RT_VOTE_SHADE_G
#define RT_VOTE_SHADE_G 48
#endif
#ifndef RT_VOTE_LEAF_G
#define RT_VOTE_LEAF_G 12 /* round 4 sweep with RT_NODE_KEEP_G (profiles/r04_experiments/ab_votes*.log): cover 300.9 -> 294.7 ms at 12 / 12 */
#endif
// the lean general family (no media / textures: the Cornell box): shade block 42 % of the time, quorum sweep 32..64 -> 56
#ifndef RT_VOTE_SHADE_LEAN
#define RT_VOTE_SHADE_LEAN 56
#endif
#ifndef RT_NODE_KEEP_G
#define RT_NODE_KEEP_G 12 /* node array in global memory (8 until round 4; 16 / 12: 295.4, 16 / 16: 298.7, 24 / 16: 301.4, 32 / 16: 306.1, 8 / 24: 310.6) */
#endif
// ... and when the node array lives in LDS a node step is cheap enough to leave the inner loop earlier (round 5, the cover with its
// binary16 tree in LDS, 300 spp: 12: 81.4 ms, 8: 80.7, 6: 80.2, 4: 80.1, 20: 83.5; profiles/r05_logs/ab_cover_votes*.log)
#ifndef RT_NODE_KEEP_G_LDS
#define RT_NODE_KEEP_G_LDS 6
#endif
#ifndef RT_WAVES_PER_EU
#define RT_WAVES_PER_EU 4 /* spheres-only kernel: 105 VGPRs fit 4 waves per SIMD */
#endif
#ifndef RT_WAVES_PER_EU_LEAN
#define RT_WAVES_PER_EU_LEAN 4 /* general prims without media / textures: 128 VGPRs, four 256-thread groups per CU (+15 % on the Cornell box over 3) */
#endif
#ifndef RT_WAVES_PER_EU_GENERAL
#define RT_WAVES_PER_EU_GENERAL 3 /* media over general boundaries: 168 VGPRs + 144 B of scratch per lane beat 214 VGPRs at 2 waves per
                                     SIMD without scratch (instanced scene 46.7 vs 55.8 ms); 4 waves (128 VGPRs, 320 B) lose: 65.3 */
#endif
#ifndef RT_WAVES_PER_EU_MEDIUM
#define RT_WAVES_PER_EU_MEDIUM 4
#endif

// iterations of the unit-ball rejection sampler per shade block (0 = run it to the end, like the reference's loop);
// see rtl::random_in_unit_sphere_bounded
#ifndef RT_BALL_ITERS
#define RT_BALL_ITERS 0
#endif

// ---- swap-at-shade (SWAP kernels): per-workgroup queues of finished segments, one per scattering material class.
// A wave that votes for the shade block parks the lanes of the other classes in LDS and pulls parked paths of ONE
// class into its free lanes, so Material::scatter runs on (nearly) homogeneous lanes; new samples are started in
// bulk the same way ("new" mode).  Nothing ever waits: a busy queue lock or a full queue just means the lane is
// shaded in place.  Which lane or wave finishes a path cannot change its result (per-sample streams and records).
// (RT_SWAP_CAP and the entry layout: rt_lds.h)
#ifndef RT_SWAP_MODE_MIN
#define RT_SWAP_MODE_MIN 56 /* a class needs this many lanes (own + parked) to be chosen over starting new samples */
#endif
#ifndef RT_SWAP_REFILL_MIN
#define RT_SWAP_REFILL_MIN 65 /* empty lanes that force a refill even in a class mode (spheres-only family: never) */
#endif
#ifndef RT_SWAP_REFILL_MIN_G
#define RT_SWAP_REFILL_MIN_G 65 /* the same for the general families */
#endif
#ifndef RT_SWAP_EARLY_RELEASE
#define RT_SWAP_EARLY_RELEASE 1
#endif
#ifndef RT_SWAP_SLEEP
#define RT_SWAP_SLEEP 2 /* s_sleep argument (x 64 clocks) while a queue lock is held by another wave */
#endif
#ifndef RT_SWAP_PROBE
#define RT_SWAP_PROBE 6 /* diagnostics: swap_cycles covers steps 1..RT_SWAP_PROBE of the swap (6 = all of it) */
#endif
#ifndef RT_SWAP_POLICY
#define RT_SWAP_POLICY 0 /* 0: the class with most lanes (own + parked); 1: the eligible class with the fullest queue */
#endif
#ifndef RT_SWAP_LOCK_TRIES
#define RT_SWAP_LOCK_TRIES 8
#endif

namespace {

// 16-bit references: one LDS word per entry = tnear truncated to its upper 16 bits | reference
// HALF: the node copy in LDS holds RtNodeH records (binary16 planes, 32 bytes: rtl::trav_node_step)
template <int BLOCK, bool HALF = false>
struct LdsStack {
    typedef RtRef16 Ref;
    static constexpr unsigned kEntryBytes = 4;
    static constexpr bool kHalfNodes = HALF;
    static constexpr bool kCubeGroups = true; // a leaf may be a cube group (rtl::trav_leaf_step; the host forms them for these walks only)
    uint32_t *base; // &stack[threadIdx.x]
    __device__ __forceinline__ void set(unsigned char *lds) { base = reinterpret_cast<uint32_t *>(lds) + threadIdx.x; }
    __device__ __forceinline__ void push(int32_t &sp, float tnear, uint32_t ref) {
        base[sp * BLOCK] = (__float_as_uint(tnear) & 0xFFFF0000u) | ref;
        ++sp;
    }
    __device__ __forceinline__ void pop(int32_t &sp, float *tnear, uint32_t *ref) {
        --sp;
        const uint32_t e = base[sp * BLOCK];
        *tnear = __uint_as_float(e & 0xFFFF0000u);
        *ref = e & 0xFFFFu;
    }
};
// 32-bit references (scenes of more than 32767 prims or nodes): two words per entry, [depth][thread] of 8 bytes
template <int BLOCK>
struct LdsStackWide {
    typedef RtRef32 Ref;
    static constexpr unsigned kEntryBytes = 8;
    uint2 *base;
    __device__ __forceinline__ void set(unsigned char *lds) { base = reinterpret_cast<uint2 *>(lds) + threadIdx.x; }
    __device__ __forceinline__ void push(int32_t &sp, float tnear, uint32_t ref) {
        base[sp * BLOCK] = make_uint2(__float_as_uint(tnear), ref);
        ++sp;
    }
    __device__ __forceinline__ void pop(int32_t &sp, float *tnear, uint32_t *ref) {
        --sp;
        const uint2 e = base[sp * BLOCK];
        *tnear = __uint_as_float(e.x);
        *ref = e.y;
    }
};
// Box-LIST scenes (<= 24 leaves, prim indices below 1 << RT_LIST_PRIM_BITS = 64: rt_types.h, the gate in rt_host.cpp reads the same
// constant): half a word per entry = ten bits of tnear (exponent + two
// mantissa bits, rounded DOWN: a lower bound, as the 16-bit form's is; tnear >= 0, so the sign bit is not stored) | six of the
// leaf's prim index.  Only the list step pushes and it pushes leaves only.  The 17 entries of the Cornell box are 8.5 KB per
// 256-thread group instead of 17: room for queues of 60 entries beside the scene's records (rtl::rec_at<true>).
template <int BLOCK>
struct LdsStackList {
    typedef RtRef16 Ref;
    static constexpr unsigned kEntryBytes = 2;
    uint16_t *base;
    __device__ __forceinline__ void set(unsigned char *lds) { base = reinterpret_cast<uint16_t *>(lds) + threadIdx.x; }
    __device__ __forceinline__ void push(int32_t &sp, float tnear, uint32_t ref) {
        base[sp * BLOCK] = (uint16_t)(((__float_as_uint(tnear) >> 21) << RT_LIST_PRIM_BITS) | (ref & ((1u << RT_LIST_PRIM_BITS) - 1u)));
        ++sp;
    }
    __device__ __forceinline__ void pop(int32_t &sp, float *tnear, uint32_t *ref) {
        --sp;
        const uint32_t e = base[sp * BLOCK];
        *tnear = __uint_as_float((e >> RT_LIST_PRIM_BITS) << 21);
        *ref = Ref::kLeaf | (e & ((1u << RT_LIST_PRIM_BITS) - 1u));
    }
};
template <int BLOCK, bool WIDE, bool LIST, bool HALF>
struct StackOf {
    typedef LdsStack<BLOCK, HALF> type;
};
template <int BLOCK>
struct StackOf<BLOCK, true, false, false> {
    typedef LdsStackWide<BLOCK> type;
};
template <int BLOCK>
struct StackOf<BLOCK, false, true, false> {
    typedef LdsStackList<BLOCK> type;
};
// workgroup size and waves per SIMD of each kernel family
__host__ __device__ constexpr int block_of(bool general, int medium) {
    return general ? (medium >= 2 ? RT_BLOCK_GENERAL : (medium == 1 ? RT_BLOCK_MEDIUM : RT_BLOCK_LEAN)) : RT_BLOCK;
}
__host__ __device__ constexpr int waves_of(bool general, int medium) {
    return general ? (medium >= 2 ? RT_WAVES_PER_EU_GENERAL : (medium == 1 ? RT_WAVES_PER_EU_MEDIUM : RT_WAVES_PER_EU_LEAN)) : RT_WAVES_PER_EU;
}

// one 32-byte aligned record per sample, two 16-byte stores: whole sectors, no read-modify-write of partially written
// lines at the memory side.  The records are written once and read once by reduce_kernel, and a lane's record lands
// 2 KB from its neighbour's (sample-major layout, pixel-major hand-out): the stores are marked non-temporal -- measured
// WRITE_SIZE per 1200x800x500 launch 20.6 GB with plain stores, 17.0 GB with nt, against 15.36 GB of records; same speed.
#ifndef RT_NT_STORE
#define RT_NT_STORE 1
#endif
// (the record's fourth word: the path's length in segments, an integer in the low half -- what a learnt tile order is made of)
__device__ __forceinline__ void store_sample(double *samples, uint32_t slot, rtl::V3 rad, int32_t segments) {
    typedef double v2d __attribute__((ext_vector_type(2)));
    v2d *o = reinterpret_cast<v2d *>(samples + (size_t)slot * 4);
    const v2d a = {rad.x, rad.y}, b = {rad.z, __builtin_bit_cast(double, (unsigned long long)(uint32_t)segments)};
#if defined(RT_PROBE_NO_SAMPLE_STORE) // timing experiment only (wrong images): what the sample-record stream costs render_kernel at most
    if (!(rad.x == -1.2345e300)) return;
#endif
    if (RT_NT_STORE) {
        __builtin_nontemporal_store(a, o);
        __builtin_nontemporal_store(b, o + 1);
    } else {
        o[0] = a;
        o[1] = b;
    }
}
// Launch fields that only the refill step needs (camera: 26 dwords, image / shard / job geometry: 17) are NOT read through the
// by-value kernel argument `L`: the compiler loads every field it sees in the prologue and keeps it in SGPRs for the whole
// persistent loop -- 66 dwords, of which 40-60 were spilled to VGPR lanes and shuffled back (v_readlane / v_writelane)
// in every shade block.  They are re-read from the kernel-argument segment (scalar loads) where they are used; the empty
// asm makes the pointer opaque at that point, so the loads can be neither hoisted out of the loop nor merged with others.
typedef const __attribute__((address_space(4))) RtLaunch *RtKernArg;
__device__ __forceinline__ RtKernArg kernarg_now() {
    RtKernArg p = (RtKernArg)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return p;
}
struct RtSampleSetup { // what rtl::start_sample reads
    RtCameraD cam;
    int32_t width, height, spp;
    uint64_t seed_mix;
};
__device__ __forceinline__ void load_sample_setup(RtKernArg K, RtSampleSetup *su) {
    for (int i = 0; i < 3; ++i) {
        su->cam.eye[i] = K->cam.eye[i];
        su->cam.lower_left[i] = K->cam.lower_left[i];
        su->cam.horizontal[i] = K->cam.horizontal[i];
        su->cam.vertical[i] = K->cam.vertical[i];
    }
    su->cam.lens_radius = K->cam.lens_radius;
    su->width = K->width;
    su->height = K->height;
    su->spp = K->spp;
    su->seed_mix = K->seed_mix;
}
__device__ __forceinline__ uint32_t lane_rank(unsigned long long mask) { // set bits below this lane
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}
// first active lane's value, for values that are the same in every lane
__device__ __forceinline__ uint32_t uniform(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

// Dynamic LDS (rt_lds.h): [the family's table][a list scene's records][stack_entries][block] traversal stack, then (LDSNODES) a
// copy of the whole node array -- node steps are a dependent pointer chase, and an LDS read returns in ~1/4 of an L2 hit -- then
// the waves' job state and the class queues.
extern __shared__ __attribute__((aligned(16))) unsigned char rt_lds[];

// LIST: the scene's leaves are a box list in LDS instead of a tree (small general scenes, rtl::trav_list_step)
// HALF: the LDS copy of the node array holds RtNodeH records (binary16 planes: half the bytes, the same slab test)
template <bool GENERAL, int MEDIUM, bool TEXTURED, bool LENS, bool COUNT, bool LDSNODES, bool SWAP, bool WIDE, bool LIST, bool RECLDS, bool HALF = false>
__global__ __launch_bounds__(block_of(GENERAL, MEDIUM), waves_of(GENERAL, MEDIUM)) void render_kernel(const RtLaunch L) {
    constexpr int kBlock = block_of(GENERAL, MEDIUM);
    // the lane program's GENERAL: 0 spheres only, 1 general prims with their records in global memory, 2 (RECLDS: box-LIST scenes
    // whose records fit) with the records in this workgroup's LDS (rt_lane.h rec_at<true>)
    constexpr int G = GENERAL ? (RECLDS ? 2 : 1) : 0;
    static_assert(!HALF || (LDSNODES && !WIDE && !LIST), "binary16 nodes exist as the LDS copy of a 16-bit tree only");
    typedef typename StackOf<kBlock, WIDE, LIST, HALF>::type Stack;
    typedef typename Stack::Ref Ref;
    // entries per class queue: the most (<= RT_SWAP_CAP) that leaves the kernel family's full occupancy resident (host)
    // (a compile-time constant for the 512-thread families: the address arithmetic of a run-time capacity costs the
    // book-one kernel 2 %)
    const uint32_t kSwapCap = SWAP ? rt_swap_cap_effective((uint32_t)kBlock, (uint32_t)L.swap_cap) : 0u;
    Stack st;
    constexpr uint32_t kStackEntry = Stack::kEntryBytes;
    const RtNode *nodes = L.nodes;
    constexpr uint32_t kNodeBytes = HALF ? (uint32_t)sizeof(RtNodeH) : (uint32_t)sizeof(RtNode);
    const uint32_t node_lds_bytes = LDSNODES ? (uint32_t)L.n_nodes * kNodeBytes : 0u;
    // ONE layout function for host and device (rt_lds.h); a launch that provides fewer bytes than it needs is refused
    // instead of run: every wave returns at once and the host reports RT_ERR_DEVICE
    constexpr uint32_t kTableBytes = rt_lds_front_bytes(MEDIUM != 0 || TEXTURED);
    const RtLdsLayout lay = rt_lds_layout((uint32_t)L.stack_entries, (uint32_t)kBlock, kStackEntry, node_lds_bytes, kSwapCap,
                                          kTableBytes + (G == 2 ? rt_lds_scene_room(L.scene_bytes) : 0u));
    st.set(rt_lds + lay.stack_off);
    const uint32_t kSwapClassBytes = lay.swap_class_bytes;
    // (the host's record offsets assume the records start right behind the family's table)
    if (lay.total > L.lds_bytes || (G == 2 && L.scene_lds_off != kTableBytes)) { // wave-uniform (kernel arguments only)
        if (threadIdx.x == 0u) atomicOr(L.status, RT_DEV_ERR_LDS_LAYOUT);
        return;
    }
    // swap queues: header {state[3], pad...}: state = entries in the queue | kSwapLock while a wave works on it; then per class RT_SWAP_F64 arrays of CAP doubles and
    // RT_SWAP_F32 arrays of CAP words (field-major: consecutive entries are consecutive addresses)
    // per-wave job state (8 words per wave): it only changes in the refill step, and as loop-carried registers its seven
    // words were copied out and back on every trip round the vote loop (25 v_mov per node-block visit)
    uint32_t *job_mem = reinterpret_cast<uint32_t *>(rt_lds + lay.job_off) + (threadIdx.x >> 6) * (RT_JOB_BYTES_PER_WAVE / 4u);
    if ((threadIdx.x & 63u) < 8u) job_mem[threadIdx.x & 63u] = (threadIdx.x & 63u) == 6u ? 1u : 0u; // job_nspp = 1, the rest 0
    // (a 1024-thread workgroup: one set of queues per eight waves, rt_lds.h rt_swap_sets; this wave's set)
    unsigned char *swap_mem = rt_lds + lay.swap_off + (rt_swap_sets((uint32_t)kBlock) > 1u ? (threadIdx.x >> 9) * lay.swap_set_bytes : 0u);
    uint32_t *swap_hdr = reinterpret_cast<uint32_t *>(swap_mem);
    if (SWAP && (threadIdx.x & 511u) < RT_SWAP_HDR_BYTES / 4u) swap_hdr[threadIdx.x & 511u] = 0u;
    if (MEDIUM != 0 || TEXTURED) // the log table (rt_libm.h) at the front of the workgroup's LDS: rtl::log_cold reads it there
        for (uint32_t i = threadIdx.x; i < RT_LDS_LOG_TABLE_BYTES / 8u; i += (uint32_t)kBlock) reinterpret_cast<double *>(rt_lds)[i] = rtm_log_tab[i];
    if (G == 2) { // the scene's records (transforms, prims, materials), packed by the host in the layout the offsets in L assume
        uint4 *dst = reinterpret_cast<uint4 *>(rt_lds + kTableBytes);
        const uint4 *src = reinterpret_cast<const uint4 *>(L.scene_blob);
        for (uint32_t i = threadIdx.x; i < L.scene_bytes / 16u; i += (uint32_t)kBlock) dst[i] = src[i];
    }
    if (LDSNODES) {
        uint4 *dst = reinterpret_cast<uint4 *>(rt_lds + lay.node_off);
        const uint4 *src = reinterpret_cast<const uint4 *>(L.nodes);
        const int n16 = L.n_nodes * (int)(kNodeBytes / 16u);
        for (int i = (int)threadIdx.x; i < n16; i += kBlock) dst[i] = src[i];
        nodes = reinterpret_cast<const RtNode *>(dst);
    }
    __syncthreads();

    rtl::PathState ps;
    rtl::Trav tv;
    tv.cur = Ref::kDone;
    tv.sp = 0;
    tv.best_prim = 0xFFFFFFFFu;
    tv.best_t = 0.0;
    bool has_path = false;
    uint32_t slot = 0; // index of this lane's sample in L.samples

    // wave-uniform job state
    bool queue_empty = false;

    unsigned long long c_nodes = 0, c_prims = 0, c_segs = 0, c_draws = 0, c_samples = 0;
    unsigned long long c_nw = 0, c_nl = 0, c_lw = 0, c_ll = 0, c_sw = 0, c_sl = 0, c_id = 0, c_il = 0, c_ie = 0;
    unsigned long long t_n = 0, t_l = 0, t_fin = 0, t_ref = 0, t_beg = 0, t0 = 0, t1 = 0;
    unsigned long long w_class = 0, w_new = 0, w_park = 0, w_pull = 0, w_busy = 0, w_scat = 0, w_off = 0, t_swap = 0; // swap diagnostics
#define RT_STAMP(v) do { if (COUNT) v = __builtin_amdgcn_s_memtime(); } while (0)
    const bool counting_lane = COUNT && (threadIdx.x & 63) == 0;
    // Counting build only: a wave that goes round its loop `watchdog_trips` times without finishing or starting one segment is
    // not going to (every trip advances at least one lane by one step of a finite traversal, or drains a queue): it sets the
    // device error word and leaves, so a livelock -- the only failure a persistent kernel cannot report otherwise -- comes
    // back as RT_ERR_DEVICE.  Wave-uniform by construction (incremented and reset on wave-uniform conditions).
    uint32_t wd_trips = 0u;

    for (;;) {
        if (COUNT && L.watchdog_trips != 0u && ++wd_trips > L.watchdog_trips) {
            if ((threadIdx.x & 63u) == 0u) atomicOr(L.status, RT_DEV_ERR_WATCHDOG);
            break;
        }
        const bool is_done = tv.cur == Ref::kDone;
        const bool is_leaf = tv.cur >= Ref::kLeaf && tv.cur < Ref::kDone;
        const bool is_node = tv.cur < Ref::kLeaf;
        const unsigned long long mS = __ballot(is_done), mL = __ballot(is_leaf), mN = __ballot(is_node);
        if ((mS | mL | mN) == 0ull) { // every lane is dead
            if (!SWAP) break;
            // ... but paths may still be parked in the workgroup's queues: the shade block below pulls them
            const uint32_t parked = __hip_atomic_load(&swap_hdr[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) |
                                    __hip_atomic_load(&swap_hdr[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) |
                                    __hip_atomic_load(&swap_hdr[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (parked == 0u) break;
        }
        const int nS = __popcll(mS), nL = __popcll(mL), nN = __popcll(mN);

        constexpr int kVoteShade = GENERAL ? (MEDIUM == 0 ? RT_VOTE_SHADE_LEAN : RT_VOTE_SHADE_G) : RT_VOTE_SHADE, kVoteLeaf = GENERAL ? RT_VOTE_LEAF_G : RT_VOTE_LEAF,
                      kNodeKeep = GENERAL ? (LDSNODES ? RT_NODE_KEEP_G_LDS : RT_NODE_KEEP_G) : RT_NODE_KEEP;
        if (nS >= kVoteShade || (nN == 0 && nL == 0)) {
            // ---------------- shade block ----------------
            if (COUNT && counting_lane) {
                ++c_sw;
                c_sl += (unsigned long long)nS;
            }
            RT_STAMP(t0);
            bool need = false;
            bool pending = false;  // the ball sampler hit its bound: the lane stays DONE and is shaded again
            bool part = is_done;   // lanes this block works for
            bool do_refill = true;
            if constexpr (SWAP) {
                constexpr uint32_t kInPlace = 3u, kEmpty = 4u, kNone = 7u;
                const uint32_t lane = threadIdx.x & 63u;
                // 1. classify.  Paths that end here without a hit record (black background, no material, a light
                //    of one colour) are settled at once, so their lanes can take a parked path below.
                uint32_t cls = kNone;
                if ((is_done && !has_path) || tv.cur == Ref::kDead) cls = kEmpty;
                if (is_done && has_path) {
                    uint32_t mat = RT_NO_MATERIAL, kind = RT_MAT_KIND_NONE;
                    if (tv.best_prim != 0xFFFFFFFFu) { // one 8-byte load: the material's kind rides in the meta word
                        const uint2 pm = *reinterpret_cast<const uint2 *>((GENERAL || MEDIUM) ? &rtl::rec_at<G == 2>(L.prim_meta, tv.best_prim) : &L.prim_meta[tv.best_prim]);
                        kind = (pm.x >> 8) & 0xFFu;
                        mat = pm.y;
                    }
                    if (kind < (uint32_t)RT_SWAP_CLASSES) {
                        cls = kind;
                    } else if (kind == RT_MAT_ISOTROPIC || (TEXTURED && kind == RT_MAT_DIFFUSE_LIGHT)) {
                        cls = kInPlace;
                    } else {
                        rtl::V3 rad = rtl::mk(0.0, 0.0, 0.0);
                        if (kind == RT_MAT_DIFFUSE_LIGHT) rad = ps.T * rtl::ld3(((GENERAL || MEDIUM) ? rtl::rec_at<G == 2>(L.materials, mat) : L.materials[mat]).rgb); // finish_segment's T * emit
                        store_sample(L.samples, slot, rad, ps.k);
                        has_path = false;
                        cls = kEmpty;
                        if (COUNT) {
                            ++c_segs;
                            c_draws += ps.g.draws;
                            ++c_samples;
                        }
                    }
                }
                if (COUNT && RT_SWAP_PROBE == 1) t_swap += __builtin_amdgcn_s_memtime() - t0;
                // 2. mode: the class with the most lanes (own + parked), or "new samples" when no class fills the wave
                const unsigned long long m0 = __ballot(cls == 0u), m1 = __ballot(cls == 1u), m2 = __ballot(cls == 2u);
                const uint32_t n0 = (uint32_t)__popcll(m0), n1 = (uint32_t)__popcll(m1), n2 = (uint32_t)__popcll(m2);
                constexpr uint32_t kSwapLock = 0x80000000u;
                const uint32_t q0 = __hip_atomic_load(&swap_hdr[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) & ~kSwapLock;
                const uint32_t q1 = __hip_atomic_load(&swap_hdr[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) & ~kSwapLock;
                const uint32_t q2 = __hip_atomic_load(&swap_hdr[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) & ~kSwapLock;
                // a class is eligible when it can (nearly) fill the wave; among the eligible ones the fullest queue goes
                // first, so the rare classes (metal, glass) are drained in whole waves instead of clogging their queue
                const uint32_t t0s = n0 + q0, t1s = n1 + q1, t2s = n2 + q2;
                uint32_t cstar = 0u, smax = t0s;
                if (t1s > smax) {
                    cstar = 1u;
                    smax = t1s;
                }
                if (t2s > smax) {
                    cstar = 2u;
                    smax = t2s;
                }
                bool eligible = smax >= (uint32_t)RT_SWAP_MODE_MIN;
                if (RT_SWAP_POLICY == 1) {
                    eligible = false;
                    uint32_t bestq = 0u;
                    if (t0s >= (uint32_t)RT_SWAP_MODE_MIN) {
                        eligible = true;
                        cstar = 0u;
                        bestq = q0;
                    }
                    if (t1s >= (uint32_t)RT_SWAP_MODE_MIN && (!eligible || q1 > bestq)) {
                        eligible = true;
                        cstar = 1u;
                        bestq = q1;
                    }
                    if (t2s >= (uint32_t)RT_SWAP_MODE_MIN && (!eligible || q2 > bestq)) {
                        eligible = true;
                        cstar = 2u;
                        bestq = q2;
                    }
                }
                const bool mode_new = !queue_empty && !eligible;
                const bool allow_push = !queue_empty; // once the job queue is dry the wave only drains
                if (COUNT && RT_SWAP_PROBE == 2) t_swap += __builtin_amdgcn_s_memtime() - t0;
                // 3. lanes 0..2 try the lock of "their" class (never wait: a busy queue is skipped this time)
                uint32_t got = 0u, cnt = 0u, my_n = 0u;
                if (lane < (uint32_t)RT_SWAP_CLASSES) {
                    my_n = lane == 0u ? n0 : (lane == 1u ? n1 : n2);
                    const uint32_t my_q = lane == 0u ? q0 : (lane == 1u ? q1 : q2);
                    const bool push_c = allow_push && (mode_new || lane != cstar) && my_n > 0u && my_q < kSwapCap;
                    const bool pull_c = !mode_new && lane == cstar && my_q > 0u;
                    if (push_c || pull_c) {
                        // One compare-and-swap both takes the lock and learns the exact count: count -> count | lock.
                        // A failed attempt returns the current word: locked -> back off, then try with that count;
                        // bounded: a holder never waits for anything, so the lock frees within a few hundred cycles.
                        uint32_t expected = my_q;
                        for (int attempt = 0; attempt < RT_SWAP_LOCK_TRIES && got == 0u; ++attempt) {
                            if (__hip_atomic_compare_exchange_strong(&swap_hdr[lane], &expected, expected | kSwapLock, __ATOMIC_ACQUIRE,
                                                                     __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) {
                                got = 1u;
                                cnt = expected;
                            } else {
                                if (expected & kSwapLock) __builtin_amdgcn_s_sleep(RT_SWAP_SLEEP);
                                expected &= ~kSwapLock;
                            }
                        }
                        if (COUNT && got == 0u) ++w_busy;
                    }
                }
                const uint32_t g0 = (uint32_t)__builtin_amdgcn_readlane((int)got, 0), g1 = (uint32_t)__builtin_amdgcn_readlane((int)got, 1),
                               g2 = (uint32_t)__builtin_amdgcn_readlane((int)got, 2);
                const uint32_t k0 = (uint32_t)__builtin_amdgcn_readlane((int)cnt, 0), k1 = (uint32_t)__builtin_amdgcn_readlane((int)cnt, 1),
                               k2 = (uint32_t)__builtin_amdgcn_readlane((int)cnt, 2);
                if (COUNT && RT_SWAP_PROBE == 3) t_swap += __builtin_amdgcn_s_memtime() - t0;
                unsigned char *qbase = swap_mem + RT_SWAP_HDR_BYTES;
                // 4. park the lanes of the other classes (every class in "new" mode)
                if (cls < (uint32_t)RT_SWAP_CLASSES && allow_push && (mode_new || cls != cstar)) {
                    // the lane's place in its class's queue: count + rank among the class's lanes.  The three masks and counts are
                    // wave-uniform (SGPRs): rank each against its own mask (v_mbcnt with scalar operands) and select the sums, instead
                    // of selecting a per-lane 64-bit mask first (VERDICT r3 #5 i)
                    const uint32_t i0 = k0 + lane_rank(m0), i1 = k1 + lane_rank(m1), i2 = k2 + lane_rank(m2);
                    const uint32_t g = cls == 0u ? g0 : (cls == 1u ? g1 : g2);
                    const uint32_t idx = cls == 0u ? i0 : (cls == 1u ? i1 : i2);
                    if (g != 0u && idx < kSwapCap) {
                        double *f64 = reinterpret_cast<double *>(qbase + cls * kSwapClassBytes) + idx;
                        uint32_t *f32 = reinterpret_cast<uint32_t *>(qbase + cls * kSwapClassBytes + RT_SWAP_F64 * 8 * kSwapCap) + idx;
                        f64[0 * kSwapCap] = ps.o.x;
                        f64[1 * kSwapCap] = ps.o.y;
                        f64[2 * kSwapCap] = ps.o.z;
                        f64[3 * kSwapCap] = ps.d.x;
                        f64[4 * kSwapCap] = ps.d.y;
                        f64[5 * kSwapCap] = ps.d.z;
                        f64[6 * kSwapCap] = ps.T.x;
                        f64[7 * kSwapCap] = ps.T.y;
                        f64[8 * kSwapCap] = ps.T.z;
                        f64[9 * kSwapCap] = __longlong_as_double((long long)ps.g.s0);
                        f64[10 * kSwapCap] = __longlong_as_double((long long)ps.g.s1);
                        f64[11 * kSwapCap] = tv.best_t;
                        if (COUNT) f64[12 * kSwapCap] = __longlong_as_double((long long)ps.g.draws);
                        if (MEDIUM) f64[13 * kSwapCap] = __longlong_as_double((long long)ps.g.base);
                        f32[0 * kSwapCap] = (uint32_t)ps.k;
                        f32[1 * kSwapCap] = tv.best_prim;
                        f32[2 * kSwapCap] = slot;
                        has_path = false;
                        cls = kEmpty;
                        if (COUNT) ++w_park;
                    }
                }
                // the parked classes are done: publish their counts and free their locks before the pull (shorter hold time)
                bool released = false;
                if (RT_SWAP_EARLY_RELEASE && got != 0u && (mode_new || lane != cstar)) {
                    __hip_atomic_store(&swap_hdr[lane], cnt + min(my_n, kSwapCap - cnt), __ATOMIC_RELEASE,
                                       __HIP_MEMORY_SCOPE_WORKGROUP);
                    released = true;
                }
                if (COUNT && RT_SWAP_PROBE == 4) t_swap += __builtin_amdgcn_s_memtime() - t0;
                // 5. pull parked paths of the chosen class into the free lanes (newest first)
                uint32_t pulled = 0u;
                {
                    const uint32_t gs = cstar == 0u ? g0 : (cstar == 1u ? g1 : g2), avail = cstar == 0u ? k0 : (cstar == 1u ? k1 : k2);
                    if (!mode_new && gs != 0u && avail > 0u) {
                        const unsigned long long mF = __ballot(cls == kEmpty);
                        pulled = min((uint32_t)__popcll(mF), avail);
                        const uint32_t r = lane_rank(mF);
                        if (cls == kEmpty && r < pulled) {
                            const uint32_t idx = avail - 1u - r;
                            const double *f64 = reinterpret_cast<const double *>(qbase + cstar * kSwapClassBytes) + idx;
                            const uint32_t *f32 =
                                reinterpret_cast<const uint32_t *>(qbase + cstar * kSwapClassBytes + RT_SWAP_F64 * 8 * kSwapCap) + idx;
                            ps.o = rtl::mk(f64[0 * kSwapCap], f64[1 * kSwapCap], f64[2 * kSwapCap]);
                            ps.d = rtl::mk(f64[3 * kSwapCap], f64[4 * kSwapCap], f64[5 * kSwapCap]);
                            ps.T = rtl::mk(f64[6 * kSwapCap], f64[7 * kSwapCap], f64[8 * kSwapCap]);
                            ps.g.s0 = (uint64_t)__double_as_longlong(f64[9 * kSwapCap]);
                            ps.g.s1 = (uint64_t)__double_as_longlong(f64[10 * kSwapCap]);
                            tv.best_t = f64[11 * kSwapCap];
                            if (COUNT) ps.g.draws = (unsigned long long)__double_as_longlong(f64[12 * kSwapCap]);
                            if (MEDIUM) ps.g.base = (uint64_t)__double_as_longlong(f64[13 * kSwapCap]);
                            ps.k = (int32_t)f32[0 * kSwapCap];
                            tv.best_prim = f32[1 * kSwapCap];
                            slot = f32[2 * kSwapCap];
                            tv.cur = Ref::kDone;
                            tv.sp = 0;
                            has_path = true;
                            cls = cstar;
                            if (COUNT) ++w_pull;
                        }
                    }
                }
                if (COUNT && RT_SWAP_PROBE == 5) t_swap += __builtin_amdgcn_s_memtime() - t0;
                // 6. publish the new counts and release (the release orders this wave's queue traffic before it)
                if (got != 0u && !released) {
                    uint32_t now = cnt;
                    if (!mode_new && lane == cstar)
                        now = cnt - pulled;
                    else
                        now = cnt + min(my_n, kSwapCap - cnt);
                    __hip_atomic_store(&swap_hdr[lane], now, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); // count without the lock bit
                }
                if (COUNT && RT_SWAP_PROBE == 6) // cycles of classification + queue traffic (steps 1-6), part of finish_cycles
                    t_swap += __builtin_amdgcn_s_memtime() - t0;
                // 7. scatter what is in registers now: the chosen class, plus whatever could not be parked
                if (COUNT && counting_lane) {
                    if (mode_new) ++w_new; else ++w_class;
                }
                if (has_path && cls <= kInPlace) {
                    rtl::V3 rad;
                    const bool fin = rtl::finish_segment<G, MEDIUM, TEXTURED>(L, &ps, tv, &rad, RT_BALL_ITERS, &pending);
                    if (COUNT && !pending) {
                        ++c_segs;
                        ++w_scat;
                        if (mode_new || cls != cstar) ++w_off;
                    }
                    if (fin) {
                        store_sample(L.samples, slot, rad, ps.k);
                        has_path = false;
                        if (COUNT) {
                            c_draws += ps.g.draws;
                            ++c_samples;
                        }
                    }
                }
                part = cls != kNone;
                need = part && !has_path;
                do_refill = mode_new || __popcll(__ballot(need)) >= (GENERAL ? RT_SWAP_REFILL_MIN_G : RT_SWAP_REFILL_MIN);
            } else if (is_done) {
                if (has_path) {
                    rtl::V3 rad;
                    const bool fin = rtl::finish_segment<G, MEDIUM, TEXTURED>(L, &ps, tv, &rad, RT_BALL_ITERS, &pending);
                    if (COUNT && !pending) ++c_segs;
                    if (fin) {
                        // one 32-byte aligned record per sample, two 16-byte stores: whole sectors,
                        // no read-modify-write of partially written lines at the memory side
                        store_sample(L.samples, slot, rad, ps.k);
                        has_path = false;
                        if (COUNT) {
                            c_draws += ps.g.draws;
                            ++c_samples;
                        }
                    }
                }
                need = !has_path;
            }
            RT_STAMP(t1);
            t_fin += t1 - t0;
            // refill: lanes without a path take the next samples of the job queue.  Executed by the
            // whole wave (wave-uniform control flow) so the job state stays identical in every lane.
            // The loop only hands out (pixel, sample, slot); the samples themselves are started once, after it: the path
            // state does not travel through the loop's back edge (the compiler copied ~36 registers per trip when it did)
            uint32_t new_x = 0, new_y = 0, new_s = 0;
            bool got = false;
            uint32_t job_next = 0, job_left = 0, job_slot0 = 0, job_x0 = 0, job_y0 = 0, job_s_first = 0, job_nspp = 1;
            uint32_t img_w = 0, img_h = 0, first_s = 0;
            if (do_refill) { // wave-uniform
                const RtKernArg K = kernarg_now();
                img_w = (uint32_t)K->width;
                img_h = (uint32_t)K->height;
                first_s = (uint32_t)K->s0;
                const uint4 ja = *reinterpret_cast<const uint4 *>(job_mem), jb = *reinterpret_cast<const uint4 *>(job_mem + 4);
                job_next = uniform(ja.x);
                job_left = uniform(ja.y);
                job_slot0 = uniform(ja.z);
                job_x0 = uniform(ja.w);
                job_y0 = uniform(jb.x);
                job_s_first = uniform(jb.y);
                job_nspp = uniform(jb.z);
            }
            while (do_refill) {
                const unsigned long long m = __ballot(need);
                if (m == 0ull) break;
                if (job_left == 0u) {
                    if (queue_empty) break;
                    uint32_t j = 0;
                    if (need && lane_rank(m) == 0u) j = atomicAdd(L.job_counter, 1u);
                    j = (uint32_t)__builtin_amdgcn_readlane((int)j, __ffsll((long long)m) - 1); // fetched by the first needing lane
                    const RtKernArg K = kernarg_now();
                    if (j >= (uint32_t)K->n_jobs) {
                        queue_empty = true;
                        break;
                    }
                    const uint32_t jobs_per_tile = (uint32_t)K->jobs_per_tile, tiles_x = (uint32_t)K->tiles_x, s_count = (uint32_t)K->s_count,
                                   job_spp = (uint32_t)K->job_spp;
                    const uint32_t kj = j / jobs_per_tile, sub = j - kj * jobs_per_tile;
                    const unsigned int *order = K->tile_order; // (wave-uniform: scalar loads)
                    const uint32_t k = order ? order[kj] : kj;
                    const uint32_t tile = (uint32_t)K->shard_index + k * (uint32_t)K->shard_count;
                    const uint32_t ty = tile / tiles_x, tx = tile - ty * tiles_x;
                    job_x0 = tx * RT_TILE_EDGE;
                    job_y0 = ty * RT_TILE_EDGE;
                    job_s_first = sub * job_spp;
                    const uint32_t nspp = min(job_spp, s_count - job_s_first);
                    job_left = nspp * RT_TILE_PIXELS;
                    job_nspp = nspp;
                    job_next = 0u;
                    job_slot0 = (k * s_count + job_s_first) * RT_TILE_PIXELS;
                }
                const uint32_t take = min((uint32_t)__popcll(m), job_left);
                const uint32_t rank = lane_rank(m);
                if (need && rank < take) {
                    const uint32_t n = job_next + rank;
                    // pixel-major order inside a job: consecutive n are consecutive samples of ONE pixel, so
                    // lanes refilled together start with near-identical rays (+1 % over sample-major)
                    const uint32_t pix = n / job_nspp, sj = n - pix * job_nspp;
                    const uint32_t x = job_x0 + (pix & 7u), y = job_y0 + (pix >> 3);
                    if (x < img_w && y < img_h) {
                        slot = job_slot0 + sj * RT_TILE_PIXELS + pix;
                        new_x = x;
                        new_y = y;
                        new_s = first_s + job_s_first + sj;
                        got = true;
                        need = false;
                    }
                    // a pixel outside the image consumes its slot and the lane asks again
                }
                job_next += take;
                job_left -= take;
            }
            if (do_refill && (threadIdx.x & 63u) == 0u) {
                *reinterpret_cast<uint4 *>(job_mem) = make_uint4(job_next, job_left, job_slot0, job_x0);
                *reinterpret_cast<uint4 *>(job_mem + 4) = make_uint4(job_y0, job_s_first, job_nspp, 0u);
            }
            if (__ballot(got) != 0ull) { // wave-uniform: the setup is read with scalar loads
                RtSampleSetup su;
                load_sample_setup(kernarg_now(), &su);
                if (got) {
                    rtl::start_sample<LENS>(su, new_x, new_y, new_s, &ps);
                    has_path = true;
                }
            }
            RT_STAMP(t0);
            t_ref += t0 - t1;
            if (part && !pending) {
                if (has_path)
                    rtl::begin_segment<G, MEDIUM, TEXTURED>(L, &ps, tv, st, &c_prims);
                else // SWAP: an empty lane waits (DONE, no path) for a parked path or the next bulk refill
                    tv.cur = (!SWAP || queue_empty) ? Ref::kDead : Ref::kDone;
            }
            if (COUNT && __ballot(part && !pending) != 0ull) wd_trips = 0u; // a segment was finished, begun, or a lane retired
            RT_STAMP(t1);
            t_beg += t1 - t0;
        } else if (nL >= kVoteLeaf || nN == 0) {
            // ---------------- leaf block ----------------
            if (COUNT && counting_lane) {
                ++c_lw;
                c_ll += (unsigned long long)nL;
            }
            RT_STAMP(t0);
            if (is_leaf) rtl::leaf_step<G, MEDIUM, TEXTURED>(L, &ps, tv, st, &c_prims);
            RT_STAMP(t1);
            t_l += t1 - t0;
        } else {
            // ---------------- node block ----------------
            RT_STAMP(t0);
            if constexpr (LIST) {
                // one step: all the leaves' boxes, every lane in lock step; the lanes leave as LEAF or DONE
                const bool at_node = tv.cur < Ref::kLeaf;
                if (COUNT) {
                    const int n_done = __popcll(__ballot(tv.cur == Ref::kDone && has_path));
                    if (counting_lane) {
                        ++c_nw;
                        c_nl += (unsigned long long)nN;
                        c_il += (unsigned long long)nL;
                        c_id += (unsigned long long)n_done;
                        c_ie += (unsigned long long)(64 - nN - nL - n_done);
                    }
                }
                if (at_node) {
                    if (COUNT) c_nodes += (unsigned long long)L.n_list;
                    rtl::trav_list_step(reinterpret_cast<const float *>(nodes), (uint32_t)L.n_list, (uint32_t)L.n_hoisted, tv, st);
                }
            } else
            for (;;) {
                const bool at_node = tv.cur < Ref::kLeaf;
                const int n = __popcll(__ballot(at_node));
                if (n == 0) break;
                if (COUNT) {
                    const int n_leaf = __popcll(__ballot(tv.cur >= Ref::kLeaf && tv.cur < Ref::kDone));
                    const int n_done = __popcll(__ballot(tv.cur == Ref::kDone && has_path));
                    if (counting_lane) {
                        ++c_nw;
                        c_nl += (unsigned long long)n;
                        c_il += (unsigned long long)n_leaf;
                        c_id += (unsigned long long)n_done;
                        c_ie += (unsigned long long)(64 - n - n_leaf - n_done);
                    }
                }
                if (at_node) {
                    if (COUNT) ++c_nodes;
                    rtl::trav_node_step<!LDSNODES>(nodes, tv, st);
                }
                if (n < kNodeKeep) break;
            }
            RT_STAMP(t1);
            t_n += t1 - t0;
        }
    }

    if (COUNT && L.counters) {
        atomicAdd(&L.counters->samples, c_samples);
        atomicAdd(&L.counters->segments, c_segs);
        atomicAdd(&L.counters->nodes_visited, c_nodes);
        atomicAdd(&L.counters->prims_tested, c_prims);
        atomicAdd(&L.counters->rng_draws, c_draws);
        if (SWAP) {
            atomicAdd(&L.counters->swap_class_mode, w_class);
            atomicAdd(&L.counters->swap_new_mode, w_new);
            atomicAdd(&L.counters->swap_parked, w_park);
            atomicAdd(&L.counters->swap_pulled, w_pull);
            atomicAdd(&L.counters->swap_lock_busy, w_busy);
            atomicAdd(&L.counters->swap_scattered, w_scat);
            atomicAdd(&L.counters->swap_off_class, w_off);
            if (counting_lane) atomicAdd(&L.counters->swap_cycles, t_swap);
        }
        if (counting_lane) {
            atomicAdd(&L.counters->node_wave, c_nw);
            atomicAdd(&L.counters->node_lane, c_nl);
            atomicAdd(&L.counters->leaf_wave, c_lw);
            atomicAdd(&L.counters->leaf_lane, c_ll);
            atomicAdd(&L.counters->shade_wave, c_sw);
            atomicAdd(&L.counters->shade_lane, c_sl);
            atomicAdd(&L.counters->node_idle_done, c_id);
            atomicAdd(&L.counters->node_idle_leaf, c_il);
            atomicAdd(&L.counters->node_idle_empty, c_ie);
            atomicAdd(&L.counters->node_cycles, t_n);
            atomicAdd(&L.counters->leaf_cycles, t_l);
            atomicAdd(&L.counters->shade_cycles, t_fin + t_ref + t_beg);
            atomicAdd(&L.counters->finish_cycles, t_fin);
            atomicAdd(&L.counters->refill_cycles, t_ref);
            atomicAdd(&L.counters->begin_cycles, t_beg);
        }
    }
}

#if RT_TU_PART == 0 || RT_TU_PART == 1
// Sum this pass's samples in sample order on top of the running sums; the last
// pass divides by spp (`pixel /= subPixelSampleCount`, examples/book-one.rs:76).
// One thread per owned pixel; consecutive threads read consecutive 32-byte records.
// COST (a render that learns its tile order, once per view: rt_api.cpp): also add up the records' fourth words, the path lengths,
// per tile (one atomic per pixel and pass).
template <bool COST>
__global__ void reduce_kernel(const double *samples, double *tiles, int n_owned_tiles, int s_count, int first_pass,
                              int last_pass, int spp, int width, int height, int tiles_x, int shard_index, int shard_count,
                              unsigned long long *tile_cost) {
    // grid-stride over the owned pixels: the launch may be the whole image (one thread per pixel) or NARROW (two 256-thread
    // groups per CU, each thread several pixels) -- the narrow form leaves the CUs to a render_kernel that runs beside it
    // (RT_FLAG_DEFERRED_OUTPUT) and still reaches the HBM roof (131 k threads x two records in flight)
    const size_t n_pix = (size_t)n_owned_tiles * RT_TILE_PIXELS, stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_pix; i += stride) {
        const size_t k = i / RT_TILE_PIXELS;
        const int pix = (int)(i % RT_TILE_PIXELS);
        const int tile = shard_index + (int)k * shard_count;
        const int x = (tile % tiles_x) * RT_TILE_EDGE + (pix & 7), y = (tile / tiles_x) * RT_TILE_EDGE + (pix >> 3);
        double *out = tiles + i * 3;
        if (x >= width || y >= height) {
            out[0] = out[1] = out[2] = 0.0;
            continue;
        }
        unsigned long long cost = 0ull;
        double r = 0.0, g = 0.0, b = 0.0;
        if (!first_pass) {
            r = out[0];
            g = out[1];
            b = out[2];
        }
        const double2 *p = reinterpret_cast<const double2 *>(samples + ((size_t)k * (size_t)s_count * RT_TILE_PIXELS + (size_t)pix) * 4);
        int s = 0;
        // (one record in flight per thread: the kernel has to fit into the 32 VGPRs per SIMD that four waves of render_kernel leave
        // free, and 131 k threads x 32 bytes in flight are enough for the HBM roof)
        for (; s < s_count; ++s) {
            const double2 a = p[0], c = p[1];
            r += a.x;
            g += a.y;
            b += c.x;
            if (COST) cost += (unsigned long long)(uint32_t)__builtin_bit_cast(unsigned long long, c.y);
            p += RT_TILE_PIXELS * 2;
        }
        if (COST) atomicAdd(&tile_cost[k], cost);
        if (last_pass) {
            const double n = (double)spp;
            r /= n;
            g /= n;
            b /= n;
        }
        out[0] = r;
        out[1] = g;
        out[2] = b;
    }
}

// The hand-out order of a later render of the same view: owned tiles by descending cost (ties: ascending index), by counting --
// thread i finds the rank of tile i among all n (n <= 65536 owned tiles, once per view).
// levels > 0: the costs are compared in that many equal steps of the largest one, so that tiles of about the same depth keep their
// ascending order (neighbouring tiles share the rays' neighbourhoods).
// tile_top_kernel (one workgroup) leaves the largest cost in tile_cost[n]; tile_order_kernel then needs one division per TILE
// (its own step) and compares the others against the step's two ends: the first form took the maximum and a 64-bit division per
// PAIR in every thread (O(n^2) divisions, ~10 ms at n = 65536).
__global__ void tile_top_kernel(unsigned long long *tile_cost, int n) {
    __shared__ unsigned long long part[1024 / 64];
    unsigned long long top = 0ull;
    for (int j = (int)threadIdx.x; j < n; j += (int)blockDim.x) top = max(top, tile_cost[j]);
    for (int off = 32; off > 0; off >>= 1) top = max(top, (unsigned long long)__shfl_xor((long long)top, off));
    if ((threadIdx.x & 63u) == 0u) part[threadIdx.x >> 6] = top;
    __syncthreads();
    if (threadIdx.x == 0u) {
        for (unsigned w = 1; w < blockDim.x / 64u; ++w) top = max(top, part[w]);
        tile_cost[n] = top;
    }
}
__global__ void tile_order_kernel(const unsigned long long *tile_cost, int n, int levels, unsigned int *order) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned long long step = levels > 0 ? tile_cost[n] / (unsigned long long)levels + 1ull : 1ull;
    const unsigned long long lo = tile_cost[i] / step * step, hi = lo + step; // the costs of this tile's step: [lo, hi)
    unsigned int rank = 0u;
    for (int j = 0; j < n; ++j) {
        const unsigned long long c = tile_cost[j];
        rank += (c >= hi || (c >= lo && j < i)) ? 1u : 0u;
    }
    order[rank] = (unsigned int)i;
}

// gathered shards -> row-major image.  One thread per (pixel, channel).
__global__ void unpack_kernel(const double *gathered, int tiles_per_shard, int shard_count, int width, int height, int tiles_x,
                              double *image) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)width * height * 3;
    if (i >= total) return;
    const int c = (int)(i % 3);
    const size_t pix = i / 3;
    const int x = (int)(pix % (size_t)width), y = (int)(pix / (size_t)width);
    const int tile = (y / RT_TILE_EDGE) * tiles_x + (x / RT_TILE_EDGE);
    const int shard = tile % shard_count, k = tile / shard_count;
    const int lane = (y % RT_TILE_EDGE) * RT_TILE_EDGE + (x % RT_TILE_EDGE);
    image[i] = gathered[(((size_t)shard * tiles_per_shard + (size_t)k) * RT_TILE_PIXELS + (size_t)lane) * 3 + (size_t)c];
}

__global__ void probe_math_kernel(const double *a, const double *b, int n, double *out_sqrt, double *out_div) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out_sqrt[i] = sqrt(a[i]);
    out_div[i] = a[i] / b[i];
}

// the product's own transcendentals (rt_libm.h, what log_cold / sphere_uv_cold / checker_sine_cold call) on n arguments
__global__ void probe_libm_kernel(int which, const double *a, const double *b, int n, double *out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double r;
    switch (which) {
    case 0: r = rtm::log(a[i]); break;
    case 1: r = rtm::sin(a[i]); break;
    case 2: r = rtm::acos(a[i]); break;
    case 4: r = rtm::cos(a[i]); break;
    case 5: r = rtm::pow(a[i], b[i]); break;
    default: r = rtm::atan2(a[i], b[i]); break;
    }
    out[i] = r;
}

#endif // parts 0 / 1

// ---- dispatch over the template instantiations ----
typedef void (*KernelFn)(const RtLaunch);

template <bool GENERAL, int MEDIUM, bool TEXTURED, bool SWAP, bool WIDE = false, bool LIST = false, bool RECLDS = false, bool HALF = false>
KernelFn pick3(bool lens, bool count, bool ldsnodes) {
    // (LIST, HALF and a tree with RECLDS imply the node copy in LDS: one instantiation, not two)
#define RT_PICK(LN, C, LD) return render_kernel<GENERAL, MEDIUM, TEXTURED, LN, C, (LD && !WIDE) || LIST || HALF || RECLDS, SWAP, WIDE, LIST, RECLDS, HALF>
    if (lens) {
        if (count) {
            if (ldsnodes) RT_PICK(true, true, true); else RT_PICK(true, true, false);
        } else {
            if (ldsnodes) RT_PICK(true, false, true); else RT_PICK(true, false, false);
        }
    } else {
        if (count) {
            if (ldsnodes) RT_PICK(false, true, true); else RT_PICK(false, true, false);
        } else {
            if (ldsnodes) RT_PICK(false, false, true); else RT_PICK(false, false, false);
        }
    }
#undef RT_PICK
}
// lds_mode: bit 0 = node array copied to LDS, bit 1 = swap-at-shade queues, bit 2 = 32-bit references (general families),
// bit 3 = box list instead of the tree (general families, 16-bit references, always in LDS), bit 4 = (list + swap only) the scene's
// records in LDS, bit 5 = (with bits 0 and 1; the family with sphere media / textures only) the LDS copy holds RtNodeH records.
// Bit 4 WITHOUT bit 3 (round 5): a small TREE scene with its records in LDS -- nodes in LDS too, swap queues, 16-bit references;
// the lean general family (compilation 1) and the one with sphere media / textures (compilation 4) have that form
#if RT_TU_PART == 0 || RT_TU_PART == 4
KernelFn pick_reclds_tree(bool lens, bool count) { return pick3<true, 1, true, true, false, false, true>(lens, count, true); } // (the lean one: part 1)
#endif
#if RT_TU_PART == 0 || RT_TU_PART == 3
// media inside the boundary of media (feature bit 16): MEDIUM = 3, the only family compiled with the nested evaluation (a real
// call per inner medium, records in scratch memory: 6 x slower than MEDIUM = 2 on the same scene, so it is kept out of it).
// Tree walk only, always with the swap queues.
KernelFn pick_nested(bool lens, bool count, int lds_mode) {
    const bool ldsnodes = (lds_mode & 1) != 0, wide = (lds_mode & 4) != 0;
    if (wide) return pick3<true, 3, true, true, true>(lens, count, false);
    return pick3<true, 3, true, true>(lens, count, ldsnodes);
}
#endif
#if RT_TU_PART == 0 || RT_TU_PART == 2
// the families with media / textures (feature bits 2, 4, 8)
KernelFn pick_media(unsigned features, bool lens, bool count, int lds_mode) {
    const bool ldsnodes = (lds_mode & 1) != 0, swap = (lds_mode & 2) != 0, wide = (lds_mode & 4) != 0, list = (lds_mode & 8) != 0;
    if (list && swap && (lds_mode & 16) != 0)
        return (features & 8u) ? pick3<true, 2, true, true, false, true, true>(lens, count, true) : pick3<true, 1, true, true, false, true, true>(lens, count, true);
    if (list) {
        if (features & 8u) return swap ? pick3<true, 2, true, true, false, true>(lens, count, true) : pick3<true, 2, true, false, false, true>(lens, count, true);
        return swap ? pick3<true, 1, true, true, false, true>(lens, count, true) : pick3<true, 1, true, false, false, true>(lens, count, true);
    }
    if (wide) { // more than 32767 prims or nodes: the node array never fits LDS
        if (features & 8u) return swap ? pick3<true, 2, true, true, true>(lens, count, false) : pick3<true, 2, true, false, true>(lens, count, false);
        return swap ? pick3<true, 1, true, true, true>(lens, count, false) : pick3<true, 1, true, false, true>(lens, count, false);
    }
    // the cover's family with its tree in binary16 (rt_api.cpp takes this form when the binary32 nodes do not fit the LDS and these do)
    if (!(features & 8u) && swap && ldsnodes && (lds_mode & 32) != 0) return pick3<true, 1, true, true, false, false, false, true>(lens, count, true);
    // media over a general boundary (bit 8): the kernel with medium_general_hit; else the one with sphere media only
    if (features & 8u) return swap ? pick3<true, 2, true, true>(lens, count, ldsnodes) : pick3<true, 2, true, false>(lens, count, ldsnodes);
    return swap ? pick3<true, 1, true, true>(lens, count, ldsnodes) : pick3<true, 1, true, false>(lens, count, ldsnodes);
}
#endif
} // namespace
#if RT_TU_PART == 2
extern "C" void *rt_pick_media_kernel(unsigned features, int lens, int count, int lds_mode) {
    return (void *)pick_media(features, lens != 0, count != 0, lds_mode);
}
#elif RT_TU_PART == 3
extern "C" void *rt_pick_nested_kernel(int lens, int count, int lds_mode) { return (void *)pick_nested(lens != 0, count != 0, lds_mode); }
#elif RT_TU_PART == 4
extern "C" void *rt_pick_reclds_tree_kernel(int lens, int count) { return (void *)pick_reclds_tree(lens != 0, count != 0); }
#elif RT_TU_PART == 1
extern "C" void *rt_pick_media_kernel(unsigned features, int lens, int count, int lds_mode);
extern "C" void *rt_pick_nested_kernel(int lens, int count, int lds_mode);
extern "C" void *rt_pick_reclds_tree_kernel(int lens, int count);
#endif
#if RT_TU_PART == 0 || RT_TU_PART == 1
namespace {
KernelFn pick(unsigned features, bool lens, bool count, int lds_mode) {
    const bool ldsnodes = (lds_mode & 1) != 0, swap = (lds_mode & 2) != 0, wide = (lds_mode & 4) != 0, list = (lds_mode & 8) != 0;
    if (features & 16u) {
#if RT_TU_PART == 1
        return (KernelFn)rt_pick_nested_kernel(lens, count, lds_mode);
#else
        return pick_nested(lens, count, lds_mode);
#endif
    }
    if (!list && !wide && swap && ldsnodes && (lds_mode & 16) != 0 && !(features & 8u)) { // a small tree scene with its records in LDS
        if ((features & ~1u) == 0u) return pick3<true, 0, false, true, false, false, true>(lens, count, true); // the lean family
#if RT_TU_PART == 1
        return (KernelFn)rt_pick_reclds_tree_kernel(lens, count);
#else
        return pick_reclds_tree(lens, count);
#endif
    }
    if ((features & ~1u) != 0u) {
#if RT_TU_PART == 1
        return (KernelFn)rt_pick_media_kernel(features, lens, count, lds_mode);
#else
        return pick_media(features, lens, count, lds_mode);
#endif
    }
    // general prims only (matrices, rectangles, cubes: the Cornell box): no medium / texture code in the kernel
    if (list && swap && (lds_mode & 16) != 0) return pick3<true, 0, false, true, false, true, true>(lens, count, true);
    if (list) return swap ? pick3<true, 0, false, true, false, true>(lens, count, true) : pick3<true, 0, false, false, false, true>(lens, count, true);
    if (wide) return swap ? pick3<true, 0, false, true, true>(lens, count, false) : pick3<true, 0, false, false, true>(lens, count, false);
    if (features == 1u) return swap ? pick3<true, 0, false, true>(lens, count, ldsnodes) : pick3<true, 0, false, false>(lens, count, ldsnodes);
    return swap ? pick3<false, 0, false, true>(lens, count, ldsnodes) : pick3<false, 0, false, false>(lens, count, ldsnodes);
}
} // namespace

extern "C" int rt_kernel_block_size(unsigned features);

// feature bits: 1 = general prims, 2 = media, 4 = textured, 8 = media over a general boundary.  `blocks` persistent workgroups,
// `lds_bytes` of dynamic LDS (stack + optional node copy).
extern "C" int rt_launch_render(const RtLaunch *L, unsigned features, int lens, int count, int ldsnodes, int blocks,
                                unsigned lds_bytes, void *stream) {
    KernelFn k = pick(features, lens != 0, count != 0, ldsnodes);
    if (lds_bytes > 48u * 1024u) { // dynamic LDS beyond the default limit must be requested explicitly
        hipError_t ea = hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (ea != hipSuccess) return (int)ea;
    }
    hipLaunchKernelGGL(k, dim3((unsigned)blocks), dim3((unsigned)rt_kernel_block_size(features)), lds_bytes, (hipStream_t)stream, *L);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

// threads per workgroup of the kernel family that serves these feature bits (see pick())
extern "C" int rt_kernel_block_size(unsigned features) {
    return features == 0u ? RT_BLOCK : (features == 1u ? RT_BLOCK_LEAN : ((features & 8u) ? RT_BLOCK_GENERAL : RT_BLOCK_MEDIUM));
}
// waves per SIMD that family is compiled for: (that x 4 SIMDs x 64) / block size workgroups share a CU's LDS
extern "C" int rt_kernel_waves_per_simd(unsigned features) {
    return features == 0u ? RT_WAVES_PER_EU
                          : (features == 1u ? RT_WAVES_PER_EU_LEAN : ((features & 8u) ? RT_WAVES_PER_EU_GENERAL : RT_WAVES_PER_EU_MEDIUM));
}
// occupancy-derived size of the persistent grid
extern "C" int rt_persistent_blocks(unsigned features, int lens, int count, int ldsnodes, unsigned lds_bytes, int *blocks_per_cu,
                                    int *n_cu) {
    KernelFn k = pick(features, lens != 0, count != 0, ldsnodes);
    int per_cu = 0;
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)k, rt_kernel_block_size(features), lds_bytes);
    if (e != hipSuccess) return (int)e;
    int dev = 0;
    hipDeviceProp_t prop;
    if ((e = hipGetDevice(&dev)) != hipSuccess) return (int)e;
    if ((e = hipGetDeviceProperties(&prop, dev)) != hipSuccess) return (int)e;
    *blocks_per_cu = per_cu;
    *n_cu = prop.multiProcessorCount;
    return 0;
}

// narrow_blocks > 0: that many 256-thread groups (grid-stride) instead of one thread per pixel
// tile_cost: null, or n_owned_tiles sums of path lengths to add this pass's to
extern "C" int rt_launch_reduce(const double *samples, double *tiles, int n_owned_tiles, int s_count, int first_pass, int last_pass,
                                int spp, int width, int height, int shard_index, int shard_count, int narrow_blocks,
                                unsigned long long *tile_cost, void *stream) {
    const size_t n = (size_t)n_owned_tiles * RT_TILE_PIXELS;
    const int tiles_x = (width + RT_TILE_EDGE - 1) / RT_TILE_EDGE;
    size_t blocks = (n + 255) / 256;
    if (narrow_blocks > 0 && (size_t)narrow_blocks < blocks) blocks = (size_t)narrow_blocks;
    if (tile_cost)
        hipLaunchKernelGGL(reduce_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, samples, tiles, n_owned_tiles,
                           s_count, first_pass, last_pass, spp, width, height, tiles_x, shard_index, shard_count, tile_cost);
    else
        hipLaunchKernelGGL(reduce_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, samples, tiles, n_owned_tiles,
                           s_count, first_pass, last_pass, spp, width, height, tiles_x, shard_index, shard_count, tile_cost);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

// tile_cost: n sums of path lengths and one more word (the largest of them, written here)
extern "C" int rt_launch_tile_order(unsigned long long *tile_cost, int n, int levels, unsigned int *order, void *stream) {
    hipLaunchKernelGGL(tile_top_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, tile_cost, n);
    hipLaunchKernelGGL(tile_order_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, tile_cost, n, levels, order);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

extern "C" int rt_launch_unpack(const double *gathered, int tiles_per_shard, int shard_count, int width, int height,
                                double *image, void *stream) {
    const size_t total = (size_t)width * height * 3;
    const int tiles_x = (width + RT_TILE_EDGE - 1) / RT_TILE_EDGE;
    hipLaunchKernelGGL(unpack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, gathered,
                       tiles_per_shard, shard_count, width, height, tiles_x, image);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

extern "C" int rt_launch_probe_libm(int which, const double *a, const double *b, int n, double *out, void *stream) {
    hipLaunchKernelGGL(probe_libm_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, which, a, b, n, out);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

extern "C" int rt_launch_probe_math(const double *a, const double *b, int n, double *out_sqrt, double *out_div, void *stream) {
    hipLaunchKernelGGL(probe_math_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a, b, n, out_sqrt,
                       out_div);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}
#endif // parts 0 / 1
