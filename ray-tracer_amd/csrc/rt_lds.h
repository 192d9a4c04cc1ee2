// rt_lds.h -- THE layout of render_kernel's dynamic LDS, one definition for the host (rt_api.cpp sizes the launch with
// it) and the device (rt_kernels.hip carves the regions out of it).
//
// Why it exists (docs/experiments.md section 2, "the 05:58 abort of round 2"): the kernel placed a new region (the waves' job state)
// between the node copy and the swap queues while the host still sized the launch without it; the queues' last 128 bytes then
// lay outside the workgroup's allocation and aliased the next workgroup's traversal stack -- a persistent kernel whose
// waves only leave when their paths are finished turns that into a hang, not a wrong pixel.  With one function there is no
// second place to forget; the kernel additionally refuses to run (error word -> RT_ERR_DEVICE) when the bytes it was
// launched with are fewer than the layout needs.
//
//   [ traversal stack: stack_entries x block x entry_bytes ][ node copy or box list (optional) ][ job state: 32 B per wave ]
//   per set of queues (one per eight waves of a 1024-thread workgroup, else one):
//   [ swap header 32 B ][ class 0 | class 1 | class 2 : RT_SWAP_F64 arrays of cap doubles, then RT_SWAP_F32 arrays of cap words ]
#ifndef RT_LDS_H
#define RT_LDS_H

#include <stdint.h>

#if defined(__HIPCC__)
#define RT_LDS_HD __host__ __device__ inline
#else
#define RT_LDS_HD inline
#endif

#ifndef RT_SWAP_CAP
#define RT_SWAP_CAP 64 /* entries per class queue at most; fewer when four 256-thread groups have to share a CU's LDS */
#endif
#define RT_SWAP_CLASSES 3  /* lambertian, metal, dielectric (RT_MAT_* values 0..2): queues of finished segments waiting to be scattered */
// A fourth queue in the same format -- rays whose next segment has been set up and that wait for a lane, so that lanes which finish
// their traversal early park their hit and take one instead of idling until the wave's shade quorum -- was built and measured in
// round 3 (profiles/r03_experiments/ray_exchange_kernel.patch, docs/experiments.md section 2): slower on every scene (cornell 2533 -> 2282,
// cover 2121 -> 2077 Msamples/s).  The layout has the three class queues and nothing else: there is no queue count for the host and
// the kernel to disagree about.
#define RT_SWAP_F64 14     /* o, d, T (9), s0, s1, best_t, draws, base */
#define RT_SWAP_F32 3      /* k, best_prim, slot */
#define RT_SWAP_HDR_BYTES 32u
#define RT_SWAP_ENTRY_BYTES ((unsigned)(RT_SWAP_F64 * 8 + RT_SWAP_F32 * 4))
#define RT_JOB_BYTES_PER_WAVE 32u
#define RT_LDS_GRANULE 512u /* LDS is handed out in 512-byte granules */
#define RT_LDS_PER_CU (160u * 1024u)

struct RtLdsLayout {
    uint32_t stack_off, node_off, job_off, swap_off; // byte offsets of the regions
    uint32_t swap_class_bytes;                       // one class queue
    uint32_t swap_set_bytes;                         // one set of queues (header + classes); a workgroup has rt_swap_sets() of them
    uint32_t total;                                  // bytes the launch has to provide
};

// One 1024-thread workgroup per CU (the family with sphere media / textures since round 5) would serve sixteen waves with ONE set of
// class queues, i.e. three locks: measured on the cover, 1.86 M lock attempts given up per launch against 0.4 M with four 256-thread
// groups, and 56 % of all scatters off class.  A workgroup therefore keeps one SET of queues per eight waves (the share a 512-thread
// group's waves have): wave w parks into and pulls from set w / 8 only, each set with its own header and locks.
RT_LDS_HD constexpr uint32_t rt_swap_sets(uint32_t block_threads) { return block_threads >= 1024u ? block_threads / 512u : 1u; }
// the capacity the kernel really uses: compiled in for the families of 512 threads and more (run-time address arithmetic costs the
// book-one kernel 2 %, the cover's 1.5 %: profiles/r05_logs/ab_cover_caps.log),
// the launch's value (EVEN: a class is 124 x cap bytes and its binary64 arrays have to stay 8-byte aligned) for the others
RT_LDS_HD constexpr uint32_t rt_swap_cap_effective(uint32_t block_threads, uint32_t launch_cap) {
    return block_threads >= 512u ? (uint32_t)RT_SWAP_CAP : (launch_cap & ~1u);
}

// The kernel families with media keep glibc's 2 KB log table (rt_libm.h: 128 x {invc, logc}) at the very front of their LDS: a
// free-flight draw takes its logarithm in every medium a segment crosses, and from global memory -- behind a 217 KB node array that
// washes through the L1 -- the table's entry was an L2 access on the critical path of a call (book-two cover: +2 %).
#define RT_LDS_LOG_TABLE_BYTES 2048u
RT_LDS_HD constexpr uint32_t rt_lds_front_bytes(bool media_family) { return media_family ? RT_LDS_LOG_TABLE_BYTES : 0u; }
// Box-LIST scenes whose records -- transforms, prims, materials -- fit (RT_LIST_SCENE_MAX, and the workgroup's LDS share with the
// smallest queues: rt_api.cpp render_range): the records follow the table, in whole 256-byte steps (the stack behind them stays
// 256-byte aligned).  Other LIST scenes keep their records in global memory.
RT_LDS_HD constexpr uint32_t rt_lds_scene_room(uint32_t scene_bytes) { return (scene_bytes + 255u) & ~255u; }

// front_bytes: rt_lds_front_bytes of the family; node_bytes: 0 when the node array stays in global memory; swap_cap: 0 for the
// kernels without swap queues
RT_LDS_HD constexpr RtLdsLayout rt_lds_layout(uint32_t stack_entries, uint32_t block_threads, uint32_t stack_entry_bytes, uint32_t node_bytes,
                                              uint32_t swap_cap, uint32_t front_bytes) {
    RtLdsLayout l{};
    l.stack_off = front_bytes; // (a multiple of 256)
    l.node_off = l.stack_off + stack_entries * block_threads * stack_entry_bytes; // multiple of 256: uint4 copies stay aligned
    l.job_off = l.node_off + ((node_bytes + 15u) & ~15u);
    l.swap_off = l.job_off + (block_threads / 64u) * RT_JOB_BYTES_PER_WAVE;
    l.swap_class_bytes = RT_SWAP_ENTRY_BYTES * swap_cap;
    l.swap_set_bytes = swap_cap ? RT_SWAP_HDR_BYTES + (uint32_t)RT_SWAP_CLASSES * l.swap_class_bytes : 0u; // header + three class queues
    l.total = l.swap_off + rt_swap_sets(block_threads) * l.swap_set_bytes;
    return l;
}
// every region starts where its widest access needs it to
RT_LDS_HD constexpr bool rt_lds_layout_aligned(const RtLdsLayout &l) {
    return l.node_off % 16u == 0u && l.job_off % 16u == 0u && l.swap_off % 8u == 0u && l.swap_class_bytes % 8u == 0u && l.swap_set_bytes % 8u == 0u;
}

// The largest EVEN capacity (<= RT_SWAP_CAP) with which `groups_per_cu` workgroups of this shape still share one CU's LDS;
// 16 when even that does not fit (fewer groups will be resident).
RT_LDS_HD constexpr uint32_t rt_swap_cap_that_fits(uint32_t stack_entries, uint32_t block_threads, uint32_t stack_entry_bytes, uint32_t node_bytes,
                                                   uint32_t groups_per_cu, uint32_t front_bytes) {
    if (block_threads >= 512u) return (uint32_t)RT_SWAP_CAP;
    const uint32_t share = (RT_LDS_PER_CU / (groups_per_cu ? groups_per_cu : 1u)) & ~(RT_LDS_GRANULE - 1u);
    const uint32_t other = rt_lds_layout(stack_entries, block_threads, stack_entry_bytes, node_bytes, 0u, front_bytes).total + RT_SWAP_HDR_BYTES;
    const uint32_t per_entry = (uint32_t)RT_SWAP_CLASSES * RT_SWAP_ENTRY_BYTES;
    if (other + 16u * per_entry > share) return 16u;
    const uint32_t cap = (share - other) / per_entry;
    return (cap < (uint32_t)RT_SWAP_CAP ? cap : (uint32_t)RT_SWAP_CAP) & ~1u;
}

static_assert(rt_lds_layout_aligned(rt_lds_layout(24, 512, 4, 31 * 1024, RT_SWAP_CAP, 0)), "book-one shape");
static_assert(rt_lds_layout_aligned(rt_lds_layout(17, 256, 4, 648, 38, RT_LDS_LOG_TABLE_BYTES)), "list shape, even capacity");
static_assert(rt_lds_layout_aligned(rt_lds_layout(13, 256, 8, 0, 16, 0)), "wide references");
static_assert(rt_lds_layout(10, 256, 4, 0, 0, 0).total == 10 * 256 * 4 + 4 * RT_JOB_BYTES_PER_WAVE, "no queues: stack + job state");
static_assert(rt_lds_layout_aligned(rt_lds_layout(14, 1024, 4, 1406 * 32, RT_SWAP_CAP, RT_LDS_LOG_TABLE_BYTES)) &&
              rt_lds_layout(14, 1024, 4, 1406 * 32, RT_SWAP_CAP, RT_LDS_LOG_TABLE_BYTES).total <= RT_LDS_PER_CU, "the book-two cover: two sets of queues beside its binary16 tree");

// device error word (RtLaunch::status): set by the kernel, turned into RT_ERR_DEVICE by the host
#define RT_DEV_OK 0u
#define RT_DEV_ERR_LDS_LAYOUT 1u /* launched with fewer LDS bytes than rt_lds_layout() needs */
#define RT_DEV_ERR_WATCHDOG 2u   /* counting build: a wave went round its loop without finishing a segment for too long */

#endif
