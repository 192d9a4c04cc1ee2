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
//                    segment; a finished path stores its radiance and the lane
//                    immediately takes the next sample of the wave's job queue
//                    (jobs = 8x8-pixel tile x job_spp samples, drawn from one
//                    device-wide atomic counter).
//   The expensive blocks only run when enough lanes have queued up for them (or
//   nothing else can run), so each block executes at high SIMD occupancy instead
//   of every lane dragging the other 63 through its own branch.  Lanes never idle
//   at sample, pixel or tile boundaries; a wave retires only when the job queue
//   is empty.  Scheduling never changes a result: each path consumes its own
//   random stream (include/rt_rng.h) and every sample is an independent value.
//   The traversal stack lives in LDS, [depth][thread], conflict-free ds_read/write_b32.
//
// reduce_kernel -- pixel = (((s_0 + s_1) + s_2) + ...) / spp in sample order, the
//   rounding of `pixel += color(...)` in examples/book-one.rs:69-76.  Per-sample
//   radiance takes one 32-byte record per sample (15.4 GB for 1200x800x500; the
//   MI355X has 288 GB) written once and read once, noise next to the traversal.
//
// No MFMA anywhere: the workload has no dense contraction.

#include <hip/hip_runtime.h>

#include "rt_lane.h"
#include "rt_types.h"

#ifndef RT_BLOCK
#define RT_BLOCK 512 /* threads per workgroup: 8 waves share one LDS copy of the node array */
#endif

// vote thresholds (lanes).  A block runs when at least this many lanes wait for
// it, or when no cheaper block has work.
#ifndef RT_VOTE_SHADE
#define RT_VOTE_SHADE 48
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
#define RT_VOTE_LEAF_G 16
#endif
#ifndef RT_NODE_KEEP_G
#define RT_NODE_KEEP_G 8
#endif
#ifndef RT_WAVES_PER_EU
#define RT_WAVES_PER_EU 4 /* spheres-only kernel: 105 VGPRs fit 4 waves per SIMD */
#endif
#ifndef RT_WAVES_PER_EU_GENERAL
#define RT_WAVES_PER_EU_GENERAL 2 /* general kernel (matrices, cubes, media, textures) needs the registers */
#endif

namespace {

struct LdsStack {
    uint32_t *base; // &stack[threadIdx.x]
    __device__ __forceinline__ void push(int32_t &sp, uint32_t v) {
        base[sp * RT_BLOCK] = v;
        ++sp;
    }
    __device__ __forceinline__ uint32_t pop(int32_t &sp) {
        --sp;
        return base[sp * RT_BLOCK];
    }
};

__device__ __forceinline__ uint32_t lane_rank(unsigned long long mask) { // set bits below this lane
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}
__device__ __forceinline__ uint32_t uniform(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

// Dynamic LDS: [stack_entries][RT_BLOCK] traversal stack, then (LDSNODES) a copy of the
// whole node array: node steps are a dependent pointer chase, and an LDS read returns
// in ~1/4 of an L2 hit.
extern __shared__ __attribute__((aligned(16))) unsigned char rt_lds[];

template <bool GENERAL, bool MEDIUM, bool TEXTURED, bool LENS, bool COUNT, bool LDSNODES>
__global__ __launch_bounds__(RT_BLOCK, (GENERAL ? RT_WAVES_PER_EU_GENERAL : RT_WAVES_PER_EU)) void render_kernel(const RtLaunch L) {
    uint32_t *stack_mem = reinterpret_cast<uint32_t *>(rt_lds);
    LdsStack st;
    st.base = stack_mem + threadIdx.x;
    const RtNode *nodes = L.nodes;
    if (LDSNODES) {
        uint4 *dst = reinterpret_cast<uint4 *>(rt_lds + (size_t)L.stack_entries * RT_BLOCK * sizeof(uint32_t));
        const uint4 *src = reinterpret_cast<const uint4 *>(L.nodes);
        const int n16 = L.n_nodes * (int)(sizeof(RtNode) / 16);
        for (int i = (int)threadIdx.x; i < n16; i += RT_BLOCK) dst[i] = src[i];
        __syncthreads();
        nodes = reinterpret_cast<const RtNode *>(dst);
    }

    rtl::PathState ps;
    rtl::Trav tv;
    tv.cur = RT_CUR_DONE;
    tv.sp = 0;
    tv.best_prim = 0xFFFFFFFFu;
    tv.best_sub = 0u;
    tv.best_t = 0.0;
    bool has_path = false;
    uint32_t slot = 0; // index of this lane's sample in L.samples

    // wave-uniform job state
    uint32_t job_next = 0, job_left = 0, job_slot0 = 0, job_x0 = 0, job_y0 = 0, job_s_first = 0, job_nspp = 1;
    bool queue_empty = false;

    unsigned long long c_nodes = 0, c_prims = 0, c_segs = 0, c_draws = 0, c_samples = 0;
    unsigned long long c_nw = 0, c_nl = 0, c_lw = 0, c_ll = 0, c_sw = 0, c_sl = 0;
    unsigned long long t_n = 0, t_l = 0, t_s = 0, t_fin = 0, t_ref = 0, t_beg = 0, t0 = 0, t1 = 0;
#define RT_STAMP(v) do { if (COUNT) v = __builtin_amdgcn_s_memtime(); } while (0)
    const bool counting_lane = COUNT && (threadIdx.x & 63) == 0;

    for (;;) {
        const bool is_done = tv.cur == RT_CUR_DONE;
        const bool is_leaf = (tv.cur & (RT_REF_LEAF | RT_CUR_DONE | RT_CUR_DEAD)) == RT_REF_LEAF;
        const bool is_node = tv.cur < RT_REF_LEAF;
        const unsigned long long mS = __ballot(is_done), mL = __ballot(is_leaf), mN = __ballot(is_node);
        if ((mS | mL | mN) == 0ull) break; // every lane is dead
        const int nS = __popcll(mS), nL = __popcll(mL), nN = __popcll(mN);

        constexpr int kVoteShade = GENERAL ? RT_VOTE_SHADE_G : RT_VOTE_SHADE, kVoteLeaf = GENERAL ? RT_VOTE_LEAF_G : RT_VOTE_LEAF,
                      kNodeKeep = GENERAL ? RT_NODE_KEEP_G : RT_NODE_KEEP;
        if (nS >= kVoteShade || (nN == 0 && nL == 0)) {
            // ---------------- shade block ----------------
            if (COUNT && counting_lane) {
                ++c_sw;
                c_sl += (unsigned long long)nS;
            }
            RT_STAMP(t0);
            bool need = false;
            if (is_done) {
                if (has_path) {
                    if (COUNT) ++c_segs;
                    rtl::V3 rad;
                    if (rtl::finish_segment<GENERAL, MEDIUM, TEXTURED>(L, &ps, tv, &rad)) {
                        // one 32-byte aligned record per sample, two 16-byte stores: whole sectors,
                        // no read-modify-write of partially written lines at the memory side
                        double2 *o = reinterpret_cast<double2 *>(L.samples + (size_t)slot * 4);
                        o[0] = make_double2(rad.x, rad.y);
                        o[1] = make_double2(rad.z, 0.0);
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
            for (;;) {
                const unsigned long long m = __ballot(need);
                if (m == 0ull) break;
                if (job_left == 0u) {
                    if (queue_empty) break;
                    uint32_t j = 0;
                    if (need && lane_rank(m) == 0u) j = atomicAdd(L.job_counter, 1u);
                    j = (uint32_t)__builtin_amdgcn_readlane((int)j, __ffsll((long long)m) - 1); // fetched by the first needing lane
                    if (j >= (uint32_t)L.n_jobs) {
                        queue_empty = true;
                        break;
                    }
                    const uint32_t k = j / (uint32_t)L.jobs_per_tile, sub = j - k * (uint32_t)L.jobs_per_tile;
                    const uint32_t tile = (uint32_t)L.shard_index + k * (uint32_t)L.shard_count;
                    const uint32_t ty = tile / (uint32_t)L.tiles_x, tx = tile - ty * (uint32_t)L.tiles_x;
                    job_x0 = tx * RT_TILE_EDGE;
                    job_y0 = ty * RT_TILE_EDGE;
                    job_s_first = sub * (uint32_t)L.job_spp;
                    const uint32_t nspp = min((uint32_t)L.job_spp, (uint32_t)L.s_count - job_s_first);
                    job_left = nspp * RT_TILE_PIXELS;
                    job_nspp = nspp;
                    job_next = 0u;
                    job_slot0 = (k * (uint32_t)L.s_count + job_s_first) * RT_TILE_PIXELS;
                }
                const uint32_t take = min((uint32_t)__popcll(m), job_left);
                const uint32_t rank = lane_rank(m);
                if (need && rank < take) {
                    const uint32_t n = job_next + rank;
                    // pixel-major order inside a job: consecutive n are consecutive samples of ONE pixel, so
                    // lanes refilled together start with near-identical rays (+1 % over sample-major)
                    const uint32_t pix = n / job_nspp, sj = n - pix * job_nspp;
                    const uint32_t x = job_x0 + (pix & 7u), y = job_y0 + (pix >> 3);
                    if (x < (uint32_t)L.width && y < (uint32_t)L.height) {
                        slot = job_slot0 + sj * RT_TILE_PIXELS + pix;
                        rtl::start_sample<LENS>(L, x, y, (uint32_t)L.s0 + job_s_first + sj, &ps);
                        has_path = true;
                        need = false;
                    }
                    // a pixel outside the image consumes its slot and the lane asks again
                }
                job_next += take;
                job_left -= take;
            }
            RT_STAMP(t0);
            t_ref += t0 - t1;
            if (is_done) {
                if (has_path)
                    rtl::begin_segment<GENERAL, MEDIUM, TEXTURED>(L, &ps, tv, st, &c_prims);
                else
                    tv.cur = RT_CUR_DEAD;
            }
            RT_STAMP(t1);
            t_beg += t1 - t0;
        } else if (nL >= kVoteLeaf || nN == 0) {
            // ---------------- leaf block ----------------
            if (COUNT && counting_lane) {
                ++c_lw;
                c_ll += (unsigned long long)nL;
            }
            RT_STAMP(t0);
            if (is_leaf) rtl::leaf_step<GENERAL, MEDIUM, TEXTURED>(L, &ps, tv, st, &c_prims);
            RT_STAMP(t1);
            t_l += t1 - t0;
        } else {
            // ---------------- node block ----------------
            RT_STAMP(t0);
            for (;;) {
                const bool at_node = tv.cur < RT_REF_LEAF;
                const int n = __popcll(__ballot(at_node));
                if (n == 0) break;
                if (COUNT && counting_lane) {
                    ++c_nw;
                    c_nl += (unsigned long long)n;
                }
                if (at_node) {
                    if (COUNT) ++c_nodes;
                    rtl::trav_node_step(nodes, tv, st);
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
        if (counting_lane) {
            atomicAdd(&L.counters->node_wave, c_nw);
            atomicAdd(&L.counters->node_lane, c_nl);
            atomicAdd(&L.counters->leaf_wave, c_lw);
            atomicAdd(&L.counters->leaf_lane, c_ll);
            atomicAdd(&L.counters->shade_wave, c_sw);
            atomicAdd(&L.counters->shade_lane, c_sl);
            atomicAdd(&L.counters->node_cycles, t_n);
            atomicAdd(&L.counters->leaf_cycles, t_l);
            atomicAdd(&L.counters->shade_cycles, t_fin + t_ref + t_beg);
            atomicAdd(&L.counters->finish_cycles, t_fin);
            atomicAdd(&L.counters->refill_cycles, t_ref);
            atomicAdd(&L.counters->begin_cycles, t_beg);
        }
    }
}

// Sum this pass's samples in sample order on top of the running sums; the last
// pass divides by spp (`pixel /= subPixelSampleCount`, examples/book-one.rs:76).
// One thread per owned pixel; consecutive threads read consecutive 32-byte records.
__global__ void reduce_kernel(const double *samples, double *tiles, int n_owned_tiles, int s_count, int first_pass,
                              int last_pass, int spp, int width, int height, int tiles_x, int shard_index, int shard_count) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)n_owned_tiles * RT_TILE_PIXELS) return;
    const size_t k = i / RT_TILE_PIXELS;
    const int pix = (int)(i % RT_TILE_PIXELS);
    const int tile = shard_index + (int)k * shard_count;
    const int x = (tile % tiles_x) * RT_TILE_EDGE + (pix & 7), y = (tile / tiles_x) * RT_TILE_EDGE + (pix >> 3);
    double *out = tiles + i * 3;
    if (x >= width || y >= height) {
        out[0] = out[1] = out[2] = 0.0;
        return;
    }
    double r = 0.0, g = 0.0, b = 0.0;
    if (!first_pass) {
        r = out[0];
        g = out[1];
        b = out[2];
    }
    const double2 *p = reinterpret_cast<const double2 *>(samples + ((size_t)k * (size_t)s_count * RT_TILE_PIXELS + (size_t)pix) * 4);
    for (int s = 0; s < s_count; ++s) {
        const double2 a = p[0], c = p[1];
        r += a.x;
        g += a.y;
        b += c.x;
        p += RT_TILE_PIXELS * 2;
    }
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

// ---- dispatch over the template instantiations ----
typedef void (*KernelFn)(const RtLaunch);

template <bool GENERAL, bool MEDIUM, bool TEXTURED>
KernelFn pick3(bool lens, bool count, bool ldsnodes) {
#define RT_PICK(LN, C, LD) return render_kernel<GENERAL, MEDIUM, TEXTURED, LN, C, LD>
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
KernelFn pick(unsigned features, bool lens, bool count, bool ldsnodes) {
    return features == 0u ? pick3<false, false, false>(lens, count, ldsnodes) : pick3<true, true, true>(lens, count, ldsnodes);
}

} // namespace

// feature bits: 1 = general prims, 2 = media, 4 = textured.  `blocks` persistent workgroups,
// `lds_bytes` of dynamic LDS (stack + optional node copy).
extern "C" int rt_launch_render(const RtLaunch *L, unsigned features, int lens, int count, int ldsnodes, int blocks,
                                unsigned lds_bytes, void *stream) {
    KernelFn k = pick(features, lens != 0, count != 0, ldsnodes != 0);
    if (lds_bytes > 48u * 1024u) { // dynamic LDS beyond the default limit must be requested explicitly
        hipError_t ea = hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (ea != hipSuccess) return (int)ea;
    }
    hipLaunchKernelGGL(k, dim3((unsigned)blocks), dim3(RT_BLOCK), lds_bytes, (hipStream_t)stream, *L);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

extern "C" int rt_kernel_block_size(void) { return RT_BLOCK; }

// occupancy-derived size of the persistent grid
extern "C" int rt_persistent_blocks(unsigned features, int lens, int count, int ldsnodes, unsigned lds_bytes, int *blocks_per_cu,
                                    int *n_cu) {
    KernelFn k = pick(features, lens != 0, count != 0, ldsnodes != 0);
    int per_cu = 0;
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)k, RT_BLOCK, lds_bytes);
    if (e != hipSuccess) return (int)e;
    int dev = 0;
    hipDeviceProp_t prop;
    if ((e = hipGetDevice(&dev)) != hipSuccess) return (int)e;
    if ((e = hipGetDeviceProperties(&prop, dev)) != hipSuccess) return (int)e;
    *blocks_per_cu = per_cu;
    *n_cu = prop.multiProcessorCount;
    return 0;
}

extern "C" int rt_launch_reduce(const double *samples, double *tiles, int n_owned_tiles, int s_count, int first_pass, int last_pass,
                                int spp, int width, int height, int shard_index, int shard_count, void *stream) {
    const size_t n = (size_t)n_owned_tiles * RT_TILE_PIXELS;
    const int tiles_x = (width + RT_TILE_EDGE - 1) / RT_TILE_EDGE;
    hipLaunchKernelGGL(reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, samples, tiles,
                       n_owned_tiles, s_count, first_pass, last_pass, spp, width, height, tiles_x, shard_index, shard_count);
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

extern "C" int rt_launch_probe_math(const double *a, const double *b, int n, double *out_sqrt, double *out_div, void *stream) {
    hipLaunchKernelGGL(probe_math_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a, b, n, out_sqrt,
                       out_div);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

