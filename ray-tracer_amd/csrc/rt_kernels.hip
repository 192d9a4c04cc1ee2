// rt_kernels.hip -- HIP kernels for gfx950 (MI355X).  The only compute path of
// librt_mi355x.so: there is no CPU fallback.
//
// render_kernel: one 64-lane wavefront owns one 8x8 pixel tile; lane = pixel.
// Every lane runs ONE flattened loop over (sample, path segment): when a path
// ends the lane folds its radiance into the pixel sum -- in sample order, so the
// sum rounds exactly like `pixel += color(...)` in examples/book-one.rs:69-75 --
// and regenerates the next sample's camera ray in place.  Lanes never wait for
// each other at sample boundaries; the wave retires when its 64 pixels are done.
// The BVH traversal stack lives in LDS, laid out [depth][thread] so that a
// push/pop by all 64 lanes is one conflict-free ds_write/ds_read_b32.
// No MFMA anywhere: the workload has no dense contraction.

#include <hip/hip_runtime.h>

#include "rt_lane.h"
#include "rt_types.h"

#define RT_BLOCK 256 /* 4 waves = 4 tiles per workgroup */

namespace {

struct LdsStack {
    int32_t *base; // &stack[threadIdx.x]
    int sp;
    __device__ __forceinline__ void reset() { sp = 0; }
    __device__ __forceinline__ void push(int32_t v) {
        base[sp * RT_BLOCK] = v;
        ++sp;
    }
    __device__ __forceinline__ int32_t pop() {
        --sp;
        return base[sp * RT_BLOCK];
    }
    __device__ __forceinline__ bool empty() const { return sp == 0; }
};

template <bool GENERAL, bool MEDIUM, bool TEXTURED, bool LENS, bool COUNT>
__global__ __launch_bounds__(RT_BLOCK) void render_kernel(const RtLaunch L) {
    __shared__ int32_t stack_mem[RT_STACK_DEPTH * RT_BLOCK];

    const int lane = threadIdx.x & 63;
    const int k = blockIdx.x * (RT_BLOCK / 64) + (threadIdx.x >> 6); // ordinal of the owned tile
    if (k >= L.n_owned_tiles) return;                                // wave-uniform
    const int tile = L.shard_index + k * L.shard_count;
    const int tx = tile % L.tiles_x, ty = tile / L.tiles_x;
    const uint32_t x = (uint32_t)(tx * RT_TILE_EDGE + (lane & 7));
    const uint32_t y = (uint32_t)(ty * RT_TILE_EDGE + (lane >> 3));
    const bool inside = x < (uint32_t)L.width && y < (uint32_t)L.height;

    LdsStack st;
    st.base = stack_mem + threadIdx.x;
    st.sp = 0;

    rtl::V3 acc = rtl::mk(0.0, 0.0, 0.0);
    rtl::PathState ps;
    unsigned long long nodes = 0, prims = 0, segs = 0, draws = 0, witers = 0;

    int s = inside ? 0 : L.spp;
    bool fresh = true;
    while (s < L.spp) {
        if (fresh) {
            rtl::start_sample<LENS>(L, x, y, (uint32_t)s, &ps);
            fresh = false;
        }
        if (COUNT) {
            ++segs;
            // count this wave-iteration once: by the lowest active lane
            if (__ffsll((unsigned long long)__ballot(1)) - 1 == lane) ++witers;
        }
        const bool done = rtl::advance_segment<GENERAL, MEDIUM, TEXTURED>(L, &ps, st, &nodes, &prims);
        if (done) {
            acc = acc + ps.Lsum; // pixel += color(...)
            if (COUNT) draws += ps.g.draws;
            ++s;
            fresh = true;
        }
    }

    double *o = L.out + ((size_t)k * RT_TILE_PIXELS + (size_t)lane) * 3;
    if (inside) {
        const double n = (double)L.spp; // pixel /= subPixelSampleCount
        o[0] = acc.x / n;
        o[1] = acc.y / n;
        o[2] = acc.z / n;
    } else {
        o[0] = 0.0;
        o[1] = 0.0;
        o[2] = 0.0;
    }
    if (COUNT && L.counters) {
        if (inside) atomicAdd(&L.counters->samples, (unsigned long long)L.spp);
        atomicAdd(&L.counters->segments, segs);
        atomicAdd(&L.counters->nodes_visited, nodes);
        atomicAdd(&L.counters->prims_tested, prims);
        atomicAdd(&L.counters->rng_draws, draws);
        atomicAdd(&L.counters->wave_iterations, witers);
        atomicAdd(&L.counters->lane_iterations, segs);
    }
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

template <bool GENERAL, bool MEDIUM, bool TEXTURED>
hipError_t launch3(const RtLaunch &L, bool lens, bool count, hipStream_t stream) {
    const int blocks = (L.n_owned_tiles + (RT_BLOCK / 64) - 1) / (RT_BLOCK / 64);
    if (blocks <= 0) return hipSuccess;
    dim3 g((unsigned)blocks), b(RT_BLOCK);
    if (lens) {
        if (count)
            hipLaunchKernelGGL((render_kernel<GENERAL, MEDIUM, TEXTURED, true, true>), g, b, 0, stream, L);
        else
            hipLaunchKernelGGL((render_kernel<GENERAL, MEDIUM, TEXTURED, true, false>), g, b, 0, stream, L);
    } else {
        if (count)
            hipLaunchKernelGGL((render_kernel<GENERAL, MEDIUM, TEXTURED, false, true>), g, b, 0, stream, L);
        else
            hipLaunchKernelGGL((render_kernel<GENERAL, MEDIUM, TEXTURED, false, false>), g, b, 0, stream, L);
    }
    return hipGetLastError();
}

} // namespace

// feature bits: 1 = general prims, 2 = media, 4 = textured
extern "C" int rt_launch_render(const RtLaunch *L, unsigned features, int lens, int count, void *stream) {
    hipStream_t s = (hipStream_t)stream;
    hipError_t e;
    if (features == 0u)
        e = launch3<false, false, false>(*L, lens != 0, count != 0, s);
    else
        e = launch3<true, true, true>(*L, lens != 0, count != 0, s);
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
