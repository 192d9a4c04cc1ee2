// rt_scene_priv.h -- layout of the opaque rt_scene handle.  Private to the
// library (rt_api.cpp); tests/lane_emul.cpp includes it to read the committed
// flat scene on the host.
#ifndef RT_SCENE_PRIV_H
#define RT_SCENE_PRIV_H

#include "rt_host.h"
#include "../../include/rt_mi355x.h"

#include <hip/hip_runtime_api.h>

#include <mutex>
#include <vector>

struct rt_scene {
    rt::SceneIR ir;
    rt::FlatScene flat;
    bool committed = false;
    int device = -1;
    void *d_nodes = nullptr, *d_nodes_half = nullptr, *d_prim_meta = nullptr, *d_prim_geo = nullptr, *d_prim_extra = nullptr, *d_xforms = nullptr,
         *d_materials = nullptr, *d_textures = nullptr, *d_blob = nullptr, *d_scene_blob = nullptr;
    size_t device_bytes = 0;
    size_t workspace_limit = 0; // rt_scene_set_workspace_limit; 0 = default (rt_api.cpp sample_workspace_cap)
    size_t workspace_default = 0; // the default, once computed
    std::mutex mu;        // guards the fields below
    std::mutex render_mu; // serialises rt_render calls: they share the sample workspace
    // cached occupancy query of the last kernel variant used
    unsigned occ_key = 0xFFFFFFFFu, occ_lds = 0;
    int occ_per_cu = 0, occ_n_cu = 0;
    // A render in flight owns one slot: the per-sample radiance workspace (grown on demand, reused between renders), the
    // job counter of its launches, the HIP events bracketing each render_kernel launch, and `done`, recorded behind its
    // last kernel.  Two slots let consecutive renders on different streams overlap (the next render's workgroups fill
    // the CUs while the previous one's few 100-segment paths finish); a caller that waits for each render only ever
    // touches slot 0.
    struct RenderSlot {
        void *d_samples = nullptr;
        size_t samples_bytes = 0;
        void *d_job_counter = nullptr;
        hipEvent_t done = nullptr;
        bool used = false;
        bool deferred = false; // its last reduce ran on the scene's post stream (RT_FLAG_DEFERRED_OUTPUT)
        std::vector<hipEvent_t> events;
        int events_used = 0;
    };
    hipStream_t post_stream = nullptr; // deferred reduce_kernel launches
    // the learnt hand-out order of the owned tiles of ONE view (include/rt_mi355x.h RT_TILE_ORDER_*): written once by the render
    // that learns it (behind both slots' earlier renders), read by later renders of the same view only after `ready` has fired
    struct TileOrder {
        void *d_cost = nullptr, *d_order = nullptr; // n uint64 sums of path lengths; n uint32 owned-tile indices
        int capacity = 0, n = 0;
        uint64_t key = 0;   // of the view it belongs to (0: none)
        bool complete = false; // `ready` has been seen fired
        hipEvent_t ready = nullptr;
    } tile_order;
    RenderSlot slots[2];
    int last_slot = 0;
    bool timed = false;
    rt_launch_config last_launch{}; // of the last render_range call

    void release_device() {
        if (device >= 0) {
            (void)hipSetDevice(device);
            for (void **p : {&d_nodes, &d_nodes_half, &d_prim_meta, &d_prim_geo, &d_prim_extra, &d_xforms, &d_materials, &d_textures, &d_blob, &d_scene_blob}) {
                if (*p) (void)hipFree(*p);
                *p = nullptr;
            }
            if (post_stream) (void)hipStreamDestroy(post_stream);
            post_stream = nullptr;
            if (tile_order.d_cost) (void)hipFree(tile_order.d_cost);
            if (tile_order.d_order) (void)hipFree(tile_order.d_order);
            if (tile_order.ready) (void)hipEventDestroy(tile_order.ready);
            tile_order = TileOrder();
            for (RenderSlot &sl : slots) {
                if (sl.d_samples) (void)hipFree(sl.d_samples);
                if (sl.d_job_counter) (void)hipFree(sl.d_job_counter);
                if (sl.done) (void)hipEventDestroy(sl.done);
                for (hipEvent_t e : sl.events) (void)hipEventDestroy(e);
                sl = RenderSlot();
            }
        }
        device_bytes = 0;
    }
};

#endif
