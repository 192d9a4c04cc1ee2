// rt_scene_priv.h -- layout of the opaque rt_scene handle.  Private to the
// library (rt_api.cpp); tests/lane_emul.cpp includes it to read the committed
// flat scene on the host.
#ifndef RT_SCENE_PRIV_H
#define RT_SCENE_PRIV_H

#include "rt_host.h"

#include <hip/hip_runtime_api.h>

#include <mutex>
#include <vector>

struct rt_scene {
    rt::SceneIR ir;
    rt::FlatScene flat;
    bool committed = false;
    int device = -1;
    void *d_nodes = nullptr, *d_prim_meta = nullptr, *d_prim_geo = nullptr, *d_prim_extra = nullptr, *d_xforms = nullptr,
         *d_materials = nullptr, *d_textures = nullptr, *d_blob = nullptr;
    size_t device_bytes = 0;
    std::mutex mu;        // guards the fields below
    std::mutex render_mu; // serialises rt_render calls: they share the sample workspace
    // cached occupancy query of the last kernel variant used
    unsigned occ_key = 0xFFFFFFFFu, occ_lds = 0;
    int occ_per_cu = 0, occ_n_cu = 0;
    // per-sample radiance workspace (grown on demand, reused between renders) + job counter
    void *d_samples = nullptr;
    size_t samples_bytes = 0;
    void *d_job_counter = nullptr;
    // HIP events bracketing each render_kernel launch of the last render
    std::vector<hipEvent_t> events;
    int events_used = 0;
    bool timed = false;

    void release_device() {
        if (device >= 0) {
            (void)hipSetDevice(device);
            for (void **p : {&d_nodes, &d_prim_meta, &d_prim_geo, &d_prim_extra, &d_xforms, &d_materials, &d_textures, &d_blob,
                             &d_samples, &d_job_counter}) {
                if (*p) (void)hipFree(*p);
                *p = nullptr;
            }
            for (hipEvent_t e : events) (void)hipEventDestroy(e);
            events.clear();
            samples_bytes = 0;
        }
        device_bytes = 0;
    }
};

#endif
