// rt_scene_priv.h -- layout of the opaque rt_scene handle.  Private to the
// library (rt_api.cpp); tests/lane_emul.cpp includes it to read the committed
// flat scene on the host.
#ifndef RT_SCENE_PRIV_H
#define RT_SCENE_PRIV_H

#include "rt_host.h"

#include <hip/hip_runtime_api.h>

#include <mutex>

struct rt_scene {
    rt::SceneIR ir;
    rt::FlatScene flat;
    bool committed = false;
    int device = -1;
    void *d_nodes = nullptr, *d_prims = nullptr, *d_xforms = nullptr, *d_materials = nullptr, *d_textures = nullptr,
         *d_blob = nullptr;
    size_t device_bytes = 0;
    std::mutex mu;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool have_events = false, timed = false;

    void release_device() {
        if (device >= 0) {
            (void)hipSetDevice(device);
            for (void **p : {&d_nodes, &d_prims, &d_xforms, &d_materials, &d_textures, &d_blob}) {
                if (*p) (void)hipFree(*p);
                *p = nullptr;
            }
            if (have_events) {
                (void)hipEventDestroy(ev0);
                (void)hipEventDestroy(ev1);
                have_events = false;
            }
        }
        device_bytes = 0;
    }
};

#endif
