// common.hpp -- shared driver of the three example programs: the reference's `main()`s
// (examples/book-one.rs:25-101, cornell-box.rs:24-204, main.rs:38-154) have every parameter
// as a literal; here they are flags with the reference's values as defaults.
#ifndef RT_EXAMPLES_COMMON_HPP
#define RT_EXAMPLES_COMMON_HPP

#include "../ray_tracer.hpp"
#include "../../../include/rt_rng.h"

#include <cstdio>
#include <unistd.h>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <exception>
#include <string>
#include <thread>
#include <typeinfo>
#include <vector>

namespace rtx {
using namespace ray_tracer;

struct Options {
    int width, height, spp, depth = 100, device = 0;
    int gpus = 1; // one host thread + one committed copy of the scene per GPU, shards of 8x8 tiles dealt tile_id % gpus
    uint64_t seed = 1, scene_seed = 1;
    std::string out = "-";
    std::string checkpoint; // raw sums + progress, rewritten after every pass; an existing matching file is resumed
    int passes = 1;
    bool describe = false;
    bool png_on_stdout = false; // examples/main.rs writes its PNG to stdout (`> image.png`); the other drivers print P3 text
};

inline Options parse(int argc, char **argv, int w, int h, int spp) {
    Options o;
    o.width = w;
    o.height = h;
    o.spp = spp;
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        auto next = [&]() -> const char * {
            if (i + 1 >= argc) {
                std::fprintf(stderr, "missing value for %s\n", a.c_str());
                std::exit(2);
            }
            return argv[++i];
        };
        if (a == "--width") o.width = std::atoi(next());
        else if (a == "--height") o.height = std::atoi(next());
        else if (a == "--spp") o.spp = std::atoi(next());
        else if (a == "--depth") o.depth = std::atoi(next());
        else if (a == "--device") o.device = std::atoi(next());
        else if (a == "--gpus") o.gpus = std::atoi(next());
        else if (a == "--seed") o.seed = std::strtoull(next(), nullptr, 10);
        else if (a == "--scene-seed") o.scene_seed = std::strtoull(next(), nullptr, 10);
        else if (a == "--out") o.out = next();
        else if (a == "--passes") o.passes = std::atoi(next());
        else if (a == "--checkpoint") o.checkpoint = next();
        else if (a == "--describe") o.describe = true;
        else {
            std::fprintf(stderr, "usage: %s [--width W] [--height H] [--spp N] [--depth D] [--seed S] [--scene-seed S] "
                                 "[--device I] [--gpus N] [--out file.ppm|file.png|-] [--passes N] [--checkpoint file] [--describe]\n", argv[0]);
            std::exit(2);
        }
    }
    return o;
}

// host-side generator of include/rt_rng.h in place of thread_rng()
struct SceneRng {
    rt_rng g;
    explicit SceneRng(uint64_t seed) { rt_rng_init(&g, seed, RT_RNG_SCENE_STREAM); }
    double gen_range(double lo, double hi) { return rt_rng_gen_range(&g, lo, hi); }
};

// canonical text dump of the sprite list (tests compare it with the Python scene description)
inline void describe_texture(const Texture *t) {
    if (auto s = dynamic_cast<const SolidColor *>(t)) std::printf("solid(%.17g,%.17g,%.17g)", s->color.x, s->color.y, s->color.z);
    else if (auto c = dynamic_cast<const CheckerTexture *>(t)) {
        std::printf("checker(");
        describe_texture(c->black.get());
        std::printf(",");
        describe_texture(c->white.get());
        std::printf(")");
    } else if (auto im = dynamic_cast<const ImageTexture *>(t)) {
        unsigned long long sum = 0;
        for (size_t i = 0; i < im->rgb.size(); ++i) sum = sum * 1315423911ull + im->rgb[i];
        std::printf("image(%d,%d,%llu)", im->w, im->h, sum);
    }
}
inline void describe(const std::vector<SpritePtr> &sprites, const PerspectiveCamera &cam) {
    for (const SpritePtr &sp : sprites) {
        const Geometry *g = sp->geometry_.get();
        if (auto s = dynamic_cast<const Sphere *>(g)) std::printf("sphere(%.17g)", s->radius);
        else if (auto r = dynamic_cast<const Rectangle *>(g)) std::printf("rectangle(%.17g,%.17g)", r->width, r->height);
        else if (auto c = dynamic_cast<const Cube *>(g)) std::printf("cube(%.17g,%.17g,%.17g)", c->width, c->height, c->depth);
        else if (auto m = dynamic_cast<const ConstantMedium *>(g))
            std::printf("medium(sphere(%.17g),%.17g)", dynamic_cast<const Sphere *>(m->boundary.get())->radius, m->density);
        else std::printf("none");
        std::printf(" ");
        const Material *mt = sp->material_.get();
        if (auto l = dynamic_cast<const Lambertian *>(mt)) {
            std::printf("lambertian(");
            describe_texture(l->albedo.get());
            std::printf(")");
        } else if (auto me = dynamic_cast<const Metal *>(mt)) {
            std::printf("metal(");
            describe_texture(me->albedo.get());
            std::printf(",%.17g)", me->fuzziness);
        } else if (auto d = dynamic_cast<const Dielectric *>(mt)) std::printf("dielectric(%.17g)", d->refractive);
        else if (auto dl = dynamic_cast<const DiffuseLight *>(mt)) {
            std::printf("diffuse_light(");
            describe_texture(dl->emission.get());
            std::printf(")");
        } else if (auto is = dynamic_cast<const Isotropic *>(mt)) {
            std::printf("isotropic(");
            describe_texture(is->albedo.get());
            std::printf(")");
        } else std::printf("none");
        for (double v : sp->transform_.a) std::printf(" %.17g", v);
        std::printf("\n");
    }
    std::printf("camera");
    for (double v : cam.c.eye) std::printf(" %.17g", v);
    for (double v : cam.c.lower_left) std::printf(" %.17g", v);
    for (double v : cam.c.horizontal) std::printf(" %.17g", v);
    for (double v : cam.c.vertical) std::printf(" %.17g", v);
    std::printf(" %.17g\n", cam.c.lens_radius);
}

inline int run(const Options &o, const std::vector<SpritePtr> &sprites, const PerspectiveCamera &camera) {
    if (o.describe) {
        describe(sprites, camera);
        auto w = BoundingVolumeHierarchyNode::make(sprites, -1);
        if (!w) return 1;
        rt_scene_info i = w->info();
        std::printf("info prims=%d hoisted=%d nodes=%d child_prims=%d\n", i.n_prims, i.n_hoisted, i.n_nodes, i.n_child_prims);
        return 0;
    }
    auto world = BoundingVolumeHierarchyNode::make(sprites, o.device); // .unwrap() in the reference
    if (!world) {
        std::fprintf(stderr, "empty scene\n");
        return 1;
    }
    std::vector<Vec3> buffer;
    if (o.gpus > 1) {
        // The reference deals rows to host threads (examples/book-one.rs:56-65); here tiles are dealt to GPUs
        // (rt_render_sharded: a host thread per committed copy of the scene, shards written straight into one image --
        // their pixels are disjoint, so there is nothing to merge and no collective); the image does not depend on N.
        const int n_dev = rt_device_count();
        if (n_dev < 1) {
            std::fprintf(stderr, "no HIP device\n");
            return 1;
        }
        if (o.passes > 1 || !o.checkpoint.empty()) {
            std::fprintf(stderr, "--gpus cannot be combined with --passes / --checkpoint\n");
            return 2;
        }
        // one committed copy per GPU; devices wrap around (--gpus 2 on a one-GPU box renders both shards on device 0)
        std::vector<BoundingVolumeHierarchyNode> worlds;
        worlds.push_back(*world);
        for (int r = 1; r < o.gpus; ++r) worlds.push_back(world->clone((o.device + r) % n_dev));
        buffer = render_sharded(worlds, camera, o.width, o.height, o.spp, o.depth, o.seed);
    } else if (o.passes <= 1 && o.checkpoint.empty()) {
        buffer = render(*world, camera, o.width, o.height, o.spp, o.depth, o.seed);
    } else {
        // progressive: the reference's only progress report is main.rs printing finished rows to stderr
        // (examples/main.rs:123-125) and it cannot resume; here every pass can be checkpointed
        // checkpoint = header + raw sums.  The header names the render AND the scene (a resume with another --scene-seed, or
        // another driver at the same size, must not add samples of a different picture); the file is replaced atomically
        // (tmp + rename) and a file that does not match or is damaged is refused, never overwritten.
        // ... and the library's DEVICE code (the "kernels <hash>" token of rt_version: sample streams, keying and arithmetic live
        // there): sums made by other kernels are never continued.
        struct Header {
            uint64_t magic, version, width, height, spp, depth, seed, scene_seed, scene_hash, camera_hash, kernels_hash, s_done;
        } h{0x3250545252ull, 3, (uint64_t)o.width, (uint64_t)o.height, (uint64_t)o.spp, (uint64_t)o.depth, o.seed, o.scene_seed,
            world->hash(), 0, 0, 0};
        {
            const std::string v = rt_version();
            const size_t at = v.find("kernels ");
            const std::string token = at == std::string::npos ? v : v.substr(at, v.find(' ', at + 8) - at);
            uint64_t kh = 0xcbf29ce484222325ull;
            for (unsigned char c : token) kh = (kh ^ c) * 0x100000001b3ull;
            h.kernels_hash = kh;
        }
        {
            uint64_t ch = 0xcbf29ce484222325ull;
            const unsigned char *cb = (const unsigned char *)&camera.c;
            for (size_t i = 0; i < sizeof camera.c; ++i) ch = (ch ^ cb[i]) * 0x100000001b3ull;
            h.camera_hash = ch;
        }
        std::vector<double> sums((size_t)o.width * o.height * 3, 0.0);
        int done = 0;
        if (!o.checkpoint.empty()) {
            if (FILE *f = std::fopen(o.checkpoint.c_str(), "rb")) {
                Header g{};
                const bool header_ok = std::fread(&g, sizeof g, 1, f) == 1;
                const bool same = header_ok && g.magic == h.magic && g.version == h.version && g.width == h.width && g.height == h.height &&
                                  g.spp == h.spp && g.depth == h.depth && g.seed == h.seed && g.scene_seed == h.scene_seed &&
                                  g.scene_hash == h.scene_hash && g.camera_hash == h.camera_hash && g.kernels_hash == h.kernels_hash &&
                                  g.s_done <= h.spp;
                const bool whole = same && std::fread(sums.data(), sizeof(double), sums.size(), f) == sums.size();
                std::fclose(f);
                if (!same) {
                    std::fprintf(stderr, "checkpoint %s belongs to another render (size, spp, depth, seed, scene, camera or the library's kernels differ) -- "
                                         "not touching it; remove it or pass another --checkpoint\n", o.checkpoint.c_str());
                    return 3;
                }
                if (!whole) {
                    std::fprintf(stderr, "checkpoint %s is truncated -- not touching it\n", o.checkpoint.c_str());
                    return 3;
                }
                done = (int)g.s_done;
                std::fprintf(stderr, "resuming %s at %d / %d spp\n", o.checkpoint.c_str(), done, o.spp);
            }
        }
        const int passes = o.passes < 1 ? 1 : o.passes;
        const int per = (o.spp + passes - 1) / passes;
        while (done < o.spp) {
            const int end = done + per < o.spp ? done + per : o.spp;
            render_progressive(*world, camera, o.width, o.height, o.spp, o.depth, o.seed, done, end, sums);
            done = end;
            std::fprintf(stderr, "%d / %d spp\n", done, o.spp);
            if (!o.checkpoint.empty()) {
                const std::string tmp = o.checkpoint + ".tmp";
                FILE *f = std::fopen(tmp.c_str(), "wb");
                h.s_done = (uint64_t)done;
                bool ok = f != nullptr;
                ok = ok && std::fwrite(&h, sizeof h, 1, f) == 1;
                ok = ok && std::fwrite(sums.data(), sizeof(double), sums.size(), f) == sums.size();
                ok = ok && std::fflush(f) == 0;
                ok = ok && ::fsync(::fileno(f)) == 0; // on the disk BEFORE it takes the old file's name: a power loss leaves one whole file
                if (f) ok = (std::fclose(f) == 0) && ok;
                ok = ok && std::rename(tmp.c_str(), o.checkpoint.c_str()) == 0;
                if (!ok) {
                    std::fprintf(stderr, "cannot write checkpoint %s (the previous one, if any, is intact)\n", o.checkpoint.c_str());
                    std::remove(tmp.c_str());
                    return 3;
                }
            }
        }
        buffer.resize((size_t)o.width * o.height);
        const double n = (double)o.spp;
        for (size_t i = 0; i < buffer.size(); ++i) buffer[i] = Vec3(sums[i * 3] / n, sums[i * 3 + 1] / n, sums[i * 3 + 2] / n);
    }
    const std::string path = o.out == "-" ? "/dev/stdout" : o.out;
    if ((o.out == "-" && o.png_on_stdout) || (path.size() > 4 && path.compare(path.size() - 4, 4, ".png") == 0))
        write_png(path, buffer, o.width, o.height); // RGBA8 like examples/main.rs:105-135
    else
        write_ppm(path, buffer, o.width, o.height); // P3 text like println! in the reference
    return 0;
}

} // namespace rtx
#endif
