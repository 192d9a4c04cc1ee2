// ray_tracer.hpp -- C++17 host-side mirror of the reference's builder API over the
// C ABI of librt_mi355x (include/rt_mi355x.h).  Header-only.  Names, argument meaning
// and error behaviour follow aiifabbf/ray-tracer so the example drivers can be restated
// one to one (the reference is Rust; no Rust toolchain exists in the build image):
//
//   Vec3, Mat4::{identity,translation,rotation,multiplied}       src/vec3.rs, src/mat4.rs
//   SolidColor / CheckerTexture / ImageTexture                   src/material.rs:196-271
//   Lambertian / Metal / Dielectric / DiffuseLight / Isotropic   src/material.rs:24-326
//   Sphere / Rectangle / Cube / ConstantMedium                   src/geometry.rs, src/volume.rs
//   TransformedGeometry, NodeGeometry (a node as a geometry)     src/geometry.rs:185-246, src/sprite.rs:87-93
//   Sprite::builder().geometry().material().transform().build() src/sprite.rs:22-72
//   BoundingVolumeHierarchyNode::make(objects) -> optional       src/optimize.rs:366 (None on empty input)
//   PerspectiveCamera(eye, center, up, fov, aspect, focus, lens) src/camera.rs:25-33
//   render(world, camera, W, H, spp, max_depth, seed)            the loop of examples/book-one.rs:56-88
//
// Shared ownership: the reference shares geometries / materials by Arc::clone; here a
// std::shared_ptr is interned by identity, so sharing one object yields one record.
#ifndef RAY_TRACER_HPP
#define RAY_TRACER_HPP

#include "../../include/rt_mi355x.h"

#include <array>
#include <cstdint>
#include <map>
#include <memory>
#include <optional>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace ray_tracer {

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string &m) : std::runtime_error(m), code(c) {}
};
inline int check(int rc) {
    if (rc < 0) throw Error(rc, rt_last_error());
    return rc;
}

struct Vec3 {
    double x = 0, y = 0, z = 0;
    Vec3() = default;
    Vec3(double x_, double y_, double z_) : x(x_), y(y_), z(z_) {}
    static Vec3 ex() { return {1, 0, 0}; }
    static Vec3 ey() { return {0, 1, 0}; }
    static Vec3 ez() { return {0, 0, 1}; }
    std::array<double, 3> arr() const { return {x, y, z}; }
};

struct Mat4 { // column-major, src/mat4.rs:5-17
    std::array<double, 16> a{};
    static Mat4 identity() {
        Mat4 m;
        rt_mat4_identity(m.a.data());
        return m;
    }
    static Mat4 translation(Vec3 t) {
        Mat4 m;
        auto v = t.arr();
        rt_mat4_translation(v.data(), m.a.data());
        return m;
    }
    static Mat4 rotation(double radians, Vec3 axis) {
        Mat4 m;
        auto v = axis.arr();
        rt_mat4_rotation(radians, v.data(), m.a.data());
        return m;
    }
    Mat4 multiplied(const Mat4 &o) const { // self * other
        Mat4 m;
        rt_mat4_multiplied(a.data(), o.a.data(), m.a.data());
        return m;
    }
    double determinant() const { return rt_mat4_determinant(a.data()); }
    std::optional<Mat4> inversed() const {
        Mat4 m;
        if (rt_mat4_inversed(a.data(), m.a.data()) != RT_OK) return std::nullopt;
        return m;
    }
};
inline double to_radians(double deg) { return deg * (3.14159265358979323846264338327950288 / 180.0); }

// ---- textures ----
struct Texture {
    virtual ~Texture() = default;
    virtual int record(rt_scene *s, std::map<const void *, int> &seen) const = 0;
};
using TexturePtr = std::shared_ptr<const Texture>;
template <class T>
int intern(rt_scene *s, std::map<const void *, int> &seen, const T *obj) {
    auto it = seen.find(obj);
    if (it != seen.end()) return it->second;
    int id = obj->record(s, seen);
    seen[obj] = id;
    return id;
}
struct SolidColor : Texture {
    Vec3 color;
    explicit SolidColor(Vec3 c) : color(c) {}
    int record(rt_scene *s, std::map<const void *, int> &) const override {
        auto v = color.arr();
        return check(rt_add_texture_solid(s, v.data()));
    }
};
struct CheckerTexture : Texture {
    TexturePtr black, white;
    CheckerTexture(TexturePtr b, TexturePtr w) : black(std::move(b)), white(std::move(w)) {}
    int record(rt_scene *s, std::map<const void *, int> &seen) const override {
        int b = intern(s, seen, black.get()), w = intern(s, seen, white.get());
        return check(rt_add_texture_checker(s, b, w));
    }
};
// the reference's ImageTexture holds a closure; its one use is the nearest-texel lookup of an
// RGB8 image (examples/main.rs:267-280), which is what crosses the boundary
struct ImageTexture : Texture {
    std::vector<uint8_t> rgb;
    int w, h;
    ImageTexture(std::vector<uint8_t> data, int w_, int h_) : rgb(std::move(data)), w(w_), h(h_) {}
    int record(rt_scene *s, std::map<const void *, int> &) const override {
        return check(rt_add_texture_image_rgb8(s, rgb.data(), w, h));
    }
};
inline TexturePtr solid(Vec3 c) { return std::make_shared<SolidColor>(c); } // impl Into<Arc<dyn Texture>> for Vec3

// ---- materials ----
struct Material {
    virtual ~Material() = default;
    virtual int record(rt_scene *s, std::map<const void *, int> &seen) const = 0;
};
using MaterialPtr = std::shared_ptr<const Material>;
struct Lambertian : Material {
    TexturePtr albedo;
    explicit Lambertian(TexturePtr t) : albedo(std::move(t)) {}
    explicit Lambertian(Vec3 c) : albedo(solid(c)) {}
    int record(rt_scene *s, std::map<const void *, int> &seen) const override {
        return check(rt_add_material_lambertian(s, intern(s, seen, albedo.get())));
    }
};
struct Metal : Material {
    TexturePtr albedo;
    double fuzziness;
    Metal(TexturePtr t, double f) : albedo(std::move(t)), fuzziness(f) {}
    Metal(Vec3 c, double f) : albedo(solid(c)), fuzziness(f) {}
    int record(rt_scene *s, std::map<const void *, int> &seen) const override {
        return check(rt_add_material_metal(s, intern(s, seen, albedo.get()), fuzziness));
    }
};
struct Dielectric : Material {
    double refractive;
    explicit Dielectric(double r) : refractive(r) {}
    int record(rt_scene *s, std::map<const void *, int> &) const override { return check(rt_add_material_dielectric(s, refractive)); }
};
struct DiffuseLight : Material {
    TexturePtr emission;
    explicit DiffuseLight(TexturePtr t) : emission(std::move(t)) {}
    explicit DiffuseLight(Vec3 c) : emission(solid(c)) {}
    int record(rt_scene *s, std::map<const void *, int> &seen) const override {
        return check(rt_add_material_diffuse_light(s, intern(s, seen, emission.get())));
    }
};
struct Isotropic : Material {
    TexturePtr albedo;
    explicit Isotropic(TexturePtr t) : albedo(std::move(t)) {}
    explicit Isotropic(Vec3 c) : albedo(solid(c)) {}
    int record(rt_scene *s, std::map<const void *, int> &seen) const override {
        return check(rt_add_material_isotropic(s, intern(s, seen, albedo.get())));
    }
};

// ---- geometries ----
struct Geometry {
    virtual ~Geometry() = default;
    virtual int record(rt_scene *s, std::map<const void *, int> &seen) const = 0;
};
using GeometryPtr = std::shared_ptr<const Geometry>;
struct Sphere : Geometry {
    double radius;
    explicit Sphere(double r) : radius(r) {}
    int record(rt_scene *s, std::map<const void *, int> &) const override { return check(rt_add_geometry_sphere(s, radius)); }
};
struct Rectangle : Geometry {
    double width, height;
    Rectangle(double w, double h) : width(w), height(h) {}
    int record(rt_scene *s, std::map<const void *, int> &) const override { return check(rt_add_geometry_rectangle(s, width, height)); }
};
// BoundingVolumeHierarchyNode::new(Cube::new(w, h, d)) as the examples wrap it
struct Cube : Geometry {
    double width, height, depth;
    Cube(double w, double h, double d) : width(w), height(h), depth(d) {}
    int record(rt_scene *s, std::map<const void *, int> &) const override { return check(rt_add_geometry_cube(s, width, height, depth)); }
};
struct ConstantMedium : Geometry {
    GeometryPtr boundary;
    double density;
    ConstantMedium(GeometryPtr b, double d) : boundary(std::move(b)), density(d) {}
    int record(rt_scene *s, std::map<const void *, int> &seen) const override {
        return check(rt_add_geometry_constant_medium(s, intern(s, seen, boundary.get()), density));
    }
};

// TransformedGeometry::new(geometry, M) (src/geometry.rs:185-246)
struct TransformedGeometry : Geometry {
    GeometryPtr geometry;
    Mat4 transform;
    TransformedGeometry(GeometryPtr g, const Mat4 &m) : geometry(std::move(g)), transform(m) {}
    int record(rt_scene *s, std::map<const void *, int> &seen) const override {
        return check(rt_add_geometry_transformed(s, intern(s, seen, geometry.get()), transform.a.data()));
    }
};

// ---- Sprite + builder (src/sprite.rs:22-72) ----
struct Sprite {
    GeometryPtr geometry_; // Option<Arc<T>>
    MaterialPtr material_; // Option<Arc<U>>
    Mat4 transform_ = Mat4::identity();
    struct Builder;
    static Builder builder();
    // records geometry, material and the sprite itself; returns the sprite id
    int record(rt_scene *s, std::map<const void *, int> &seen) const {
        int g = geometry_ ? intern(s, seen, geometry_.get()) : -1;
        int m = material_ ? intern(s, seen, material_.get()) : -1;
        return check(rt_add_sprite(s, g, m, transform_.a.data()));
    }
};
// BoundingVolumeHierarchyNode::new(vec![sprites...]) used as the GEOMETRY of a sprite -- instancing
// (src/sprite.rs:87-93 with T = BoundingVolumeHierarchyNode, examples/cornell-box.rs:85-101).  Every sprite that carries
// the same NodeGeometry is one more instance of it.
struct NodeGeometry : Geometry {
    std::vector<std::shared_ptr<Sprite>> children;
    explicit NodeGeometry(std::vector<std::shared_ptr<Sprite>> c) : children(std::move(c)) {}
    int record(rt_scene *s, std::map<const void *, int> &seen) const override {
        std::vector<int> ids;
        for (const auto &c : children) ids.push_back(c->record(s, seen));
        return check(rt_add_geometry_bvh(s, ids.data(), (int)ids.size()));
    }
};
struct Sprite::Builder {
    Sprite sprite;
    Builder geometry(GeometryPtr g) && {
        sprite.geometry_ = std::move(g);
        return std::move(*this);
    }
    Builder material(MaterialPtr m) && {
        sprite.material_ = std::move(m);
        return std::move(*this);
    }
    Builder transform(const Mat4 &m) && {
        sprite.transform_ = m;
        return std::move(*this);
    }
    std::shared_ptr<Sprite> build() && { return std::make_shared<Sprite>(std::move(sprite)); }
};
inline Sprite::Builder Sprite::builder() { return Builder{}; }
using SpritePtr = std::shared_ptr<Sprite>;

// ---- camera ----
struct PerspectiveCamera {
    rt_camera c{};
    PerspectiveCamera(Vec3 eye, Vec3 center, Vec3 up, double fov, double aspect, double focusDistance, double lensRadius) {
        auto e = eye.arr(), ce = center.arr(), u = up.arr();
        check(rt_camera_perspective(&c, e.data(), ce.data(), u.data(), fov, aspect, focusDistance, lensRadius));
    }
};

// ---- world ----
class BoundingVolumeHierarchyNode {
  public:
    // BoundingVolumeHierarchyNode::new(objects): nullopt for an empty list (src/optimize.rs:367-370).
    // Nested nodes without a transform (examples/main.rs:191,303) are the same flat set of sprites:
    // pass them all.  device = -1 records and flattens only (no GPU needed).
    static std::optional<BoundingVolumeHierarchyNode> make(const std::vector<SpritePtr> &objects, int device = 0) {
        BoundingVolumeHierarchyNode w;
        w.scene_.reset(rt_scene_create(), rt_scene_destroy);
        std::map<const void *, int> seen;
        for (const SpritePtr &sp : objects) sp->record(w.scene_.get(), seen);
        int rc = rt_scene_commit(w.scene_.get(), device);
        if (rc == RT_ERR_EMPTY) return std::nullopt;
        check(rc);
        return w;
    }
    rt_scene *raw() const { return scene_.get(); }
    // a second committed copy on another device (rt_scene_clone): the Arc<world> every worker thread holds upstream
    BoundingVolumeHierarchyNode clone(int device) const {
        BoundingVolumeHierarchyNode w;
        rt_scene *c = rt_scene_clone(scene_.get(), device);
        if (!c) throw Error(RT_ERR_DEVICE, rt_last_error());
        w.scene_.reset(c, rt_scene_destroy);
        return w;
    }
    uint64_t hash() const {
        uint64_t h = 0;
        check(rt_scene_hash(scene_.get(), &h));
        return h;
    }
    // the per-sample workspace the renders keep on the device: its limit per render slot (0 = default), its release, its size
    void set_workspace_limit(size_t bytes) const { check(rt_scene_set_workspace_limit(scene_.get(), bytes)); }
    void trim() const { check(rt_scene_trim(scene_.get())); }
    size_t workspace_bytes() const { return rt_scene_workspace_bytes(scene_.get()); }
    // waits for the renders launched on this world; throws when a kernel reported a device error word
    void status() const { check(rt_render_status(scene_.get())); }
    rt_scene_info info() const {
        rt_scene_info i;
        check(rt_scene_get_info(scene_.get(), &i));
        return i;
    }

  private:
    std::shared_ptr<rt_scene> scene_;
};

// the per-pixel sampling loop of the drivers; buffer[y*W + x], y up (examples/book-one.rs:53,87)
inline std::vector<Vec3> render(const BoundingVolumeHierarchyNode &world, const PerspectiveCamera &camera, int width, int height,
                                int subPixelSampleCount, int maxDepth, uint64_t seed) {
    std::vector<double> rgb((size_t)width * height * 3);
    rt_render_params p{width, height, subPixelSampleCount, maxDepth, seed, 0, 1, 0u};
    check(rt_render(world.raw(), &camera.c, &p, rgb.data(), nullptr));
    std::vector<Vec3> out((size_t)width * height);
    for (size_t i = 0; i < out.size(); ++i) out[i] = Vec3(rgb[i * 3], rgb[i * 3 + 1], rgb[i * 3 + 2]);
    return out;
}
// One shard of render(): only the pixels of 8x8 tiles with tile_id % shard_count == shard_index are written into
// rgb (width*height*3 doubles, y up) -- several worlds (one per GPU, one host thread each) can fill one image.
inline void render_shard(const BoundingVolumeHierarchyNode &world, const PerspectiveCamera &camera, int width, int height,
                         int subPixelSampleCount, int maxDepth, uint64_t seed, int shard_index, int shard_count, double *rgb) {
    rt_render_params p{width, height, subPixelSampleCount, maxDepth, seed, shard_index, shard_count, 0u};
    check(rt_render(world.raw(), &camera.c, &p, rgb, nullptr));
}
// The reference's thread fan-out (examples/book-one.rs:52-88) across GPUs: one committed copy of the world per device,
// tiles dealt tile_id % worlds.size(), one host thread per copy inside the library (rt_render_sharded); bit-identical to render()
inline std::vector<Vec3> render_sharded(const std::vector<BoundingVolumeHierarchyNode> &worlds, const PerspectiveCamera &camera, int width,
                                        int height, int subPixelSampleCount, int maxDepth, uint64_t seed) {
    std::vector<double> rgb((size_t)width * height * 3, 0.0);
    std::vector<rt_scene *> raw;
    for (const BoundingVolumeHierarchyNode &w : worlds) raw.push_back(w.raw());
    rt_render_params p{width, height, subPixelSampleCount, maxDepth, seed, 0, 1, 0u};
    check(rt_render_sharded(raw.data(), (int)raw.size(), &camera.c, &p, rgb.data()));
    std::vector<Vec3> out((size_t)width * height);
    for (size_t i = 0; i < out.size(); ++i) out[i] = Vec3(rgb[i * 3], rgb[i * 3 + 1], rgb[i * 3 + 2]);
    return out;
}
// Progressive form of render(): continues the raw per-pixel sums (width*height*3 doubles, y up) with samples
// [s_begin, s_end) of the subPixelSampleCount-sample render; divide by subPixelSampleCount after the last
// range.  Bit-identical to one render() call (rt_render_progressive).
inline void render_progressive(const BoundingVolumeHierarchyNode &world, const PerspectiveCamera &camera, int width, int height,
                               int subPixelSampleCount, int maxDepth, uint64_t seed, int s_begin, int s_end, std::vector<double> &sums) {
    sums.resize((size_t)width * height * 3);
    rt_render_params p{width, height, subPixelSampleCount, maxDepth, seed, 0, 1, 0u};
    check(rt_render_progressive(world.raw(), &camera.c, &p, s_begin, s_end, sums.data()));
}
inline void write_ppm(const std::string &path, const std::vector<Vec3> &buffer, int width, int height) {
    std::vector<double> rgb(buffer.size() * 3);
    for (size_t i = 0; i < buffer.size(); ++i) {
        rgb[i * 3] = buffer[i].x;
        rgb[i * 3 + 1] = buffer[i].y;
        rgb[i * 3 + 2] = buffer[i].z;
    }
    check(rt_write_ppm_p3(path.c_str(), rgb.data(), width, height));
}
// examples/main.rs:105-135: RGBA8 PNG (the writer of the book-two cover driver)
inline void write_png(const std::string &path, const std::vector<Vec3> &buffer, int width, int height) {
    std::vector<double> rgb(buffer.size() * 3);
    for (size_t i = 0; i < buffer.size(); ++i) {
        rgb[i * 3] = buffer[i].x;
        rgb[i * 3 + 1] = buffer[i].y;
        rgb[i * 3 + 2] = buffer[i].z;
    }
    check(rt_write_png_rgba8(path.c_str(), rgb.data(), width, height));
}

} // namespace ray_tracer
#endif
