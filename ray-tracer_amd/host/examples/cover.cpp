// cover.cpp -- examples/main.rs (book-two cover, finalScene() :156-330, camera :49-61)
// restated over the C++ facade.  The reference writes a PNG through the `image` crate
// (out of scope); this driver writes the P3 PPM of the other two examples.
// ./earthmap.jpg is not part of the reference repository: a deterministic procedural
// texture goes through the same nearest-texel rule (examples/main.rs:267-280).
#include "common.hpp"
using namespace rtx;

static std::shared_ptr<ImageTexture> earth_texture(int w = 1024, int h = 512) {
    std::vector<uint8_t> img((size_t)w * h * 3);
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            const long lat = ((long)y * 180) / h, lon = ((long)x * 360) / w;
            const bool land = (((lon / 30) + (lat / 20)) % 2 == 0) && lat > 20 && lat < 160;
            uint8_t r = land ? (uint8_t)(60 + (lat % 64)) : 20;
            uint8_t g = land ? (uint8_t)(140 - (lat % 40)) : (uint8_t)(60 + (lon % 50));
            uint8_t b = land ? 50 : (uint8_t)(160 + (lat % 80));
            if (lat <= 12 || lat >= 168) {
                r = 235;
                g = 240;
                b = 245;
            }
            uint8_t *p = &img[((size_t)y * w + x) * 3];
            p[0] = r;
            p[1] = g;
            p[2] = b;
        }
    return std::make_shared<ImageTexture>(std::move(img), w, h);
}

static std::vector<SpritePtr> finalScene(uint64_t scene_seed) {
    SceneRng generator(scene_seed);
    auto groundMaterial = std::make_shared<Lambertian>(Vec3(0.48, 0.83, 0.53));
    std::vector<SpritePtr> cubes, res;
    for (int i = 0; i < 20; ++i)
        for (int j = 0; j < 20; ++j) {
            double w = 100.0, x0 = -1000.0 + (double)i * w, y0 = 0.0, z0 = -1000.0 + (double)j * w;
            double x1 = x0 + w, y1 = generator.gen_range(1.0, 101.0), z1 = z0 + w;
            cubes.push_back(Sprite::builder().geometry(std::make_shared<Cube>(x1 - x0, y1 - y0, z1 - z0)).material(groundMaterial)
                                .transform(Mat4::translation(Vec3((x0 + x1) / 2.0, (y0 + y1) / 2.0, (z0 + z1) / 2.0))).build());
        }
    auto light = Sprite::builder().geometry(std::make_shared<Rectangle>(300.0, 265.0)).material(std::make_shared<DiffuseLight>(Vec3(7.0, 7.0, 7.0)))
                     .transform(Mat4::translation(Vec3(273.0, 554.0, 279.5)).multiplied(Mat4::rotation(to_radians(90.0), Vec3::ex()))).build();
    auto movingSphere = Sprite::builder().geometry(std::make_shared<Sphere>(50.0)).material(std::make_shared<Lambertian>(Vec3(0.7, 0.3, 0.1)))
                            .transform(Mat4::translation(Vec3(400.0, 400.0, 200.0))).build();
    auto glassSphere = Sprite::builder().geometry(std::make_shared<Sphere>(50.0)).material(std::make_shared<Dielectric>(1.5))
                           .transform(Mat4::translation(Vec3(260.0, 150.0, 45.0))).build();
    auto metalSphere = Sprite::builder().geometry(std::make_shared<Sphere>(50.0)).material(std::make_shared<Metal>(Vec3(0.8, 0.8, 0.9), 1.0))
                           .transform(Mat4::translation(Vec3(0.0, 150.0, 145.0))).build();
    auto blueSphereSurface = Sprite::builder().geometry(std::make_shared<Sphere>(70.0)).material(std::make_shared<Dielectric>(1.5))
                                 .transform(Mat4::translation(Vec3(360.0, 150.0, 145.0))).build();
    auto blueSphereMedium = Sprite::builder().geometry(std::make_shared<ConstantMedium>(std::make_shared<Sphere>(70.0 - 1e-6), 0.03))
                                .material(std::make_shared<Isotropic>(Vec3(0.2, 0.4, 0.9))).transform(Mat4::translation(Vec3(360.0, 150.0, 145.0))).build();
    auto fog = Sprite::builder().geometry(std::make_shared<ConstantMedium>(std::make_shared<Sphere>(5000.0), 0.0001))
                   .material(std::make_shared<Isotropic>(Vec3(1.0, 1.0, 1.0))).build();
    auto earth = Sprite::builder().geometry(std::make_shared<Sphere>(100.0)).material(std::make_shared<Lambertian>(TexturePtr(earth_texture())))
                     .transform(Mat4::translation(Vec3(400.0, 200.0, 400.0))).build();
    auto whiteMaterial = std::make_shared<Lambertian>(Vec3(0.73, 0.73, 0.73));
    auto s10 = std::make_shared<Sphere>(10.0);
    std::vector<SpritePtr> spheres;
    for (int k = 0; k < 1000; ++k) {
        double x = generator.gen_range(0.0, 165.0), y = generator.gen_range(0.0, 165.0), z = generator.gen_range(0.0, 165.0);
        spheres.push_back(Sprite::builder().geometry(s10).material(whiteMaterial).transform(Mat4::translation(Vec3(x - 100.0, y + 270.0, z + 395.0))).build());
    }
    // examples/main.rs:316-327; the nested BVH nodes (cubes, spheres) carry no transform -> same flat set
    res = cubes;
    for (auto &s : {light, movingSphere, glassSphere, metalSphere, blueSphereSurface, blueSphereMedium, earth, fog}) res.push_back(s);
    res.insert(res.end(), spheres.begin(), spheres.end());
    return res;
}

int main(int argc, char **argv) {
    try {
        Options o = parse(argc, argv, 800, 800, 1000);
        o.png_on_stdout = true; // examples/main.rs:105-140: `cargo run --release --example main > image.png`
        PerspectiveCamera camera(Vec3(555.0 / 2.0 + 200.0, 550.0 / 2.0, -600.0), Vec3(555.0 / 2.0, 555.0 / 2.0, 0.0), Vec3(0.0, 1.0, 0.0),
                                 to_radians(40.0), (double)o.width / (double)o.height, 10.0, 0.0);
        return run(o, finalScene(o.scene_seed), camera);
    } catch (const Error &e) {
        std::fprintf(stderr, "error %d: %s\n", e.code, e.what());
        return 1;
    }
}
