// book_one.cpp -- examples/book-one.rs restated over the C++ facade: randomScene()
// (:103-205), camera (:35-47), the sampling loop (:56-88, now one rt_render call on the
// MI355X) and the P3 output (:28-30,90-100).  Defaults are the reference's literals
// (1600x800, 100 spp, depth 100); BASELINE configs[1] is --width 1200 --height 800 --spp 500.
#include "common.hpp"

#include <cmath>
using namespace rtx;

static std::vector<SpritePtr> randomScene(uint64_t scene_seed) {
    std::vector<SpritePtr> scene;
    auto sph1000 = std::make_shared<Sphere>(1000.0), sph2000 = std::make_shared<Sphere>(2000.0);
    auto sph02 = std::make_shared<Sphere>(0.2), sph1 = std::make_shared<Sphere>(1.0);
    scene.push_back(Sprite::builder().geometry(sph1000).material(std::make_shared<Lambertian>(Vec3(0.5, 0.5, 0.5)))
                        .transform(Mat4::translation(Vec3(0.0, -1000.0, 0.0))).build()); // the ground is a huge sphere
    scene.push_back(Sprite::builder().geometry(sph2000).material(std::make_shared<DiffuseLight>(Vec3(0.5, 0.7, 1.0))).build()); // sky
    SceneRng generator(scene_seed);
    for (int a = -11; a < 11; ++a)
        for (int b = -11; b < 11; ++b) {
            double whichMaterial = generator.gen_range(0.0, 1.0);
            double cx = (double)a + 0.9 * generator.gen_range(0.0, 1.0);
            double cz = (double)b + 0.9 * generator.gen_range(0.0, 1.0);
            double dx = cx - 4.0, dy = 0.2 - 0.2, dz = cz - 0.0;
            if (std::sqrt(dx * dx + dy * dy + dz * dz) > 0.9) {
                MaterialPtr m;
                if (whichMaterial < 0.3) {
                    double r = generator.gen_range(0.0, 1.0), g = generator.gen_range(0.0, 1.0), bl = generator.gen_range(0.0, 1.0);
                    m = std::make_shared<Lambertian>(Vec3(r * r, g * g, bl * bl));
                } else if (whichMaterial < 0.6) {
                    double r = generator.gen_range(0.5, 1.0), g = generator.gen_range(0.5, 1.0), bl = generator.gen_range(0.5, 1.0);
                    double fuzziness = generator.gen_range(0.0, 0.5);
                    m = std::make_shared<Metal>(Vec3(r, g, bl), fuzziness);
                } else {
                    m = std::make_shared<Dielectric>(1.5);
                }
                scene.push_back(Sprite::builder().geometry(sph02).material(m).transform(Mat4::translation(Vec3(cx, 0.2, cz))).build());
            }
        }
    scene.push_back(Sprite::builder().geometry(sph1).material(std::make_shared<Lambertian>(Vec3(0.4, 0.2, 0.1)))
                        .transform(Mat4::translation(Vec3(-4.0, 1.0, 0.0))).build());
    scene.push_back(Sprite::builder().geometry(sph1).material(std::make_shared<Metal>(Vec3(0.7, 0.6, 0.5), 0.0))
                        .transform(Mat4::translation(Vec3(4.0, 1.0, 0.0))).build());
    scene.push_back(Sprite::builder().geometry(sph1).material(std::make_shared<Dielectric>(1.5))
                        .transform(Mat4::translation(Vec3(0.0, 1.0, 0.0))).build());
    return scene;
}

int main(int argc, char **argv) {
    try {
        Options o = parse(argc, argv, 1600, 800, 100);
        PerspectiveCamera camera(Vec3(13.0, 2.0, 3.0), Vec3(0.0, 0.0, 0.0), Vec3(0.0, 1.0, 0.0), to_radians(20.0),
                                 (double)o.width / (double)o.height, 10.0, 0.05);
        return run(o, randomScene(o.scene_seed), camera);
    } catch (const Error &e) {
        std::fprintf(stderr, "error %d: %s\n", e.code, e.what());
        return 1;
    }
}
