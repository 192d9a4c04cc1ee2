// cornell_box.cpp -- examples/cornell-box.rs:24-204 restated over the C++ facade.
// Reference literals: 800x800, 1000 spp, depth 100; BASELINE configs[2] is 600x600x1000.
#include "common.hpp"
using namespace rtx;

int main(int argc, char **argv) {
    try {
        Options o = parse(argc, argv, 800, 800, 1000);
        auto red = std::make_shared<Lambertian>(Vec3(0.65, 0.05, 0.05));
        auto white = std::make_shared<Lambertian>(Vec3(0.73, 0.73, 0.73));
        auto green = std::make_shared<Lambertian>(Vec3(0.12, 0.45, 0.15));
        auto light = std::make_shared<DiffuseLight>(Vec3(15.0, 15.0, 15.0));
        auto tr = [](Vec3 t, double deg, Vec3 axis) { return Mat4::translation(t).multiplied(Mat4::rotation(to_radians(deg), axis)); };
        auto rect = [](double w, double h) { return std::make_shared<Rectangle>(w, h); };
        std::vector<SpritePtr> world;
        world.push_back(Sprite::builder().geometry(rect(555.0, 555.0)).material(green)
                            .transform(tr(Vec3(555.0, 555.0 / 2.0, 555.0 / 2.0), -90.0, Vec3::ey())).build());
        world.push_back(Sprite::builder().geometry(rect(555.0, 555.0)).material(red)
                            .transform(tr(Vec3(0.0, 555.0 / 2.0, 555.0 / 2.0), 90.0, Vec3::ey())).build());
        world.push_back(Sprite::builder().geometry(rect(130.0, 105.0)).material(light)
                            .transform(tr(Vec3(555.0 / 2.0, 554.0, 555.0 / 2.0), 90.0, Vec3::ex())).build());
        world.push_back(Sprite::builder().geometry(rect(555.0, 555.0)).material(white)
                            .transform(tr(Vec3(555.0 / 2.0, 0.0, 555.0 / 2.0), -90.0, Vec3::ex())).build());
        world.push_back(Sprite::builder().geometry(rect(555.0, 555.0)).material(white)
                            .transform(tr(Vec3(555.0 / 2.0, 555.0, 555.0 / 2.0), 90.0, Vec3::ex())).build());
        world.push_back(Sprite::builder().geometry(rect(555.0, 556.0)).material(white)
                            .transform(tr(Vec3(555.0 / 2.0, 555.0 / 2.0, 555.0), 180.0, Vec3::ey())).build());
        world.push_back(Sprite::builder().geometry(std::make_shared<Cube>(165.0, 165.0, 165.0)).material(white)
                            .transform(tr(Vec3(212.5, 82.5, 147.5), -18.0, Vec3::ey())).build());
        world.push_back(Sprite::builder().geometry(std::make_shared<Cube>(165.0, 330.0, 165.0)).material(white)
                            .transform(tr(Vec3(347.5, 165.0, 377.5), 15.0, Vec3::ey())).build());
        PerspectiveCamera camera(Vec3(555.0 / 2.0, 555.0 / 2.0, -800.0), Vec3(555.0 / 2.0, 555.0 / 2.0, 0.0), Vec3(0.0, 1.0, 0.0),
                                 to_radians(40.0), (double)o.width / (double)o.height, 10.0, 0.0);
        return run(o, world, camera);
    } catch (const Error &e) {
        std::fprintf(stderr, "error %d: %s\n", e.code, e.what());
        return 1;
    }
}
