// The random-spheres scene of "Ray Tracing in One Weekend" (the reference's examples/book-one.rs) on the MI355X, written
// against this crate's copy of the reference's API.  Scene data -- sizes, colours, the order of the random draws -- follow
// examples/book-one.rs:103-205, so that with SCENE_SEED = 1 this is the scene the C++ and Python drivers of the repository render
// (scenes.book_one(1)); the code around the data is this example's own (README.md has the three-edit recipe that turns the
// reference's own file into a driver of this crate instead).
extern crate ray_tracer;

use ray_tracer::camera::PerspectiveCamera;
use ray_tracer::geometry::Sphere;
use ray_tracer::mat4::Mat4;
use ray_tracer::material::{Dielectric, DiffuseLight, Lambertian, Material, Metal};
use ray_tracer::optimize::{AxisAlignedBoundingBox, Bound, BoundingVolumeHierarchyNode};
use ray_tracer::sprite::Sprite;
use ray_tracer::util::{write_ppm, HostRng};
use ray_tracer::vec3::Vec3;

use std::sync::Arc;

type Object = Arc<dyn Bound<AxisAlignedBoundingBox>>;

const SCENE_SEED: u64 = 1;
const RENDER_SEED: u64 = 1;

/// a sphere of `radius` at `at`: `Sprite<Sphere, U>` behind the list's trait object
fn ball<U: Material + 'static>(radius: f64, material: U, at: Vec3) -> Object {
    Arc::new(Sprite::builder().geometry(Sphere::new(radius).into()).material(Arc::new(material)).transform(Mat4::translation(at)).build())
}

fn random_scene(seed: u64) -> Vec<Object> {
    let mut generator = HostRng::new(seed);
    let mut scene: Vec<Object> = Vec::new();
    // the ground is a sphere of radius 1000 below the origin; the sky is an emitting sphere around everything (no transform)
    scene.push(ball(1000.0, Lambertian::new(Vec3::new(0.5, 0.5, 0.5)), Vec3::new(0.0, -1000.0, 0.0)));
    scene.push(Arc::new(Sprite::builder().geometry(Sphere::new(2000.0).into()).material(DiffuseLight::new(Vec3::new(0.5, 0.7, 1.0)).into()).build()));
    for a in -11..11 {
        for b in -11..11 {
            let which = generator.gen_range(0.0, 1.0);
            let center = Vec3::new(a as f64 + 0.9 * generator.gen_range(0.0, 1.0), 0.2, b as f64 + 0.9 * generator.gen_range(0.0, 1.0));
            if (center - Vec3::new(4.0, 0.2, 0.0)).length() <= 0.9 {
                continue; // too close to the big metal sphere
            }
            if which < 0.3 {
                let c = Vec3::new(generator.gen_range(0.0, 1.0), generator.gen_range(0.0, 1.0), generator.gen_range(0.0, 1.0));
                scene.push(ball(0.2, Lambertian::new(c * c), center));
            } else if which < 0.6 {
                let c = Vec3::new(generator.gen_range(0.5, 1.0), generator.gen_range(0.5, 1.0), generator.gen_range(0.5, 1.0));
                let fuzziness = generator.gen_range(0.0, 0.5);
                scene.push(ball(0.2, Metal::new(c, fuzziness), center));
            } else {
                scene.push(ball(0.2, Dielectric::new(1.5), center));
            }
        }
    }
    scene.push(ball(1.0, Lambertian::new(Vec3::new(0.4, 0.2, 0.1)), Vec3::new(-4.0, 1.0, 0.0)));
    scene.push(ball(1.0, Metal::new(Vec3::new(0.7, 0.6, 0.5), 0.0), Vec3::new(4.0, 1.0, 0.0)));
    scene.push(ball(1.0, Dielectric::new(1.5), Vec3::new(0.0, 1.0, 0.0)));
    scene
}

fn main() {
    // BASELINE configs[1]: 1200 x 800, 500 samples per pixel, depth 100
    let (width, height, samples, depth) = (1200usize, 800usize, 500usize, 100usize);
    let world = BoundingVolumeHierarchyNode::new(random_scene(SCENE_SEED)).unwrap();
    let camera = PerspectiveCamera::new(
        Vec3::new(13.0, 2.0, 3.0),
        Vec3::new(0.0, 0.0, 0.0),
        Vec3::new(0.0, 1.0, 0.0),
        20.0_f64.to_radians(),
        width as f64 / height as f64,
        10.0,
        0.05,
    );
    // the reference's worker threads, its `color(&ray, world, 100)` per sample and its mpsc fan-in (examples/book-one.rs:50-88)
    let buffer = world.render(&camera, width, height, samples, depth, RENDER_SEED).unwrap();
    write_ppm("/dev/stdout", &buffer).unwrap(); // the P3 text of examples/book-one.rs:28-30,90-100
}
