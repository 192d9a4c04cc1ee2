// The Cornell box of the reference's examples/cornell-box.rs on the MI355X, written against this crate's copy of the reference's
// API.  Scene data (sizes, colours, positions, angles, the order of the eight objects) follow examples/cornell-box.rs:31-140 =
// scenes.cornell() of the repository; the code around the data is this example's own.
extern crate ray_tracer;

use ray_tracer::camera::PerspectiveCamera;
use ray_tracer::geometry::{Cube, Rectangle};
use ray_tracer::mat4::Mat4;
use ray_tracer::material::{DiffuseLight, Lambertian, Material};
use ray_tracer::optimize::{AxisAlignedBoundingBox, Bound, BoundingVolumeHierarchyNode};
use ray_tracer::sprite::Sprite;
use ray_tracer::util::write_ppm;
use ray_tracer::vec3::Vec3;

use std::sync::Arc;

type Object = Arc<dyn Bound<AxisAlignedBoundingBox>>;

const RENDER_SEED: u64 = 1;

/// a rectangle (normal +z in its own frame) turned by `degrees` about `axis`, then moved to `at`
fn panel<U: Material + 'static>(width: f64, height: f64, material: &Arc<U>, at: Vec3, degrees: f64, axis: Vec3) -> Object {
    let transform = Mat4::translation(at).multiplied(&Mat4::rotation(degrees.to_radians(), axis));
    Arc::new(Sprite::builder().geometry(Rectangle::new(width, height).into()).material(material.clone()).transform(transform).build())
}

/// a cube: the node of its six faces as the sprite's geometry, exactly as the reference's drivers wrap `Cube::new`
fn block(width: f64, height: f64, depth: f64, material: &Arc<Lambertian>, at: Vec3, degrees: f64) -> Object {
    let faces: Vec<Object> = Cube::new(width, height, depth).into_iter().map(|v| Arc::new(v) as Object).collect();
    Arc::new(
        Sprite::builder()
            .geometry(BoundingVolumeHierarchyNode::new(faces).unwrap().into())
            .material(material.clone())
            .transform(Mat4::translation(at).multiplied(&Mat4::rotation(degrees.to_radians(), Vec3::ey())))
            .build(),
    )
}

fn main() {
    // BASELINE configs[2]: 600 x 600, 1000 samples per pixel, depth 100
    let (width, height, samples, depth) = (600usize, 600usize, 1000usize, 100usize);
    let red = Arc::new(Lambertian::new(Vec3::new(0.65, 0.05, 0.05)));
    let white = Arc::new(Lambertian::new(Vec3::new(0.73, 0.73, 0.73)));
    let green = Arc::new(Lambertian::new(Vec3::new(0.12, 0.45, 0.15)));
    let lamp = Arc::new(DiffuseLight::new(Vec3::new(15.0, 15.0, 15.0)));
    let s = 555.0;
    let world: Vec<Object> = vec![
        panel(s, s, &green, Vec3::new(s, s / 2.0, s / 2.0), -90.0, Vec3::ey()),
        panel(s, s, &red, Vec3::new(0.0, s / 2.0, s / 2.0), 90.0, Vec3::ey()),
        panel(130.0, 105.0, &lamp, Vec3::new(s / 2.0, 554.0, s / 2.0), 90.0, Vec3::ex()),
        panel(s, s, &white, Vec3::new(s / 2.0, 0.0, s / 2.0), -90.0, Vec3::ex()),     // floor
        panel(s, s, &white, Vec3::new(s / 2.0, s, s / 2.0), 90.0, Vec3::ex()),        // ceiling
        panel(s, 556.0, &white, Vec3::new(s / 2.0, s / 2.0, s), 180.0, Vec3::ey()),   // back wall
        block(165.0, 165.0, 165.0, &white, Vec3::new(212.5, 82.5, 147.5), -18.0),
        block(165.0, 330.0, 165.0, &white, Vec3::new(347.5, 165.0, 377.5), 15.0),
    ];
    let world = BoundingVolumeHierarchyNode::new(world).unwrap();
    let camera = PerspectiveCamera::new(
        Vec3::new(s / 2.0, s / 2.0, -800.0),
        Vec3::new(s / 2.0, s / 2.0, 0.0),
        Vec3::new(0.0, 1.0, 0.0),
        40.0_f64.to_radians(),
        width as f64 / height as f64,
        10.0,
        0.0,
    );
    let buffer = world.render(&camera, width, height, samples, depth, RENDER_SEED).unwrap();
    write_ppm("/dev/stdout", &buffer).unwrap();
}
