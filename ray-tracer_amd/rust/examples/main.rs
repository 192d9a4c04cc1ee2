// The cover of "Ray Tracing: The Next Week" (the reference's examples/main.rs) on the MI355X, written against this crate's copy
// of the reference's API: a floor of 400 cubes in a node, a lamp, five spheres, a glass ball filled with a blue medium, a fog over
// everything, a textured earth and a node of 1000 small spheres -- the two nodes as MEMBERS of the world's list, as upstream.
// Scene data and the order of the random draws follow examples/main.rs:156-330 = scenes.cover(1) of the repository; the code
// around the data is this example's own.
//
// [dev-dependencies] image = "*"   (as upstream: only to read ./earthmap.jpg; the PNG is written by the library)
extern crate ray_tracer;

use ray_tracer::camera::PerspectiveCamera;
use ray_tracer::geometry::{Cube, Rectangle, Sphere};
use ray_tracer::mat4::Mat4;
use ray_tracer::material::{Dielectric, DiffuseLight, ImageTexture, Isotropic, Lambertian, Material, Metal, Texture};
use ray_tracer::optimize::{AxisAlignedBoundingBox, Bound, BoundingVolumeHierarchyNode};
use ray_tracer::sprite::Sprite;
use ray_tracer::util::{write_png, HostRng};
use ray_tracer::vec3::Vec3;
use ray_tracer::volume::ConstantMedium;

use std::sync::Arc;

type Object = Arc<dyn Bound<AxisAlignedBoundingBox>>;

const SCENE_SEED: u64 = 1;
const RENDER_SEED: u64 = 1;

fn ball<U: Material + 'static>(radius: f64, material: Arc<U>, at: Vec3) -> Object {
    Arc::new(Sprite::builder().geometry(Sphere::new(radius).into()).material(material).transform(Mat4::translation(at)).build())
}

fn final_scene(seed: u64) -> Vec<Object> {
    let mut generator = HostRng::new(seed);

    // the floor: 20 x 20 boxes of random height, each the node of a Cube's six faces under a translation; all of them in one node
    let ground = Arc::new(Lambertian::new(Vec3::new(0.48, 0.83, 0.53)));
    let mut cubes: Vec<Object> = Vec::new();
    for i in 0..20 {
        for j in 0..20 {
            let w = 100.0;
            let (x0, y0, z0) = (-1000.0 + i as f64 * w, 0.0, -1000.0 + j as f64 * w);
            let (x1, y1, z1) = (x0 + w, generator.gen_range(1.0, 101.0), z0 + w);
            let faces: Vec<Object> = Cube::new(x1 - x0, y1 - y0, z1 - z0).into_iter().map(|v| Arc::new(v) as Object).collect();
            let node = BoundingVolumeHierarchyNode::new(faces).unwrap();
            cubes.push(Arc::new(
                Sprite::builder()
                    .geometry(Arc::new(node))
                    .material(ground.clone())
                    .transform(Mat4::translation(Vec3::new((x0 + x1) / 2.0, (y0 + y1) / 2.0, (z0 + z1) / 2.0)))
                    .build(),
            ));
        }
    }
    let cubes: Object = Arc::new(BoundingVolumeHierarchyNode::new(cubes).unwrap());

    let light: Object = Arc::new(
        Sprite::builder()
            .geometry(Rectangle::new(300.0, 265.0).into())
            .material(Arc::new(DiffuseLight::new(Vec3::new(7.0, 7.0, 7.0))))
            .transform(Mat4::translation(Vec3::new(273.0, 554.0, 279.5)).multiplied(&Mat4::rotation(90.0_f64.to_radians(), Vec3::ex())))
            .build(),
    );
    let moving = ball(50.0, Arc::new(Lambertian::new(Vec3::new(0.7, 0.3, 0.1))), Vec3::new(400.0, 400.0, 200.0));
    let glass = ball(50.0, Dielectric::new(1.5).into(), Vec3::new(260.0, 150.0, 45.0));
    let metal = ball(50.0, Metal::new(Vec3::new(0.8, 0.8, 0.9), 1.0).into(), Vec3::new(0.0, 150.0, 145.0));

    // the blue ball: a glass surface, and inside it (1e-6 smaller) a medium that scatters blue
    let blue_at = Vec3::new(360.0, 150.0, 145.0);
    let blue_surface = ball(70.0, Arc::new(Dielectric::new(1.5)), blue_at);
    let blue_medium: Object = Arc::new(
        Sprite::builder()
            .geometry(ConstantMedium::new(Sphere::new(70.0 - 1e-6).into(), 0.03).into())
            .material(Isotropic::new(Vec3::new(0.2, 0.4, 0.9)).into())
            .transform(Mat4::translation(blue_at))
            .build(),
    );
    // a thin fog in a sphere of radius 5000 around the origin (no transform)
    let fog: Object = Arc::new(
        Sprite::builder()
            .geometry(ConstantMedium::new(Sphere::new(5000.0).into(), 0.0001).into())
            .material(Isotropic::new(Vec3::new(1.0, 1.0, 1.0)).into())
            .build(),
    );

    // the earth: the reference's closure (nearest texel of ./earthmap.jpg), tabulated for the device at the image's own size
    let image = Arc::new(image::open("./earthmap.jpg").unwrap());
    let (image_width, image_height) = (image.as_rgb8().unwrap().width(), image.as_rgb8().unwrap().height());
    let mapping = move |uv: &(f64, f64)| -> Vec3 {
        let texels = image.as_rgb8().unwrap();
        let p = texels.get_pixel((uv.0 * texels.width() as f64) as u32, ((1.0 - uv.1) * texels.height() as f64) as u32);
        Vec3::new(p[0] as f64 / 255.0, p[1] as f64 / 255.0, p[2] as f64 / 255.0)
    };
    let earth_texture: Arc<dyn Texture> = Arc::new(ImageTexture::new(mapping).resolution(image_width, image_height));
    let earth = ball(100.0, Arc::new(Lambertian::new(earth_texture)), Vec3::new(400.0, 200.0, 400.0));

    // the foam: 1000 small spheres sharing one material, in a node of their own
    let white = Arc::new(Lambertian::new(Vec3::new(0.73, 0.73, 0.73)));
    let mut spheres: Vec<Object> = Vec::new();
    for _ in 0..1000 {
        let (x, y, z) = (generator.gen_range(0.0, 165.0), generator.gen_range(0.0, 165.0), generator.gen_range(0.0, 165.0));
        spheres.push(ball(10.0, white.clone(), Vec3::new(x - 100.0, y + 270.0, z + 395.0)));
    }
    let spheres: Object = Arc::new(BoundingVolumeHierarchyNode::new(spheres).unwrap());

    vec![cubes, light, moving, glass, metal, blue_surface, blue_medium, earth, fog, spheres]
}

fn main() {
    // BASELINE configs[3]: 800 x 800, 1000 samples per pixel, depth 100
    let (width, height, samples, depth) = (800usize, 800usize, 1000usize, 100usize);
    let world = BoundingVolumeHierarchyNode::new(final_scene(SCENE_SEED)).unwrap();
    let camera = PerspectiveCamera::new(
        Vec3::new(555.0 / 2.0 + 200.0, 550.0 / 2.0, -600.0),
        Vec3::new(555.0 / 2.0, 555.0 / 2.0, 0.0),
        Vec3::new(0.0, 1.0, 0.0),
        40.0_f64.to_radians(),
        width as f64 / height as f64,
        10.0,
        0.0,
    );
    let buffer = world.render(&camera, width, height, samples, depth, RENDER_SEED).unwrap();
    write_png("/dev/stdout", &buffer).unwrap(); // the RGBA8 PNG of examples/main.rs:105-135
}
