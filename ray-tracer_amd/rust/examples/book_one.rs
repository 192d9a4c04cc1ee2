// examples/book-one.rs of the reference with the sampling loop moved to the MI355X.
// Scene construction keeps the reference's structure (examples/book-one.rs:103-205); the
// unseedable thread_rng() is replaced by any seeded generator of the caller's choice
// (a tiny SplitMix64 here -- this driver is NOT the pinned scene of the test-suite, which
// uses include/rt_rng.h through the C++ / Python drivers).
use ray_tracer_mi355x::*;
use std::sync::Arc;

struct SplitMix64(u64);
impl SplitMix64 {
    fn next_f64(&mut self) -> f64 {
        self.0 = self.0.wrapping_add(0x9E3779B97F4A7C15);
        let mut z = self.0;
        z = (z ^ (z >> 30)).wrapping_mul(0xBF58476D1CE4E5B9);
        z = (z ^ (z >> 27)).wrapping_mul(0x94D049BB133111EB);
        ((z ^ (z >> 31)) >> 11) as f64 * (1.0 / 9007199254740992.0)
    }
    fn gen_range(&mut self, lo: f64, hi: f64) -> f64 {
        lo + (hi - lo) * self.next_f64()
    }
}

fn translation(t: Vec3) -> Mat4 {
    let mut m = [0.0; 16];
    m[0] = 1.0;
    m[5] = 1.0;
    m[10] = 1.0;
    m[15] = 1.0;
    m[12] = t.x;
    m[13] = t.y;
    m[14] = t.z;
    m
}

fn random_scene(seed: u64) -> Vec<Sprite> {
    let mut g = SplitMix64(seed);
    let small = Arc::new(Geometry::Sphere(0.2));
    let big = Arc::new(Geometry::Sphere(1.0));
    let mut scene = vec![
        Sprite::builder()
            .geometry(Arc::new(Geometry::Sphere(1000.0)))
            .material(Arc::new(Material::Lambertian(Vec3::new(0.5, 0.5, 0.5).into())))
            .transform(translation(Vec3::new(0.0, -1000.0, 0.0)))
            .build(),
        Sprite::builder()
            .geometry(Arc::new(Geometry::Sphere(2000.0)))
            .material(Arc::new(Material::DiffuseLight(Vec3::new(0.5, 0.7, 1.0).into())))
            .build(),
    ];
    for a in -11..11 {
        for b in -11..11 {
            let which = g.gen_range(0.0, 1.0);
            let center = Vec3::new(a as f64 + 0.9 * g.gen_range(0.0, 1.0), 0.2, b as f64 + 0.9 * g.gen_range(0.0, 1.0));
            let (dx, dz) = (center.x - 4.0, center.z);
            if (dx * dx + dz * dz).sqrt() > 0.9 {
                let material = if which < 0.3 {
                    let (r, gr, bl) = (g.gen_range(0.0, 1.0), g.gen_range(0.0, 1.0), g.gen_range(0.0, 1.0));
                    Material::Lambertian(Vec3::new(r * r, gr * gr, bl * bl).into())
                } else if which < 0.6 {
                    let albedo = Vec3::new(g.gen_range(0.5, 1.0), g.gen_range(0.5, 1.0), g.gen_range(0.5, 1.0));
                    Material::Metal(albedo.into(), g.gen_range(0.0, 0.5))
                } else {
                    Material::Dielectric(1.5)
                };
                scene.push(Sprite::builder().geometry(small.clone()).material(Arc::new(material)).transform(translation(center)).build());
            }
        }
    }
    scene.push(Sprite::builder().geometry(big.clone()).material(Arc::new(Material::Lambertian(Vec3::new(0.4, 0.2, 0.1).into())))
        .transform(translation(Vec3::new(-4.0, 1.0, 0.0))).build());
    scene.push(Sprite::builder().geometry(big.clone()).material(Arc::new(Material::Metal(Vec3::new(0.7, 0.6, 0.5).into(), 0.0)))
        .transform(translation(Vec3::new(4.0, 1.0, 0.0))).build());
    scene.push(Sprite::builder().geometry(big).material(Arc::new(Material::Dielectric(1.5)))
        .transform(translation(Vec3::new(0.0, 1.0, 0.0))).build());
    scene
}

fn main() {
    let (width, height) = (1600usize, 800usize);
    let world = BoundingVolumeHierarchyNode::new(&random_scene(1), 0).unwrap().unwrap();
    let camera = PerspectiveCamera::new(
        Vec3::new(13.0, 2.0, 3.0),
        Vec3::new(0.0, 0.0, 0.0),
        Vec3::new(0.0, 1.0, 0.0),
        (20.0f64).to_radians(),
        width as f64 / height as f64,
        10.0,
        0.05,
    );
    // was: cpuCount threads, rows y % cpuCount, 100 x color(&ray, world, 100) per pixel, mpsc channel
    let buffer = world.render(&camera, width, height, 100, 100, 1).unwrap();
    write_ppm("/dev/stdout", &buffer, width, height).unwrap();
}
