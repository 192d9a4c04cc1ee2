//! Host-side helpers of a driver: the seeded generator for SCENE construction and the two output writers.
//!
//! Upstream draws the scene's random numbers from `rand::thread_rng()`, which cannot be seeded (examples/book-one.rs:117-160):
//! the reference never renders the same scene twice.  `HostRng` is the scene stream of include/rt_rng.h -- xoroshiro128+ seeded by
//! SplitMix64, `gen_range(low, high)` with rand 0.7's `UniformFloat::sample_single` arithmetic -- under the method name the
//! examples already call, so `let mut generator = thread_rng();` becomes `let mut generator = HostRng::new(seed);` and every
//! `generator.gen_range(a, b)` line stays.  With the same seed the Rust, C++ and Python drivers build the same scene.
use crate::ffi;
use crate::gpu::{check, Error};
use crate::vec3::Vec3;

use std::ffi::CString;

const GAMMA: u64 = 0x9E37_79B9_7F4A_7C15;
const SCENE_STREAM: u64 = (1u64 << 40) - 1; // RT_RNG_SCENE_STREAM: never a sample's stream

fn mix64(mut z: u64) -> u64 {
    z = (z ^ (z >> 30)).wrapping_mul(0xBF58_476D_1CE4_E5B9);
    z = (z ^ (z >> 27)).wrapping_mul(0x94D0_49BB_1331_11EB);
    z ^ (z >> 31)
}

pub struct HostRng {
    s0: u64,
    s1: u64,
}
impl HostRng {
    pub fn new(seed: u64) -> Self {
        let base = mix64(seed).wrapping_add((SCENE_STREAM << 24).wrapping_mul(GAMMA));
        let mut s0 = mix64(base.wrapping_add(GAMMA));
        let s1 = mix64(base.wrapping_add(GAMMA.wrapping_mul(2)));
        if (s0 | s1) == 0 {
            s0 = GAMMA;
        }
        HostRng { s0, s1 }
    }
    fn next_u64(&mut self) -> u64 {
        let (s0, mut s1) = (self.s0, self.s1);
        let r = s0.wrapping_add(s1);
        s1 ^= s0;
        self.s0 = s0.rotate_left(24) ^ s1 ^ (s1 << 16);
        self.s1 = s1.rotate_left(37);
        r
    }
    /// `Rng::gen_range(low, high)` of rand 0.7 for f64: `v12 * (high - low) + (low - (high - low))`, retried while `>= high`
    pub fn gen_range(&mut self, low: f64, high: f64) -> f64 {
        let scale = high - low;
        let offset = low - scale;
        loop {
            let v12 = f64::from_bits((self.next_u64() >> 12) | 0x3FF0_0000_0000_0000);
            let res = v12 * scale + offset; // two roundings: Rust never fuses
            if res < high {
                return res;
            }
        }
    }
}

fn flat(buffer: &[Vec<Vec3>]) -> Vec<f64> {
    buffer.iter().flat_map(|row| row.iter().flat_map(|p| vec![p.r(), p.g(), p.b()])).collect()
}

/// the P3 text the `println!` loop of examples/book-one.rs:28-30,90-100 prints, byte for byte (`buffer[y][x]`, y up)
pub fn write_ppm(path: &str, buffer: &[Vec<Vec3>]) -> Result<(), Error> {
    let (height, width) = (buffer.len(), buffer.first().map_or(0, |r| r.len()));
    let c = CString::new(path).unwrap();
    check(unsafe { ffi::rt_write_ppm_p3(c.as_ptr(), flat(buffer).as_ptr(), width as i32, height as i32) })?;
    Ok(())
}

/// the RGBA8 PNG examples/main.rs:105-135 builds with the `image` crate (`put_pixel(x, height - 1 - y, ..)`)
pub fn write_png(path: &str, buffer: &[Vec<Vec3>]) -> Result<(), Error> {
    let (height, width) = (buffer.len(), buffer.first().map_or(0, |r| r.len()));
    let c = CString::new(path).unwrap();
    check(unsafe { ffi::rt_write_png_rgba8(c.as_ptr(), flat(buffer).as_ptr(), width as i32, height as i32) })?;
    Ok(())
}
