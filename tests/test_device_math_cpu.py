"""The arithmetic forms only DEVICE code takes -- rt_lane.h's shared-reciprocal divisions (Vec3 / f64, sphere_roots,
world_roots_rcp + sphere_t), include/rt_rng.h's funnel-shift rotations and rt_u64_to_pm1 from bits -- executed on the CPU.

tests/lane_emul.cpp is built a second time with -DRT_EMULATE_DEVICE_MATH: the `#if defined(RT_DEVICE_MATH)` branches are then
compiled for the host with portable stand-ins for three intrinsics.  The stand-in for v_rcp_f64 is the exact reciprocal SPOILT to
the ISA manual's error bound (2^-23 relative) downwards, upwards, or either way by a hash of the operand: the refinement has to
reach the correctly rounded quotient from any such seed, so these tests do not depend on what one implementation of the
instruction returns.  What they cannot show is that the hardware's v_rcp_f64 keeps that bound and that v_fma_f64 / v_mul_f64
round as IEEE says: tests/test_gpu_parity.py::test_device_sqrt_div_correctly_rounded and the bit-exact image tests do that on
the MI355X."""
import numpy as np
import pytest

RCP_MODES = (0, 1, 2)


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def same_bits(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return bool(np.all((bits(a) == bits(b)) | (np.isnan(a) & np.isnan(b))))


def spread(rng, n, lo, hi):
    """numbers of either sign with binary exponents uniform in [lo, hi]"""
    return rng.uniform(1.0, 2.0, n) * np.exp2(rng.integers(lo, hi + 1, n).astype(np.float64)) * rng.choice([-1.0, 1.0], n)


@pytest.mark.parametrize("mode", RCP_MODES)
def test_vec3_division_is_the_correctly_rounded_quotient(lane_devmath, mode):
    rng = np.random.default_rng(100 + mode)
    n = 400_000
    lane_devmath.set_rcp_mode(mode)
    with np.errstate(all="ignore"):
        # the ranges the renderer lives in, the whole guarded range, and the guard's edges with operands on either side of them
        for (alo, ahi), (slo, shi) in [((-30, 30), (-30, 30)), ((-500, 256), (-500, 256)), ((-520, 270), (-520, 270)),
                                       ((-1074, 1023), (-1074, 1023))]:
            a = np.stack([spread(rng, n, alo, ahi) for _ in range(3)], axis=1)
            s = spread(rng, n, slo, shi)
            before = lane_devmath.rcp_calls()
            assert same_bits(lane_devmath.div3(a, s), a / s[:, None])
            taken = lane_devmath.rcp_calls() - before
            assert taken > 0 if (alo, ahi) != (-1074, 1023) else taken < n  # the fast branch ran; over the full range it mostly must not
        # zeros, infinities, NaNs, denormals: all through the ordinary division
        special = np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 5e-324, -2.2e-308, 1.7e308, 1.0, -3.0])
        a = np.stack(np.meshgrid(special, special, special, indexing="ij"), axis=-1).reshape(-1, 3)
        for sv in special:
            assert same_bits(lane_devmath.div3(a, np.full(len(a), sv)), a / sv)
        # quotients that overflow or go denormal although every operand is inside the guard
        a = np.stack([spread(rng, n, 200, 256), spread(rng, n, -500, -400), spread(rng, n, -10, 10)], axis=1)
        s = np.concatenate([spread(rng, n // 2, -500, -400), spread(rng, n - n // 2, 200, 256)])
        assert same_bits(lane_devmath.div3(a, s), a / s[:, None])


def test_guard_overflow_and_denormal_quotients_are_out_of_reach():
    """The guard's own arithmetic: exponents in [2^-500, 2^256] bound every quotient by 2^-757 < |q| < 2^757 -- no overflow, no
    denormal result, and div_scale / div_fixup of the full expansion would have been the identity (they act on operands
    outside roughly 2^-767 .. 2^767 and on quotients near the ends of the range)."""
    assert 256 + 1 - (-500) < 1023 - 52 and -500 - (256 + 1) > -1022 + 52


@pytest.mark.parametrize("mode", RCP_MODES)
def test_sphere_roots_match_plain_division(lane_devmath, mode):
    rng = np.random.default_rng(200 + mode)
    n = 400_000
    lane_devmath.set_rcp_mode(mode)
    with np.errstate(all="ignore"):
        for (nlo, nhi), (dlo, dhi) in [((-30, 40), (-20, 20)), ((-200, 256), (-100, 100)), ((-300, 300), (-110, 110))]:
            n1, n2, den = spread(rng, n, nlo, nhi), spread(rng, n, nlo, nhi), np.abs(spread(rng, n, dlo, dhi))
            before = lane_devmath.rcp_calls()
            got = lane_devmath.sphere_roots(n1, n2, den)
            assert lane_devmath.rcp_calls() > before
            want = np.stack([n1 / den, n2 / den], axis=1)
            # numerators below 2^-500 of the denominator's scale ("cancellation dust", rt_lane.h) may differ in their low bits, both
            # far below the 1e-6 a root is compared with; everything else has to be the same bits
            dust = np.abs(want) < 2.0 ** -300
            assert np.all((bits(got) == bits(want)) | dust)
            assert np.all(np.abs(got[dust]) < 2.0 ** -290)
        z = np.array([0.0, -0.0, np.nan, np.inf, 1.0])
        n1, n2, den = [v.ravel() for v in np.meshgrid(z, z, z, indexing="ij")]
        got, want = lane_devmath.sphere_roots(n1, n2, den), np.stack([n1 / den, n2 / den], axis=1)
        # a zero numerator over an ordinary denominator is dust too: the refinement returns +0 where the division keeps -0 (a root
        # is only ever compared with 1e-6, which neither zero exceeds)
        assert np.all((bits(got) == bits(want)) | (np.isnan(got) & np.isnan(want)) | ((got == 0.0) & (want == 0.0)))


@pytest.mark.parametrize("mode", RCP_MODES)
def test_world_sphere_with_the_segments_shared_reciprocal(lane_emul, lane_devmath, mode):
    """Sphere::hit's t for a world-space sphere: the per-segment refined reciprocal (device form) against the host form (two plain
    divisions), on rays that hit, graze and miss, with |d| far from 1 and origins far from the sphere."""
    rng = np.random.default_rng(300 + mode)
    n = 300_000
    lane_devmath.set_rcp_mode(mode)
    radius = np.exp(rng.uniform(-7, 7, n))
    target = rng.standard_normal((n, 3)) * radius[:, None] * rng.choice([0.2, 0.9, 0.999999, 1.000001, 1.5], (n, 1))
    oc = rng.standard_normal((n, 3)) * radius[:, None] * np.exp(rng.uniform(-3, 6, (n, 1)))
    d = (target - oc) * np.exp(rng.uniform(-20, 20, (n, 1)))
    before = lane_devmath.rcp_calls()
    got, want = lane_devmath.sphere_t_world(oc, d, radius), lane_emul.sphere_t_world(oc, d, radius)
    assert lane_devmath.rcp_calls() - before == n and lane_emul.rcp_calls() == 0
    assert same_bits(got, want)
    hits = np.isfinite(want).mean()
    assert 0.3 < hits < 0.9
    # outside the per-segment guard (|d|^2 beyond 2^100) the shared reciprocal is NaN and the per-test path decides: same t again
    d_far = d * 2.0 ** 60
    assert same_bits(lane_devmath.sphere_t_world(oc, d_far, radius), lane_emul.sphere_t_world(oc, d_far, radius))


def test_rotations_and_pm1_from_bits(lane_emul, lane_devmath):
    rng = np.random.default_rng(7)
    x = rng.integers(0, 2 ** 64, 1_000_000, dtype=np.uint64)
    x[:8] = [0, 1, 2 ** 63, 2 ** 64 - 1, 0x7FF, 0x800, 2 ** 63 + 0x7FF, 2 ** 64 - 0x800]
    dev, host = lane_devmath.rng_forms(x), lane_emul.rng_forms(x)
    for k, got, want in zip((24, 37, 16), dev[:3], host[:3]):
        assert np.array_equal(got, want) and np.array_equal(got, (x << np.uint64(k)) | (x >> np.uint64(64 - k)))
    assert same_bits(dev[3], host[3])
    assert same_bits(dev[3], (x >> np.uint64(11)).astype(np.float64) * 2.0 ** -53 * 2.0 - 1.0)  # random::<f64>() * 2.0 - 1.0, src/util.rs:8-12
    assert dev[3].min() == -1.0 and dev[3].max() < 1.0


def _scene_cases(scenes):
    return [("book-one", scenes.book_one(1, 1.5), 48, 32, 3, 50), ("cornell", scenes.cornell(1.0), 32, 32, 6, 100),
            ("cover", scenes.cover(1, 1.0), 32, 32, 3, 100), ("nested media", scenes.nested_media(), 32, 24, 4, 50),
            ("deep chains", scenes.deep_chains(), 32, 24, 3, 50)]


@pytest.mark.parametrize("mode", RCP_MODES)
@pytest.mark.parametrize("case", range(5))
def test_scenes_with_device_arithmetic_match_the_oracle(scenes, oracle, lane_devmath, case, mode):
    """Whole images through the device forms against the oracle's plain arithmetic: same bits, same segment counts -- and the fast
    branches demonstrably ran.  The binary32 reciprocal of the culling boxes (v_rcp_f32, 1 ulp) errs in the same chosen
    direction as the binary64 one: the boxes' slack has to cover it, or a culled primitive would show as a different pixel."""
    name, desc, W, H, spp, depth = _scene_cases(scenes)[case]
    lane_devmath.set_rcp_mode(mode)
    sc, cam = scenes.build_product(desc, device=-1)
    before = lane_devmath.rcp_calls()
    img, cnt, _ = lane_devmath.render(sc, cam, W, H, spp, depth, 11)
    ref, ocnt = oracle.build_oracle(desc).render(W, H, spp, depth, 11, iterative=True, nthreads=8, counters=True)
    assert lane_devmath.rcp_calls() - before > cnt["segments"], name  # at least the segment's reciprocal and one normalisation each
    assert np.array_equal(img, ref), name
    assert cnt["segments"] == ocnt["segments"], name


def test_libm_calls_of_a_pixel_are_recorded(scenes, lane_emul):
    """The hooks tests/sweeps/libm_attribution.py stands on: the harness records the arguments its lane program passes to log / sin /
    atan2 / acos (the only functions whose device results may differ from the host's), and can move every log by an ulp to
    show how little of that reaches a picture."""
    sc, cam = scenes.build_product(scenes.cover(1, 1.0), device=-1)
    W = H = 48
    calls = np.concatenate([lane_emul.trace_pixel(sc, cam, W, H, 3, 50, 5, x, y) for x, y in ((30, 20), (24, 24), (10, 40), (36, 12))])
    fns = set(int(f) for f in calls[:, 1])
    assert 0 in fns  # the fog: one keyed free-flight draw per segment at least
    import math  # the C library's functions (numpy's vectorised ones are another implementation, an ulp apart here and there)
    for f, host in ((0, math.log), (1, math.sin), (3, math.acos)):
        c = calls[calls[:, 1] == f]
        assert same_bits(c[:, 4], [host(v) for v in c[:, 2]])
    c = calls[calls[:, 1] == 2]
    assert same_bits(c[:, 4], [math.atan2(a, b) for a, b in c[:, 2:4]])
    assert np.all((calls[calls[:, 1] == 0][:, 2] > 0.0) & (calls[calls[:, 1] == 0][:, 2] < 1.0))  # log of a uniform draw
    base, *_ = lane_emul.render(sc, cam, W, H, 2, 50, 5)
    try:
        lane_emul.set_log_perturbation(1)
        moved, *_ = lane_emul.render(sc, cam, W, H, 2, 50, 5)
    finally:
        lane_emul.set_log_perturbation(0)
    again, *_ = lane_emul.render(sc, cam, W, H, 2, 50, 5)
    assert np.array_equal(base, again)
    changed = np.abs(moved - base).max(axis=2) > 0.0
    assert changed.mean() < 0.5 and np.abs(moved - base).mean() <= 1e-4  # an ulp in every free flight stays far inside the stated bar
