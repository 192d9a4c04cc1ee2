"""ctypes binding of tests/_build/liblane_emul*.so (host compiles of the device lane program; tests only).

load(name) binds one build; the module-level names are those of the host-form build (liblane_emul.so)."""
import ctypes as C
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
# the product library must be loaded first: liblane_emul only reads the committed flat scene
import types


def load(name):
    """bind tests/_build/<name> -> namespace with render, ball_check, medium_forms, lds_layout and the arithmetic probes"""
    _LIB = C.CDLL(str(ROOT / "tests" / "_build" / name))
    _DP = C.POINTER(C.c_double)
    _LIB.lane_emul_render.restype = C.c_int
    _LIB.lane_emul_render.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_int, C.c_int,
                                      C.c_int, C.c_int, _DP, _DP, C.c_int, C.c_int, C.POINTER(C.c_ulonglong), C.POINTER(C.c_int)]


    def render(scene, cam, W, H, spp, max_depth, seed=1, region=None, sample_pixel=None):
        """-> image (H,W,3), counters dict, stack high-water mark, [per-sample radiance of sample_pixel]."""
        x0, y0, x1, y1 = region if region else (0, 0, W, H)
        out = np.zeros((H, W, 3))
        cnt = (C.c_ulonglong * 5)()
        hw = C.c_int()
        samples = np.zeros((spp, 3)) if sample_pixel else None
        sx, sy = sample_pixel if sample_pixel else (-1, -1)
        rc = _LIB.lane_emul_render(scene._h, C.addressof(cam.c), W, H, spp, max_depth, seed, x0, y0, x1, y1,
                                   out.ctypes.data_as(_DP), samples.ctypes.data_as(_DP) if sample_pixel else None, sx, sy, cnt,
                                   C.byref(hw))
        assert rc == 0
        names = ("samples", "segments", "nodes_visited", "prims_tested", "rng_draws")
        res = (out, dict(zip(names, [int(c) for c in cnt])), hw.value)
        return res + (samples,) if sample_pixel else res


    _LIB.lane_emul_half_nodes.restype = None
    _LIB.lane_emul_half_nodes.argtypes = [C.c_int]
    _LIB.lane_emul_half_tree_check.restype = C.c_int
    _LIB.lane_emul_half_tree_check.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
    _LIB.lane_emul_half_toward.restype = C.c_float
    _LIB.lane_emul_half_toward.argtypes = [C.c_float, C.c_int]


    def half_nodes(on):
        """walk trees through their binary16 form (FlatScene::nodes_half, RtNodeH) where it exists, as the device does from LDS"""
        _LIB.lane_emul_half_nodes(1 if on else 0)


    def half_tree_check(scene):
        """-> (violations, nodes): every binary16 plane outside its binary32 plane by at most a step; -1: the scene has no such tree"""
        n = C.c_int()
        bad = _LIB.lane_emul_half_tree_check(scene._h, C.byref(n))
        return bad, n.value


    def half_toward(x, up):
        return float(_LIB.lane_emul_half_toward(float(x), 1 if up else 0))


    _LIB.lane_emul_ball_check.restype = C.c_int
    _LIB.lane_emul_ball_check.argtypes = [C.c_uint64, C.c_uint64, C.c_int, _DP, _DP, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]


    def ball_check(seed, stream, max_iter):
        """-> (calls, reference point, bounded point, reference (s0, s1, draws), bounded (s0, s1, draws))."""
        p, q = np.zeros(3), np.zeros(3)
        a, b = (C.c_uint64 * 3)(), (C.c_uint64 * 3)()
        calls = _LIB.lane_emul_ball_check(seed, stream, max_iter, p.ctypes.data_as(_DP), q.ctypes.data_as(_DP), a, b)
        return calls, p, q, tuple(int(v) for v in a), tuple(int(v) for v in b)


    _LIB.lane_emul_medium_forms.restype = None
    _LIB.lane_emul_medium_forms.argtypes = [_DP, _DP, C.c_double, C.c_double, C.c_uint64, C.c_uint32, C.c_uint32, _DP]


    def medium_forms(oc, d, radius, density, base=12345, segment=0, slot=0):
        """-> ((hit, t) of the traversal form, (hit, t) of the record form) of ConstantMedium<Sphere>::hit."""
        oc, d, out = np.ascontiguousarray(oc, dtype=np.float64), np.ascontiguousarray(d, dtype=np.float64), np.zeros(4)
        _LIB.lane_emul_medium_forms(oc.ctypes.data_as(_DP), d.ctypes.data_as(_DP), radius, density, base, segment, slot, out.ctypes.data_as(_DP))
        return (bool(out[0]), float(out[1])), (bool(out[2]), float(out[3]))


    _LIB.lane_emul_lds_layout.restype = None
    _LIB.lane_emul_lds_layout.argtypes = [C.c_uint] * 6 + [C.POINTER(C.c_uint)]


    def lds_layout(stack_entries, block, entry_bytes, node_bytes, groups_per_cu, front_bytes=0):
        """rt_lds.h for a launch shape -> dict of offsets, total, aligned, the queue capacity that fits and the one the kernel uses"""
        out = (C.c_uint * 9)()
        _LIB.lane_emul_lds_layout(stack_entries, block, entry_bytes, node_bytes, groups_per_cu, front_bytes, out)
        names = ("stack_off", "node_off", "job_off", "swap_off", "swap_class_bytes", "total", "aligned", "cap", "cap_effective")
        return dict(zip(names, [int(v) for v in out]))

    _LIB.lane_emul_scene_blob_check.restype = C.c_int
    _LIB.lane_emul_scene_blob_check.argtypes = [C.c_void_p, C.POINTER(C.c_uint), C.POINTER(C.c_int)]

    def scene_blob_check(scene):
        """the records a box-LIST kernel keeps in LDS (FlatScene::scene_blob) -> (verdict, bytes, n_list): verdict 0 = every array
        is in the blob byte for byte at a 16-byte offset, -1 = the scene has no blob"""
        nbytes, n_list = C.c_uint(), C.c_int()
        v = _LIB.lane_emul_scene_blob_check(scene._h, C.byref(nbytes), C.byref(n_list))
        return int(v), int(nbytes.value), int(n_list.value)

    _U64P = C.POINTER(C.c_uint64)
    _LIB.lane_emul_device_math.restype = C.c_int
    _LIB.lane_emul_set_rcp_mode.argtypes = [C.c_int]
    _LIB.lane_emul_rcp_calls.restype = C.c_ulonglong
    _LIB.lane_emul_world_hit.restype = C.c_int
    _LIB.lane_emul_world_hit.argtypes = [C.c_void_p, C.c_void_p, _DP, _DP, C.c_uint64, C.c_uint64, _DP]

    def world_hit(scene, cam, o, d, seed=1, stream=0):
        """the nearest hit of the ray o + t d as the lane program finds it -> (t, prim) or None"""
        o, d, out = _d(o), _d(d), np.zeros(2)
        rc = _LIB.lane_emul_world_hit(scene._h, C.addressof(cam.c), o.ctypes.data_as(_DP), d.ctypes.data_as(_DP), seed, stream, out.ctypes.data_as(_DP))
        assert rc >= 0
        return (float(out[0]), int(out[1])) if rc else None
    _LIB.lane_emul_set_log_perturbation.argtypes = [C.c_int]
    _LIB.lane_emul_trace_pixel.restype = C.c_long
    _LIB.lane_emul_trace_pixel.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_int, C.c_int, _DP, C.c_long]

    def trace_pixel(scene, cam, W, H, spp, max_depth, seed, x, y, max_calls=1 << 16):
        """the lane program's libm calls for pixel (x, y), in program order -> (n, 5) array of {sample, fn (0 log, 1 sin, 2 atan2, 3 acos), a, b, host result}"""
        out = np.zeros((max_calls, 5))
        n = _LIB.lane_emul_trace_pixel(scene._h, C.addressof(cam.c), W, H, spp, max_depth, seed, x, y, out.ctypes.data_as(_DP), max_calls)
        assert 0 <= n <= max_calls, n
        return out[:n]
    _LIB.lane_emul_trace_segments.restype = C.c_long
    _LIB.lane_emul_trace_segments.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_int, C.c_int, _DP, C.c_long]

    def trace_segments(scene, cam, W, H, spp, max_depth, seed, x, y, max_rows=1 << 16):
        """the segments of pixel (x, y)'s samples as the lane program walks them -> (n, 9) array of {sample, o[3], d[3], t (inf: no hit), prim}"""
        out = np.zeros((max_rows, 9))
        n = _LIB.lane_emul_trace_segments(scene._h, C.addressof(cam.c), W, H, spp, max_depth, seed, x, y, out.ctypes.data_as(_DP), max_rows)
        assert 0 <= n <= max_rows, n
        return out[:n]
    _LIB.lane_emul_div3.argtypes = [C.c_long, _DP, _DP, _DP]
    _LIB.lane_emul_sphere_roots.argtypes = [C.c_long, _DP, _DP, _DP, _DP]
    _LIB.lane_emul_sphere_t_world.argtypes = [C.c_long, _DP, _DP, _DP, _DP]
    _LIB.lane_emul_rng_forms.argtypes = [C.c_long, _U64P, _U64P, _U64P, _U64P, _DP]

    def _d(a):
        return np.ascontiguousarray(a, dtype=np.float64)

    def div3(a, s):
        """(n,3) / (n,) through the lane program's Vec3 / f64"""
        a, s = _d(a), _d(s)
        out = np.empty_like(a)
        _LIB.lane_emul_div3(len(s), a.ctypes.data_as(_DP), s.ctypes.data_as(_DP), out.ctypes.data_as(_DP))
        return out

    def sphere_roots(n1, n2, den):
        n1, n2, den = _d(n1), _d(n2), _d(den)
        out = np.empty((len(den), 2))
        _LIB.lane_emul_sphere_roots(len(den), n1.ctypes.data_as(_DP), n2.ctypes.data_as(_DP), den.ctypes.data_as(_DP), out.ctypes.data_as(_DP))
        return out

    def sphere_t_world(oc, d, radius):
        oc, d, radius = _d(oc), _d(d), _d(radius)
        out = np.empty(len(radius))
        _LIB.lane_emul_sphere_t_world(len(radius), oc.ctypes.data_as(_DP), d.ctypes.data_as(_DP), radius.ctypes.data_as(_DP), out.ctypes.data_as(_DP))
        return out

    def rng_forms(x):
        """-> rotl(x, 24), rotl(x, 37), rotl(x, 16), rt_u64_to_pm1(x)"""
        x = np.ascontiguousarray(x, dtype=np.uint64)
        r = [np.empty_like(x) for _ in range(3)]
        pm1 = np.empty(len(x))
        _LIB.lane_emul_rng_forms(len(x), x.ctypes.data_as(_U64P), *[v.ctypes.data_as(_U64P) for v in r], pm1.ctypes.data_as(_DP))
        return r[0], r[1], r[2], pm1

    ns = types.SimpleNamespace(render=render, ball_check=ball_check, medium_forms=medium_forms, lds_layout=lds_layout, div3=div3,
                               sphere_roots=sphere_roots, sphere_t_world=sphere_t_world, rng_forms=rng_forms,
                               device_math=bool(_LIB.lane_emul_device_math()), set_rcp_mode=_LIB.lane_emul_set_rcp_mode,
                               rcp_calls=lambda: int(_LIB.lane_emul_rcp_calls()), trace_pixel=trace_pixel, trace_segments=trace_segments, world_hit=world_hit, scene_blob_check=scene_blob_check, set_log_perturbation=_LIB.lane_emul_set_log_perturbation, half_nodes=half_nodes,
                               half_tree_check=half_tree_check, half_toward=half_toward, name=name)
    return ns


globals().update(vars(load("liblane_emul.so")))
