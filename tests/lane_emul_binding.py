"""ctypes binding of tests/_build/liblane_emul.so (host compile of the device lane program; tests only)."""
import ctypes as C
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
# the product library must be loaded first: liblane_emul only reads the committed flat scene
_LIB = C.CDLL(str(ROOT / "tests" / "_build" / "liblane_emul.so"))
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


def lds_layout(stack_entries, block, entry_bytes, node_bytes, groups_per_cu, n_queues=3):
    """rt_lds.h for a launch shape -> dict of offsets, total, aligned, the queue capacity that fits and the one the kernel uses"""
    out = (C.c_uint * 9)()
    _LIB.lane_emul_lds_layout(stack_entries, block, entry_bytes, node_bytes, groups_per_cu, n_queues, out)
    names = ("stack_off", "node_off", "job_off", "swap_off", "swap_class_bytes", "total", "aligned", "cap", "cap_effective")
    return dict(zip(names, [int(v) for v in out]))
