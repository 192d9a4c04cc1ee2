"""ray-tracer_amd -- harness glue (ctypes + numpy) around librt_mi355x.so.

The product is the C-ABI shared library built from ``csrc/`` (hand-written HIP
kernels for gfx950 + host flattener); this package only loads it and mirrors the
reference's builder vocabulary (Sprite / Sphere / Lambertian / PerspectiveCamera,
reference src/sprite.rs:22-72, src/geometry.rs, src/material.rs, src/camera.rs:25-33)
so that tests and bench.py read like the reference's example drivers.

There is no Python or CPU rendering path: if the library is missing, or no HIP
device is present, rendering raises.

The directory name contains a hyphen, so load it with ``importlib`` (see
``load_package`` in ``__graft_entry__.py``) under the module name ``ray_tracer_amd``.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
# RT_MI355X_LIB: alternative build of the same library (A/B tuning experiments only)
LIB_PATH = Path(os.environ.get("RT_MI355X_LIB") or (_HERE / "lib" / "librt_mi355x.so"))

RT_TILE = 8
RT_FLAG_COUNTERS = 1
RT_FLAG_DEFERRED_OUTPUT = 2
RT_FLAG_ASCENDING_TILES = 4
RT_TILE_ORDER_ASCENDING, RT_TILE_ORDER_LEARNT, RT_TILE_ORDER_LEARNING = 0, 1, 2
RT_FEAT_SPHERE_T, RT_FEAT_GENERAL, RT_FEAT_MEDIUM, RT_FEAT_TEXTURED, RT_FEAT_LENS, RT_FEAT_MEDIUM_GENERAL, RT_FEAT_WIDE = 1, 2, 4, 8, 16, 32, 64
RT_FEAT_DEEP_CHAIN = FEAT_DEEP_CHAIN = 128
RT_FEAT_MEDIUM_NESTED = 256
ERR_INVALID, ERR_EMPTY, ERR_DEVICE, ERR_UNSUPPORTED, ERR_STATE = -1, -2, -3, -4, -5  # include/rt_mi355x.h RT_ERR_*


class RtError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"librt_mi355x error {code}: {msg}")
        self.code = code


class rt_camera(C.Structure):
    _fields_ = [("eye", C.c_double * 3), ("lower_left", C.c_double * 3), ("horizontal", C.c_double * 3),
                ("vertical", C.c_double * 3), ("lens_radius", C.c_double)]


class rt_render_params(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("spp", C.c_int), ("max_depth", C.c_int),
                ("seed", C.c_uint64), ("shard_index", C.c_int), ("shard_count", C.c_int), ("flags", C.c_uint)]


class rt_counters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("samples", "segments", "nodes_visited", "prims_tested", "rng_draws",
                                            "node_wave", "node_lane", "leaf_wave", "leaf_lane", "shade_wave", "shade_lane",
                                            "node_cycles", "leaf_cycles", "shade_cycles", "finish_cycles", "refill_cycles",
                                            "begin_cycles", "swap_class_mode", "swap_new_mode", "swap_parked", "swap_pulled",
                                            "swap_lock_busy", "swap_scattered", "swap_off_class", "swap_cycles",
                                            "node_idle_done", "node_idle_leaf", "node_idle_empty")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


class rt_scene_info(C.Structure):
    _fields_ = [("n_prims", C.c_int), ("n_child_prims", C.c_int), ("n_hoisted", C.c_int), ("n_nodes", C.c_int),
                ("max_depth", C.c_int),
                ("n_materials", C.c_int), ("n_textures", C.c_int), ("n_xforms", C.c_int), ("node_bytes", C.c_int),
                ("prim_bytes", C.c_int), ("material_bytes", C.c_int), ("feature_mask", C.c_uint),
                ("device_bytes", C.c_size_t), ("n_list", C.c_int)]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


class rt_launch_config(C.Structure):
    _fields_ = [("blocks", C.c_int), ("block_threads", C.c_int), ("lds_bytes", C.c_uint), ("blocks_per_cu", C.c_int),
                ("n_cu", C.c_int), ("passes", C.c_int), ("n_jobs", C.c_int), ("job_spp", C.c_int),
                ("kernel_features", C.c_uint), ("lds_nodes", C.c_int), ("swap", C.c_int), ("workspace_bytes", C.c_size_t),
                ("swap_cap", C.c_int), ("waves_per_simd", C.c_int), ("tile_order", C.c_int), ("records_in_lds", C.c_int)]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


_D = C.c_double
_DP = C.POINTER(C.c_double)
_VP = C.c_void_p

# every symbol include/rt_mi355x.h declares: name -> (restype, argtypes)
ABI = {
    "rt_last_error": (C.c_char_p, []),
    "rt_version": (C.c_char_p, []),
    "rt_device_count": (C.c_int, []),
    "rt_mat4_identity": (None, [_DP]),
    "rt_mat4_translation": (None, [_DP, _DP]),
    "rt_mat4_rotation": (None, [_D, _DP, _DP]),
    "rt_mat4_multiplied": (None, [_DP, _DP, _DP]),
    "rt_mat4_determinant": (_D, [_DP]),
    "rt_mat4_inversed": (C.c_int, [_DP, _DP]),
    "rt_scene_create": (_VP, []),
    "rt_scene_destroy": (None, [_VP]),
    "rt_add_texture_solid": (C.c_int, [_VP, _DP]),
    "rt_add_texture_checker": (C.c_int, [_VP, C.c_int, C.c_int]),
    "rt_add_texture_image_rgb8": (C.c_int, [_VP, C.POINTER(C.c_uint8), C.c_int, C.c_int]),
    "rt_add_material_lambertian": (C.c_int, [_VP, C.c_int]),
    "rt_add_material_metal": (C.c_int, [_VP, C.c_int, _D]),
    "rt_add_material_dielectric": (C.c_int, [_VP, _D]),
    "rt_add_material_diffuse_light": (C.c_int, [_VP, C.c_int]),
    "rt_add_material_isotropic": (C.c_int, [_VP, C.c_int]),
    "rt_add_geometry_sphere": (C.c_int, [_VP, _D]),
    "rt_add_geometry_rectangle": (C.c_int, [_VP, _D, _D]),
    "rt_add_geometry_cube": (C.c_int, [_VP, _D, _D, _D]),
    "rt_add_geometry_constant_medium": (C.c_int, [_VP, C.c_int, _D]),
    "rt_add_geometry_transformed": (C.c_int, [_VP, C.c_int, _DP]),
    "rt_add_geometry_bvh": (C.c_int, [_VP, C.POINTER(C.c_int), C.c_int]),
    "rt_add_sprite": (C.c_int, [_VP, C.c_int, C.c_int, _DP]),
    "rt_scene_commit": (C.c_int, [_VP, C.c_int]),
    "rt_camera_perspective": (C.c_int, [C.POINTER(rt_camera), _DP, _DP, _DP, _D, _D, _D, _D]),
    "rt_render": (C.c_int, [_VP, C.POINTER(rt_camera), C.POINTER(rt_render_params), _DP, C.POINTER(rt_counters)]),
    "rt_scene_clone": (_VP, [_VP, C.c_int]),
    "rt_render_sharded": (C.c_int, [C.POINTER(_VP), C.c_int, C.POINTER(rt_camera), C.POINTER(rt_render_params), _DP]),
    "rt_render_progressive": (C.c_int, [_VP, C.POINTER(rt_camera), C.POINTER(rt_render_params), C.c_int, C.c_int, _DP]),
    "rt_shard_tile_count": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "rt_render_tiles_device": (C.c_int, [_VP, C.POINTER(rt_camera), C.POINTER(rt_render_params), _VP, _VP, _VP]),
    "rt_unpack_tiles_device": (C.c_int, [_VP, C.c_int, C.c_int, C.c_int, C.c_int, _VP, _VP]),
    "rt_pack_tiles_host": (C.c_int, [_DP, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _DP]),
    "rt_unpack_tiles_host": (C.c_int, [_DP, C.c_int, C.c_int, C.c_int, C.c_int, _DP]),
    "rt_render_status": (C.c_int, [_VP]),
    "rt_render_wait_output": (C.c_int, [_VP, _VP]),
    "rt_scene_set_workspace_limit": (C.c_int, [_VP, C.c_size_t]),
    "rt_scene_trim": (C.c_int, [_VP]),
    "rt_scene_workspace_bytes": (C.c_size_t, [_VP]),
    "rt_last_kernel_ms": (C.c_int, [_VP, C.POINTER(C.c_float)]),
    "rt_last_launch_config": (C.c_int, [_VP, C.POINTER(rt_launch_config)]),
    "rt_scene_plan_launch": (C.c_int, [_VP, C.POINTER(rt_launch_config)]),
    "rt_scene_tile_order": (C.c_int, [_VP, _VP, _VP, C.c_int]),
    "rt_tonemap_rgb8": (None, [_DP, C.c_size_t, C.POINTER(C.c_uint8)]),
    "rt_write_ppm_p3": (C.c_int, [C.c_char_p, _DP, C.c_int, C.c_int]),
    "rt_write_png_rgba8": (C.c_int, [C.c_char_p, _DP, C.c_int, C.c_int]),
    "rt_tonemap_png8": (None, [_DP, C.c_size_t, C.POINTER(C.c_uint8)]),
    "rt_scene_get_info": (C.c_int, [_VP, C.POINTER(rt_scene_info)]),
    "rt_scene_copy_nodes": (C.c_int, [_VP, _DP, C.c_int]),
    "rt_scene_hash": (C.c_int, [_VP, C.POINTER(C.c_uint64)]),
    "rt_scene_prim_bounds": (C.c_int, [_VP, C.c_int, _DP]),
    "rt_scene_prim_group": (C.c_int, [_VP, C.c_int]),
    "rt_probe_device_math": (C.c_int, [C.c_int, _DP, _DP, C.c_int, _DP, _DP]),
    "rt_probe_device_libm": (C.c_int, [C.c_int, C.c_int, _DP, _DP, C.c_int, _DP]),
}

_lib = None


def _share_torch_hip_runtime():
    """One HIP runtime per process.  The library needs `libamdhip64.so.7`; PyTorch-ROCm brings its own copy and asks for it as
    `libamdhip64.so`, so a process that loads this library BEFORE importing torch ends up with two runtimes, and the one
    initialised second finds no GPU (seen on the MI355X box: "No HIP GPUs are available" from torch after the library's first
    render).  Loaded first, torch's copy carries the soname this library asks for and both share it -- so when torch is installed
    but not imported yet, its runtime is loaded here (torch itself is not imported)."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        return
    if spec is None or not spec.origin:
        return
    hip = Path(spec.origin).parent / "lib" / "libamdhip64.so"
    if hip.exists():
        C.CDLL(str(hip), mode=C.RTLD_GLOBAL)


def lib() -> C.CDLL:
    """Load librt_mi355x.so (fails loudly if it has not been built)."""
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise FileNotFoundError(
                f"{LIB_PATH} is missing: build it with `make -C ray-tracer_amd/csrc` "
                "(or __graft_entry__.build()); there is no fallback path")
        _share_torch_hip_runtime()
        L = C.CDLL(str(LIB_PATH))
        for name, (res, args) in ABI.items():
            fn = getattr(L, name)  # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def _check(rc: int) -> int:
    if rc < 0:
        raise RtError(rc, lib().rt_last_error().decode())
    return rc


def _dp(a: np.ndarray):
    return a.ctypes.data_as(_DP)


def _vec(v, n=3):
    a = np.ascontiguousarray(v, dtype=np.float64)
    assert a.shape == (n,)
    return a


# ---------------------------------------------------------------- Mat4 (src/mat4.rs)
class Mat4:
    """Column-major 4x4, the subset the examples use (src/mat4.rs:21-47,52-80,85-143)."""

    def __init__(self, a):
        self.a = np.ascontiguousarray(a, dtype=np.float64).reshape(16)

    @staticmethod
    def identity():
        out = np.empty(16)
        lib().rt_mat4_identity(_dp(out))
        return Mat4(out)

    @staticmethod
    def translation(offset):
        out = np.empty(16)
        lib().rt_mat4_translation(_dp(_vec(offset)), _dp(out))
        return Mat4(out)

    @staticmethod
    def rotation(radians, axis):
        out = np.empty(16)
        lib().rt_mat4_rotation(float(radians), _dp(_vec(axis)), _dp(out))
        return Mat4(out)

    def multiplied(self, other: "Mat4"):
        out = np.empty(16)
        lib().rt_mat4_multiplied(_dp(self.a), _dp(other.a), _dp(out))
        return Mat4(out)

    def determinant(self):
        return float(lib().rt_mat4_determinant(_dp(self.a)))

    def inversed(self):
        out = np.empty(16)
        rc = lib().rt_mat4_inversed(_dp(self.a), _dp(out))
        return None if rc != 0 else Mat4(out)


# ---------------------------------------------------------------- scene
class Scene:
    """Records textures / materials / geometries / sprites through the C ABI."""

    def __init__(self, handle=None, committed=False):
        self._h = handle if handle is not None else lib().rt_scene_create()
        self.committed = committed

    def clone(self, device: int) -> "Scene":
        """a copy of this scene's description, committed on `device` when this one is committed (rt_scene_clone)"""
        h = lib().rt_scene_clone(self._h, device)
        if not h:
            raise RtError(-3, lib().rt_last_error().decode())
        return Scene(h, committed=self.committed)

    def status(self):
        """wait for the renders launched on this scene and raise RtError if a kernel reported a device error (rt_render_status)"""
        _check(lib().rt_render_status(self._h))

    def set_workspace_limit(self, n_bytes: int):
        _check(lib().rt_scene_set_workspace_limit(self._h, int(n_bytes)))

    def trim(self):
        """give the per-sample workspace back (rt_scene_trim)"""
        _check(lib().rt_scene_trim(self._h))

    def workspace_bytes(self) -> int:
        return int(lib().rt_scene_workspace_bytes(self._h))

    def close(self):
        if self._h:
            lib().rt_scene_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # textures
    def solid(self, rgb):
        return _check(lib().rt_add_texture_solid(self._h, _dp(_vec(rgb))))

    def checker(self, black, white):
        return _check(lib().rt_add_texture_checker(self._h, black, white))

    def image(self, rgb8: np.ndarray):
        a = np.ascontiguousarray(rgb8, dtype=np.uint8)
        h, w, c = a.shape
        assert c == 3
        return _check(lib().rt_add_texture_image_rgb8(self._h, a.ctypes.data_as(C.POINTER(C.c_uint8)), w, h))

    # materials
    def lambertian(self, tex):
        return _check(lib().rt_add_material_lambertian(self._h, tex))

    def metal(self, tex, fuzz):
        return _check(lib().rt_add_material_metal(self._h, tex, float(fuzz)))

    def dielectric(self, refractive):
        return _check(lib().rt_add_material_dielectric(self._h, float(refractive)))

    def diffuse_light(self, tex):
        return _check(lib().rt_add_material_diffuse_light(self._h, tex))

    def isotropic(self, tex):
        return _check(lib().rt_add_material_isotropic(self._h, tex))

    # geometries
    def sphere(self, r):
        return _check(lib().rt_add_geometry_sphere(self._h, float(r)))

    def rectangle(self, w, h):
        return _check(lib().rt_add_geometry_rectangle(self._h, float(w), float(h)))

    def cube(self, w, h, d):
        return _check(lib().rt_add_geometry_cube(self._h, float(w), float(h), float(d)))

    def constant_medium(self, boundary, density):
        return _check(lib().rt_add_geometry_constant_medium(self._h, boundary, float(density)))

    def transformed(self, geometry, transform):
        """TransformedGeometry::new(geometry, M) (src/geometry.rs:185-246)"""
        m = None if transform is None else _dp(np.ascontiguousarray(transform, dtype=np.float64).reshape(16))
        return _check(lib().rt_add_geometry_transformed(self._h, geometry, m))

    def bvh(self, sprites):
        """BoundingVolumeHierarchyNode::new(sprites) as a geometry: the sprites are moved into the node (instancing)"""
        arr = (C.c_int * len(sprites))(*[int(v) for v in sprites])
        return _check(lib().rt_add_geometry_bvh(self._h, arr, len(sprites)))

    def sprite(self, geometry, material, transform=None):
        m = None if transform is None else _dp(np.ascontiguousarray(transform, dtype=np.float64).reshape(16))
        return _check(lib().rt_add_sprite(self._h, -1 if geometry is None else geometry,
                                          -1 if material is None else material, m))

    def commit(self, device: int = 0):
        _check(lib().rt_scene_commit(self._h, device))
        self.committed = True
        return self

    def info(self) -> dict:
        i = rt_scene_info()
        _check(lib().rt_scene_get_info(self._h, C.byref(i)))
        return i.as_dict()

    def scene_hash(self) -> int:
        h = C.c_uint64()
        _check(lib().rt_scene_hash(self._h, C.byref(h)))
        return int(h.value)

    def nodes(self) -> np.ndarray:
        n = self.info()["n_nodes"]
        out = np.zeros((n, 28))
        _check(lib().rt_scene_copy_nodes(self._h, _dp(out), n))
        return out

    def prim_bounds(self, i) -> np.ndarray:
        out = np.zeros(6)
        _check(lib().rt_scene_prim_bounds(self._h, i, _dp(out)))
        return out

    def prim_group(self, i) -> int:
        """1: prim i is a leaf of its own; 6: the head of a cube group (one leaf for prims i .. i + 5); 0: another face of a group"""
        return _check(lib().rt_scene_prim_group(self._h, i))

    # rendering
    def render(self, cam: "Camera", width, height, spp, max_depth, seed=1, shard=(0, 1), counters=False):
        """rt_render -> (H, W, 3) float64 image, y up (row 0 = bottom); optionally counters dict."""
        p = rt_render_params(width, height, spp, max_depth, seed, shard[0], shard[1], 0)
        out = np.zeros((max(height, 1), max(width, 1), 3))  # the library validates the real values
        cnt = rt_counters() if counters else None
        _check(lib().rt_render(self._h, C.byref(cam.c), C.byref(p), _dp(out), C.byref(cnt) if counters else None))
        return (out, cnt.as_dict()) if counters else out

    def render_progressive(self, cam: "Camera", width, height, spp, max_depth, seed, s_begin, s_end, sums: np.ndarray, shard=(0, 1)):
        """Continue the raw per-pixel sums with samples [s_begin, s_end) of the spp-sample render (in place)."""
        p = rt_render_params(width, height, spp, max_depth, seed, shard[0], shard[1], 0)
        assert sums.dtype == np.float64 and sums.shape == (height, width, 3) and sums.flags.c_contiguous
        _check(lib().rt_render_progressive(self._h, C.byref(cam.c), C.byref(p), s_begin, s_end, _dp(sums)))
        return sums

    def render_tiles_device(self, cam: "Camera", width, height, spp, max_depth, seed, shard, d_out_ptr, d_counters_ptr=None,
                            stream_ptr=None, flags=0):
        p = rt_render_params(width, height, spp, max_depth, seed, shard[0], shard[1], flags)
        _check(lib().rt_render_tiles_device(self._h, C.byref(cam.c), C.byref(p), d_out_ptr, d_counters_ptr, stream_ptr))

    def wait_output(self, stream_ptr=None):
        """make `stream` wait for the output of the last render_tiles_device(flags=RT_FLAG_DEFERRED_OUTPUT) (rt_render_wait_output)"""
        _check(lib().rt_render_wait_output(self._h, stream_ptr))

    def last_kernel_ms(self) -> float:
        ms = C.c_float()
        _check(lib().rt_last_kernel_ms(self._h, C.byref(ms)))
        return float(ms.value)

    def last_launch_config(self) -> dict:
        lc = rt_launch_config()
        _check(lib().rt_last_launch_config(self._h, C.byref(lc)))
        return lc.as_dict()

    def plan_launch(self) -> dict:
        """what a render of this scene would launch (rt_scene_plan_launch): the same decision a render makes, without a device"""
        lc = rt_launch_config()
        _check(lib().rt_scene_plan_launch(self._h, C.byref(lc)))
        return lc.as_dict()

    def tile_order(self, capacity: int = 65536):
        """the learnt hand-out order of the owned tiles of the current view and the tiles' summed path lengths
        (rt_scene_tile_order; waits for the device), or None when there is none"""
        order = np.zeros(capacity, dtype=np.uint32)
        cost = np.zeros(capacity, dtype=np.uint64)
        n = lib().rt_scene_tile_order(self._h, order.ctypes.data_as(C.c_void_p), cost.ctypes.data_as(C.c_void_p), capacity)
        if n < 0:
            _check(n)
        return None if n == 0 else (order[:n].copy(), cost[:n].copy())


class Camera:
    """PerspectiveCamera::new(eye, center, up, fov, aspect, focusDistance, lensRadius) (src/camera.rs:25-33)."""

    def __init__(self, eye, center, up, fov, aspect, focus_distance, lens_radius):
        self.c = rt_camera()
        self.args = (tuple(eye), tuple(center), tuple(up), float(fov), float(aspect), float(focus_distance), float(lens_radius))
        _check(lib().rt_camera_perspective(C.byref(self.c), _dp(_vec(eye)), _dp(_vec(center)), _dp(_vec(up)), float(fov),
                                           float(aspect), float(focus_distance), float(lens_radius)))


def render_sharded(scenes, cam: "Camera", width, height, spp, max_depth, seed=1) -> np.ndarray:
    """rt_render_sharded: one host thread per committed scene copy, tiles dealt tile_id % len(scenes)"""
    p = rt_render_params(width, height, spp, max_depth, seed, 0, 1, 0)
    out = np.zeros((height, width, 3))
    arr = (_VP * len(scenes))(*[s._h for s in scenes])
    _check(lib().rt_render_sharded(arr, len(scenes), C.byref(cam.c), C.byref(p), _dp(out)))
    return out


def shard_tile_count(width, height, shard_index, shard_count) -> int:
    return _check(lib().rt_shard_tile_count(width, height, shard_index, shard_count))


def pack_tiles_host(image: np.ndarray, shard_index: int, shard_count: int, tiles_padded: int) -> np.ndarray:
    """this shard's pixels of a row-major image in the packed layout of render_tiles_device (rt_pack_tiles_host)"""
    a = np.ascontiguousarray(image, dtype=np.float64)
    h, w, _ = a.shape
    out = np.empty((tiles_padded, 64, 3))
    _check(lib().rt_pack_tiles_host(_dp(a), w, h, shard_index, shard_count, tiles_padded, _dp(out)))
    return out


def unpack_tiles_host(gathered: np.ndarray, tiles_per_shard_padded: int, shard_count: int, width: int, height: int) -> np.ndarray:
    """gathered shards [shard][tile][64][3] -> row-major image (rt_unpack_tiles_host)"""
    g = np.ascontiguousarray(gathered, dtype=np.float64)
    assert g.size == shard_count * tiles_per_shard_padded * 64 * 3
    out = np.zeros((height, width, 3))
    _check(lib().rt_unpack_tiles_host(_dp(g), tiles_per_shard_padded, shard_count, width, height, _dp(out)))
    return out


def unpack_tiles_device(d_gathered_ptr, tiles_per_shard_padded, shard_count, width, height, d_image_ptr, stream_ptr=None):
    _check(lib().rt_unpack_tiles_device(d_gathered_ptr, tiles_per_shard_padded, shard_count, width, height, d_image_ptr,
                                        stream_ptr))


def tonemap_rgb8(img: np.ndarray) -> np.ndarray:
    a = np.ascontiguousarray(img, dtype=np.float64)
    out = np.zeros(a.shape, dtype=np.uint8)
    lib().rt_tonemap_rgb8(_dp(a), a.size // 3, out.ctypes.data_as(C.POINTER(C.c_uint8)))
    return out


def write_ppm_p3(path, img: np.ndarray):
    a = np.ascontiguousarray(img, dtype=np.float64)
    h, w, _ = a.shape
    _check(lib().rt_write_ppm_p3(os.fsencode(str(path)), _dp(a), w, h))


def write_png_rgba8(path, img: np.ndarray):
    """examples/main.rs:105-135: RGBA8 PNG, rows top-down, channel = min(sqrt(c) * 255, 255) as u8"""
    a = np.ascontiguousarray(img, dtype=np.float64)
    h, w, _ = a.shape
    _check(lib().rt_write_png_rgba8(os.fsencode(str(path)), _dp(a), w, h))


def tonemap_png8(img: np.ndarray) -> np.ndarray:
    a = np.ascontiguousarray(img, dtype=np.float64)
    out = np.zeros(a.shape, dtype=np.uint8)
    lib().rt_tonemap_png8(_dp(a), a.size // 3, out.ctypes.data_as(C.POINTER(C.c_uint8)))
    return out


HASHED_SOURCES = ("csrc/rt_kernels.hip", "csrc/rt_lane.h", "csrc/rt_libm.h", "csrc/rt_libm_tables.h", "csrc/rt_types.h", "csrc/rt_lds.h", "csrc/rt_host.cpp", "csrc/rt_host.h", "csrc/rt_api.cpp",
                  "csrc/rt_scene_priv.h", "../include/rt_mi355x.h", "../include/rt_rng.h", "csrc/Makefile")


def version() -> str:
    return lib().rt_version().decode()


def build_hash() -> str:
    """source hash the loaded library was built from (rt_version)"""
    return version().rsplit("src ", 1)[-1]


def kernel_hash() -> str:
    """hash of the device sources alone (rt_version): profiles of the kernels stay valid across host-side changes"""
    v = version()
    return v.split("kernels ", 1)[1].split(" ", 1)[0] if "kernels " in v else build_hash()


def source_hash() -> str:
    """the same hash computed from the tree (csrc/Makefile: SRC_HASH)"""
    import hashlib
    h = hashlib.sha256()
    for rel in HASHED_SOURCES:
        h.update((_HERE / rel).read_bytes())
    return h.hexdigest()[:16]


def device_count() -> int:
    return int(lib().rt_device_count())


LIBM_FUNCTIONS = ("log", "sin", "acos", "atan2", "cos", "pow")


def probe_device_libm(which: str, a: np.ndarray, b: np.ndarray = None, device=0) -> np.ndarray:
    """the restated host libm on the device (csrc/rt_libm.h): log / sin / acos / atan2 (a, b) -- what the kernels call -- and cos / pow (a, b)"""
    a = np.ascontiguousarray(a, dtype=np.float64)
    b = np.ascontiguousarray(a if b is None else b, dtype=np.float64)
    out = np.zeros_like(a)
    _check(lib().rt_probe_device_libm(device, LIBM_FUNCTIONS.index(which), _dp(a), _dp(b), a.size, _dp(out)))
    return out


def probe_device_math(a: np.ndarray, b: np.ndarray, device=0):
    a = np.ascontiguousarray(a, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    s = np.zeros_like(a)
    d = np.zeros_like(a)
    _check(lib().rt_probe_device_math(device, _dp(a), _dp(b), a.size, _dp(s), _dp(d)))
    return s, d
