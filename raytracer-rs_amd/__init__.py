"""ctypes bindings of the MI355X path-tracing backend.

Two shared libraries, both built in-tree by the top-level Makefile:

* ``lib/librt_hip.so``  — the device library behind ``include/rt_api.h`` (HIP kernels for gfx950 + C ABI).
* ``lib/librt_host.so`` — the host scene library behind ``include/rt_host.h`` (Scene/World/Mesh/Material/Light
  mirror of the reference, OBJ loader, scene catalogue, flattening).

There is no CPU fallback: rendering needs ``librt_hip.so`` and a GPU, and loading fails loudly otherwise.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
REPO_ROOT = os.path.dirname(_HERE)
LIB_DIR = os.path.join(_HERE, "lib")
ASSET_DIR = os.path.join(REPO_ROOT, "assets")

RT_OK = 0
RT_ERR_INVALID, RT_ERR_UNSUPPORTED, RT_ERR_DEVICE, RT_ERR_NOMEM = 1, 2, 3, 4
MAT_DIFFUSE, MAT_METAL, MAT_DIELECTRIC, MAT_MICROFACET = 0, 1, 2, 3
LIGHT_POINT, LIGHT_DISK, LIGHT_POLYGON = 0, 1, 2
SAMPLER_WHITE, SAMPLER_UNIFORM, SAMPLER_JITTERED, SAMPLER_BLUE = 1, 2, 3, 4
OBJ_TRIANGLE, OBJ_SPHERE, OBJ_CYLINDER, OBJ_MESH = 0, 1, 2, 3
ACCEL_NAIVE, ACCEL_BOX, ACCEL_BVH = 0, 1, 2
SHADER_NORMAL, SHADER_WHITTED, SHADER_PATH, SHADER_MIS = 0, 1, 2, 3
FLAG_TIMING, FLAG_COUNT_WORK = 1, 2
FLAG_RUSSIAN_ROULETTE, FLAG_COSINE_DIFFUSE = 4, 8  # options the reference does not have (off by default)
RR_MIN_DEPTH = 3
SCENE_BVH_ON_DEVICE = 1

c_float_p = C.POINTER(C.c_float)
c_int32_p = C.POINTER(C.c_int32)


class Material(C.Structure):
    _fields_ = [("kind", C.c_int32), ("albedo", C.c_float * 3), ("f0", C.c_float * 3), ("roughness", C.c_float),
                ("metallic", C.c_float), ("ref_idx", C.c_float)]


class Light(C.Structure):
    _fields_ = [("kind", C.c_int32), ("origin", C.c_float * 3), ("radius", C.c_float), ("spectrum", C.c_float * 3),
                ("tri_first", C.c_int32), ("tri_count", C.c_int32), ("sampler_kind", C.c_int32)]


class Object(C.Structure):
    _fields_ = [("kind", C.c_int32), ("first", C.c_int32), ("count", C.c_int32), ("accel", C.c_int32)]


class SceneDesc(C.Structure):
    _fields_ = [("n_tri", C.c_int32), ("tri_v", c_float_p), ("tri_mat", c_int32_p),
                ("n_sph", C.c_int32), ("sph", c_float_p), ("sph_mat", c_int32_p),
                ("n_cyl", C.c_int32), ("cyl", c_float_p), ("cyl_mat", c_int32_p),
                ("n_mat", C.c_int32), ("mats", C.POINTER(Material)),
                ("n_light", C.c_int32), ("lights", C.POINTER(Light)),
                ("n_light_tri", C.c_int32), ("light_tri_v", c_float_p),
                ("n_obj", C.c_int32), ("objects", C.POINTER(Object))]


class Camera(C.Structure):
    _fields_ = [("origin", C.c_float * 3), ("llc", C.c_float * 3), ("horizontal", C.c_float * 3), ("vertical", C.c_float * 3)]


class Params(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("spp", C.c_int32), ("spp_begin", C.c_int32),
                ("spp_end", C.c_int32), ("tile_size", C.c_int32), ("shard_rank", C.c_int32), ("shard_count", C.c_int32),
                ("shader", C.c_int32), ("pixel_sampler", C.c_int32), ("light_samples", C.c_int32), ("max_depth", C.c_int32),
                ("t_min", C.c_float), ("t_max", C.c_float), ("seed", C.c_uint32), ("flags", C.c_uint32)]


class SceneInfo(C.Structure):
    _fields_ = [("n_triangles", C.c_int32), ("n_nodes", C.c_int32), ("n_orders", C.c_int32), ("n_top", C.c_int32),
                ("bvh_on_device", C.c_int32), ("n_loose", C.c_int32), ("bvh_build_ms", C.c_double)]


class Stats(C.Structure):
    _fields_ = [("camera_samples", C.c_uint64), ("rays_closest", C.c_uint64), ("rays_shadow", C.c_uint64),
                ("nodes_visited", C.c_uint64), ("tris_tested", C.c_uint64), ("iterations", C.c_uint64),
                ("launches_extend", C.c_uint64), ("launches_shade", C.c_uint64), ("launches_shadow", C.c_uint64),
                ("launches_raygen", C.c_uint64), ("ms_total", C.c_double), ("ms_extend", C.c_double),
                ("ms_shade", C.c_double), ("ms_shadow", C.c_double), ("ms_raygen", C.c_double), ("device_bytes", C.c_uint64), ("rays_shadow_inline", C.c_uint64), ("nodes_carrying", C.c_uint64),
                ("nodes_visited_shadow", C.c_uint64), ("tris_tested_shadow", C.c_uint64), ("tris_scanned", C.c_uint64)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


def default_params(width: int, height: int, spp: int, **kw) -> Params:
    """The reference's constants (src/lib.rs:17-24, src/shader.rs:94): NS2=9, depth 40, T_MIN=1e-3, T_MAX=1e7."""
    p = Params(width=width, height=height, spp=spp, spp_begin=0, spp_end=spp, tile_size=0, shard_rank=0, shard_count=1,
               shader=SHADER_MIS, pixel_sampler=SAMPLER_BLUE, light_samples=9, max_depth=40, t_min=0.001, t_max=10000000.0,
               seed=1, flags=0)
    for k, v in kw.items():
        if not hasattr(p, k):
            raise AttributeError(k)
        setattr(p, k, v)
    return p


class RtError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"rt error {code}: {msg}")
        self.code = code


# ----------------------------------------------------------------------------------------------- host library
_host = None


def host_lib() -> C.CDLL:
    global _host
    if _host is None:
        path = os.path.join(LIB_DIR, "librt_host.so")
        if not os.path.exists(path):
            raise ImportError(f"{path} is missing: run `make` at the repository root")
        lib = C.CDLL(path)
        lib.rth_last_error.restype = C.c_char_p
        lib.rth_scene_create.argtypes = [C.c_char_p, C.c_float, C.c_int32, C.c_char_p, C.POINTER(C.c_void_p)]
        lib.rth_scene_create.restype = C.c_int32
        lib.rth_scene_destroy.argtypes = [C.c_void_p]
        lib.rth_scene_destroy.restype = None
        lib.rth_scene_desc.argtypes = [C.c_void_p]
        lib.rth_scene_desc.restype = C.POINTER(SceneDesc)
        lib.rth_scene_camera.argtypes = [C.c_void_p]
        lib.rth_scene_camera.restype = C.POINTER(Camera)
        lib.rth_camera_new.argtypes = [c_float_p, c_float_p, c_float_p, C.c_float, C.c_float, C.POINTER(Camera)]
        lib.rth_camera_new.restype = None
        lib.rth_transform_points.argtypes = [c_float_p, C.c_int32, C.c_float, c_float_p, C.c_float, C.c_float, C.c_float]
        lib.rth_transform_points.restype = None
        lib.rth_load_obj.argtypes = [C.c_char_p, C.POINTER(c_float_p)]
        lib.rth_load_obj.restype = C.c_int32
        lib.rth_free.argtypes = [C.c_void_p]
        lib.rth_free.restype = None
        lib.rth_write_ppm.argtypes = [C.c_char_p, C.c_int32, C.c_int32, c_int32_p]
        lib.rth_write_ppm.restype = C.c_int32
        lib.rth_write_png.argtypes = [C.c_char_p, C.c_int32, C.c_int32, c_int32_p]
        lib.rth_write_png.restype = C.c_int32
        lib.rth_upscale_nearest.argtypes = [C.c_int32, C.c_int32, C.c_int32, c_int32_p, c_int32_p]
        lib.rth_upscale_nearest.restype = C.c_int32
        lib.rth_accum_save.argtypes = [C.c_char_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_uint32, c_float_p]
        lib.rth_accum_save.restype = C.c_int32
        lib.rth_accum_load.argtypes = [C.c_char_p, C.c_int32, C.c_int32, C.c_int32, C.c_uint32, C.POINTER(C.c_int32), c_float_p]
        lib.rth_accum_load.restype = C.c_int32
        lib.rth_render_tag.argtypes = [C.c_char_p]
        lib.rth_render_tag.restype = C.c_uint32
        lib.rth_accum_save_tagged.argtypes = [C.c_char_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_uint32, C.c_uint32, c_float_p]
        lib.rth_accum_save_tagged.restype = C.c_int32
        lib.rth_accum_load_tagged.argtypes = [C.c_char_p, C.c_int32, C.c_int32, C.c_int32, C.c_uint32, C.c_uint32, C.POINTER(C.c_int32), c_float_p]
        lib.rth_accum_load_tagged.restype = C.c_int32
        _host = lib
    return _host


def _fp(a: np.ndarray):
    return a.ctypes.data_as(c_float_p)


def _ip(a: np.ndarray):
    return a.ctypes.data_as(c_int32_p)


class HostScene:
    """A scene of the catalogue (include/rt_host.h), flattened.  `.desc` / `.camera` are ctypes views."""

    def __init__(self, name: str, aspect: float = 1.0, light_sampler: int = SAMPLER_BLUE, asset_dir: Optional[str] = None):
        lib = host_lib()
        h = C.c_void_p()
        rc = lib.rth_scene_create(name.encode(), aspect, light_sampler, (asset_dir or ASSET_DIR).encode(), C.byref(h))
        if rc != RT_OK:
            raise RtError(rc, lib.rth_last_error().decode())
        self._h = h
        self.name = name
        self.desc = lib.rth_scene_desc(h).contents
        self.camera = lib.rth_scene_camera(h).contents

    def close(self):
        if self._h:
            host_lib().rth_scene_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class NumpyScene:
    """A scene description assembled from numpy arrays (for tests and custom callers).  Keeps the arrays alive."""

    def __init__(self, tri_v=None, tri_mat=None, sph=None, sph_mat=None, cyl=None, cyl_mat=None, mats=(), lights=(),
                 light_tri_v=None, objects=None):
        def f32(a, w):
            a = np.zeros((0, w), np.float32) if a is None else np.ascontiguousarray(a, np.float32).reshape(-1, w)
            return a

        def i32(a):
            return np.zeros((0,), np.int32) if a is None else np.ascontiguousarray(a, np.int32).reshape(-1)

        self.tri_v, self.tri_mat = f32(tri_v, 9), i32(tri_mat)
        self.sph, self.sph_mat = f32(sph, 4), i32(sph_mat)
        self.cyl, self.cyl_mat = f32(cyl, 5), i32(cyl_mat)
        self.light_tri_v = f32(light_tri_v, 9)
        self.mats = (Material * max(1, len(mats)))(*mats)
        self.lights = (Light * max(1, len(lights)))(*lights)
        if objects is None:  # default world order: every triangle, then spheres, then cylinders, as single objects
            objects = [Object(OBJ_TRIANGLE, i, 1, 0) for i in range(len(self.tri_mat))]
            objects += [Object(OBJ_SPHERE, i, 1, 0) for i in range(len(self.sph_mat))]
            objects += [Object(OBJ_CYLINDER, i, 1, 0) for i in range(len(self.cyl_mat))]
        self.objects = (Object * max(1, len(objects)))(*objects)
        d = SceneDesc()
        d.n_tri, d.tri_v, d.tri_mat = len(self.tri_mat), _fp(self.tri_v), _ip(self.tri_mat)
        d.n_sph, d.sph, d.sph_mat = len(self.sph_mat), _fp(self.sph), _ip(self.sph_mat)
        d.n_cyl, d.cyl, d.cyl_mat = len(self.cyl_mat), _fp(self.cyl), _ip(self.cyl_mat)
        d.n_mat, d.mats = len(mats), self.mats
        d.n_light, d.lights = len(lights), self.lights
        d.n_light_tri, d.light_tri_v = len(self.light_tri_v), _fp(self.light_tri_v)
        d.n_obj, d.objects = len(objects), self.objects
        self.desc = d


def diffuse(r, g, b) -> Material:
    return Material(MAT_DIFFUSE, (r, g, b), (0, 0, 0), 0, 0, 0)


def metal(r, g, b) -> Material:
    return Material(MAT_METAL, (r, g, b), (0, 0, 0), 0, 0, 0)


def dielectric(ref_idx) -> Material:
    return Material(MAT_DIELECTRIC, (0, 0, 0), (0, 0, 0), 0, 0, ref_idx)


def microfacet(f0, roughness, metallic, attenuation) -> Material:
    return Material(MAT_MICROFACET, tuple(attenuation), tuple(f0), roughness, metallic, 0)


def camera_new(lookfrom, lookat, vup, vfov_deg: float, aspect: float) -> Camera:
    cam = Camera()
    a = [np.asarray(v, np.float32) for v in (lookfrom, lookat, vup)]
    host_lib().rth_camera_new(_fp(a[0]), _fp(a[1]), _fp(a[2]), vfov_deg, aspect, C.byref(cam))
    return cam


def load_obj(path: str) -> np.ndarray:
    """Triangles of an OBJ file as an (n, 9) float32 array (src/obj_loader.rs:16-47)."""
    lib = host_lib()
    p = c_float_p()
    n = lib.rth_load_obj(path.encode(), C.byref(p))
    if n < 0:
        raise RtError(-n, lib.rth_last_error().decode())
    out = np.ctypeslib.as_array(p, shape=(n, 9)).copy() if n else np.zeros((0, 9), np.float32)
    lib.rth_free(p)
    return out


def transform_points(pts: np.ndarray, scale: float, disp, x: float, y: float, z: float) -> np.ndarray:
    """Mesh::transform (src/mesh.rs:23-39) on an (n, 3) array; returns a new array."""
    out = np.ascontiguousarray(pts, np.float32).reshape(-1, 3).copy()
    d = np.asarray(disp, np.float32)
    host_lib().rth_transform_points(_fp(out), len(out), scale, _fp(d), x, y, z)
    return out


def write_ppm(path: str, rgb_i32: np.ndarray) -> None:
    h, w, _ = rgb_i32.shape
    a = np.ascontiguousarray(rgb_i32, np.int32)
    rc = host_lib().rth_write_ppm(path.encode(), w, h, _ip(a))
    if rc != RT_OK:
        raise RtError(rc, host_lib().rth_last_error().decode())


def write_png(path: str, rgb_i32: np.ndarray) -> None:
    """rth_write_png: 8-bit RGB PNG of the integers of resolve(); values outside 0..255 saturate."""
    h, w, _ = rgb_i32.shape
    a = np.ascontiguousarray(rgb_i32, np.int32)
    rc = host_lib().rth_write_png(path.encode(), w, h, _ip(a))
    if rc != RT_OK:
        raise RtError(rc, host_lib().rth_last_error().decode())


def upscale_nearest(rgb_i32: np.ndarray, rate: int) -> np.ndarray:
    """rth_upscale_nearest: every pixel repeated rate x rate times (resize.py:5-11)."""
    h, w, _ = rgb_i32.shape
    a = np.ascontiguousarray(rgb_i32, np.int32)
    out = np.empty((h * max(rate, 0), w * max(rate, 0), 3), np.int32)
    rc = host_lib().rth_upscale_nearest(w, h, rate, _ip(a), _ip(out))
    if rc != RT_OK:
        raise RtError(rc, host_lib().rth_last_error().decode())
    return out


def save_accum(path: str, accum: np.ndarray, spp: int, spp_done: int, seed: int = 1) -> None:
    """rth_accum_save: the sum buffer and the number of samples already in it, to be picked up by load_accum."""
    h, w, _ = accum.shape
    a = np.ascontiguousarray(accum, np.float32)
    rc = host_lib().rth_accum_save(path.encode(), w, h, spp, spp_done, seed, _fp(a))
    if rc != RT_OK:
        raise RtError(rc, host_lib().rth_last_error().decode())


def load_accum(path: str, width: int, height: int, spp: int, seed: int = 1) -> Tuple[np.ndarray, int]:
    """rth_accum_load: (sum buffer, samples done); RtError if the file was saved for another frame, sample count or seed."""
    a = np.empty((height, width, 3), np.float32)
    done = C.c_int32()
    rc = host_lib().rth_accum_load(path.encode(), width, height, spp, seed, C.byref(done), _fp(a))
    if rc != RT_OK:
        raise RtError(rc, host_lib().rth_last_error().decode())
    return a, done.value


# ----------------------------------------------------------------------------------------------- device library
_hip = None


def _share_hip_runtime_with_torch() -> None:
    """A process must hold ONE HIP runtime.  PyTorch wheels carry their own libamdhip64.so (same soname as the system
    one); if librt_hip.so pulled in the system copy first, a later `import torch` would find no GPU.  So when torch
    is installed but not yet imported, its copy is loaded first and both sides resolve to it."""
    import importlib.util
    import sys

    if "torch" in sys.modules or os.environ.get("RT_SYSTEM_HIP"):
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def hip_lib() -> C.CDLL:
    """Loads librt_hip.so.  Raises ImportError if it has not been built: there is no CPU substitute."""
    global _hip
    if _hip is None:
        path = os.environ.get("RT_HIP_LIB") or os.path.join(LIB_DIR, "librt_hip.so")  # RT_HIP_LIB: A/B builds of the same ABI
        if not os.path.exists(path):
            raise ImportError(f"{path} is missing: run `make` at the repository root (needs hipcc)")
        _share_hip_runtime_with_torch()
        lib = C.CDLL(path)
        lib.rt_last_error.restype = C.c_char_p
        lib.rt_context_create.argtypes = [C.c_int32, C.POINTER(C.c_void_p)]
        lib.rt_context_create.restype = C.c_int32
        lib.rt_context_destroy.argtypes = [C.c_void_p]
        lib.rt_context_destroy.restype = C.c_int32
        lib.rt_scene_create.argtypes = [C.c_void_p, C.POINTER(SceneDesc), C.POINTER(C.c_void_p)]
        lib.rt_scene_create.restype = C.c_int32
        lib.rt_scene_create_ex.argtypes = [C.c_void_p, C.POINTER(SceneDesc), C.c_uint32, C.POINTER(C.c_void_p)]
        lib.rt_scene_create_ex.restype = C.c_int32
        lib.rt_scene_get_info.argtypes = [C.c_void_p, C.POINTER(SceneInfo)]
        lib.rt_scene_get_info.restype = C.c_int32
        lib.rt_scene_destroy.argtypes = [C.c_void_p]
        lib.rt_scene_destroy.restype = C.c_int32
        lib.rt_render.argtypes = [C.c_void_p, C.POINTER(Camera), C.POINTER(Params), C.c_void_p, C.c_int32, C.POINTER(Stats)]
        lib.rt_render.restype = C.c_int32
        lib.rt_render_static.argtypes = [C.c_void_p, C.POINTER(Camera), C.POINTER(Params), C.c_void_p, C.POINTER(Stats)]
        lib.rt_render_static.restype = C.c_int32
        lib.rt_primary_hits.argtypes = [C.c_void_p, C.POINTER(Camera), C.POINTER(Params), c_int32_p, c_float_p]
        lib.rt_primary_hits.restype = C.c_int32
        lib.rt_resolve.argtypes = [c_float_p, C.c_int64, C.c_int32, c_int32_p, C.POINTER(C.c_uint8)]
        lib.rt_resolve.restype = C.c_int32
        lib.rt_resolve_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p]
        lib.rt_resolve_device.restype = C.c_int32
        lib.rt_accum_create.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]
        lib.rt_accum_create.restype = C.c_int32
        lib.rt_accum_destroy.argtypes = [C.c_void_p, C.c_void_p]
        lib.rt_accum_destroy.restype = C.c_int32
        lib.rt_accum_download.argtypes = [C.c_void_p, C.c_void_p, c_float_p, C.c_int64]
        lib.rt_accum_download.restype = C.c_int32
        lib.rt_reduce_accum.argtypes = [C.POINTER(C.c_void_p), C.c_int32, C.POINTER(C.c_void_p), C.c_int64, C.c_int32]
        lib.rt_reduce_accum.restype = C.c_int32
        lib.rt_context_stream.argtypes = [C.c_void_p]
        lib.rt_context_stream.restype = C.c_void_p
        lib.rt_params_default.argtypes = [C.POINTER(Params), C.c_int32, C.c_int32, C.c_int32]
        lib.rt_params_default.restype = None
        lib.rt_bvh_build_host.argtypes = [c_float_p, C.c_int32, C.c_int32, C.c_int32, c_float_p, c_float_p, c_int32_p, c_int32_p, c_int32_p]
        lib.rt_bvh_build_host.restype = C.c_int32
        lib.rt_bvh_build_host_compact.argtypes = [c_float_p, C.c_int32, C.c_int32, C.POINTER(C.c_uint32), c_float_p, c_int32_p, c_int32_p]
        lib.rt_bvh_build_host_compact.restype = C.c_int32
        _hip = lib
    return _hip


def bvh_build_host(tri_v: np.ndarray, max_top: int = 32, ordered: bool = True):
    """rt_bvh_build_host: the device library's BVH for `tri_v` ((n, 9) float32), built on the host (no GPU needed).

    Returns (nodes, tris, n_top): nodes is (n_orders, n_nodes, 8) float32 with columns xmin, xmax, ymin, ymax, zmin,
    zmax, link, miss (the last two int32 bit patterns), tris is (n, 12) float32 in leaf order."""
    tri_v = np.ascontiguousarray(tri_v, np.float32).reshape(-1, 9)
    n = tri_v.shape[0]
    nn, nt, no = C.c_int32(), C.c_int32(), C.c_int32()
    lib = hip_lib()
    _check(lib.rt_bvh_build_host(_fp(tri_v), n, max_top, int(ordered), None, None, C.byref(nn), C.byref(nt), C.byref(no)))
    nodes = np.zeros((no.value, nn.value, 8), np.float32)
    tris = np.zeros((n, 12), np.float32)
    _check(lib.rt_bvh_build_host(_fp(tri_v), n, max_top, int(ordered), _fp(nodes), _fp(tris), C.byref(nn), C.byref(nt), C.byref(no)))
    return nodes, tris, nt.value


def bvh_build_host_compact(tri_v: np.ndarray, ordered: bool = True):
    """rt_bvh_build_host_compact: (cnodes (n_orders, n_nodes, 4) uint32, q_params (6,) float32 = q_lo xyz, q_scale xyz)."""
    tri_v = np.ascontiguousarray(tri_v, np.float32).reshape(-1, 9)
    nn, no = C.c_int32(), C.c_int32()
    q = np.zeros(6, np.float32)
    lib = hip_lib()
    _check(lib.rt_bvh_build_host_compact(_fp(tri_v), tri_v.shape[0], int(ordered), None, _fp(q), C.byref(nn), C.byref(no)))
    cn = np.zeros((no.value, nn.value, 4), np.uint32)
    _check(lib.rt_bvh_build_host_compact(_fp(tri_v), tri_v.shape[0], int(ordered), cn.ctypes.data_as(C.POINTER(C.c_uint32)), _fp(q), C.byref(nn), C.byref(no)))
    return cn, q


def _check(rc: int):
    if rc != RT_OK:
        raise RtError(rc, hip_lib().rt_last_error().decode())


class Context:
    """One per device / process (rt_context_create)."""

    def __init__(self, device: int = 0):
        import weakref

        self._scenes = weakref.WeakSet()  # scenes must be destroyed before their context
        self._h = C.c_void_p()
        _check(hip_lib().rt_context_create(device, C.byref(self._h)))
        self.device = device

    @property
    def stream(self) -> int:
        return hip_lib().rt_context_stream(self._h) or 0

    def close(self):
        if self._h:
            for sc in list(self._scenes):
                sc.close()
            hip_lib().rt_context_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DeviceAccum:
    """rt_accum_create: a zeroed (H, W, 3) float32 sum buffer on a context's device, owned by the library."""

    def __init__(self, ctx: Context, width: int, height: int):
        self.ctx, self.width, self.height = ctx, width, height
        self._p = C.c_void_p()
        _check(hip_lib().rt_accum_create(ctx._h, width, height, C.byref(self._p)))

    @property
    def ptr(self) -> int:
        return self._p.value or 0

    def download(self) -> np.ndarray:
        out = np.empty((self.height, self.width, 3), np.float32)
        _check(hip_lib().rt_accum_download(self.ctx._h, self._p, _fp(out), out.size))
        return out

    def close(self):
        if self._p and self.ctx._h:
            hip_lib().rt_accum_destroy(self.ctx._h, self._p)
        self._p = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def reduce_accum_devices(accums, root: int = 0) -> None:
    """rt_reduce_accum: accums[root] += the other buffers (DeviceAccum objects of one frame size, one per context of this
    process); RCCL between devices, a kernel between buffers that share a device."""
    n = len(accums)
    ctxs = (C.c_void_p * n)(*[a.ctx._h for a in accums])
    ptrs = (C.c_void_p * n)(*[a._p for a in accums])
    _check(hip_lib().rt_reduce_accum(ctxs, n, ptrs, accums[0].width * accums[0].height * 3, root))


class DeviceScene:
    """Scene uploaded to the GPU (rt_scene_create): BVH built, buffers resident."""

    def __init__(self, ctx: Context, desc: SceneDesc, keepalive=None, bvh_on_device: Optional[bool] = None):
        self.ctx = ctx
        self._keep = keepalive
        self._h = C.c_void_p()
        if bvh_on_device is None:  # rt_scene_create: host builder unless RT_BVH_DEVICE=1
            _check(hip_lib().rt_scene_create(ctx._h, C.byref(desc), C.byref(self._h)))
        else:
            _check(hip_lib().rt_scene_create_ex(ctx._h, C.byref(desc), SCENE_BVH_ON_DEVICE if bvh_on_device else 0, C.byref(self._h)))
        ctx._scenes.add(self)

    @property
    def info(self) -> SceneInfo:
        i = SceneInfo()
        _check(hip_lib().rt_scene_get_info(self._h, C.byref(i)))
        return i

    def render(self, cam: Camera, params: Params, accum: Optional[np.ndarray] = None) -> Tuple[np.ndarray, Stats]:
        """Adds the selected samples into a host (H, W, 3) float32 sum buffer (created zeroed if None)."""
        if accum is None:
            accum = np.zeros((params.height, params.width, 3), np.float32)
        assert accum.dtype == np.float32 and accum.flags.c_contiguous and accum.size == params.width * params.height * 3
        st = Stats()
        _check(hip_lib().rt_render(self._h, C.byref(cam), C.byref(params), accum.ctypes.data_as(C.c_void_p), 0, C.byref(st)))
        return accum, st

    def render_device(self, cam: Camera, params: Params, accum_ptr: int) -> Stats:
        """Same, into a device buffer of width*height*3 floats given by address (e.g. torch.Tensor.data_ptr())."""
        st = Stats()
        _check(hip_lib().rt_render(self._h, C.byref(cam), C.byref(params), C.c_void_p(accum_ptr), 1, C.byref(st)))
        return st

    def render_static(self, cam: Camera, params: Params, accum_ptr: int) -> Stats:
        """rt_render_static: as render_device, but in the library's fixed device array (no allocation on the call)."""
        st = Stats()
        _check(hip_lib().rt_render_static(self._h, C.byref(cam), C.byref(params), C.c_void_p(accum_ptr), C.byref(st)))
        return st

    def primary_hits(self, cam: Camera, params: Params) -> Tuple[np.ndarray, np.ndarray]:
        prim = np.empty((params.height, params.width), np.int32)
        t = np.empty((params.height, params.width), np.float32)
        _check(hip_lib().rt_primary_hits(self._h, C.byref(cam), C.byref(params), _ip(prim), _fp(t)))
        return prim, t

    def close(self):
        if self._h:
            hip_lib().rt_scene_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def resolve(accum: np.ndarray, spp: int) -> Tuple[np.ndarray, np.ndarray]:
    """src/bin/main.rs:45-50 on a host sum buffer: (raw i32 image, clamped uint8 image)."""
    a = np.ascontiguousarray(accum, np.float32)
    n = a.size // 3
    raw = np.empty(a.shape, np.int32)
    u8 = np.empty(a.shape, np.uint8)
    _check(hip_lib().rt_resolve(_fp(a), n, spp, _ip(raw), u8.ctypes.data_as(C.POINTER(C.c_uint8))))
    return raw, u8


from .sharding import ShardPlan, plan_shards, reduce_accum  # noqa: E402,F401
