"""ctypes wrapper of the CPU oracle (oracle/build/liboracle_rt.so).

TEST INFRASTRUCTURE: imported only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
The product package never imports this module.
"""
from __future__ import annotations

import ctypes as C
import importlib
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "build", "liboracle_rt.so")

rt = importlib.import_module("raytracer-rs_amd")  # struct definitions only (rt_api.h PODs)

_lib = None
c_float_p = C.POINTER(C.c_float)
c_int32_p = C.POINTER(C.c_int32)


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: run `make -C oracle`")
        L = C.CDLL(LIB_PATH)
        L.orc_last_error.restype = C.c_char_p
        L.orc_scene_create.argtypes = [C.POINTER(rt.SceneDesc), C.c_int32, C.POINTER(C.c_void_p)]
        L.orc_scene_create.restype = C.c_int32
        L.orc_set_estimator_terms.argtypes = [C.c_int32]
        L.orc_set_estimator_terms.restype = None
        L.orc_scene_destroy.argtypes = [C.c_void_p]
        L.orc_scene_destroy.restype = None
        L.orc_render.argtypes = [C.c_void_p, C.POINTER(rt.Camera), C.POINTER(rt.Params), c_float_p, C.POINTER(rt.Stats), C.c_int32]
        L.orc_render.restype = C.c_int32
        L.orc_primary_hits.argtypes = [C.c_void_p, C.POINTER(rt.Camera), C.POINTER(rt.Params), c_int32_p, c_float_p]
        L.orc_primary_hits.restype = C.c_int32
        L.orc_resolve.argtypes = [c_float_p, C.c_int64, C.c_int32, c_int32_p, C.POINTER(C.c_uint8)]
        L.orc_resolve.restype = C.c_int32
        L.orc_camera_new.argtypes = [c_float_p, c_float_p, c_float_p, C.c_float, C.c_float, C.POINTER(rt.Camera)]
        L.orc_camera_new.restype = None
        L.orc_get_ray.argtypes = [C.POINTER(rt.Camera), C.c_float, C.c_float, c_float_p, c_float_p]
        L.orc_get_ray.restype = None
        L.orc_world_hit.argtypes = [C.c_void_p, c_float_p, c_float_p, C.c_float, C.c_float, c_float_p, c_int32_p, c_float_p]
        L.orc_world_hit.restype = C.c_int32
        L.orc_tri_hit.argtypes = [c_float_p, c_float_p, c_float_p, C.c_float, C.c_float, C.c_int32, c_float_p]
        L.orc_tri_hit.restype = C.c_int32
        L.orc_tri_area.argtypes = [c_float_p]
        L.orc_tri_area.restype = C.c_float
        L.orc_tri_normal.argtypes = [c_float_p, c_float_p]
        L.orc_tri_normal.restype = None
        L.orc_ref_hit_box.argtypes = [c_float_p, c_float_p, c_float_p, c_float_p, C.c_float, C.c_float]
        L.orc_ref_hit_box.restype = C.c_int32
        L.orc_lights_hit.argtypes = [C.c_void_p, c_float_p, c_float_p, C.c_float, C.c_float, c_float_p]
        L.orc_lights_hit.restype = C.c_int32
        L.orc_lights_pdf.argtypes = [C.c_void_p, c_float_p, c_float_p, C.c_float, C.c_float]
        L.orc_lights_pdf.restype = C.c_float
        L.orc_light_info.argtypes = [C.c_void_p, C.c_int32, c_float_p, c_float_p]
        L.orc_light_info.restype = None
        L.orc_mat_brdf.argtypes = [C.POINTER(rt.Material), c_float_p, c_float_p, c_float_p, C.c_float, c_float_p]
        L.orc_mat_brdf.restype = None
        L.orc_mat_pdf.argtypes = [C.POINTER(rt.Material), c_float_p, c_float_p, c_float_p, C.c_float]
        L.orc_mat_pdf.restype = C.c_float
        L.orc_mat_scatter_d.argtypes = [C.POINTER(rt.Material), c_float_p, c_float_p, c_float_p, c_float_p, C.c_float,
                                        C.c_uint32, C.c_uint32, C.c_uint32, c_float_p]
        L.orc_mat_scatter_d.restype = C.c_int32
        L.orc_rng_u32.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32), C.c_uint32, C.c_uint32]
        L.orc_rng_u32.restype = C.c_uint32
        L.orc_rng_f32.argtypes = [C.c_uint32]
        L.orc_rng_f32.restype = C.c_float
        L.orc_sincos_2pi.argtypes = [C.c_float, c_float_p, c_float_p]
        L.orc_sincos_2pi.restype = None
        L.orc_sampler_points.argtypes = [C.c_int32, C.c_float, C.c_int32, C.c_uint32, C.c_uint32, c_float_p]
        L.orc_sampler_points.restype = C.c_int32
        L.orc_disk_sample_d.argtypes = [C.c_int32, C.c_float, C.c_int32, C.c_uint32, C.c_uint32, C.c_uint32, c_float_p]
        L.orc_disk_sample_d.restype = C.c_int32
        L.orc_num_threads.restype = C.c_int32
        _lib = L
    return _lib


def _f3(v):
    a = np.asarray(v, np.float32).reshape(-1).copy()
    return a, a.ctypes.data_as(c_float_p)


def _check(rc):
    if rc != 0:
        raise rt.RtError(rc, lib().orc_last_error().decode())


class OracleScene:
    """accel = 0: every mesh brute force (the definition).  accel = 1: reference-shaped BVH per mesh (same results)."""

    def __init__(self, desc, accel: int = 0, keepalive=None):
        self._keep = keepalive
        self._h = C.c_void_p()
        _check(lib().orc_scene_create(C.byref(desc), accel, C.byref(self._h)))

    def render(self, cam, params, accum=None, threads: int = 0):
        if accum is None:
            accum = np.zeros((params.height, params.width, 3), np.float32)
        st = rt.Stats()
        _check(lib().orc_render(self._h, C.byref(cam), C.byref(params), accum.ctypes.data_as(c_float_p), C.byref(st), threads))
        return accum, st

    def primary_hits(self, cam, params):
        prim = np.empty((params.height, params.width), np.int32)
        t = np.empty((params.height, params.width), np.float32)
        _check(lib().orc_primary_hits(self._h, C.byref(cam), C.byref(params), prim.ctypes.data_as(c_int32_p), t.ctypes.data_as(c_float_p)))
        return prim, t

    def world_hit(self, o, d, t_min=0.001, t_max=1e7):
        (_, po), (_, pd) = _f3(o), _f3(d)
        oa, da = _f3(o), _f3(d)
        t = C.c_float()
        prim = C.c_int32()
        out6 = np.zeros(6, np.float32)
        hit = lib().orc_world_hit(self._h, oa[1], da[1], t_min, t_max, C.byref(t), C.byref(prim), out6.ctypes.data_as(c_float_p))
        if not hit:
            return None
        return {"t": t.value, "prim": prim.value, "p": out6[:3].copy(), "n": out6[3:].copy()}

    def lights_hit(self, o, d, t_min=0.001, t_max=1e7):
        oa, da = _f3(o), _f3(d)
        t = C.c_float()
        li = lib().orc_lights_hit(self._h, oa[1], da[1], t_min, t_max, C.byref(t))
        return (li, t.value) if li >= 0 else None

    def lights_pdf(self, o, d, t_min=0.001, t_max=1e7):
        oa, da = _f3(o), _f3(d)
        return lib().orc_lights_pdf(self._h, oa[1], da[1], t_min, t_max)

    def light_info(self, i):
        area = C.c_float()
        n = np.zeros(3, np.float32)
        lib().orc_light_info(self._h, i, C.byref(area), n.ctypes.data_as(c_float_p))
        return area.value, n

    def close(self):
        if self._h:
            lib().orc_scene_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def resolve(accum, spp):
    a = np.ascontiguousarray(accum, np.float32)
    raw = np.empty(a.shape, np.int32)
    u8 = np.empty(a.shape, np.uint8)
    _check(lib().orc_resolve(a.ctypes.data_as(c_float_p), a.size // 3, spp, raw.ctypes.data_as(c_int32_p), u8.ctypes.data_as(C.POINTER(C.c_uint8))))
    return raw, u8


def camera_new(lookfrom, lookat, vup, vfov, aspect):
    cam = rt.Camera()
    a, b, c = _f3(lookfrom), _f3(lookat), _f3(vup)
    lib().orc_camera_new(a[1], b[1], c[1], vfov, aspect, C.byref(cam))
    return cam


def get_ray(cam, s, t):
    o = np.zeros(3, np.float32)
    d = np.zeros(3, np.float32)
    lib().orc_get_ray(C.byref(cam), s, t, o.ctypes.data_as(c_float_p), d.ctypes.data_as(c_float_p))
    return o, d


def tri_hit(v9, o, d, t_min=0.001, t_max=1e7, both=False):
    va, oa, da = _f3(v9), _f3(o), _f3(d)
    t = C.c_float()
    return t.value if lib().orc_tri_hit(va[1], oa[1], da[1], t_min, t_max, int(both), C.byref(t)) else None


def tri_area(v9):
    return lib().orc_tri_area(_f3(v9)[1])


def tri_normal(v9):
    n = np.zeros(3, np.float32)
    va = _f3(v9)
    lib().orc_tri_normal(va[1], n.ctypes.data_as(c_float_p))
    return n


def ref_hit_box(mn, mx, o, d, t_min=0.001, t_max=1e7):
    a, b, c, e = _f3(mn), _f3(mx), _f3(o), _f3(d)
    return bool(lib().orc_ref_hit_box(a[1], b[1], c[1], e[1], t_min, t_max))


def mat_brdf(m, din, dout, n, t_min=0.001):
    a, b, c = _f3(din), _f3(dout), _f3(n)
    out = np.zeros(3, np.float32)
    lib().orc_mat_brdf(C.byref(m), a[1], b[1], c[1], t_min, out.ctypes.data_as(c_float_p))
    return out


def mat_pdf(m, din, dout, n, t_min=0.001):
    a, b, c = _f3(din), _f3(dout), _f3(n)
    return lib().orc_mat_pdf(C.byref(m), a[1], b[1], c[1], t_min)


def mat_scatter_d(m, o, d, p, n, seed=1, pixel=0, sample=0, t_min=0.001):
    a, b, c, e = _f3(o), _f3(d), _f3(p), _f3(n)
    out = np.zeros(3, np.float32)
    ok = lib().orc_mat_scatter_d(C.byref(m), a[1], b[1], c[1], e[1], t_min, seed, pixel, sample, out.ctypes.data_as(c_float_p))
    return out if ok else None


def rng_u32(seed, pixel, sample, branches=(), purpose=1, index=0):
    arr = (C.c_uint32 * max(1, len(branches)))(*branches)
    return lib().orc_rng_u32(seed, pixel, sample, len(branches), arr, purpose, index)


def rng_f32(u):
    return lib().orc_rng_f32(u)


def sincos_2pi(r):
    s, c = C.c_float(), C.c_float()
    lib().orc_sincos_2pi(r, C.byref(s), C.byref(c))
    return s.value, c.value


def sampler_points(kind, rng_range, rate, seed=1, pixel=0):
    out = np.zeros((max(rate, 1), 2), np.float32)
    n = lib().orc_sampler_points(kind, rng_range, rate, seed, pixel, out.ctypes.data_as(c_float_p))
    return out[:n].copy()


def disk_sample_d(kind, radius, ns2, seed=1, pixel=0, sample=0):
    ab = np.zeros(2, np.float32)
    ok = lib().orc_disk_sample_d(kind, radius, ns2, seed, pixel, sample, ab.ctypes.data_as(c_float_p))
    return ab if ok else None


def num_threads():
    return lib().orc_num_threads()
