"""ctypes loader for oracle/liboracle.so -- TEST INFRASTRUCTURE.

Importable only from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  PARITY UNPINNED: see oracle/pt_oracle.c.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
LIB_PATH = _HERE / "liboracle.so"

RNG_XORSHIFT = 0
RNG_LIBC = 1


class OracleParams(C.Structure):
    _fields_ = [
        ("spp", C.c_uint32),
        ("max_bounces", C.c_uint32),
        ("fov_rad", C.c_float),
        ("seed", C.c_uint32),
        ("sample_offset", C.c_uint32),
        ("rng_mode", C.c_uint32),
    ]


_lib = None


def build():
    subprocess.run(["make", "-C", os.fspath(_HERE)], check=True,
                   stdout=subprocess.DEVNULL)


def lib():
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            build()
        h = C.CDLL(os.fspath(LIB_PATH))
        from path_tracer_c_amd._abi import RayHit, Scene, Vec3
        h.oracle_trace_ray.restype = RayHit
        h.oracle_trace_ray.argtypes = [Vec3, Vec3, C.POINTER(Scene)]
        h.oracle_trace_rays.restype = None
        h.oracle_trace_rays.argtypes = [C.POINTER(Scene), C.c_void_p, C.c_void_p,
                                        C.c_size_t, C.c_void_p]
        h.oracle_render_rows.restype = C.c_uint64
        h.oracle_render_rows.argtypes = [C.POINTER(Scene), C.c_void_p, C.c_size_t,
                                         C.c_size_t, C.c_size_t, C.c_size_t,
                                         C.POINTER(OracleParams)]
        h.oracle_accumulate_rows.restype = C.c_uint64
        h.oracle_accumulate_rows.argtypes = h.oracle_render_rows.argtypes
        h.oracle_sample.restype = None
        h.oracle_sample.argtypes = [C.POINTER(Scene), C.c_size_t, C.c_size_t,
                                    C.c_int, C.c_int, C.c_uint32,
                                    C.POINTER(OracleParams), C.c_void_p]
        h.oracle_sample_rays.restype = C.c_size_t
        h.oracle_sample_rays.argtypes = [C.POINTER(Scene), C.c_size_t, C.c_size_t,
                                         C.c_int, C.c_int, C.c_uint32,
                                         C.POINTER(OracleParams), C.c_void_p, C.c_size_t]
        h.oracle_srandom.restype = None
        h.oracle_srandom.argtypes = [C.c_uint]
        h.oracle_render_mt.restype = C.c_uint64
        h.oracle_render_mt.argtypes = [C.POINTER(Scene), C.c_void_p, C.c_size_t,
                                       C.c_size_t, C.c_size_t, C.c_size_t,
                                       C.c_size_t, C.POINTER(OracleParams),
                                       C.c_uint]
        h.oracle_glsl_render_rows.restype = C.c_uint64
        h.oracle_glsl_render_rows.argtypes = [C.POINTER(Scene), C.c_void_p, C.c_void_p,
                                              C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t,
                                              C.c_uint32, C.c_uint32, C.c_float, C.c_uint32]
        _lib = h
    return _lib


def params(spp=16, max_bounces=4, fov_rad=None, seed=1, sample_offset=0,
           rng_mode=RNG_XORSHIFT) -> OracleParams:
    if fov_rad is None:
        # FOV of src/renderer.c:7 (M_PI / 180.f * 90.0f) narrowed to float,
        # the same value ptgpu_params_default() uses
        fov_rad = float(np.float32(np.pi / 180.0 * 90.0))
    return OracleParams(spp, max_bounces, fov_rad, seed, sample_offset, rng_mode)


def render_rows(scene, width, height, row_begin=0, row_end=None, frame=None, **kw):
    """Rows [row_begin, row_end) into a (h, w, 3) uint8 frame; returns
    (frame, rays)."""
    if row_end is None:
        row_end = height
    if frame is None:
        frame = np.zeros((height, width, 3), dtype=np.uint8)
    p = params(**kw)
    rays = lib().oracle_render_rows(scene.ptr(), frame.ctypes.data_as(C.c_void_p),
                                    width, height, row_begin, row_end, C.byref(p))
    return frame, int(rays)


def accumulate_rows(scene, accum, width, height, row_begin=0, row_end=None, **kw):
    if row_end is None:
        row_end = height
    assert accum.dtype == np.float32 and accum.shape == (height, width, 4)
    p = params(**kw)
    return int(lib().oracle_accumulate_rows(
        scene.ptr(), accum.ctypes.data_as(C.c_void_p), width, height,
        row_begin, row_end, C.byref(p)))


def render_mt(scene, width, height, threads, row_begin=0, row_end=None,
              row_step=1, frame=None, **kw):
    if row_end is None:
        row_end = height
    if frame is None:
        frame = np.zeros((height, width, 3), dtype=np.uint8)
    p = params(**kw)
    rays = lib().oracle_render_mt(scene.ptr(), frame.ctypes.data_as(C.c_void_p),
                                  width, height, row_begin, row_end, row_step,
                                  C.byref(p), threads)
    return frame, int(rays)


def glsl_render_rows(scene, width, height, row_begin=0, row_end=None, spp=16, max_bounces=4,
                     fov_rad=None, seed=1, want_mean=False):
    """The shader restatement (oracle/pt_glsl.c): (frame, mean or None, rays)."""
    if row_end is None:
        row_end = height
    if fov_rad is None:
        fov_rad = float(np.float32(np.pi / 180.0 * 90.0))
    frame = np.zeros((height, width, 3), dtype=np.uint8)
    mean = np.zeros((height, width, 3), dtype=np.float32) if want_mean else None
    rays = lib().oracle_glsl_render_rows(
        scene.ptr(), frame.ctypes.data_as(C.c_void_p),
        mean.ctypes.data_as(C.c_void_p) if want_mean else None, width, height, row_begin, row_end,
        spp, max_bounces, fov_rad, seed)
    return frame, mean, int(rays)


def trace_rays(scene, origins, directions):
    from path_tracer_c_amd._abi import RayHit
    from path_tracer_c_amd.host import hits_to_arrays
    o = np.ascontiguousarray(origins, dtype=np.float32).reshape(-1, 3)
    d = np.ascontiguousarray(directions, dtype=np.float32).reshape(-1, 3)
    n = o.shape[0]
    hits = (RayHit * n)()
    lib().oracle_trace_rays(scene.ptr(), o.ctypes.data_as(C.c_void_p),
                            d.ctypes.data_as(C.c_void_p), n, C.cast(hits, C.c_void_p))
    return hits_to_arrays(hits, n)


def sample(scene, width, height, sx, sy, index, **kw):
    p = params(**kw)
    out = (C.c_float * 3)()
    lib().oracle_sample(scene.ptr(), width, height, sx, sy, index, C.byref(p),
                        C.cast(out, C.c_void_p))
    return np.array(out[:], dtype=np.float32)


def sample_rays(scene, width, height, sx, sy, index, max_rays=64, **kw):
    """(origins, directions) of every trace_ray call of one sample."""
    p = params(**kw)
    buf = np.zeros((max_rays, 6), dtype=np.float32)
    n = lib().oracle_sample_rays(scene.ptr(), width, height, sx, sy, index, C.byref(p),
                                 buf.ctypes.data_as(C.c_void_p), max_rays)
    return buf[:n, :3].copy(), buf[:n, 3:].copy()
