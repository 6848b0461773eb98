"""ctypes view of the C-ABI in include/ptgpu.h and include/path_tracer_api.h.

This module is plumbing for the Python test / bench harness: it loads
``path_tracer_c_amd/lib/libptgpu.so`` (built by ``make`` /
``__graft_entry__.build()``) and declares the argument types of every exported
entry point.  It never substitutes another implementation: if the library is
missing, importing :func:`lib` raises, and if HIP is unusable the calls return
``PTGPU_ERR_NO_DEVICE`` which :func:`check` turns into an exception.

Struct layouts mirror the reference headers (include/shapes.h:6-36,
include/scene.h:7-15, include/ray.h:7-12, include/renderer.h:6-11,
include/threadpool.h:6-13 of the reference) via include/path_tracer_api.h.
"""
from __future__ import annotations

import ctypes as C
import importlib.util
import os
import sys
from pathlib import Path

_PKG = Path(__file__).resolve().parent
LIB_PATH = _PKG / "lib" / "libptgpu.so"

PTGPU_OK = 0
PTGPU_ERR_INVALID = -1
PTGPU_ERR_NO_DEVICE = -2
PTGPU_ERR_HIP = -3
PTGPU_ERR_NOMEM = -4
PTGPU_ERR_LIMIT = -5

FLAG_COMPACT_ROWS = 0x1
FLAG_SCENE_GLOBAL = 0x2
FLAG_FORCE_TREE = 0x4
FLAG_NO_RAY_COUNT = 0x8
FLAG_SCAN_PLAIN = 0x10
FLAG_SCENE_LDS = 0x40
FLAG_PIXEL_OWNER = 0x80
FLAG_SAMPLE_QUEUE = 0x100
FLAG_NO_PRIMARY_SHARE = 0x200
FLAG_SCAN_BVH = 0x400
FLAG_SCAN_SCREENED = 0x20
FLAG_WHOLE_WALKS = 0x800
FLAG_FAST_MATH = 0x1000
FLAG_GLSL_SEMANTICS = 0x2000
MULTI_PEER_COPY = 0x1
MULTI_RCCL_LOOPBACK = 0x2

MAX_BOUNCES_PATH = 62
MAX_BOUNCES_TREE = 14


class Vec3(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float)]

    def __init__(self, x=0.0, y=0.0, z=0.0):
        super().__init__(float(x), float(y), float(z))

    def tuple(self):
        return (self.x, self.y, self.z)


class Sphere(C.Structure):
    _fields_ = [("center", Vec3), ("radius", C.c_float)]


class Triangle(C.Structure):
    _fields_ = [("v0", Vec3), ("v1", Vec3), ("v2", Vec3)]


class ShapeMaterial(C.Structure):
    _fields_ = [
        ("albedo", Vec3),
        ("roughness", C.c_float),
        ("metallicity", C.c_float),
        ("emission_color", Vec3),
        ("emission_strength", C.c_float),
        ("transparency", C.c_float),
        ("refractive_index", C.c_float),
    ]


class _ShapeUnion(C.Union):
    _fields_ = [("sphere", Sphere), ("triangle", Triangle)]


class Shape(C.Structure):
    _anonymous_ = ("u",)
    _fields_ = [("tag", C.c_int), ("material", C.c_size_t), ("u", _ShapeUnion)]


class Scene(C.Structure):
    _fields_ = [
        ("objects", C.POINTER(Shape)),
        ("num_objects", C.c_size_t),
        ("max_objects", C.c_size_t),
        ("materials", C.POINTER(ShapeMaterial)),
        ("num_materials", C.c_size_t),
        ("max_materials", C.c_size_t),
        ("sky_color", Vec3),
    ]


class RayHit(C.Structure):
    _fields_ = [
        ("distance", C.c_float),
        ("point", Vec3),
        ("normal", Vec3),
        ("material", C.c_size_t),
    ]


class RenderArgs(C.Structure):
    _fields_ = [
        ("screen_y", C.c_int),
        ("framew", C.c_size_t),
        ("frameh", C.c_size_t),
        ("frame", C.POINTER(C.c_uint8)),
        ("world", C.POINTER(Scene)),
    ]


class Vector(C.Structure):
    _fields_ = [("data", C.POINTER(C.c_void_p)), ("size", C.c_size_t),
                ("capacity", C.c_size_t)]


class ThreadPool(C.Structure):
    # 192 bytes on x86-64 glibc: 2 vectors, mutex (40), 2 condvars (48), bool
    _fields_ = [
        ("threads", Vector),
        ("tasks", Vector),
        ("tasks_mutex", C.c_ubyte * 40),
        ("tasks_available_cond", C.c_ubyte * 48),
        ("tasks_exhausted_cond", C.c_ubyte * 48),
        ("threads_running", C.c_bool),
    ]


class Params(C.Structure):
    _fields_ = [
        ("spp", C.c_uint32),
        ("max_bounces", C.c_uint32),
        ("fov_rad", C.c_float),
        ("seed", C.c_uint32),
        ("sample_offset", C.c_uint32),
        ("row_begin", C.c_uint32),
        ("num_rows", C.c_uint32),
        ("tile_rows", C.c_uint32),
        ("tile_stride", C.c_uint32),
        ("flags", C.c_uint32),
        ("block_threads", C.c_uint32),
        ("settle_lanes", C.c_uint32),
        ("lanes_per_pixel", C.c_uint32),
    ]


class Stats(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("samples", C.c_uint64),
                ("wave_iters", C.c_uint64), ("launches", C.c_uint64),
                ("scans", C.c_uint64), ("kernel_ms", C.c_double),
                ("hbm_bytes", C.c_uint64)]


assert C.sizeof(Vec3) == 12 and C.sizeof(Sphere) == 16
assert C.sizeof(Triangle) == 36 and C.sizeof(ShapeMaterial) == 44
assert C.sizeof(Shape) == 56 and C.sizeof(Scene) == 64
assert C.sizeof(RayHit) == 40 and C.sizeof(RenderArgs) == 40
assert C.sizeof(ThreadPool) == 192

TASK_FN = C.CFUNCTYPE(None, C.c_void_p)

# name -> (restype, argtypes): every symbol include/*.h declares.
SIGNATURES = {
    # include/path_tracer_api.h (legacy surface)
    "scene_init": (None, [C.POINTER(Scene)]),
    "scene_add_material": (C.c_size_t, [C.POINTER(Scene), Vec3, C.c_float,
                                        C.c_float, Vec3, C.c_float, C.c_float,
                                        C.c_float]),
    "scene_add_sphere": (None, [C.POINTER(Scene), Vec3, C.c_float, C.c_size_t]),
    "scene_add_triangle": (None, [C.POINTER(Scene), Vec3, Vec3, Vec3, C.c_size_t]),
    "scene_destroy": (None, [C.POINTER(Scene)]),
    "trace_ray": (RayHit, [Vec3, Vec3, C.POINTER(Scene)]),
    "render_task": (None, [C.POINTER(RenderArgs)]),
    "vector_init": (None, [C.POINTER(Vector)]),
    "vector_init_with_capacity": (None, [C.POINTER(Vector), C.c_size_t]),
    "vector_free": (None, [C.POINTER(Vector)]),
    "vector_push": (None, [C.POINTER(Vector), C.c_void_p]),
    "vector_pop": (C.c_void_p, [C.POINTER(Vector)]),
    "threadpool_init": (None, [C.POINTER(ThreadPool), C.c_size_t]),
    "threadpool_destroy": (None, [C.POINTER(ThreadPool)]),
    "threadpool_add_task": (None, [C.POINTER(ThreadPool), C.c_void_p, C.c_void_p]),
    "threadpool_wait_for_tasks": (None, [C.POINTER(ThreadPool)]),
    "write_bitmap": (None, [C.c_char_p, C.c_uint, C.c_uint,
                            C.POINTER(C.c_uint8), C.c_bool]),
    # include/ptgpu.h (explicit MI355X API)
    "ptgpu_params_default": (C.c_int, [C.POINTER(Params)]),
    "ptgpu_params_from_env": (C.c_int, [C.POINTER(Params)]),
    "ptgpu_device_count": (C.c_int, []),
    "ptgpu_set_device": (C.c_int, [C.c_int]),
    "ptgpu_scene_upload": (C.c_int, [C.POINTER(Scene), C.POINTER(C.c_void_p)]),
    "ptgpu_scene_free": (None, [C.c_void_p]),
    "ptgpu_scene_is_branching": (C.c_int, [C.c_void_p]),
    "ptgpu_scene_schedule": (C.c_int, [C.c_void_p]),
    "ptgpu_scene_scan": (C.c_int, [C.c_void_p]),
    "ptgpu_scene_hierarchy": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "ptgpu_release_scratch": (C.c_int, []),
    "ptgpu_scene_tune": (C.c_int, [C.c_void_p, C.c_size_t, C.c_size_t, C.POINTER(Params)]),
    "ptgpu_render_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t,
                                      C.c_size_t, C.POINTER(Params), C.c_void_p]),
    "ptgpu_render_frame": (C.c_int, [C.POINTER(Scene), C.c_void_p, C.c_size_t,
                                     C.c_size_t, C.POINTER(Params)]),
    "ptgpu_render_frame_multi": (C.c_int, [C.POINTER(Scene), C.c_void_p, C.c_size_t,
                                           C.c_size_t, C.POINTER(Params),
                                           C.POINTER(C.c_int), C.c_int, C.c_uint32]),
    "ptgpu_multi_create": (C.c_int, [C.POINTER(C.c_int), C.c_int, C.c_uint32,
                                     C.POINTER(C.c_void_p)]),
    "ptgpu_multi_upload": (C.c_int, [C.c_void_p, C.POINTER(Scene)]),
    "ptgpu_multi_render": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t,
                                     C.POINTER(Params), C.c_uint32]),
    "ptgpu_multi_transport": (C.c_char_p, [C.c_void_p]),
    "ptgpu_multi_destroy": (None, [C.c_void_p]),
    "ptgpu_render_device_multi": (C.c_int, [C.POINTER(Scene), C.c_void_p, C.c_size_t,
                                            C.c_size_t, C.POINTER(Params),
                                            C.POINTER(C.c_int), C.c_int, C.c_uint32]),
    "ptgpu_device_alloc": (C.c_int, [C.c_size_t, C.POINTER(C.c_void_p)]),
    "ptgpu_device_free": (C.c_int, [C.c_void_p]),
    "ptgpu_device_upload": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "ptgpu_device_download": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "ptgpu_device_synchronize": (C.c_int, []),
    "ptgpu_accumulate_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t,
                                          C.c_size_t, C.POINTER(Params),
                                          C.c_void_p]),
    "ptgpu_resolve_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t,
                                       C.c_void_p]),
    "ptgpu_trace_rays": (C.c_int, [C.POINTER(Scene), C.c_void_p, C.c_void_p,
                                   C.c_size_t, C.c_void_p]),
    "ptgpu_trace_rays_scan": (C.c_int, [C.POINTER(Scene), C.c_void_p, C.c_void_p,
                                        C.c_size_t, C.c_void_p, C.c_uint32]),
    "ptgpu_trace_rays_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                          C.c_void_p, C.c_uint32, C.c_void_p]),
    "ptgpu_stats_reset": (C.c_int, []),
    "ptgpu_stats_read": (C.c_int, [C.POINTER(Stats)]),
    "ptgpu_last_error": (C.c_char_p, []),
    "ptgpu_version": (C.c_char_p, []),
    "ptgpu_legacy_frames": (C.c_uint32, []),
    "ptgpu_build_config_scene": (C.c_int, [C.c_int, C.POINTER(Scene)]),
}

_lib = None


def _share_hip_runtime_with_torch():
    """libptgpu.so needs libamdhip64.so.7.  PyTorch-ROCm wheels bundle their
    own copy under torch/lib with the same SONAME; a process that loaded the
    system copy first leaves torch with a mismatched runtime
    (torch.cuda.is_available() turns False).  So, when torch is installed,
    load ITS copy first -- without importing torch -- and let the dynamic
    linker resolve libptgpu.so's dependency to it.  Either import order then
    works and both sides share one HIP runtime, streams and allocations."""
    if "torch" in sys.modules:
        return  # torch's runtime is already the one in the process
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    bundled = Path(spec.origin).parent / "lib" / "libamdhip64.so"
    if bundled.exists():
        try:
            C.CDLL(os.fspath(bundled), mode=C.RTLD_GLOBAL)
        except OSError:
            pass  # fall back to the system runtime via libptgpu.so's NEEDED


class PtgpuError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"ptgpu error {code}: {message}")
        self.code = code


def lib() -> C.CDLL:
    """The product library.  Raises if it has not been built."""
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise OSError(
                f"{LIB_PATH} is missing: run `make` (or __graft_entry__.build()); "
                "there is no Python or CPU fallback for the render path")
        _share_hip_runtime_with_torch()
        handle = C.CDLL(os.fspath(LIB_PATH))
        for name, (restype, argtypes) in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError if a symbol is missing
            fn.restype = restype
            fn.argtypes = argtypes
        _lib = handle
    return _lib


def check(code: int) -> int:
    if code < 0:
        raise PtgpuError(code, lib().ptgpu_last_error().decode("utf-8", "replace"))
    return code


def default_params(**overrides) -> Params:
    p = Params()
    check(lib().ptgpu_params_default(C.byref(p)))
    for k, v in overrides.items():
        setattr(p, k, v)
    return p
