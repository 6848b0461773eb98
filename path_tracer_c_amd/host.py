"""Host-side helpers over the C-ABI: scene construction, frame rendering,
device residency.  Everything here calls into libptgpu.so; numpy / torch only
own buffers.

The reference's host surface is C (scene_init / scene_add_* / render_task /
threadpool_* / write_bitmap, include/scene.h:17-30, include/renderer.h:13,
include/threadpool.h:15-20, include/bitmap.h:4-5); `HostScene` is a thin
lifetime wrapper around exactly those calls, not a re-implementation.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _abi
from ._abi import (Params, RayHit, RenderArgs, Scene, Stats, ThreadPool, Vec3,
                   check, default_params, lib)

# BASELINE.json's configurations: (scene id, width, height, spp, bounces, seed)
CONFIGS = {
    1: dict(scene=1, width=256, height=256, spp=16, max_bounces=4, seed=12345),
    2: dict(scene=2, width=1920, height=1080, spp=256, max_bounces=8, seed=1),
    3: dict(scene=2, width=1920, height=1080, spp=4096, max_bounces=8, seed=1),
    4: dict(scene=2, width=3840, height=2160, spp=1024, max_bounces=8, seed=1),
    5: dict(scene=5, width=1920, height=1080, spp=256, max_bounces=8, seed=1),
}


class HostScene:
    """Owns a `scene` struct built with the legacy scene_* calls."""

    def __init__(self, config: int | None = None):
        self.c = Scene()
        self._alive = False
        if config is None:
            lib().scene_init(C.byref(self.c))
            self.c.sky_color = Vec3(0, 0, 0)
        else:
            check(lib().ptgpu_build_config_scene(int(config), C.byref(self.c)))
        self._alive = True

    # -- builders (include/scene.h:17-30) ---------------------------------
    def set_sky(self, r, g, b):
        self.c.sky_color = Vec3(r, g, b)
        return self

    def add_material(self, albedo, roughness=1.0, metallicity=0.0,
                     emission_color=(0, 0, 0), emission_strength=0.0,
                     transparency=0.0, refractive_index=1.0) -> int:
        return int(lib().scene_add_material(
            C.byref(self.c), Vec3(*albedo), roughness, metallicity,
            Vec3(*emission_color), emission_strength, transparency,
            refractive_index))

    def add_sphere(self, center, radius, material):
        lib().scene_add_sphere(C.byref(self.c), Vec3(*center), radius, material)
        return self

    def add_triangle(self, v0, v1, v2, material):
        lib().scene_add_triangle(C.byref(self.c), Vec3(*v0), Vec3(*v1),
                                 Vec3(*v2), material)
        return self

    @property
    def num_objects(self):
        return int(self.c.num_objects)

    @property
    def num_materials(self):
        return int(self.c.num_materials)

    def counts(self):
        """(spheres, triangles)"""
        tri = sum(1 for i in range(self.num_objects) if self.c.objects[i].tag == 1)
        return self.num_objects - tri, tri

    def ptr(self):
        return C.byref(self.c)

    def close(self):
        if self._alive:
            lib().scene_destroy(C.byref(self.c))
            self._alive = False

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def make_params(**kw) -> Params:
    """ptgpu_params with the reference's defaults, overridden by keywords."""
    return default_params(**kw)


def render_frame(scene: HostScene, width: int, height: int, **kw) -> np.ndarray:
    """ptgpu_render_frame into a fresh (height, width, 3) uint8 array.
    Rows outside the call's row set are left zero."""
    p = make_params(**kw)
    frame = np.zeros((height, width, 3), dtype=np.uint8)
    check(lib().ptgpu_render_frame(scene.ptr(), frame.ctypes.data_as(C.c_void_p),
                                   width, height, C.byref(p)))
    return frame


def render_frame_legacy(scene: HostScene, width: int, height: int,
                        rows=None) -> np.ndarray:
    """The reference's driver loop: threadpool_add_task(render_task) per row,
    threadpool_wait_for_tasks.  Settings come from PTGPU_* variables."""
    L = lib()
    frame = np.zeros((height, width, 3), dtype=np.uint8)
    rows = list(range(height)) if rows is None else list(rows)
    args = (RenderArgs * len(rows))()
    fptr = frame.ctypes.data_as(C.POINTER(C.c_uint8))
    pool = ThreadPool()
    L.threadpool_init(C.byref(pool), 4)
    task = C.cast(L.render_task, C.c_void_p)
    for i, y in enumerate(rows):
        args[i].screen_y = y
        args[i].framew = width
        args[i].frameh = height
        args[i].frame = fptr
        args[i].world = C.pointer(scene.c)
        L.threadpool_add_task(C.byref(pool), task, C.cast(C.byref(args[i]), C.c_void_p))
    L.threadpool_wait_for_tasks(C.byref(pool))
    L.threadpool_destroy(C.byref(pool))
    return frame


def trace_rays(scene: HostScene, origins, directions, scan_flags: int = 0):
    """Batched trace_ray: returns (distance, point, normal, material) arrays.
    scan_flags: 0 plain scan, FLAG_SCAN_SCREENED, FLAG_SCAN_BVH."""
    o = np.ascontiguousarray(origins, dtype=np.float32).reshape(-1, 3)
    d = np.ascontiguousarray(directions, dtype=np.float32).reshape(-1, 3)
    n = o.shape[0]
    hits = (RayHit * n)()
    check(lib().ptgpu_trace_rays_scan(scene.ptr(), o.ctypes.data_as(C.c_void_p),
                                      d.ctypes.data_as(C.c_void_p), n,
                                      C.cast(hits, C.c_void_p), scan_flags))
    return hits_to_arrays(hits, n)


def hits_to_arrays(hits, n):
    raw = np.frombuffer(hits, dtype=np.uint8).reshape(n, 40)
    dist = raw[:, 0:4].copy().view(np.float32).reshape(n)
    point = raw[:, 4:16].copy().view(np.float32).reshape(n, 3)
    normal = raw[:, 16:28].copy().view(np.float32).reshape(n, 3)
    material = raw[:, 32:40].copy().view(np.uint64).reshape(n)
    return dist, point, normal, material


def read_stats() -> Stats:
    s = Stats()
    check(lib().ptgpu_stats_read(C.byref(s)))
    return s


def reset_stats():
    check(lib().ptgpu_stats_reset())


class DeviceScene:
    """A scene resident in HBM (ptgpu_scene_upload / ptgpu_scene_free)."""

    def __init__(self, scene: HostScene):
        self.handle = C.c_void_p()
        check(lib().ptgpu_scene_upload(scene.ptr(), C.byref(self.handle)))

    @property
    def branching(self) -> bool:
        return bool(check(lib().ptgpu_scene_is_branching(self.handle)))

    @property
    def schedule(self) -> str:
        """Schedule the library remembered for this scene."""
        return ("unmeasured", "pixel-owner", "sample-queue")[
            check(lib().ptgpu_scene_schedule(self.handle))]

    @property
    def scan(self) -> str:
        """Scan form the library remembered for this scene."""
        return ("by object count", "plain", "screened linear", "bvh")[
            check(lib().ptgpu_scene_scan(self.handle))]

    @property
    def hierarchy(self) -> tuple[int, int]:
        """(inner nodes, depth) of the BVH built at upload; (0, 0) without one."""
        nodes, depth = C.c_uint32(0), C.c_uint32(0)
        check(lib().ptgpu_scene_hierarchy(self.handle, C.byref(nodes), C.byref(depth)))
        return nodes.value, depth.value

    def tune(self, width: int, height: int, params: Params | None = None):
        """ptgpu_scene_tune: measure scan form / schedule once (synchronous)."""
        check(lib().ptgpu_scene_tune(self.handle, width, height,
                                     C.byref(params) if params is not None else None))
        return self

    def trace(self, d_origins_ptr: int, d_directions_ptr: int, n: int, d_hits_ptr: int,
              scan_flags: int = 0, stream: int = 0):
        """ptgpu_trace_rays_device: n x 3 float origins / directions and n
        40-byte ray_hit records, all in HBM; asynchronous on `stream`."""
        check(lib().ptgpu_trace_rays_device(self.handle, C.c_void_p(d_origins_ptr),
                                            C.c_void_p(d_directions_ptr), n,
                                            C.c_void_p(d_hits_ptr), scan_flags,
                                            C.c_void_p(stream)))

    def render(self, d_frame_ptr: int, width: int, height: int, params: Params,
               stream: int = 0):
        check(lib().ptgpu_render_device(self.handle, C.c_void_p(d_frame_ptr),
                                        width, height, C.byref(params),
                                        C.c_void_p(stream)))

    def accumulate(self, d_accum_ptr: int, width: int, height: int,
                   params: Params, stream: int = 0):
        check(lib().ptgpu_accumulate_device(self.handle, C.c_void_p(d_accum_ptr),
                                            width, height, C.byref(params),
                                            C.c_void_p(stream)))

    def close(self):
        if self.handle:
            lib().ptgpu_scene_free(self.handle)
            self.handle = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class MultiDevice:
    """ptgpu_multi_*: one process, several GPUs, the frame assembled in HBM on
    devices[0] by one grouped ncclSend/ncclRecv (or, with peer_copy=True, by
    peer copies: a rehearsal that also runs with repeated ordinals)."""

    def __init__(self, devices, peer_copy=False, rccl_loopback=False):
        from ._abi import MULTI_PEER_COPY, MULTI_RCCL_LOOPBACK
        self.handle = C.c_void_p()
        arr = (C.c_int * len(devices))(*devices)
        flags = (MULTI_PEER_COPY if peer_copy else 0) | (MULTI_RCCL_LOOPBACK if rccl_loopback else 0)
        check(lib().ptgpu_multi_create(arr, len(devices), flags, C.byref(self.handle)))

    @property
    def transport(self) -> str:
        return lib().ptgpu_multi_transport(self.handle).decode()

    def upload(self, scene: HostScene):
        check(lib().ptgpu_multi_upload(self.handle, scene.ptr()))
        return self

    def render(self, d_frame_ptr: int, width: int, height: int, params: Params, tile_rows=0):
        check(lib().ptgpu_multi_render(self.handle, C.c_void_p(d_frame_ptr), width, height,
                                       C.byref(params), tile_rows))

    def close(self):
        if self.handle:
            lib().ptgpu_multi_destroy(self.handle)
            self.handle = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def resolve(d_accum_ptr: int, d_frame_ptr: int, npixels: int, stream: int = 0):
    check(lib().ptgpu_resolve_device(C.c_void_p(d_accum_ptr),
                                     C.c_void_p(d_frame_ptr), npixels,
                                     C.c_void_p(stream)))


def write_bitmap(path, frame: np.ndarray, y_inverted=True):
    """include/bitmap.h:4-5.  `frame` is (h, w, 3) uint8, row 0 = top."""
    frame = np.ascontiguousarray(frame, dtype=np.uint8)
    h, w, _ = frame.shape
    lib().write_bitmap(os.fsencode(path), w, h,
                       frame.ctypes.data_as(C.POINTER(C.c_uint8)), y_inverted)
