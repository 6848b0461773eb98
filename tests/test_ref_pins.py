"""Pins held by the reference ITSELF (CPU only, no GPU).

`oracle/ref.mk` compiles the three reference sources that build from their own
files -- /root/reference/src/{bitmap,vector,threadpool}.c, unmodified, plain
gcc, no stand-in headers -- into oracle/_ref/libref_host.so, and a probe of the
reference's own headers into oracle/_ref/libref_probe.so.  These tests hold the
re-written host C of this repo (path_tracer_c_amd/csrc/host/{bitmap,vector}.c),
the legacy struct layouts of include/path_tracer_api.h and the task order of
the threadpool shim (csrc/ptgpu_abi.hip) to them, byte for byte.

What this pins: the I/O formats either side of the hot path (BMP writeback,
the pointer vector, the caller-allocated pool struct, LIFO task order).
What it cannot pin: the hot path's arithmetic (src/renderer.c, src/ray.c need
cglm, which is absent) -- that stays "parity unpinned" (DESIGN.md section 1).

The whole module is skipped when oracle/_ref is neither prebuilt nor buildable
(no /root/reference, e.g. a checkout elsewhere).
"""
import ctypes as C
import os
import subprocess
import threading
from pathlib import Path

import numpy as np
import pytest

from path_tracer_c_amd._abi import Vector  # a plain ctypes struct; importing loads no library

ROOT = Path(__file__).resolve().parent.parent
REF_DIR = ROOT / "oracle" / "_ref"
REFERENCE = Path("/root/reference")


def _ensure_ref():
    host, probe = REF_DIR / "libref_host.so", REF_DIR / "libref_probe.so"
    if not (host.exists() and probe.exists()) and (REFERENCE / "src" / "bitmap.c").exists():
        subprocess.run(["make", "-C", str(ROOT / "oracle"), "-f", "ref.mk"], check=True,
                       capture_output=True)
    if not (host.exists() and probe.exists()):
        pytest.skip("oracle/_ref is not built and /root/reference is absent",
                    allow_module_level=True)
    return host, probe


_HOST_SO, _PROBE_SO = _ensure_ref()


@pytest.fixture(scope="module")
def ref():
    """The reference's own bitmap.c / vector.c / threadpool.c."""
    h = C.CDLL(os.fspath(_HOST_SO), mode=os.RTLD_LOCAL | os.RTLD_NOW)
    h.write_bitmap.restype = None
    h.write_bitmap.argtypes = [C.c_char_p, C.c_uint, C.c_uint, C.POINTER(C.c_uint8), C.c_bool]
    for name in ("vector_init", "vector_free"):
        getattr(h, name).restype = None
        getattr(h, name).argtypes = [C.POINTER(Vector)]
    h.vector_init_with_capacity.restype = None
    h.vector_init_with_capacity.argtypes = [C.POINTER(Vector), C.c_size_t]
    h.vector_push.restype = None
    h.vector_push.argtypes = [C.POINTER(Vector), C.c_void_p]
    h.vector_pop.restype = C.c_void_p
    h.vector_pop.argtypes = [C.POINTER(Vector)]
    h.threadpool_init.restype = None
    h.threadpool_init.argtypes = [C.c_void_p, C.c_size_t]
    h.threadpool_add_task.restype = None
    h.threadpool_add_task.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    h.threadpool_destroy.restype = None
    h.threadpool_destroy.argtypes = [C.c_void_p]
    return h


def _probe(path):
    p = C.CDLL(os.fspath(path), mode=os.RTLD_LOCAL | os.RTLD_NOW)
    for name in ("ref_sizeof_vector", "ref_sizeof_threadpool", "ref_alignof_threadpool",
                 "ref_num_fields"):
        getattr(p, name).restype = C.c_size_t
        getattr(p, name).argtypes = []
    p.ref_field_name.restype = C.c_char_p
    p.ref_field_name.argtypes = [C.c_size_t]
    for name in ("ref_field_offset", "ref_field_size"):
        getattr(p, name).restype = C.c_size_t
        getattr(p, name).argtypes = [C.c_size_t]
    fields = {p.ref_field_name(i).decode(): (p.ref_field_offset(i), p.ref_field_size(i))
              for i in range(p.ref_num_fields())}
    return dict(vector=p.ref_sizeof_vector(), threadpool=p.ref_sizeof_threadpool(),
                align=p.ref_alignof_threadpool(), fields=fields)


# ---- (i) write_bitmap: src/bitmap.c:5-78 ---------------------------------------

# every padding class (3w mod 4 = 0..3), single row / column, the empty picture
_BMP_SIZES = [(1, 1), (2, 3), (3, 2), (4, 4), (5, 3), (6, 1), (7, 5), (8, 2), (9, 9),
              (1, 17), (33, 1), (64, 48), (127, 3), (0, 0), (0, 4), (5, 0)]


@pytest.mark.parametrize("y_inverted", [True, False])
@pytest.mark.parametrize("w,h", _BMP_SIZES)
def test_write_bitmap_bytes_equal_the_reference(pt, ref, tmp_path, w, h, y_inverted):
    rng = np.random.default_rng(w * 131 + h * 7 + int(y_inverted))
    frame = rng.integers(0, 256, size=max(w * h * 3, 1), dtype=np.uint8)
    ptr = frame.ctypes.data_as(C.POINTER(C.c_uint8))
    a, b = tmp_path / "ref.bmp", tmp_path / "ours.bmp"
    ref.write_bitmap(os.fsencode(a), w, h, ptr, y_inverted)
    pt.lib().write_bitmap(os.fsencode(b), w, h, ptr, y_inverted)
    got, want = b.read_bytes(), a.read_bytes()
    assert len(want) == 54 + ((3 * w + 3) // 4 * 4) * h
    assert got == want


def test_write_bitmap_of_a_rendered_size_frame_equals_the_reference(pt, ref, tmp_path):
    """The reference's own window size (src/main.c: 1280x800), as the offline
    render writes it: y_inverted = true."""
    w, h = 1280, 800
    frame = np.random.default_rng(5).integers(0, 256, size=w * h * 3, dtype=np.uint8)
    ptr = frame.ctypes.data_as(C.POINTER(C.c_uint8))
    a, b = tmp_path / "ref.bmp", tmp_path / "ours.bmp"
    ref.write_bitmap(os.fsencode(a), w, h, ptr, True)
    pt.lib().write_bitmap(os.fsencode(b), w, h, ptr, True)
    assert b.read_bytes() == a.read_bytes()


# ---- (ii) vector_*: src/vector.c:3-31 -------------------------------------------

def _vector_trace(lib, ops, initial_capacity):
    """Runs `ops` (('push', value) / ('pop',)) and records what a caller can
    see after each: popped value, size, capacity, and the live elements."""
    v = Vector()
    if initial_capacity is None:
        lib.vector_init(C.byref(v))
    else:
        lib.vector_init_with_capacity(C.byref(v), initial_capacity)
    trace = [("init", v.size, v.capacity)]
    for op in ops:
        if op[0] == "push":
            lib.vector_push(C.byref(v), C.c_void_p(op[1]))
            out = None
        else:
            out = lib.vector_pop(C.byref(v))
        live = tuple(v.data[i] for i in range(v.size))
        trace.append((op[0], out, v.size, v.capacity, live))
    lib.vector_free(C.byref(v))
    return trace


@pytest.mark.parametrize("initial_capacity", [None, 1, 3, 8, 100])
def test_vector_sequences_equal_the_reference(pt, ref, initial_capacity):
    rng = np.random.default_rng(11 if initial_capacity is None else initial_capacity)
    ops, depth = [("pop",)], 0                       # pop of an empty vector -> NULL
    for _ in range(600):
        if depth == 0 or rng.random() < 0.6:
            ops.append(("push", int(rng.integers(1, 2**48))))
            depth += 1
        else:
            ops.append(("pop",))
            depth -= 1
    ops += [("pop",)] * (depth + 2)                  # drain, then two pops too many
    ours_lib = pt.lib()
    ours_lib.vector_pop.restype = C.c_void_p
    want = _vector_trace(ref, ops, initial_capacity)
    got = _vector_trace(ours_lib, ops, initial_capacity)
    assert got == want
    assert want[0] == ("init", 0, 8 if initial_capacity is None else initial_capacity)


# ---- (iii) struct layouts: include/threadpool.h:6-13, include/vector.h:5-9 ------

def test_pool_and_vector_layouts_equal_what_the_reference_headers_compile_to(pt, tmp_path):
    ours_so = tmp_path / "libours_probe.so"
    subprocess.run(["gcc", "-std=gnu2x", "-O2", "-fPIC", "-shared", f"-I{ROOT / 'include'}",
                    "-o", os.fspath(ours_so), os.fspath(ROOT / "oracle" / "ref_probe.c")],
                   check=True)
    want, got = _probe(_PROBE_SO), _probe(ours_so)
    assert got == want
    assert want["threadpool"] == 192 and want["vector"] == 24
    assert want["fields"]["threadpool.threads_running"] == (184, 1)
    # and the ctypes mirror the Python harness uses
    tp = pt._abi.ThreadPool
    assert C.sizeof(tp) == want["threadpool"] and C.sizeof(pt._abi.Vector) == want["vector"]
    for name, (off, size) in want["fields"].items():
        struct, field = name.split(".")
        desc = getattr(tp if struct == "threadpool" else pt._abi.Vector, field)
        assert (desc.offset, desc.size) == (off, size), name


# ---- (iv) task order of a one-thread pool: src/threadpool.c:14-41,84-95 ---------

TASK_FN = C.CFUNCTYPE(None, C.c_void_p)


def test_one_thread_pool_runs_queued_tasks_last_in_first_out_like_the_reference(pt, ref):
    """The reference's worker pops the NEWEST task (vector_pop,
    src/threadpool.c:28).  With one worker held inside a first task while k more
    are queued, the order is fully determined: k-1 ... 0.  The shim runs
    ordinary (non-render_task) tasks the same way, on worker threads of its own
    (csrc/host/threadpool.c): the same procedure on both pools, same order.
    The reference's threadpool_wait_for_tasks is not called: it waits on a
    condition variable without a predicate (src/threadpool.c:97-101, SURVEY
    3.4) and can miss its wake-up; the test waits on its own event instead."""
    k = 12

    def run_on(init, add, destroy, pool):
        gate, started, done = threading.Event(), threading.Event(), threading.Event()
        order = []

        def body(arg):
            tag = int(arg or 0)
            if tag == 1000:
                started.set()
                gate.wait(10)
                return
            order.append(tag - 1)
            if len(order) == k:
                done.set()

        fn = TASK_FN(body)
        init(pool, 1)
        add(pool, C.cast(fn, C.c_void_p), C.c_void_p(1000))
        assert started.wait(10)           # the one worker is inside the first task: nothing waits for wait()
        for i in range(k):
            add(pool, C.cast(fn, C.c_void_p), C.c_void_p(i + 1))
        gate.set()
        assert done.wait(10)
        # one worker, now idle in pthread_cond_wait: destroy's single signal
        # (src/threadpool.c:61) reaches it and the join returns
        destroy(pool)
        return order

    want = run_on(ref.threadpool_init, ref.threadpool_add_task, ref.threadpool_destroy,
                  (C.c_ubyte * 192)())
    assert want == list(range(k - 1, -1, -1))
    lib = pt.lib()
    ours = pt._abi.ThreadPool()
    got = run_on(lambda p, n: lib.threadpool_init(C.byref(p), n),
                 lambda p, f, a: lib.threadpool_add_task(C.byref(p), f, a),
                 lambda p: lib.threadpool_destroy(C.byref(p)), ours)
    assert got == want
