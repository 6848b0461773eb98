#!/usr/bin/env python3
"""The REFERENCE's own scheduler driving this library's render_task.

oracle/_ref/libref_host.so is /root/reference/src/threadpool.c (+ vector.c,
bitmap.c) compiled unmodified (oracle/ref.mk).  Here its threadpool_init /
threadpool_add_task run 8 worker threads that call libptgpu.so's render_task,
one row per task -- the reference's CPU driver loop with only
the task body replaced -- and the frame is compared with ptgpu_render_frame.

The reference's threadpool_wait_for_tasks can return early or miss its wake-up
and its threadpool_destroy hangs with more than one worker, so
completion is counted by the task wrapper and the process leaves through
os._exit without destroying the pool.  Exit 0 = frames equal, 77 = no _ref build.
"""
import ctypes as C
import os
import sys
import threading
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
import path_tracer_c_amd as pt  # noqa: E402

TASK_FN = C.CFUNCTYPE(None, C.c_void_p)


def main():
    so = ROOT / "oracle" / "_ref" / "libref_host.so"
    if not so.exists():
        print("oracle/_ref/libref_host.so is not built")
        return 77
    ref = C.CDLL(os.fspath(so), mode=os.RTLD_LOCAL | os.RTLD_NOW)
    ref.threadpool_init.argtypes = [C.c_void_p, C.c_size_t]
    ref.threadpool_add_task.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    os.environ.update(PTGPU_SPP="8", PTGPU_BOUNCES="4", PTGPU_SEED="21")
    L, A = pt.lib(), pt._abi
    w, h = 160, 100
    done = [0]
    lock = threading.Lock()
    workers = set()

    def body(arg):                      # runs on the reference's worker threads
        L.render_task(C.cast(arg, C.POINTER(A.RenderArgs)))
        with lock:
            done[0] += 1
            workers.add(threading.get_ident())

    fn = TASK_FN(body)
    status = 0
    for cfg in (0, 2):
        with pt.HostScene(cfg) as s:
            want = pt.render_frame(s, w, h, spp=8, max_bounces=4, seed=21)
            frame = np.zeros((h, w, 3), dtype=np.uint8)
            args = (A.RenderArgs * h)()
            for y in range(h):
                args[y].screen_y = y
                args[y].framew, args[y].frameh = w, h
                args[y].frame = frame.ctypes.data_as(C.POINTER(C.c_uint8))
                args[y].world = C.pointer(s.c)
            pool = (C.c_ubyte * 192)()          # the reference's struct, caller-allocated
            ref.threadpool_init(pool, 8)
            done[0] = 0
            t0 = time.perf_counter()
            for y in range(h):
                ref.threadpool_add_task(pool, fn, C.cast(C.byref(args[y]), C.c_void_p))
            while done[0] < h and time.perf_counter() - t0 < 60:
                time.sleep(0.001)
            ok = done[0] == h and np.array_equal(frame, want)
            print(f"scene {cfg}: {done[0]}/{h} rows rendered by the reference's pool on {len(workers)} worker "
                  f"threads in {1e3 * (time.perf_counter() - t0):.1f} ms; frame equal to ptgpu_render_frame: {ok}",
                  flush=True)
            status |= 0 if ok else 1
    os._exit(status)                     # no threadpool_destroy: it would hang


if __name__ == "__main__":
    sys.exit(main())
