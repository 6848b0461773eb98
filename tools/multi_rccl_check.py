#!/usr/bin/env python3
"""ptgpu_multi_render over RCCL against the single-device frame.  Exit 0 = equal.

    python tools/multi_rccl_check.py [n_devices]     >= 2 GPUs in one process (77 = fewer)
    python tools/multi_rccl_check.py --loopback n    ONE GPU: n shards on device 0, the
                                                     communicator has one rank and every
                                                     shard travels by ncclSend/ncclRecv to
                                                     self (PTGPU_MULTI_RCCL_LOOPBACK): the
                                                     library calls and the stream ordering
                                                     on real hardware, without a second GPU

Run by tests/test_parity_gpu.py in a child process with a timeout, so that a
collective that ever hung could not hang the test suite."""
import ctypes as C
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import path_tracer_c_amd as pt  # noqa: E402


def main():
    L = pt.lib()
    have = L.ptgpu_device_count()
    loopback = len(sys.argv) > 1 and sys.argv[1] == "--loopback"
    if loopback:
        devices = [0] * (int(sys.argv[2]) if len(sys.argv) > 2 else 3)
    else:
        want = int(sys.argv[1]) if len(sys.argv) > 1 else min(have, 8)
        if have < 2 or want < 2:
            print(f"{have} GPU(s): the RCCL gather needs two")
            return 77
        devices = list(range(min(want, have)))
    w, h = (1920, 1080) if not loopback else (641, 360)
    p = pt.make_params(spp=32, max_bounces=8, seed=3)
    pt._abi.check(L.ptgpu_set_device(0))
    d_full, d_multi = C.c_void_p(), C.c_void_p()
    pt._abi.check(L.ptgpu_device_alloc(3 * w * h, C.byref(d_full)))
    pt._abi.check(L.ptgpu_device_alloc(3 * w * h, C.byref(d_multi)))
    full = np.zeros((h, w, 3), np.uint8)
    multi = np.zeros_like(full)
    with pt.HostScene(2) as s, pt.DeviceScene(s) as d:
        d.render(d_full.value, w, h, p)
        pt._abi.check(L.ptgpu_device_synchronize())
        pt._abi.check(L.ptgpu_device_download(full.ctypes.data_as(C.c_void_p), d_full, full.nbytes))
        t0 = time.perf_counter()
        with pt.host.MultiDevice(devices, rccl_loopback=loopback) as md:
            t1 = time.perf_counter()
            assert md.transport == "rccl"
            md.upload(s)
            for tile in (0, 4):
                md.render(d_multi.value, w, h, p, tile)      # first call: pilots, code load
                t2 = time.perf_counter()
                md.render(d_multi.value, w, h, p, tile)
                t3 = time.perf_counter()
                pt._abi.check(L.ptgpu_device_download(multi.ctypes.data_as(C.c_void_p), d_multi,
                                                      multi.nbytes))
                if not np.array_equal(multi, full):
                    print(f"tile_rows {tile}: {int((multi != full).sum())} bytes differ")
                    return 1
                print(f"{len(devices)} {'shards on one device (loopback)' if loopback else 'devices'}, tile_rows {tile or 8}: frame equal to the single-device "
                      f"frame; {1e3 * (t3 - t2):.2f} ms per frame (communicators {1e3 * (t1 - t0):.0f} ms)")
    return 0


if __name__ == "__main__":
    sys.exit(main())
