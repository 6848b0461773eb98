#!/usr/bin/env python3
"""Tolerance mode (PTGPU_FLAG_FAST_MATH) against the exact GPU frame: share of
bytes within +-1, RMSE in 8-bit levels, mean signed difference, and the time
of both builds.  Usage: tools/fast_math_parity.py [config ...]"""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

import path_tracer_c_amd as pt  # noqa: E402

CASES = {
    "c1": dict(scene=1, w=256, h=256, spp=16, b=4, seed=12345),
    "demo": dict(scene=0, w=1280, h=800, spp=16, b=4, seed=1),
    "c2": dict(scene=2, w=1920, h=1080, spp=256, b=8, seed=1),
    "c5": dict(scene=5, w=1920, h=1080, spp=256, b=8, seed=1),
}


def render(d, c, flags):
    p = pt.make_params(spp=c["spp"], max_bounces=c["b"], seed=c["seed"], flags=flags)
    out = torch.zeros((c["h"], c["w"], 3), dtype=torch.uint8, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    d.render(out.data_ptr(), c["w"], c["h"], p, st)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    d.render(out.data_ptr(), c["w"], c["h"], p, st)
    torch.cuda.synchronize()
    return out.cpu().numpy(), (time.perf_counter() - t0) * 1e3


def main():
    pt._abi.check(pt.lib().ptgpu_set_device(0))
    for name in (sys.argv[1:] or list(CASES)):
        c = CASES[name]
        with pt.HostScene(c["scene"]) as s, pt.DeviceScene(s) as d:
            exact, ms_e = render(d, c, pt._abi.FLAG_NO_PRIMARY_SHARE)
            fast, ms_f = render(d, c, pt._abi.FLAG_NO_PRIMARY_SHARE | pt._abi.FLAG_FAST_MATH)
        diff = fast.astype(np.int32) - exact.astype(np.int32)
        print(f"{name}: within+-1 {np.mean(np.abs(diff) <= 1):.6f}  exact-equal {np.mean(diff == 0):.4f}  "
              f"rmse {np.sqrt(np.mean(diff.astype(np.float64) ** 2)):.4f}  mean {diff.mean():+.5f}  "
              f"max |d| {np.abs(diff).max()}  exact {ms_e:.2f} ms  fast {ms_f:.2f} ms  "
              f"({ms_e / ms_f:.2f}x)", flush=True)


if __name__ == "__main__":
    main()
