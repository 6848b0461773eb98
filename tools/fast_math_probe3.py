#!/usr/bin/env python3
"""Rays per sample by bounce limit, exact vs tolerance build, configuration 5."""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

import path_tracer_c_amd as pt  # noqa: E402

A = pt._abi


def render(d, w, h, spp, b, flags):
    pt.host.reset_stats()
    p = pt.make_params(spp=spp, max_bounces=b, seed=1, flags=flags)
    out = torch.zeros((h, w, 3), dtype=torch.uint8, device="cuda")
    d.render(out.data_ptr(), w, h, p, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    st = pt.host.read_stats()
    return out.cpu().numpy().astype(np.float64), st.rays / max(1, st.samples)


def main():
    pt._abi.check(pt.lib().ptgpu_set_device(0))
    w, h, spp = 960, 540, 32
    base = A.FLAG_NO_PRIMARY_SHARE | A.FLAG_SCAN_BVH | A.FLAG_SAMPLE_QUEUE
    with pt.HostScene(5) as s, pt.DeviceScene(s) as d:
        for b in (0, 1, 2, 4, 8):
            e, re_ = render(d, w, h, spp, b, base)
            f, rf = render(d, w, h, spp, b, base | A.FLAG_FAST_MATH)
            print(f"bounces {b}: rays/sample exact {re_:.5f} fast {rf:.5f} (diff {rf - re_:+.5f}); "
                  f"mean byte exact {e.mean():.3f} fast {f.mean():.3f}", flush=True)


if __name__ == "__main__":
    main()
