"""Writes tests/golden/full_frames.json: SHA-256 digests of the BASELINE configurations' frames
at FULL size, rendered on an MI355X by this repo's HIP path (run on a GPU box:
`python tests/golden/make_full_frame_digests.py`).

These are regression pins, not an independent check: the CPU oracle cannot render a 1080p /
256-spp frame in seconds, so the full-size parity tests compare a few rows of each frame with the
oracle (tests/test_parity_gpu.py, test_full_size_*), and the digests recorded here -- taken from a
build whose rows passed those tests -- hold the *rest* of every frame to the same bytes from then
on, whatever scan form, schedule or kernel change comes later.  Eight horizontal bands are
digested as well, so that a mismatch says where.
"""
import hashlib
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))


def digests(frame):
    h = frame.shape[0]
    bands = [hashlib.sha256(frame[h * k // 8: h * (k + 1) // 8].tobytes()).hexdigest()[:16] for k in range(8)]
    return {"sha256": hashlib.sha256(frame.tobytes()).hexdigest(), "bands": bands}


def render_full(pt, torch, key, passes=1, flags=0):
    """One full-size frame of configuration `key` through the device-resident entry points."""
    import numpy as np
    c = pt.CONFIGS[key]
    w, h = c["width"], c["height"]
    stream = torch.cuda.current_stream().cuda_stream
    with pt.HostScene(c["scene"]) as s, pt.DeviceScene(s) as d:
        out = torch.zeros((h, w, 3), dtype=torch.uint8, device="cuda")
        kw = dict(max_bounces=c["max_bounces"], seed=c["seed"], flags=flags)
        if passes == 1:
            d.render(out.data_ptr(), w, h, pt.make_params(spp=c["spp"], **kw), stream)
        else:
            acc = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")
            per = c["spp"] // passes
            for k in range(passes):
                d.accumulate(acc.data_ptr(), w, h, pt.make_params(spp=per, sample_offset=k * per, **kw), stream)
            pt.host.resolve(acc.data_ptr(), out.data_ptr(), w * h, stream)
        torch.cuda.synchronize()
        return out.cpu().numpy()


FRAMES = {"c1": (1, 1), "c2": (2, 1), "c3": (3, 16), "c4": (4, 1), "c5": (5, 1)}


def main():
    import torch
    import path_tracer_c_amd as pt
    table = {}
    for name, (key, passes) in FRAMES.items():
        c = pt.CONFIGS[key]
        table[name] = dict(width=c["width"], height=c["height"], spp=c["spp"],
                           max_bounces=c["max_bounces"], seed=c["seed"], passes=passes,
                           **digests(render_full(pt, torch, key, passes)))
        print(name, table[name]["sha256"][:16], flush=True)
    out = Path(__file__).with_name("full_frames.json")
    out.write_text(json.dumps(table, indent=1) + "\n")
    print("wrote", out)


if __name__ == "__main__":
    main()
