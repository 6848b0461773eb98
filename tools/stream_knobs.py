"""Configuration 5, sample queue + resumable walks, every segment walked: time of a
64-spp frame for one setting of the stream kernel's knobs (read from the environment
when the library starts a stream pass):
    PTGPU_LEAF_LANES=n PTGPU_RESOLVE_LANES=n PTGPU_WALK_BUDGET=n python tools/stream_knobs.py"""
import os
import sys
sys.path.insert(0, '.')
import torch
import path_tracer_c_amd as pt

c = pt.CONFIGS[5]
w, h = c['width'], c['height']
with pt.HostScene(c['scene']) as s, pt.DeviceScene(s) as d:
    st = torch.cuda.current_stream().cuda_stream
    out = torch.zeros((h, w, 3), dtype=torch.uint8, device='cuda')
    p = pt.make_params(spp=64, max_bounces=c['max_bounces'], seed=1, flags=0x700)
    d.render(out.data_ptr(), w, h, p, st); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): d.render(out.data_ptr(), w, h, p, st)
    e1.record(); torch.cuda.synchronize()
    knobs = {k: os.environ.get(k, '-') for k in ('PTGPU_LEAF_LANES', 'PTGPU_RESOLVE_LANES', 'PTGPU_WALK_BUDGET')}
    print(f"{knobs}  {e0.elapsed_time(e1) / 3:8.2f} ms per 64 spp  checksum {int(out.sum())}", flush=True)
