"""Sample-queue schedule with the BVH scan on configuration 5: resumable walks
(pt_streamkernel) against whole walks, over the service threshold.
    [PTGPU_WALK_BUDGET=n] python tools/stream_time.py [spp]"""
import sys
sys.path.insert(0, '.')
import torch
import path_tracer_c_amd as pt

c = pt.CONFIGS[5]
w, h = c['width'], c['height']
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 64
with pt.HostScene(c['scene']) as s, pt.DeviceScene(s) as d:
    st = torch.cuda.current_stream().cuda_stream
    ref = torch.zeros((h, w, 3), dtype=torch.uint8, device='cuda')
    out = torch.zeros_like(ref)
    p = pt.make_params(spp=spp, max_bounces=c['max_bounces'], seed=1, flags=0x480, lanes_per_pixel=1)
    d.render(ref.data_ptr(), w, h, p, st); torch.cuda.synchronize()
    for name, flags, settle, block in [("whole walks", 0xD00, 0, 0)] + [(f"stream, service at {t}", 0x500, t, 0) for t in (16, 24, 32)] + \
            [(f"stream, every ray walked, service at {t}", 0x700, t, 0) for t in (24, 32)] + \
            [(f"stream, block {b}", 0x500, 24, b) for b in (128, 256)] + [(f"stream+LDS, block {b}", 0x540, 24, b) for b in (512, 1024)]:
        p = pt.make_params(spp=spp, max_bounces=c['max_bounces'], seed=1, flags=flags, settle_lanes=settle, block_threads=block)
        d.render(out.data_ptr(), w, h, p, st); torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        pt.host.reset_stats()
        e0.record()
        for _ in range(2): d.render(out.data_ptr(), w, h, p, st)
        e1.record(); torch.cuda.synchronize()
        stt = pt.host.read_stats()
        ms = e0.elapsed_time(e1) / 2
        print(f"{name:40s} {ms:8.2f} ms  {stt.rays / 2 / ms / 1e6:6.2f} Gray/s  same {bool(torch.equal(out, ref))}", flush=True)
