"""lanes_per_pixel sweep of the pixel-owner schedule: ms per frame by scene and k.
    python tools/sweep_lanes.py [c1 c2 ...]"""
import sys
sys.path.insert(0, '.')
import torch
import path_tracer_c_amd as pt

names = sys.argv[1:] or ["c1", "c2", "c5"]
for name in names:
    c = pt.CONFIGS[int(name[1:])]
    w, h, spp = c['width'], c['height'], min(c['spp'], 256)
    if name == 'c4': spp = 64
    with pt.HostScene(c['scene']) as s, pt.DeviceScene(s) as d:
        st = torch.cuda.current_stream().cuda_stream
        ref = torch.zeros((h, w, 3), dtype=torch.uint8, device='cuda')
        out = torch.zeros_like(ref)
        p = pt.make_params(spp=spp, max_bounces=c['max_bounces'], seed=1)
        d.render(ref.data_ptr(), w, h, p, st); torch.cuda.synchronize()   # settles the scan form
        for share, flags in (("shared primary", 0x80), ("every ray scanned", 0x280)):
            for k in (1, 2, 4, 8, 16, 32, 64):
                if k > spp:
                    continue
                p = pt.make_params(spp=spp, max_bounces=c['max_bounces'], seed=1, flags=flags, lanes_per_pixel=k)
                d.render(out.data_ptr(), w, h, p, st); torch.cuda.synchronize()
                e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                pt.host.reset_stats()
                e0.record()
                for _ in range(3): d.render(out.data_ptr(), w, h, p, st)
                e1.record(); torch.cuda.synchronize()
                stt = pt.host.read_stats()
                util = stt.rays / (64.0 * stt.wave_iters) if stt.wave_iters else 0.0
                print(f"{name} {w}x{h} {spp} spp scan {d.scan}  {share:18s} k={k:2d}  {e0.elapsed_time(e1)/3:8.2f} ms  lane use {util:.3f}  same {bool(torch.equal(out, ref))}", flush=True)
