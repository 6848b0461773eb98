"""Which of the BVH forms agrees with the linear (screened) scan on configuration 5?"""
import os
import sys
sys.path.insert(0, '.')
import torch
import path_tracer_c_amd as pt

c = pt.CONFIGS[5]
w, h = c['width'], c['height']
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 64
with pt.HostScene(c['scene']) as s, pt.DeviceScene(s) as d:
    st = torch.cuda.current_stream().cuda_stream
    frames = {}
    for name, flags, duo in (("screened linear", 0x320, '0'), ("stream", 0x700, '0'), ("whole walks", 0xF00, '0'), ("duo", 0x700, '1')):
        os.environ['PTGPU_DUO'] = duo
        out = torch.zeros((h, w, 3), dtype=torch.uint8, device='cuda')
        p = pt.make_params(spp=spp, max_bounces=c['max_bounces'], seed=1, flags=flags)
        pt.host.reset_stats()
        d.render(out.data_ptr(), w, h, p, st); torch.cuda.synchronize()
        stt = pt.host.read_stats()
        frames[name] = out
        diff = (out != frames["screened linear"]).any(dim=2).nonzero()
        print(f"{name:16s} rays {stt.rays} differing pixels vs linear: {diff.shape[0]} {diff[:8].tolist()}", flush=True)
