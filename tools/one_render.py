"""One configuration rendered a few times with fixed flags -- the thing to put behind rocprofv3 --pmc.
    python tools/one_render.py c5 0x500 [spp] [frames] [lanes_per_pixel] [block_threads]"""
import sys
sys.path.insert(0, '.')
import torch
import path_tracer_c_amd as pt

c = pt.CONFIGS[int(sys.argv[1][1:])]
flags = int(sys.argv[2], 0)
spp = int(sys.argv[3]) if len(sys.argv) > 3 else c['spp']
frames = int(sys.argv[4]) if len(sys.argv) > 4 else 2
lanes = int(sys.argv[5]) if len(sys.argv) > 5 else 0
block = int(sys.argv[6]) if len(sys.argv) > 6 else 0
w, h = c['width'], c['height']
with pt.HostScene(c['scene']) as s, pt.DeviceScene(s) as d:
    st = torch.cuda.current_stream().cuda_stream
    out = torch.zeros((h, w, 3), dtype=torch.uint8, device='cuda')
    p = pt.make_params(spp=spp, max_bounces=c['max_bounces'], seed=c['seed'], flags=flags, lanes_per_pixel=lanes, block_threads=block)
    pt.host.reset_stats()
    for _ in range(frames):
        d.render(out.data_ptr(), w, h, p, st)
    torch.cuda.synchronize()
    stt = pt.host.read_stats()
    print(f"{sys.argv[1]} flags {flags:#x} spp {spp}: {frames} frames, rays {stt.rays}, samples {stt.samples}, schedule {d.schedule}, scan {d.scan}")
