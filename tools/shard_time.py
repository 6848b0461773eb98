"""Time one rank's shard of the C2 frame (as bench.py --gpus N would render it) on one GPU."""
import sys
sys.path.insert(0, '.')
import torch
import path_tracer_c_amd as pt
from path_tracer_c_amd import tiles
c = pt.CONFIGS[2]; w, h = c['width'], c['height']
with pt.HostScene(c['scene']) as s, pt.DeviceScene(s) as d:
    st = torch.cuda.current_stream().cuda_stream
    full = torch.zeros((h, w, 3), dtype=torch.uint8, device='cuda')
    d.render(full.data_ptr(), w, h, pt.make_params(spp=c['spp'], max_bounces=8, seed=1, flags=0x200), st); torch.cuda.synchronize()
    for world in (1, 2, 4, 8):
        sh = tiles.shard_rows(h, world, 0, 8)
        pad = tiles.max_shard_rows(h, world, 8)
        out = torch.zeros((pad, w, 3), dtype=torch.uint8, device='cuda')
        for name, flags, k in (("default", 0x201, 0), ("owner", 0x281, 1), ("queue", 0x301, 1),
                               ("owner k=2", 0x281, 2), ("owner k=4", 0x281, 4), ("owner k=8", 0x281, 8),
                               ("owner k=16", 0x281, 16), ("owner k=32", 0x281, 32)):
            p = pt.make_params(spp=c['spp'], max_bounces=8, seed=1, row_begin=sh.row_begin, num_rows=sh.num_rows,
                               tile_rows=8 if world > 1 else 0, tile_stride=sh.tile_stride, flags=flags,
                               lanes_per_pixel=k)
            d.render(out.data_ptr(), w, h, p, st); torch.cuda.synchronize()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3): d.render(out.data_ptr(), w, h, p, st)
            e1.record(); torch.cuda.synchronize()
            ok = bool(torch.equal(out[:sh.num_rows], full[sh.global_rows()]))
            print(f"world {world}: rank-0 shard {sh.num_rows} rows  {name:10s} {e0.elapsed_time(e1)/3:7.2f} ms  (x{world} = {e0.elapsed_time(e1)/3*world:6.1f})  same-as-full {ok}", flush=True)
