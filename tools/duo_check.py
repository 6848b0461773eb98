"""Two paths per lane (pt_duokernel, PTGPU_DUO=1) against the stream kernel on configuration 5:
same frame, same ray counts, time of both.
    python tools/duo_check.py [spp] [flags]"""
import os
import sys
sys.path.insert(0, '.')
import torch
import path_tracer_c_amd as pt

c = pt.CONFIGS[5]
w, h = c['width'], c['height']
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 64
flag_sets = [int(sys.argv[2], 0)] if len(sys.argv) > 2 else [0x700, 0x500]
with pt.HostScene(c['scene']) as s, pt.DeviceScene(s) as d:
    st = torch.cuda.current_stream().cuda_stream
    for flags in flag_sets:
        frames = {}
        for duo in ('0', '1'):
            os.environ['PTGPU_DUO'] = duo
            out = torch.zeros((h, w, 3), dtype=torch.uint8, device='cuda')
            p = pt.make_params(spp=spp, max_bounces=c['max_bounces'], seed=1, flags=flags)
            d.render(out.data_ptr(), w, h, p, st); torch.cuda.synchronize()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            pt.host.reset_stats()
            e0.record()
            for _ in range(3): d.render(out.data_ptr(), w, h, p, st)
            e1.record(); torch.cuda.synchronize()
            stt = pt.host.read_stats()
            ms = e0.elapsed_time(e1) / 3
            frames[duo] = out
            print(f"flags {flags:#x} duo {duo}: {ms:8.2f} ms per {spp} spp  {stt.rays / 3 / ms / 1e6:6.2f} Gray/s  rays {stt.rays // 3} scans {stt.scans // 3} "
                  f"samples {stt.samples // 3} wave_iters {stt.wave_iters // 3}", flush=True)
        print(f"flags {flags:#x}: frames equal {bool(torch.equal(frames['0'], frames['1']))}", flush=True)
