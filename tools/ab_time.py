"""A/B timing of a library variant (tools/exp_lib.sh <lib> -- python3 tools/ab_time.py): configuration 2
(256 spp, every ray scanned) and configuration 5 (64 spp, every segment walked), frame checksums printed
so that a variant that changes the picture shows."""
import sys
sys.path.insert(0, '.')
import torch
import path_tracer_c_amd as pt
A = pt._abi
for cfg, spp, flags in ((2, 256, A.FLAG_NO_PRIMARY_SHARE), (5, 64, A.FLAG_NO_PRIMARY_SHARE)):
    c = pt.CONFIGS[cfg]; w, h = c['width'], c['height']
    with pt.HostScene(c['scene']) as s, pt.DeviceScene(s) as d:
        st = torch.cuda.current_stream().cuda_stream
        out = torch.zeros((h, w, 3), dtype=torch.uint8, device='cuda')
        p = pt.make_params(spp=spp, max_bounces=c['max_bounces'], seed=c['seed'], flags=flags)
        d.tune(w, h, p)
        d.render(out.data_ptr(), w, h, p, st)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4):
            d.render(out.data_ptr(), w, h, p, st)
        e1.record(); torch.cuda.synchronize()
        print(f"config {cfg} {spp} spp: {e0.elapsed_time(e1)/4:.3f} ms  ({d.scan}, {d.schedule})  checksum {int(out.sum())}")
