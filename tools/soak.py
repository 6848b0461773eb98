"""Determinism soak: N full-size frames of configs 2 and 5 must be byte-identical (GPU box)."""
import sys, hashlib
sys.path.insert(0, '.')
import torch
import path_tracer_c_amd as pt
for cfg, spp, n, flagsets in ((2, 256, 12, (0, 0x100, 0x40, 0x200)), (5, 32, 8, (0, 0x80, 0x20, 0x500, 0x700, 0xD00))):
    c = pt.CONFIGS[cfg]; w, h = c['width'], c['height']
    with pt.HostScene(c['scene']) as s, pt.DeviceScene(s) as d:
        digests = set()
        out = torch.zeros((h, w, 3), dtype=torch.uint8, device='cuda')
        st = torch.cuda.current_stream().cuda_stream
        for flags in flagsets:
            for i in range(n):
                out.zero_()
                d.render(out.data_ptr(), w, h, pt.make_params(spp=spp, max_bounces=8, seed=1, flags=flags), st)
                torch.cuda.synchronize()
                digests.add(hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest())
        print(f"config {cfg}: {n * len(flagsets)} frames over flag sets {[hex(f) for f in flagsets]} -> {len(digests)} distinct image(s)")
        assert len(digests) == 1
