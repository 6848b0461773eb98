"""pt_duokernel on configuration 5 over its thresholds (PTGPU_DUO_* knobs, read per call).
    python tools/duo_sweep.py [spp]"""
import itertools
import os
import sys
sys.path.insert(0, '.')
import torch
import path_tracer_c_amd as pt

c = pt.CONFIGS[5]
w, h = c['width'], c['height']
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 16
flags = 0x700
with pt.HostScene(c['scene']) as s, pt.DeviceScene(s) as d:
    st = torch.cuda.current_stream().cuda_stream
    out = torch.zeros((h, w, 3), dtype=torch.uint8, device='cuda')
    p = pt.make_params(spp=spp, max_bounces=c['max_bounces'], seed=1, flags=flags)

    def run(env):
        for k in list(os.environ):
            if k.startswith('PTGPU_DUO'):
                del os.environ[k]
        os.environ.update(env)
        d.render(out.data_ptr(), w, h, p, st); torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3): d.render(out.data_ptr(), w, h, p, st)
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / 3, int(out.sum())

    ms, chk = run({'PTGPU_DUO': '0'})
    print(f"one path per lane: {ms:7.2f} ms  checksum {chk}", flush=True)
    base = dict(PTGPU_DUO='1')
    ms, chk2 = run(base)
    print(f"two, stream kernel's thresholds: {ms:7.2f} ms  same {chk2 == chk}", flush=True)
    for service, shade, leaf, resolve in itertools.product((40, 48, 56), (24, 32, 48), (16, 32), (4, 8, 16)):
        env = dict(base, PTGPU_DUO_SERVICE=str(service), PTGPU_DUO_SHADE=str(shade), PTGPU_DUO_LEAF=str(leaf),
                   PTGPU_DUO_RESOLVE=str(resolve))
        ms, chk2 = run(env)
        print(f"service {service} shade {shade} leaf {leaf} resolve {resolve}: {ms:7.2f} ms  same {chk2 == chk}", flush=True)
