"""Kernel-time sweep over launch variants (GPU box).  usage: sweep.py [c2|c5] ..."""
import sys, time
sys.path.insert(0, '.')
import torch
import path_tracer_c_amd as pt

def run(cfgname, spp=None, flags=0, block=0, reps=3):
    c = dict(pt.CONFIGS[int(cfgname[1])])
    if spp: c['spp'] = spp
    w, h = c['width'], c['height']
    with pt.HostScene(c['scene']) as s, pt.DeviceScene(s) as d:
        out = torch.zeros((h, w, 3), dtype=torch.uint8, device='cuda')
        p = pt.make_params(spp=c['spp'], max_bounces=c['max_bounces'], seed=c['seed'], flags=flags, block_threads=block)
        st = torch.cuda.current_stream().cuda_stream
        d.render(out.data_ptr(), w, h, p, st); torch.cuda.synchronize()
        pt.host.reset_stats()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): d.render(out.data_ptr(), w, h, p, st)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        stt = pt.host.read_stats()
        rays = stt.rays / reps
        print(f"{cfgname} spp{c['spp']} flags{flags:#x} block{block or 256}: {ms:8.2f} ms  {rays/ms/1e6:8.2f} Gray/s  util {stt.rays/(64*stt.wave_iters):.3f}", flush=True)
        return out.cpu().numpy()

if __name__ == '__main__':
    which = sys.argv[1:] or ['c2', 'c5']
    for name in which:
        spp = 256 if name == 'c2' else 32
        ref = None
        for flags in (0, 2):
            for block in (64, 128, 256, 512, 1024):
                img = run(name, spp=spp, flags=flags, block=block)
                if ref is None: ref = img
                assert (img == ref).all()
