"""Kernel-time sweeps over scheduling knobs (run on the GPU box):
    python tools/sweep_settle.py        # settle_lanes x scene storage, 4 scenes
Also provides run(), used by tools/stamps.py and tools/bvh_scaling.py."""
import sys
sys.path.insert(0, '.')
import torch
import path_tracer_c_amd as pt

def run(cfgname, w=None, h=None, spp=None, flags=0, block=0, settle=0, reps=3):
    c = dict(pt.CONFIGS[int(cfgname[1])]) if cfgname != 'c0' else dict(scene=0, width=1280, height=800, spp=64, max_bounces=4, seed=1)
    if spp: c['spp'] = spp
    if w: c['width'], c['height'] = w, h
    w, h = c['width'], c['height']
    with pt.HostScene(c['scene']) as s, pt.DeviceScene(s) as d:
        out = torch.zeros((h, w, 3), dtype=torch.uint8, device='cuda')
        p = pt.make_params(spp=c['spp'], max_bounces=c['max_bounces'], seed=c['seed'], flags=flags, block_threads=block, settle_lanes=settle)
        st = torch.cuda.current_stream().cuda_stream
        d.render(out.data_ptr(), w, h, p, st); torch.cuda.synchronize()
        pt.host.reset_stats()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): d.render(out.data_ptr(), w, h, p, st)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        stt = pt.host.read_stats()
        print(f"{cfgname} {w}x{h} spp{c['spp']} flags{flags:#x} block{block or 256} settle{settle}: {ms:8.2f} ms  {stt.rays/reps/ms/1e6:8.2f} Gray/s  util {stt.rays/(64*stt.wave_iters):.3f}", flush=True)
        return out.cpu().numpy()

if __name__ == '__main__':
    for name, kw in (('c2', {}), ('c5', dict(spp=32)), ('c1', dict(w=1920, h=1080, spp=256)), ('c0', {})):
        ref = None
        for flags in (0, 2):
            for settle in (1, 8, 16, 32, 64, 0):
                img = run(name, flags=flags, settle=settle, block=256 if flags == 0 else 64, **kw)
                if ref is None: ref = img
                assert (img == ref).all(), "image depends on settle_lanes!"
