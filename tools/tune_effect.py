"""What ptgpu_scene_tune buys on configuration 2 and 5: frame time untuned (the
asynchronous entry point goes by object count), untuned with 8 lanes per pixel forced, tuned."""
import sys
sys.path.insert(0, '.')
import torch
import path_tracer_c_amd as pt


def frame_ms(d, w, h, p, n=3):
    st = torch.cuda.current_stream().cuda_stream
    out = torch.zeros((h, w, 3), dtype=torch.uint8, device='cuda')
    d.render(out.data_ptr(), w, h, p, st); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        d.render(out.data_ptr(), w, h, p, st)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for cfg in (2, 5):
    c = pt.CONFIGS[cfg]
    w, h = c['width'], c['height']
    kw = dict(spp=c['spp'], max_bounces=c['max_bounces'], seed=c['seed'], flags=pt._abi.FLAG_NO_PRIMARY_SHARE)
    with pt.HostScene(c['scene']) as s:
        s.add_sphere((1e6, 1e6, -1e6), 0.5, 0)      # content no other run has tuned (choices are memoised by content)
        with pt.DeviceScene(s) as d:
            a = frame_ms(d, w, h, pt.make_params(**kw))
            sched_a = (d.schedule, d.scan)
            b = frame_ms(d, w, h, pt.make_params(lanes_per_pixel=8, **kw)) if cfg == 2 else float('nan')
            d.tune(w, h, pt.make_params(**kw))
            t = frame_ms(d, w, h, pt.make_params(**kw))
            print(f"config {cfg}: untuned {a:.2f} ms {sched_a}; untuned, 8 lanes per pixel forced {b:.2f} ms; "
                  f"tuned {t:.2f} ms ({d.schedule}, {d.scan})", flush=True)
