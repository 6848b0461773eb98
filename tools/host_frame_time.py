"""PCIe-inclusive time of the host-buffer entry points on BASELINE config 2."""
import os, sys, time
sys.path.insert(0, '.')
import path_tracer_c_amd as pt
c = pt.CONFIGS[2]
with pt.HostScene(c['scene']) as s:
    kw = dict(spp=c['spp'], max_bounces=c['max_bounces'], seed=c['seed'])
    pt.render_frame(s, c['width'], c['height'], **kw)
    t = []
    for _ in range(5):
        t0 = time.perf_counter(); pt.render_frame(s, c['width'], c['height'], **kw); t.append(time.perf_counter() - t0)
    print(f"ptgpu_render_frame (upload + kernel + 6.2 MB download): {min(t)*1e3:.2f} ms min, {sum(t)/len(t)*1e3:.2f} ms mean")
    os.environ.update(PTGPU_SPP=str(c['spp']), PTGPU_BOUNCES=str(c['max_bounces']), PTGPU_SEED=str(c['seed']))
    pt.host.render_frame_legacy(s, c['width'], c['height'])
    t = []
    for _ in range(5):
        t0 = time.perf_counter(); pt.host.render_frame_legacy(s, c['width'], c['height']); t.append(time.perf_counter() - t0)
    print(f"threadpool_add_task x1080 + wait_for_tasks (legacy API):   {min(t)*1e3:.2f} ms min, {sum(t)/len(t)*1e3:.2f} ms mean")
