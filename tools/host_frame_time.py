"""PCIe-inclusive time of the host-buffer entry points: BASELINE config 2, and the reference's own
scene at its window size and settings (1280x800, 16 spp, 4 bounces: src/main.c:17-18, src/renderer.c:8-9)."""
import os, sys, time
sys.path.insert(0, '.')
import path_tracer_c_amd as pt

def measure(scene_id, w, h, spp, bounces, seed, label):
    with pt.HostScene(scene_id) as s:
        kw = dict(spp=spp, max_bounces=bounces, seed=seed)
        pt.render_frame(s, w, h, **kw)
        t = []
        for _ in range(8):
            t0 = time.perf_counter(); pt.render_frame(s, w, h, **kw); t.append(time.perf_counter() - t0)
        print(f"{label}: ptgpu_render_frame (upload + kernel + {3*w*h/1e6:.1f} MB download): {min(t)*1e3:.3f} ms min, {sum(t)/len(t)*1e3:.3f} ms mean")
        os.environ.update(PTGPU_SPP=str(spp), PTGPU_BOUNCES=str(bounces), PTGPU_SEED=str(seed))
        pt.host.render_frame_legacy(s, w, h)
        t = []
        for _ in range(8):
            t0 = time.perf_counter(); pt.host.render_frame_legacy(s, w, h); t.append(time.perf_counter() - t0)
        print(f"{label}: threadpool_add_task x{h} + wait_for_tasks (legacy API): {min(t)*1e3:.3f} ms min, {sum(t)/len(t)*1e3:.3f} ms mean")

c = pt.CONFIGS[2]
measure(c['scene'], c['width'], c['height'], c['spp'], c['max_bounces'], c['seed'], "config 2")
measure(0, 1280, 800, 16, 4, 1, "reference scene 1280x800x16")
