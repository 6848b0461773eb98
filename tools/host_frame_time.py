"""PCIe-inclusive time of the host-buffer entry points: BASELINE config 2, and the reference's own
scene at its window size and settings (1280x800, 16 spp, 4 bounces: src/main.c:17-18, src/renderer.c:8-9)."""
import ctypes as C
import os, sys, time
sys.path.insert(0, '.')
import numpy as np
import path_tracer_c_amd as pt


def direct_calls(s, w, h):
    """render_task called directly, one row per call, top to bottom (no pool)."""
    L, A = pt.lib(), pt._abi
    frame = np.zeros((h, w, 3), dtype=np.uint8)
    args = A.RenderArgs()
    args.framew, args.frameh = w, h
    args.frame = frame.ctypes.data_as(C.POINTER(C.c_uint8))
    args.world = C.pointer(s.c)
    for y in range(h):
        args.screen_y = y
        L.render_task(C.byref(args))
    return frame

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
        pt.host.reset_stats()
        t0 = time.perf_counter(); direct_calls(s, w, h); first = time.perf_counter() - t0
        n_first = pt.host.read_stats().launches
        t = []
        pt.host.reset_stats()
        for _ in range(4):
            t0 = time.perf_counter(); direct_calls(s, w, h); t.append(time.perf_counter() - t0)
        n_later = pt.host.read_stats().launches / 4
        print(f"{label}: render_task x{h} called directly: first frame {first*1e3:.3f} ms in {n_first} launches, "
              f"the same frame again {min(t)*1e3:.3f} ms min in {n_later:g} launches (unchanged scene and settings: "
              f"rows come from the read-ahead cache); PTGPU_LEGACY_READAHEAD={os.environ.get('PTGPU_LEGACY_READAHEAD', '1')}")

c = pt.CONFIGS[2]
measure(c['scene'], c['width'], c['height'], c['spp'], c['max_bounces'], c['seed'], "config 2")
measure(0, 1280, 800, 16, 4, 1, "reference scene 1280x800x16")
