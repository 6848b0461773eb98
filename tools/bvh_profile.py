"""Diagnostic: SIMD efficiency of the three loops of the BVH scan (inner-node
steps, leaf passes, exact-test resolve passes) on configuration 5.
Needs a profiling build:  make -B EXTRA_HIPFLAGS=-DPT_BVH_PROFILE path_tracer_c_amd/lib/libptgpu.so
(rebuild without the flag afterwards; never shipped)."""
import ctypes as C
import sys
sys.path.insert(0, '.')
import torch
import path_tracer_c_amd as pt

lib = pt.lib()
fn = lib.ptgpu_debug_bvh_profile
fn.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
out = (C.c_ulonglong * 12)()
c = pt.CONFIGS[5]
w, h, spp = c['width'], c['height'], 16
if len(sys.argv) > 1:                       # a bvh_scaling scene of n objects instead
    from tools.bvh_scaling import scene
    host, w, h, spp = scene(int(sys.argv[1])), 960, 540, 4
else:
    host = pt.HostScene(c['scene'])
with host as s, pt.DeviceScene(s) as d:
    st = torch.cuda.current_stream().cuda_stream
    img = torch.zeros((h, w, 3), dtype=torch.uint8, device='cuda')
    for name, flags, k in (("queue whole", 0xD00, 1),
                           ("queue stream", 0x500, 1)):
        fn(out, 1)
        pt.host.reset_stats()
        p = pt.make_params(spp=spp, max_bounces=c['max_bounces'], seed=1, flags=flags, lanes_per_pixel=k)
        d.render(img.data_ptr(), w, h, p, st); torch.cuda.synchronize()
        fn(out, 0)
        v = [float(x) for x in out]
        stt = pt.host.read_stats()
        calls = v[7]
        if name == "queue stream":      # slots 6/7 count service stages there
            print(f"{name:12s} per ray: inner {v[0]/stt.rays:5.1f} lane-steps, leaves {v[2]/stt.rays:4.2f}, exact tests {v[4]/stt.rays:4.2f} | "
                  f"wave-passes per 64 rays: inner {64*v[1]/stt.rays:6.1f} (lane use {v[0]/(64*v[1]):.3f}), "
                  f"leaf {64*v[3]/stt.rays:5.1f} ({v[2]/(64*v[3]):.3f}), resolve {64*v[5]/stt.rays:5.1f} ({v[4]/(64*v[5]):.3f}), "
                  f"service {64*v[7]/stt.rays:5.1f} ({v[6]/(64*v[7]):.3f}) | second walks {int(v[8])}, poisoned rays {int(v[9])} of {stt.rays}", flush=True)
            continue
        print(f"{name:12s} scans/wave-call: lanes {v[6]/calls:5.1f} | inner steps: {v[1]/calls:6.1f} wave-passes, "
              f"lane use {v[0]/(64*v[1]):.3f} | leaf passes {v[3]/calls:5.1f}, lane use {v[2]/(64*v[3]):.3f} | "
              f"resolve passes {v[5]/calls:5.1f}, lane use {v[4]/(64*v[5]):.3f}", flush=True)
