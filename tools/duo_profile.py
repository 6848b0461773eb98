"""Lane use of the walk stages on configuration 5, stream kernel (one path per lane) against
pt_duokernel (two): needs the counting build
    make BUILD=build_prof LIBDIR=build_prof/lib EXTRA_HIPFLAGS=-DPT_BVH_PROFILE build_prof/lib/libptgpu.so
    tools/exp_lib.sh build_prof/lib/libptgpu.so -- python tools/duo_profile.py [spp]"""
import ctypes as C
import os
import sys
sys.path.insert(0, '.')
import torch
import path_tracer_c_amd as pt

fn = pt.lib().ptgpu_debug_bvh_profile
fn.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
out = (C.c_ulonglong * 12)()
c = pt.CONFIGS[5]
w, h = c['width'], c['height']
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 16
with pt.HostScene(c['scene']) as s, pt.DeviceScene(s) as d:
    st = torch.cuda.current_stream().cuda_stream
    img = torch.zeros((h, w, 3), dtype=torch.uint8, device='cuda')
    for flags in (0x700, 0x500):
        for duo in ('0', '1'):
            os.environ['PTGPU_DUO'] = duo
            fn(out, 1)
            pt.host.reset_stats()
            p = pt.make_params(spp=spp, max_bounces=c['max_bounces'], seed=1, flags=flags)
            d.render(img.data_ptr(), w, h, p, st); torch.cuda.synchronize()
            fn(out, 0)
            v = [float(x) for x in out]
            r = pt.host.read_stats().rays
            print(f"flags {flags:#x} {'two paths/lane' if duo == '1' else 'one path/lane '}: per ray inner {v[0]/r:5.2f} leaves {v[2]/r:4.2f} exact {v[4]/r:4.2f} | "
                  f"wave-passes per 64 rays (lane use): inner {64*v[1]/r:6.2f} ({v[0]/(64*v[1]):.3f})  leaf {64*v[3]/r:5.2f} ({v[2]/(64*v[3]):.3f})  "
                  f"exact {64*v[5]/r:5.2f} ({v[4]/(64*v[5]):.3f})  shade {64*v[7]/r:5.2f} ({v[6]/(64*v[7]):.3f})", flush=True)
