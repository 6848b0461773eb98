"""Diagnostic: on configuration 2, how often could a whole WAVE skip the sqrt + divide half of a
sphere test because every lane's root already lies beyond its best hit (the far-reject of VERDICT
r1 #3)?  Needs the counting build:
make -B EXTRA_HIPFLAGS=-DPT_FARREJECT_COUNT path_tracer_c_amd/lib/libptgpu.so  (rebuild plain afterwards)."""
import ctypes as C
import sys
sys.path.insert(0, '.')
import torch
import path_tracer_c_amd as pt

fn = pt.lib().ptgpu_debug_farreject
fn.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
out = (C.c_ulonglong * 4)()
for cfg, w, h, spp, lanes in ((2, 960, 540, 32, 8), (2, 960, 540, 32, 1), (0, 1280, 800, 16, 1), (1, 512, 512, 16, 1)):
    with pt.HostScene(cfg) as s, pt.DeviceScene(s) as d:
        img = torch.zeros((h, w, 3), dtype=torch.uint8, device='cuda')
        p = pt.make_params(spp=spp, max_bounces=8, seed=1, flags=0x290, lanes_per_pixel=lanes)   # owner, plain scan, every ray scanned
        fn(out, 1)
        d.render(img.data_ptr(), w, h, p, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        fn(out, 0)
        waves, waves_far, lanes_has, lanes_far = [float(out[i]) for i in range(4)]
        print(f"scene {cfg}, {lanes} lane(s) per pixel: sphere tests with a real discriminant in some lane: {waves:.3g} per wave; "
              f"all of those lanes beyond their best hit (the wave could skip sqrt + divide): {waves_far / waves:.4f}; "
              f"per lane: {lanes_far / lanes_has:.4f} of the tests with a real discriminant are far or behind")
