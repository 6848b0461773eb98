"""Plain vs screened linear scan around the 32-object threshold (GPU box)."""
import sys
sys.path.insert(0, '.')
from tools.bvh_scaling import scene, timeit
import path_tracer_c_amd as pt
for n in (8, 16, 24, 32, 48, 64, 128):
    with scene(n) as s, pt.DeviceScene(s) as d:
        w, h = 1920, 1080
        ms_p, rays, a = timeit(d, w, h, 0x90, spp=32)
        ms_s, _, b = timeit(d, w, h, 0xA0, spp=32)
        assert (a == b).all()
        print(f"n={n:4d}: plain {ms_p:8.2f} ms   screened {ms_s:8.2f} ms   ratio plain/screened {ms_p/ms_s:5.2f}", flush=True)
