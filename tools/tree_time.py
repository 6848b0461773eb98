"""Branching kernel (materials with 0 < transparency < 1: both children of a hit are traced)
against the path kernel on the same geometry (GPU box)."""
import sys
sys.path.insert(0, '.')
import numpy as np, torch
import path_tracer_c_amd as pt
from tools.bvh_scaling import timeit


def timeit_b(d, bounces):
    """as timeit, with a chosen bounce limit"""
    out = torch.zeros((540, 960, 3), dtype=torch.uint8, device='cuda')
    p = pt.make_params(spp=32, max_bounces=bounces, seed=1)
    st = torch.cuda.current_stream().cuda_stream
    d.render(out.data_ptr(), 960, 540, p, st); torch.cuda.synchronize()
    pt.host.reset_stats()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); d.render(out.data_ptr(), 960, 540, p, st); e1.record(); torch.cuda.synchronize()
    stt = pt.host.read_stats()
    timeit.lane_use = stt.rays / (64.0 * stt.wave_iters) if stt.wave_iters else 0.0
    return e0.elapsed_time(e1), stt.rays, None


def scene(n, branching, seed=1):
    rng = np.random.default_rng(seed)
    s = pt.HostScene()
    s.set_sky(135/255, 206/255, 235/255)
    mats = [s.add_material(tuple(rng.uniform(0.1, 0.95, 3)), roughness=float(rng.uniform(0.05, 1))) for _ in range(10)]
    mats.append(s.add_material((0.9, 0.9, 0.9), roughness=0.0))
    mats.append(s.add_material((1, 1, 1), roughness=0.0, transparency=0.5 if branching else 1.0))
    mats.append(s.add_material((0.8, 0.9, 1), roughness=0.1, transparency=0.3 if branching else 0.0))
    k = (n / 1000.0) ** (1 / 3)
    c = np.stack([rng.uniform(-20*k, 20*k, n), rng.uniform(-1, 10*k, n), rng.uniform(3, 60*k, n)], 1)
    r = rng.uniform(0.2, 0.8, n)
    m = rng.integers(0, len(mats), n)
    for i in range(n):
        s.add_sphere(tuple(c[i]), float(r[i]), int(m[i]))
    s.add_sphere((0, -1001.8, 30), 1000.0, mats[0])
    return s


def main():
  for n in (8, 100, 1000):
    for branching in (False, True):
      with scene(n, branching) as s, pt.DeviceScene(s) as d:
            ms, rays, _ = timeit(d, 960, 540, 0, spp=32) if len(sys.argv) < 2 else timeit_b(d, int(sys.argv[1]))
            print(f"n={n:5d} {'branching' if branching else 'paths    '}: {ms:8.2f} ms  {rays/ms/1e6:7.3f} Gray/s  "
                  f"{rays/(960*540*32):6.2f} rays/sample  lane use {timeit.lane_use:.3f}  schedule {d.schedule}, scan {d.scan}", flush=True)


if __name__ == '__main__':
    main()
