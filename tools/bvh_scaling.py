"""Screened linear scan vs BVH scan as the object count grows (GPU box)."""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
import path_tracer_c_amd as pt

def scene(n, seed=1):
    rng = np.random.default_rng(seed)
    s = pt.HostScene()
    s.set_sky(135/255, 206/255, 235/255)
    mats = [s.add_material(tuple(rng.uniform(0.1, 0.95, 3)), roughness=float(rng.uniform(0.05, 1))) for _ in range(12)]
    mats.append(s.add_material((0.9, 0.9, 0.9), roughness=0.0))
    mats.append(s.add_material((1, 1, 1), roughness=0.0, transparency=1.0))
    k = (n / 1000.0) ** (1 / 3)           # same density as config 5
    c = np.stack([rng.uniform(-20*k, 20*k, n), rng.uniform(-1, 10*k, n), rng.uniform(3, 60*k, n)], 1)
    r = rng.uniform(0.2, 0.8, n)
    m = rng.integers(0, len(mats), n)
    for i in range(n):
        s.add_sphere(tuple(c[i]), float(r[i]), int(m[i]))
    s.add_sphere((0, -1001.8, 30), 1000.0, mats[0])
    return s

def timeit(d, w, h, flags, spp=32):
    out = torch.zeros((h, w, 3), dtype=torch.uint8, device='cuda')
    p = pt.make_params(spp=spp, max_bounces=8, seed=1, flags=flags)
    st = torch.cuda.current_stream().cuda_stream
    d.render(out.data_ptr(), w, h, p, st); torch.cuda.synchronize()
    pt.host.reset_stats()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); d.render(out.data_ptr(), w, h, p, st); e1.record(); torch.cuda.synchronize()
    stt = pt.host.read_stats()
    timeit.lane_use = stt.rays / (64.0 * stt.wave_iters) if stt.wave_iters else 0.0
    return e0.elapsed_time(e1), stt.rays, out.cpu().numpy()

def main():
  for n in [int(x) for x in (sys.argv[1:] or [64, 96, 128, 250, 1000, 4000, 16000, 64000, 256000])]:
      t0 = time.time()
      with scene(n) as s, pt.DeviceScene(s) as d:
          tb = time.time() - t0
          w, h = 960, 540
          spp = 32 if n <= 16000 else 8
          ms_l, rays, a = timeit(d, w, h, 0x120, spp) if n <= 16000 else (float('nan'), 0, None)
          ms_b, rays, b = timeit(d, w, h, 0x500, spp)
          ms_w, _, c = timeit(d, w, h, 0xD00, spp)
          assert a is None or (a == b).all()
          assert (b == c).all()
          print(f"n={n:6d}: linear-screened {ms_l:9.2f} ms ({rays/ms_l/1e6:7.3f} Gray/s)   bvh whole walks {ms_w:9.2f} ms   "
                f"bvh resumable walks {ms_b:9.2f} ms ({rays/ms_b/1e6:7.3f} Gray/s)   linear/bvh {ms_l/ms_b:5.2f}   (build+upload {tb:.2f}s, nodes/depth {d.hierarchy})", flush=True)

if __name__ == '__main__':
    main()
