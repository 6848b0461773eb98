"""BVH scan on triangle meshes: a height field of 2 (n-1)^2 triangles plus a few spheres (GPU box).
    python tools/mesh_scaling.py [n ...]"""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
import path_tracer_c_amd as pt
from tools.bvh_scaling import timeit


def scene(n, seed=5):
    rng = np.random.default_rng(seed)
    xs = np.linspace(-12, 12, n)
    zs = np.linspace(2, 26, n)
    gx, gz = np.meshgrid(xs, zs, indexing="ij")
    hgt = -1.5 + 0.6 * np.sin(gx * 0.9) * np.cos(gz * 0.7) + rng.uniform(-0.05, 0.05, (n, n))
    s = pt.HostScene()
    s.set_sky(0.5, 0.7, 0.9)
    ground = s.add_material((0.5, 0.6, 0.3), roughness=0.9)
    shiny = s.add_material((0.9, 0.9, 0.9), roughness=0.05)
    glass = s.add_material((1, 1, 1), roughness=0.0, transparency=1.0)
    for i in range(n - 1):
        for k in range(n - 1):
            a = (xs[i], hgt[i, k], zs[k]); b = (xs[i + 1], hgt[i + 1, k], zs[k])
            c = (xs[i + 1], hgt[i + 1, k + 1], zs[k + 1]); e = (xs[i], hgt[i, k + 1], zs[k + 1])
            s.add_triangle(a, b, c, ground if (i + k) % 5 else shiny)
            s.add_triangle(a, e, c, ground)
    for _ in range(20):
        s.add_sphere(tuple(rng.uniform(-6, 6, 3) + np.array([0, 1.5, 12])), float(rng.uniform(0.3, 0.9)),
                     [shiny, glass, ground][int(rng.integers(3))])
    return s


def main():
    for n in [int(x) for x in (sys.argv[1:] or [33, 101, 317])]:
        t0 = time.time()
        with scene(n) as s:
            t1 = time.time()
            with pt.DeviceScene(s) as d:
                t2 = time.time()
                w, h, spp = 960, 540, 8
                ms_b, rays, b = timeit(d, w, h, 0x500, spp)
                line = f"{2*(n-1)**2:7d} triangles: bvh {ms_b:8.2f} ms ({rays/ms_b/1e6:6.3f} Gray/s, {rays/(w*h*spp):.2f} rays/sample)"
                if n <= 101:
                    ms_l, _, a = timeit(d, w, h, 0x120, spp)
                    assert (a == b).all()
                    line += f"   linear-screened {ms_l:9.2f} ms   ratio {ms_l/ms_b:6.1f}"
                print(line + f"   (scene {t1-t0:.1f}s, build+upload {t2-t1:.2f}s, nodes/depth {d.hierarchy})", flush=True)


if __name__ == '__main__':
    main()
