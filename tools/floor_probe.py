"""How often do the culling scans differ from the plain scan on a real picture of the kind that could provoke it:
a tilted floor of coplanar triangles (diffuse bounces off it start rays that lie almost in the plane of all the
other floor triangles), a few spheres, full path tracing.  Counts differing bytes between hierarchy / screened and
plain scan frames (GPU box)."""
import sys
sys.path.insert(0, '.')
import numpy as np
import path_tracer_c_amd as pt

A = pt._abi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
(w, h), spp = (1920, 1080) if "-hd" in sys.argv else (960, 540), int(sys.argv[2]) if len(sys.argv) > 2 else 64
rng = np.random.default_rng(5)
# rotation: floor normal tilted off every axis, so that no product in the triangle test is exactly zero
ax = np.array([0.3, 0.1, 0.2]); ang = np.linalg.norm(ax); k = ax / ang
K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
R = np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * K @ K
with pt.HostScene() as s:
    s.set_sky(0.5, 0.7, 0.9)
    floor = s.add_material((0.8, 0.8, 0.7), roughness=1.0)
    shiny = s.add_material((0.9, 0.9, 0.9), roughness=0.05)
    size = 40.0 / n
    for i in range(n):
        for j in range(n):
            x0, z0 = -20 + i * size, 2 + j * size
            p = [R @ np.array(v) for v in ((x0, -1.5, z0), (x0 + size, -1.5, z0), (x0, -1.5, z0 + size), (x0 + size, -1.5, z0 + size))]
            s.add_triangle(tuple(p[0]), tuple(p[1]), tuple(p[2]), floor)
            s.add_triangle(tuple(p[1]), tuple(p[3]), tuple(p[2]), floor)
    for _ in range(12):
        c = R @ np.array([rng.uniform(-8, 8), rng.uniform(-1, 2), rng.uniform(5, 25)])
        s.add_sphere(tuple(c), float(rng.uniform(0.4, 1.2)), shiny if rng.integers(2) else floor)
    kw = dict(spp=spp, max_bounces=8, seed=3)
    plain = pt.render_frame(s, w, h, flags=A.FLAG_SCAN_PLAIN, **kw)
    st = pt.host.read_stats()
    for name, f in (("screened", A.FLAG_SCAN_SCREENED), ("hierarchy", A.FLAG_SCAN_BVH)):
        got = pt.render_frame(s, w, h, flags=f, **kw)
        d = got != plain
        print(f"{2 * n * n} coplanar floor triangles, {w}x{h}x{spp} spp: {name} vs plain: {int(d.sum())} of {d.size} bytes differ"
              + (f" (max |diff| {int(np.abs(got.astype(int) - plain.astype(int)).max())})" if d.any() else ""), flush=True)
