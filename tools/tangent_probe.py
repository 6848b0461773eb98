"""Differential probe of the culling scans on spheres (GPU box): rays aimed to graze spheres -- tangent to within
1e-8 .. 1e-2 of the radius -- from inside, outside and from the surface, radii from 0 to huge, traced with the
plain scan (no culling), the screened scan and the hierarchy.  Any mismatch would be a hole in screen_keep or
in the node boxes (the sphere test's own formulation keeps its error ~1e-7 |dp_perp|^2, 250x inside the
screen's margin)."""
import sys
sys.path.insert(0, '.')
import numpy as np
import path_tracer_c_amd as pt

A = pt._abi
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
n_rays = 1 << 18
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
total = {"screened": 0, "bvh": 0}
for rep in range(reps):
    scale = 10 ** rng.uniform(-2, 3)
    with pt.HostScene() as s:
        s.set_sky(0.1, 0.2, 0.3)
        m = s.add_material((0.5, 0.5, 0.5))
        sph = []
        for _ in range(80):
            c = rng.normal(0, 5, 3) * scale
            r = float(rng.choice([0.0, 1e-3, 0.3, 1.0, 3.0, 50.0, 1000.0, -0.7])) * scale * rng.uniform(0.5, 2)
            s.add_sphere(tuple(c), r, m)
            sph.append((c, abs(r)))
        k = rng.integers(0, len(sph), n_rays)
        c = np.array([sph[i][0] for i in k]); r = np.array([sph[i][1] for i in k])[:, None]
        # a point on the sphere, a tangent direction there, an origin somewhere along the tangent line
        nrm = rng.normal(0, 1, (n_rays, 3)); nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
        tan = np.cross(nrm, rng.normal(0, 1, (n_rays, 3))); tan /= np.linalg.norm(tan, axis=1, keepdims=True)
        off = r * (1 + rng.choice([-1, 1], (n_rays, 1)) * 10 ** rng.uniform(-8, -2, (n_rays, 1)))   # just inside / outside
        touch = c + nrm * off
        back = 10 ** rng.uniform(-3, 3, (n_rays, 1)) * np.maximum(r, scale * 1e-3)
        origin = touch - tan * back
        kind = rng.integers(0, 3, (n_rays, 1))
        origin = np.where(kind == 0, origin, np.where(kind == 1, touch, c + nrm * r * rng.uniform(0, 1, (n_rays, 1))))  # outside / on the surface / inside
        d = tan + rng.normal(0, 1, (n_rays, 3)) * 10 ** rng.uniform(-9, -3, (n_rays, 1))
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        o32 = origin.astype(np.float32); d32 = d.astype(np.float32)
        plain = pt.trace_rays(s, o32, d32, A.FLAG_SCAN_PLAIN)
        for name, f in (("screened", A.FLAG_SCAN_SCREENED), ("bvh", A.FLAG_SCAN_BVH)):
            got = pt.trace_rays(s, o32, d32, f)
            diff = (got[0].view(np.uint32) != plain[0].view(np.uint32)) | (got[3] != plain[3])
            total[name] += int(diff.sum())
            if diff.any():
                j = int(np.flatnonzero(diff)[0])
                print(f"  rep {rep}: {name} differs, e.g. ray {j}: plain {plain[0][j]} vs {got[0][j]}, sphere r {r[j, 0]:.4g}, scale {scale:.3g}", flush=True)
    print(f"rep {rep}: scale {scale:.3g}, {n_rays} rays, hits {(plain[0] > 0).mean():.2f}", flush=True)
print(f"{reps * n_rays} grazing rays: mismatches vs the plain scan: screened {total['screened']}, hierarchy {total['bvh']}")
