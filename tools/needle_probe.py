"""Differential probe of the culling scans on thin triangles (GPU box): rays aimed at the extended edge
lines of triangles whose edges enclose a small angle -- where the reference's triangle test turns into
rounding noise -- traced with the plain scan (no culling), the screened scan and the hierarchy.  Prints
mismatches per angle class; classes below the never-culled threshold (1e-2, csrc/ptgpu_abi.hip) cannot
mismatch by construction, the ones above show how much margin the threshold has."""
import sys
sys.path.insert(0, '.')
import numpy as np
import path_tracer_c_amd as pt

A = pt._abi
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
n_rays = 1 << 18
REPS = int(sys.argv[3]) if len(sys.argv) > 3 else 6
for sin_angle in [float(x) for x in sys.argv[2].split(',')] if len(sys.argv) > 2 else (3e-1, 1e-1, 5e-2, 3e-2, 2e-2, 1.5e-2, 1.1e-2, 5e-3, 1e-5):
    bad = {"screened": 0, "bvh": 0}
    hits_beyond = 0
    for rep in range(REPS):
        scale = 10 ** rng.uniform(-1, 3)
        with pt.HostScene() as s:
            s.set_sky(0.1, 0.2, 0.3)
            m = s.add_material((0.5, 0.5, 0.5))
            tris = []
            for _ in range(70):                       # >= 64 objects: a hierarchy is built
                c = rng.normal(0, 5, 3) * scale
                e1 = rng.normal(0, 1, 3) * scale
                side = np.cross(e1, rng.normal(0, 1, 3))
                side *= np.linalg.norm(e1) / np.linalg.norm(side) * sin_angle
                e2 = e1 + side                          # tan(angle) = sin_angle exactly
                s.add_triangle(tuple(c), tuple(c + e1), tuple(c + e2), m)
                tris.append((c, e1, e2))
            s.add_sphere((0, 0, 0), 60.0 * scale, m)   # something behind everything
            k = rng.integers(0, len(tris), n_rays)
            c = np.array([tris[i][0] for i in k]); e1 = np.array([tris[i][1] for i in k]); e2 = np.array([tris[i][2] for i in k])
            lam = rng.uniform(-3, 4, (n_rays, 1))
            edge = np.where(rng.integers(0, 2, (n_rays, 1)) == 0, e1, e2)
            target = c + lam * edge + rng.normal(0, 1, (n_rays, 3)) * scale * 10 ** rng.uniform(-8, -2, (n_rays, 1))
            origin = target + rng.normal(0, 1, (n_rays, 3)) * scale * 10 ** rng.uniform(0, 2, (n_rays, 1))
            d = target - origin
            d /= np.linalg.norm(d, axis=1, keepdims=True)
            if "-inplane" in sys.argv:
                # rays lying (almost) in the aimed triangle's plane: direction = an in-plane direction tilted out of
                # the plane by 1e-7 .. 1e-2 rad, passing the triangle within an edge length or so
                nrm = np.cross(e1, e2); nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
                inp = np.cross(nrm, rng.normal(0, 1, (n_rays, 3))); inp /= np.linalg.norm(inp, axis=1, keepdims=True)
                tilt = 10 ** rng.uniform(-7, -2, (n_rays, 1)) * rng.choice([-1, 1], (n_rays, 1))
                d = inp + nrm * tilt
                d /= np.linalg.norm(d, axis=1, keepdims=True)
                through = c + e1 * rng.uniform(-1, 2, (n_rays, 1)) + e2 * rng.uniform(-1, 2, (n_rays, 1))
                origin = through - d * np.linalg.norm(e1, axis=1, keepdims=True) * 10 ** rng.uniform(0, 2, (n_rays, 1))
            o32 = origin.astype(np.float32); d32 = d.astype(np.float32)
            plain = pt.trace_rays(s, o32, d32, A.FLAG_SCAN_PLAIN)
            for name, f in (("screened", A.FLAG_SCAN_SCREENED), ("bvh", A.FLAG_SCAN_BVH)):
                got = pt.trace_rays(s, o32, d32, f)
                diff = (got[0].view(np.uint32) != plain[0].view(np.uint32)) | (got[3] != plain[3])
                bad[name] += int(diff.sum())
                if name == "screened" and diff.any() and "-v" in sys.argv:
                    for j in np.flatnonzero(diff)[:3]:
                        obj_hit = None
                        # which triangle did the plain scan hit?  (the one this ray was aimed at, most likely)
                        cc, a1, a2 = tris[k[j]]
                        hp = plain[1][j].astype(np.float64)
                        # barycentric coordinates of the plain scan's hit point in the aimed triangle (least squares)
                        M = np.stack([a1, a2], 1)
                        uv, res, *_ = np.linalg.lstsq(M, hp - cc, rcond=None)
                        nrm = np.cross(a1, a2); nrm /= np.linalg.norm(nrm)
                        print(f"   ray {j}: plain dist {plain[0][j]:.6g} vs {name} {got[0][j]:.6g}; hit point in the aimed triangle's frame: u {uv[0]:.4f} v {uv[1]:.4f}, off-plane {abs((hp - cc) @ nrm) / np.linalg.norm(a1):.2e} edges; "
                              f"|origin - v0| / |e1| {np.linalg.norm(origin[j] - cc) / np.linalg.norm(a1):.1f}, cos(ray, normal) {abs(d[j] @ nrm):.2e}, lambda aimed {lam[j, 0]:.3f}", flush=True)
            # how many of the plain scan's triangle hits lie beyond the triangle they hit (noise hits)?
            hit_tri = plain[0] > 0
            hits_beyond += int(hit_tri.sum())
    print(f"sin(angle) {sin_angle:8.1e}: {REPS * n_rays} rays, mismatches vs the plain scan: screened {bad['screened']}, hierarchy {bad['bvh']}", flush=True)
