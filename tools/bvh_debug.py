import sys
sys.path.insert(0, '.')
import numpy as np
import path_tracer_c_amd as pt
from oracle import binding as ob
from tools.bvh_scaling import scene

n = int(sys.argv[1]) if len(sys.argv) > 1 else 250
rng = np.random.default_rng(3)
with scene(n) as s:
    # rays the renderer would cast: from points on object surfaces (+0.001 offset) and from the eye
    N = 2_000_000
    # first generation from the eye
    d0 = rng.normal(size=(N, 3)).astype(np.float32); d0[:, 2] = np.abs(d0[:, 2]) + 0.3
    d0 /= np.linalg.norm(d0, axis=1, keepdims=True).astype(np.float32)
    o0 = np.zeros((N, 3), np.float32)
    dist, point, normal, mat = pt.trace_rays(s, o0, d0)
    hit = dist >= 0
    # second generation: leave the hit points in random directions
    d1 = rng.normal(size=(hit.sum(), 3)).astype(np.float32)
    d1 /= np.linalg.norm(d1, axis=1, keepdims=True).astype(np.float32)
    o1 = (point[hit] + d1 * np.float32(0.001)).astype(np.float32)
    for name, (o, d) in {"eye": (o0, d0), "bounce": (o1, d1)}.items():
        p = pt.trace_rays(s, o, d)
        b = pt.trace_rays(s, o, d, pt._abi.FLAG_SCAN_BVH)
        sc = pt.trace_rays(s, o, d, pt._abi.FLAG_SCAN_SCREENED)
        bad_b = np.flatnonzero((p[0].view(np.uint32) != b[0].view(np.uint32)) | ((p[0] >= 0) & (p[3] != b[3])))
        bad_s = np.flatnonzero((p[0].view(np.uint32) != sc[0].view(np.uint32)) | ((p[0] >= 0) & (p[3] != sc[3])))
        print(name, len(o), "rays: bvh mismatches", len(bad_b), "screened mismatches", len(bad_s))
        for k in bad_b[:5]:
            cd = ob.trace_rays(s, o[k:k+1], d[k:k+1])
            print("   ray", k, "o", o[k], "d", d[k], "plain", p[0][k], p[3][k], "bvh", b[0][k], b[3][k], "oracle", cd[0][0], cd[3][0])
