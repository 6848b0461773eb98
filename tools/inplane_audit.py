"""How close do real frames come to the one stated limit of the culling scans (DESIGN.md 4.1:
a ray lying in a triangle's plane, where the reference's triangle test is rounding noise)?
Every ray the screened / hierarchy scan answers is also put through the plain scan, and every
triangle the screen culls is asked whether the reference's test would have "hit" it.  Needs
the auditing build:
    make BUILD=build_audit LIBDIR=build_audit/lib EXTRA_HIPFLAGS=-DPT_SCAN_AUDIT build_audit/lib/libptgpu.so
    tools/exp_lib.sh build_audit/lib/libptgpu.so -- python tools/inplane_audit.py"""
import ctypes as C
import sys
sys.path.insert(0, '.')
import numpy as np
import torch
import path_tracer_c_amd as pt

A = pt._abi
fn = pt.lib().ptgpu_debug_scan_audit
fn.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
out = (C.c_ulonglong * 8)()


def floor_scene(n=40):
    """tools/floor_probe.py's scene: a tilted floor of 2 n^2 coplanar triangles and 12 spheres."""
    rng = np.random.default_rng(5)
    ax = np.array([0.3, 0.1, 0.2]); ang = np.linalg.norm(ax); k = ax / ang
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    R = np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * K @ K
    s = pt.HostScene()
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
    return s


def audit(name, host, w, h, spp, bounces=8):
    with host as s, pt.DeviceScene(s) as d:
        st = torch.cuda.current_stream().cuda_stream
        img = torch.zeros((h, w, 3), dtype=torch.uint8, device='cuda')
        for scan_name, flags in (("screened", A.FLAG_SCAN_SCREENED | A.FLAG_PIXEL_OWNER | A.FLAG_NO_PRIMARY_SHARE),
                                 ("hierarchy", A.FLAG_SCAN_BVH | A.FLAG_SAMPLE_QUEUE | A.FLAG_NO_PRIMARY_SHARE)):
            if scan_name == "hierarchy" and sum(s.counts()) < 64:
                continue    # (no hierarchy is built below 64 objects)
            fn(out, 1)
            p = pt.make_params(spp=spp, max_bounces=bounces, seed=3, flags=flags)
            d.render(img.data_ptr(), w, h, p, st); torch.cuda.synchronize()
            fn(out, 0)
            v = [int(x) for x in out]
            print(f"{name} ({s.counts()[1]} triangles, {w}x{h}x{spp} spp), {scan_name} scan: {v[0]} rays audited, "
                  f"{v[1]} answered differently from the plain scan | screen culled {v[2]} (ray, triangle) pairs: "
                  f"{v[3]} with the ray within 1e-3 rad of the plane and 3 radii, {v[4]} where the reference's test "
                  f"reports a hit, {v[5]} of them nearer than the accepted hit", flush=True)


if __name__ == '__main__':
    from tools.mesh_scaling import scene as mesh
    audit("coplanar floor", floor_scene(40), 480, 270, 16)
    audit("height field", mesh(33), 480, 270, 16)
    audit("height field", mesh(101), 480, 270, 8)
    audit("reference's scene", pt.HostScene(0), 640, 400, 16, bounces=4)
