"""Replays one trial of tests/test_parity_gpu.py::test_fast_math_randomised_scenes_are_the_same_picture_in_every_scan_form
(python tests/repro/repro_fast.py <PT_FUZZ_SEED> <trial>): renders it in every scan form and schedule, exact and tolerance
build, against the oracle, and dumps the scene to gpurun_out/repro_scene.json (tools/scene_json.py reads it back).
How tests/golden/scenes/screen_hole_r2.json was obtained."""
import os, sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import path_tracer_c_amd as gpu
import importlib.util
spec = importlib.util.spec_from_file_location("tp", "tests/test_parity_gpu.py"); tp = importlib.util.module_from_spec(spec); spec.loader.exec_module(tp)
A = gpu._abi
seed0, target = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed0 + 99)
for trial in range(target + 1):
    s = tp._random_scene(gpu, rng, branching=False)
    w, h = int(rng.integers(8, 60)), int(rng.integers(5, 36))
    kw = dict(spp=int(rng.integers(1, 4)), max_bounces=int(rng.integers(0, 9)), seed=trial)
    if trial != target:
        s.close() if hasattr(s, 'close') else s.__exit__(None, None, None)
        continue
    print("trial", trial, w, h, kw, s.counts())
    frames = {}
    for name, f in [("plain/owner", A.FLAG_SCAN_PLAIN | A.FLAG_PIXEL_OWNER), ("screened/owner", A.FLAG_SCAN_SCREENED | A.FLAG_PIXEL_OWNER),
                    ("plain/queue", A.FLAG_SCAN_PLAIN | A.FLAG_SAMPLE_QUEUE), ("screened/queue", A.FLAG_SCAN_SCREENED | A.FLAG_SAMPLE_QUEUE),
                    ("bvh/queue", A.FLAG_SCAN_BVH | A.FLAG_SAMPLE_QUEUE), ("bvh/owner", A.FLAG_SCAN_BVH | A.FLAG_PIXEL_OWNER)]:
        try:
            frames[name] = gpu.render_frame(s, w, h, flags=A.FLAG_FAST_MATH | f, **kw)
        except Exception as e:
            print(name, "error", e)
    ref = frames["plain/owner"]
    for name, fr in frames.items():
        d = np.argwhere((fr != ref).any(axis=2))
        print(name, "differs at", d.tolist()[:10], [(ref[y, x].tolist(), fr[y, x].tolist()) for y, x in d[:4]])
    exact = gpu.render_frame(s, w, h, flags=A.FLAG_SCAN_PLAIN | A.FLAG_PIXEL_OWNER, **kw)
    exact_s = gpu.render_frame(s, w, h, flags=A.FLAG_SCAN_SCREENED | A.FLAG_PIXEL_OWNER, **kw)
    print("exact plain == exact screened:", np.array_equal(exact, exact_s))
    d = np.argwhere((exact != ref).any(axis=2)); print("fast plain vs exact plain differs at", len(d), "pixels")
    from oracle import binding as oracle
    oref, _ = oracle.render_rows(s, w, h, **kw)
    for name, f in [("plain", A.FLAG_SCAN_PLAIN), ("screened", A.FLAG_SCAN_SCREENED), ("bvh", A.FLAG_SCAN_BVH)]:
        for sched, g in [("owner", A.FLAG_PIXEL_OWNER), ("queue", A.FLAG_SAMPLE_QUEUE)]:
            fr = gpu.render_frame(s, w, h, flags=f | g, **kw)
            d = np.argwhere((fr != oref).any(axis=2))
            print("EXACT", name, sched, "vs oracle differs at", d.tolist()[:6], [(oref[y, x].tolist(), fr[y, x].tolist()) for y, x in d[:3]])
    # dump the scene
    import json
    objs = []
    for i in range(s.c.num_objects):
        o = s.c.objects[i]
        if o.tag == 0:
            objs.append(dict(k="s", c=[o.sphere.center.x, o.sphere.center.y, o.sphere.center.z], r=o.sphere.radius, m=int(o.material)))
        else:
            t = o.triangle
            objs.append(dict(k="t", v=[[t.v0.x, t.v0.y, t.v0.z], [t.v1.x, t.v1.y, t.v1.z], [t.v2.x, t.v2.y, t.v2.z]], m=int(o.material)))
    mats = []
    for i in range(s.c.num_materials):
        m = s.c.materials[i]
        mats.append(dict(albedo=[m.albedo.x, m.albedo.y, m.albedo.z], roughness=m.roughness, em=[m.emission_color.x, m.emission_color.y, m.emission_color.z],
                         es=m.emission_strength, t=m.transparency, ior=m.refractive_index))
    json.dump(dict(w=w, h=h, kw=kw, sky=[s.c.sky_color.x, s.c.sky_color.y, s.c.sky_color.z], objs=objs, mats=mats), open("gpurun_out/repro_scene.json", "w"))
    print("scene dumped")
