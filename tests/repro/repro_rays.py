"""Replays one trial of tests/test_parity_gpu.py::test_randomised_rays_through_randomised_scenes_in_every_scan_form
(python tests/repro/repro_rays.py <PT_FUZZ_SEED> <trial>) and prints the rays on which a scan form differs from the oracle."""
import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import path_tracer_c_amd as gpu
from oracle import binding as oracle
import importlib.util
spec = importlib.util.spec_from_file_location("tp", "tests/test_parity_gpu.py"); tp = importlib.util.module_from_spec(spec); spec.loader.exec_module(tp)
f32 = np.float32
A = gpu._abi
seed0, target = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed0 + 31)
n = 600
for trial in range(target + 1):
    s = tp._random_scene(gpu, rng, branching=False)
    scale = float(10 ** rng.uniform(-3, 4))
    o = (rng.normal(0, 4, (n, 3)) * scale + np.array([0, 0, 4 * scale])).astype(f32)
    o[: n // 4] = 0
    d = rng.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True); d = d.astype(f32)
    d[::7] *= f32(rng.uniform(0.5, 3))
    kinds = rng.integers(0, 40, n)
    d[kinds == 0] = 0; d[kinds == 1, 0] = np.nan; o[kinds == 2, 1] = np.inf; d[kinds == 3] *= f32(1e30); o[kinds == 4] *= f32(1e20)
    if trial != target:
        s.__exit__(None, None, None)
        continue
    want = oracle.trace_rays(s, o, d)
    print("trial", trial, "scale", scale, "objects", s.counts())
    for name, f in (("plain", A.FLAG_SCAN_PLAIN), ("screened", A.FLAG_SCAN_SCREENED), ("bvh", A.FLAG_SCAN_BVH)):
        if f == A.FLAG_SCAN_BVH and s.num_objects < 64:
            continue
        got = gpu.trace_rays(s, o, d, f)
        bad = np.flatnonzero((got[0].view(np.uint32) != want[0].view(np.uint32)) & ~(np.isnan(got[0]) & np.isnan(want[0])))
        print(name, "differs on", len(bad), "rays")
        for j in bad[:8]:
            print("  ", j, "kind", kinds[j], "every7th", j % 7 == 0, "o", o[j], "d", d[j], "|d|", np.linalg.norm(d[j].astype(np.float64)),
                  "oracle", want[0][j], want[3][j], "gpu", got[0][j], got[3][j])
    # which object does the oracle hit?  one-object scenes
    import json
