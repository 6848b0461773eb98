"""Regenerates tests/golden/*.npz.

PARITY UNPINNED: the reference holds no golden vectors for this path and
cannot be built here (cglm absent), so these fixtures are outputs of THIS
repository's CPU oracle (oracle/pt_oracle.c), committed as regression pins:
they freeze the oracle's behaviour (and the scene builders') so that a change
to either is noticed, and give the GPU tests committed vectors to hit.
Run from the repo root:  python tests/golden/make_golden.py
"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))

import path_tracer_c_amd as pt  # noqa: E402
from oracle import binding as oracle  # noqa: E402

OUT = Path(__file__).resolve().parent

# (name, scene config, width, height, spp, max_bounces, seed)
FRAMES = [
    ("c1_64x64", 1, 64, 64, 16, 4, 12345),
    ("demo_80x50", 0, 80, 50, 16, 4, 12345),
    ("c2_96x54", 2, 96, 54, 8, 8, 1),
    ("c5_48x27", 5, 48, 27, 4, 8, 1),
]


def main():
    for name, cfg, w, h, spp, b, seed in FRAMES:
        with pt.HostScene(cfg) as s:
            frame, rays = oracle.render_rows(s, w, h, spp=spp, max_bounces=b, seed=seed)
        np.savez_compressed(OUT / f"frame_{name}.npz", frame=frame,
                            meta=np.array([cfg, w, h, spp, b, seed, rays], dtype=np.int64))
        print(name, rays, frame.mean())

    rng = np.random.default_rng(2024)
    for cfg in (0, 2):
        n = 512
        o = rng.uniform(-2.5, 2.5, (n, 3)).astype(np.float32)
        d = rng.normal(size=(n, 3)).astype(np.float32)
        d /= np.linalg.norm(d, axis=1, keepdims=True).astype(np.float32)
        with pt.HostScene(cfg) as s:
            dist, point, normal, mat = oracle.trace_rays(s, o, d)
        np.savez_compressed(OUT / f"rays_cfg{cfg}.npz", origins=o, directions=d,
                            distance=dist, point=point, normal=normal,
                            material=mat.astype(np.int64))
        print("rays", cfg, int((dist >= 0).sum()))


if __name__ == "__main__":
    main()
