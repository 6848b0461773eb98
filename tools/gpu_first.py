"""First-contact GPU script: parity vs oracle on small frames + rough timing."""
import sys, time
sys.path.insert(0, '.')
import numpy as np
import path_tracer_c_amd as pt
from oracle import binding as ob

L = pt.lib()
print("devices", L.ptgpu_device_count())

def cmp(cfg, w, h, spp, b, seed=3, **kw):
    with pt.HostScene(cfg) as s:
        t = time.time(); g = pt.render_frame(s, w, h, spp=spp, max_bounces=b, seed=seed, **kw); tg = time.time() - t
        t = time.time(); c, rays = ob.render_rows(s, w, h, spp=spp, max_bounces=b, seed=seed); tc = time.time() - t
    diff = (g != c)
    print(f"cfg{cfg} {w}x{h} spp{spp} b{b} {kw}: bytes differing {int(diff.sum())}/{diff.size} maxabs {int(np.abs(g.astype(int)-c).max())}  gpu {tg:.3f}s cpu {tc:.3f}s rays {rays}")
    return g, c

for cfg, (w, h, spp, b) in {1: (256, 256, 16, 4), 0: (160, 100, 16, 4), 2: (192, 108, 32, 8), 5: (96, 54, 8, 8)}.items():
    cmp(cfg, w, h, spp, b)
    cmp(cfg, w, h, spp, b, flags=pt._abi.FLAG_SCENE_GLOBAL)
    if b <= 14:
        cmp(cfg, w, h, spp, b, flags=pt._abi.FLAG_FORCE_TREE)

# odd width (byte-store fallback) + row subset
cmp(1, 101, 37, 4, 4)

# trace_rays parity
rng = np.random.default_rng(0)
for cfg in (0, 1, 2, 5):
    with pt.HostScene(cfg) as s:
        n = 20000
        o = rng.uniform(-3, 3, (n, 3)).astype(np.float32)
        d = rng.normal(size=(n, 3)).astype(np.float32); d /= np.linalg.norm(d, axis=1, keepdims=True)
        gd, gp, gn, gm = pt.trace_rays(s, o, d)
        cd, cp, cn, cm = ob.trace_rays(s, o, d)
        hit = cd >= 0
        print(f"trace cfg{cfg}: hits {int(hit.sum())} dist-eq {np.array_equal(gd.view(np.uint32), cd.view(np.uint32))} "
              f"point-eq {np.array_equal(gp[hit].view(np.uint32), cp[hit].view(np.uint32))} normal-eq {np.array_equal(gn[hit].view(np.uint32), cn[hit].view(np.uint32))} mat-eq {np.array_equal(gm[hit], cm[hit])}")

# timing C2 full size
for cfg in (2, 5):
    c = pt.CONFIGS[cfg]
    with pt.HostScene(c['scene']) as s:
        for blk in (256,):
            pt.host.reset_stats()
            t = time.time()
            g = pt.render_frame(s, c['width'], c['height'], spp=c['spp'], max_bounces=c['max_bounces'], seed=c['seed'], block_threads=blk)
            dt = time.time() - t
            st = pt.host.read_stats()
            print(f"C{cfg} full block{blk}: {dt:.3f}s rays {st.rays} => {st.rays/dt/1e6:.1f} Mray/s, rays/sample {st.rays/st.samples:.3f}, lane util {st.rays/(64*st.wave_iters):.3f}")
