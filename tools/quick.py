import sys
sys.path.insert(0, '.')
from tools.sweep_settle import run
for name, kw in (('c5', dict(spp=64)), ('c2', {}), ('c1', dict(w=1920, h=1080, spp=256)), ('c0', dict())):
    ref = None
    for flags in (0x80, 0x100, 0):
        img = run(name, flags=flags, **kw)
        if ref is None: ref = img
        assert (img == ref).all()
