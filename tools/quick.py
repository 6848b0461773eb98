import sys
sys.path.insert(0, '.')
from tools.sweep import run
for name, spp in (('c2', 256), ('c5', 32)):
    for flags in (0, 2):
        run(name, spp=spp, flags=flags, block=256 if flags == 0 else 64)
