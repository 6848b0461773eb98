import sys
sys.path.insert(0, '.')
from tools.sweep_settle import run
for flags in (0x10, 0x20, 0x12, 0x22):
    run('c5', spp=32, flags=flags, block=64 if flags & 2 else 256)
    run('c5', spp=32, flags=flags, block=256 if flags & 2 else 512)
for flags in (0x10, 0x20, 0x12, 0x22):
    run('c2', flags=flags, block=64 if flags & 2 else 256)
