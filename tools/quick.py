import sys
sys.path.insert(0, '.')
from tools.sweep_settle import run
for i in range(2):
    run('c2', flags=0x80)
run('c0', flags=0x80)
run('c1', w=1920, h=1080, spp=256, flags=0x80)
