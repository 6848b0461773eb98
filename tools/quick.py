import sys
sys.path.insert(0, '.')
from tools.sweep_settle import run
run('c5', spp=32, flags=0x500)
run('c5', spp=32, flags=0x500)
