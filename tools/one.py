import sys
sys.path.insert(0, '.')
from tools.sweep_settle import run
name = sys.argv[1]; flags = int(sys.argv[2], 0); spp = int(sys.argv[3])
run(name, flags=flags, spp=spp, reps=1)
