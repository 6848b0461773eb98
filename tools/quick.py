import sys
sys.path.insert(0, '.')
from tools.sweep_settle import run
import path_tracer_c_amd as pt
for flags in (0x80, 0x280, 0x80):
    for settle in (0, 64):
        run('c2', flags=flags, settle=settle)
        st = pt.host.read_stats(); print("   scans/rays", st.scans / st.rays)
run('c0', flags=0x80); run('c0', flags=0x280)
run('c5', spp=32, flags=0x80); run('c5', spp=32, flags=0x280)
