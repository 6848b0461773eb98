import sys
sys.path.insert(0, '.')
import numpy as np
import path_tracer_c_amd as pt
from oracle import binding as ob
from tools.bvh_scaling import scene
with scene(250) as s:
    o, d = ob.sample_rays(s, 960, 540, 945, 337, 7, max_bounces=8, seed=1)
    print(len(o), "rays")
    p = pt.trace_rays(s, o, d); b = pt.trace_rays(s, o, d, pt._abi.FLAG_SCAN_BVH); c = ob.trace_rays(s, o, d)
    for k in range(len(o)):
        print(k, "o", o[k], "d", d[k], "| plain", p[0][k], int(p[3][k]), "| bvh", b[0][k], int(b[3][k]), "| oracle", c[0][k], int(c[3][k]))
