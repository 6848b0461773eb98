"""Diagnostic: cycle shares of scan / shade / settle in the pixel-owner kernel.
Needs a stamped build:  make -B EXTRA_HIPFLAGS=-DPT_STAMPS path_tracer_c_amd/lib/libptgpu.so
(rebuild without the flag afterwards; the stamped build is slower and is never shipped)."""
import ctypes as C
import sys
sys.path.insert(0, '.')
import path_tracer_c_amd as pt
from tools.sweep_settle import run

lib = pt.lib()
fn = lib.ptgpu_debug_stamps
fn.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
out = (C.c_ulonglong * 8)()
for name, kw in (('c2', {}), ('c0', {}), ('c1', dict(w=1920, h=1080, spp=64))):
    fn(out, 1)
    run(name, flags=0x80, reps=1, **kw)
    fn(out, 0)
    scan, shade, settle, total = [float(x) for x in out][:4]
    print(f"{name}: scan {scan/total:.3f} shade {shade/total:.3f} settle {settle/total:.3f} "
          f"other {1-(scan+shade+settle)/total:.3f}")
