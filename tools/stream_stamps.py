"""Diagnostic: where a wave of the stream kernel spends its cycles (walk quanta, shading,
settling, refill; the rest is walk begin/finish and loop control).  Needs the stamped build:
make -B EXTRA_HIPFLAGS=-DPT_STAMPS path_tracer_c_amd/lib/libptgpu.so  (rebuild plain afterwards)."""
import ctypes as C
import sys
sys.path.insert(0, '.')
import torch
import path_tracer_c_amd as pt

fn = pt.lib().ptgpu_debug_stamps
fn.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
out = (C.c_ulonglong * 8)()
c = pt.CONFIGS[5]
w, h = c['width'], c['height']
with pt.HostScene(c['scene']) as s, pt.DeviceScene(s) as d:
    st = torch.cuda.current_stream().cuda_stream
    img = torch.zeros((h, w, 3), dtype=torch.uint8, device='cuda')
    for name, flags in (("primary hits shared", 0x500), ("every segment walked", 0x700)):
        p = pt.make_params(spp=16, max_bounces=8, seed=1, flags=flags)
        d.render(img.data_ptr(), w, h, p, st); torch.cuda.synchronize()
        fn(out, 1)
        d.render(img.data_ptr(), w, h, p, st); torch.cuda.synchronize()
        fn(out, 0)
        walk, shade, settle, total, refill, q_inner, q_leaf, q_exact = [float(out[i]) for i in range(8)]
        print(f"{name:22s} walk {walk/total:.3f} (inner steps {q_inner/total:.3f}, leaf visits {q_leaf/total:.3f}, "
              f"exact tests {q_exact/total:.3f})  shade {shade/total:.3f}  settle {settle/total:.3f}  "
              f"refill {refill/total:.3f}  other {1-(walk+shade+settle+refill)/total:.3f}")
