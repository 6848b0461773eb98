"""Frame time of the shader-rules mode (PTGPU_FLAG_GLSL_SEMANTICS) on configurations 2 and 5 (GPU box)."""
import sys, time
sys.path.insert(0, '.')
import torch
import path_tracer_c_amd as pt
for key in (2, 5):
    c = pt.CONFIGS[key]
    w, h = c['width'], c['height']
    with pt.HostScene(c['scene']) as s, pt.DeviceScene(s) as d:
        st = torch.cuda.current_stream().cuda_stream
        out = torch.zeros((h, w, 3), dtype=torch.uint8, device='cuda')
        spp = c['spp'] if key != 5 else 32
        p = pt.make_params(spp=spp, max_bounces=min(c['max_bounces'], 8), seed=7, flags=pt._abi.FLAG_GLSL_SEMANTICS)
        d.render(out.data_ptr(), w, h, p, st); torch.cuda.synchronize()
        pt.host.reset_stats()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(2): d.render(out.data_ptr(), w, h, p, st)
        e1.record(); torch.cuda.synchronize()
        stt = pt.host.read_stats()
        ms = e0.elapsed_time(e1) / 2
        print(f"config {key} {w}x{h} spp {spp}: GLSL-rules frame {ms:.2f} ms, rays {stt.rays/2:.0f}, {stt.rays/2/ms/1e6:.2f} Gray/s")
