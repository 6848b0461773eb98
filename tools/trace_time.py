"""Throughput of the batched closest-hit query on device-resident rays (ptgpu_trace_rays_device):
64 bytes of HBM traffic per ray (24 in, 40 out)."""
import sys
sys.path.insert(0, '.')
import torch
import path_tracer_c_amd as pt

n = 1 << 25
g = torch.Generator(device='cuda').manual_seed(1)
for cfg, flag, name in ((2, 0, "c2 scene, plain scan"), (0, 0, "demo scene, plain scan"),
                        (5, 0x400, "c5 scene, BVH scan"), (5, 0x20, "c5 scene, screened scan")):
    with pt.HostScene(cfg) as s, pt.DeviceScene(s) as d:
        st = torch.cuda.current_stream().cuda_stream
        for kind in ("coherent (eye rays of a 8192x4096 image)", "incoherent (random directions from inside the scene)"):
            if kind.startswith("coherent"):
                ix = torch.arange(n, device='cuda')
                x = (ix % 8192).float() / 8192 * 2 - 1
                y = -((ix // 8192).float() / 4096 * 2 - 1)
                dirs = torch.stack([x, y * 0.5, torch.ones_like(x)], 1)
                org = torch.zeros((n, 3), device='cuda')
            else:
                dirs = torch.randn((n, 3), device='cuda', generator=g)
                org = torch.rand((n, 3), device='cuda', generator=g) * torch.tensor([4.0, 2.0, 4.0], device='cuda') + torch.tensor([-2.0, 0.0, 3.0], device='cuda')
            dirs = (dirs / dirs.norm(dim=1, keepdim=True)).contiguous()
            org = org.contiguous()
            hits = torch.empty((n, 10), dtype=torch.float32, device='cuda')
            d.trace(org.data_ptr(), dirs.data_ptr(), n, hits.data_ptr(), flag, st); torch.cuda.synchronize()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3): d.trace(org.data_ptr(), dirs.data_ptr(), n, hits.data_ptr(), flag, st)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 3
            frac_hit = float((hits[:, 0] >= 0).float().mean())
            print(f"{name:26s} {kind:52s} {ms:8.3f} ms  {n/ms/1e6:7.2f} Gray/s  {64*n/ms/1e6:7.1f} GB/s  hit fraction {frac_hit:.2f}", flush=True)
