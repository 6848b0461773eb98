"""The reference's own scene (src/main.c:206-230; configuration 0) at its window size, 1280x800:
frame time per schedule / scan / lanes per pixel, and what the library picks by itself."""
import sys
sys.path.insert(0, '.')
import torch
import path_tracer_c_amd as pt

w, h = 1280, 800
for spp, bounces in ((16, 4), (256, 8)):
    with pt.HostScene(0) as s, pt.DeviceScene(s) as d:
        st = torch.cuda.current_stream().cuda_stream
        ref = torch.zeros((h, w, 3), dtype=torch.uint8, device='cuda')
        out = torch.zeros_like(ref)
        def run(name, **kw):
            p = pt.make_params(spp=spp, max_bounces=bounces, seed=1, **kw)
            d.render(out.data_ptr(), w, h, p, st); torch.cuda.synchronize()
            pt.host.reset_stats()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): d.render(out.data_ptr(), w, h, p, st)
            e1.record(); torch.cuda.synchronize()
            stt = pt.host.read_stats()
            ms = e0.elapsed_time(e1) / 5
            print(f"{spp:3d} spp {bounces} bounces  {name:34s} {ms:7.3f} ms  {stt.rays/5/ms/1e6:6.2f} Gray/s  "
                  f"{stt.rays/stt.samples:.2f} rays/sample  lane use {stt.rays/(64*stt.wave_iters):.3f}  "
                  f"same {bool(torch.equal(out, ref)) if name != 'library default' else '-'}", flush=True)
            return ms
        d.render(ref.data_ptr(), w, h, pt.make_params(spp=spp, max_bounces=bounces, seed=1), st); torch.cuda.synchronize()
        run("library default")
        print(f"    (picked: schedule {d.schedule}, scan {d.scan})")
        for k in (1, 2, 4, 8):
            run(f"owner plain, {k} lane(s) per pixel", flags=0x90, lanes_per_pixel=k)
        run("owner screened", flags=0xA0, lanes_per_pixel=1)
        run("queue plain", flags=0x110)
        run("queue screened", flags=0x120)
