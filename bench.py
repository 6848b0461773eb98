#!/usr/bin/env python3
"""bench.py -- the headline benchmark of the MI355X render path.

    python bench.py --gpus N --steps K --warmup W

A step is ONE FRAME of BASELINE.json's metric configuration (config 2: the
8-sphere Cornell-style scene, 1920x1080, 256 samples per pixel, 8 bounces):
the hot path -- src/renderer.c:92-116 + src/ray.c:4-73 of the reference --
over every pixel, with the scene already resident in HBM and the frame
written to HBM.  At N > 1 (launched by torch.distributed.run, one process per
GPU) the frame is cut into interleaved 8-row tiles, each rank renders its
tiles, and ONE gather over RCCL/xGMI brings them to rank 0; total work is
fixed, so scaling is "strong".

Rank 0 prints one JSON line.  `value` is Mray/s = trace_ray-equivalent path
segments (counted exactly by the kernel) per second of wall time between two
barrier + device-synchronize brackets, maximum over ranks.  The line also
carries
  roofline      the dominant kernel (pt_megakernel) against the FP32 vector
                ALU roof that bounds it (no MFMA, HBM
                is not binding), its duration measured here with HIP events
                on the launch stream, plus the algorithmic HBM rate;
  cpu_baseline  the CPU oracle (a restatement of the reference; the reference
                itself cannot be built here) timed on this machine's host
                cores on a bounded sample of the same frame, N = 1 only;
  rmse_vs_cpu_ref  per-pixel RMSE of sampled rows of the GPU frame against the
                CPU oracle (same stream: expected exactly 0) and against the
                oracle on the reference's glibc random() stream.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

import numpy as np  # noqa: E402

METRIC = "Mray/s at 1920x1080, 256 spp, 8 bounces; per-pixel RMSE vs CPU ref"
# MI355X_MICROARCH.md, chip-level parameters: "Peak FP32 (vector) 157.3 TFLOPS"
# (= 256 CUs x 4 SIMD-32 x 2 FLOP/FMA x 2.4 GHz; v_fma_f32 issues a wave64 in 2 cycles),
# HBM3E 8 TB/s.
PEAK_FP32_VECTOR_TFLOPS = 157.3
PEAK_HBM_GBS = 8000.0


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="c2", choices=["c1", "c2", "c3", "c4", "c5"],
                    help="BASELINE.json configuration; the metric is quoted on c2")
    ap.add_argument("--spp", type=int, default=0, help="override samples per pixel")
    ap.add_argument("--pass-spp", type=int, default=256,
                    help="c3 only: samples per progressive pass")
    ap.add_argument("--tile-rows", type=int, default=0)
    ap.add_argument("--block", type=int, default=0)
    ap.add_argument("--flags", type=lambda s: int(s, 0), default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo is a rehearsal mode for boxes with fewer GPUs than ranks: "
                         "ranks share devices and the gather goes through host memory")
    ap.add_argument("--verify", action="store_true",
                    help="compare the (gathered) frame with a single-launch render (untimed)")
    ap.add_argument("--out", default="", help="write the last frame as BMP (rank 0)")
    return ap.parse_args()


def flop_per_ray(n_spheres, n_triangles):
    """17 FLOP per ray-sphere test, 45 per ray-triangle
    test, ~100 per shade step."""
    return 17 * n_spheres + 45 * n_triangles + 100


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(pt, scene, cfg, gpu_frame_rows):
    """The oracle on this machine's host cores, on a bounded sample of the same
    frame: every 8th row with the GPU's stream on all cores (corrected row
    pool), plus every 128th row on the reference's glibc random() stream."""
    from oracle import binding as oracle

    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    affinity = cores
    nproc = os.cpu_count() or cores
    cores = max(1, min(cores, 16))   # a one-GPU box's CPU share is 16 cores
    w, h = cfg["width"], cfg["height"]
    kw = dict(spp=cfg["spp"], max_bounces=cfg["max_bounces"], seed=cfg["seed"])
    step = 8
    t0 = time.perf_counter()
    frame, rays = oracle.render_mt(scene, w, h, cores, row_begin=0, row_end=h,
                                   row_step=step, **kw)
    dt = time.perf_counter() - t0
    rows = list(range(0, h, step))
    out = {
        "value": rays / dt / 1e6, "unit": "Mray/s", "cores": cores, "kind": "port",
        "host": {"nproc": nproc, "affinity": affinity, "threads_used": cores, "cpu": cpu_model()},
        "sample": f"rows 0,{step},{2*step},... ({len(rows)} of {h}) of the same frame, "
                  f"{rays} rays in {dt:.2f} s; CPU oracle (restated reference, "
                  "per-pixel xorshift stream) on a corrected row-task pool",
    }
    exact = float(np.sqrt(((gpu_frame_rows(rows).astype(np.float64) - frame[rows]) ** 2).mean()))

    # the reference's own generator: process-global random() behind a lock.
    # One thread is the reference-faithful figure; the same rows on all cores
    # show the lock (slower than one thread).
    lrows = [0, h // 2, h - 1]
    oracle.lib().oracle_srandom(12345)
    lframe = np.zeros((h, w, 3), dtype=np.uint8)
    lrays = 0
    t0 = time.perf_counter()
    for y in lrows:
        _, n = oracle.render_rows(scene, w, h, y, y + 1, frame=lframe,
                                  rng_mode=oracle.RNG_LIBC, **kw)
        lrays += n
    ldt = time.perf_counter() - t0
    t0 = time.perf_counter()
    _, mrays = oracle.render_mt(scene, w, h, cores, row_begin=h // 2, row_end=h // 2 + 2,
                                row_step=1, rng_mode=oracle.RNG_LIBC, **kw)
    mdt = time.perf_counter() - t0
    out["libc_random"] = {
        "value": lrays / ldt / 1e6, "unit": "Mray/s", "cores": 1,
        "sample": f"{len(lrows)} rows, {lrays} rays in {ldt:.2f} s, one thread; glibc "
                  "random() as the reference uses it",
        "all_cores": {"value": mrays / mdt / 1e6, "cores": cores,
                      "sample": f"2 rows, {mrays} rays in {mdt:.2f} s (global lock in "
                                "random(): anti-scaling, SURVEY.md section 6)"},
    }
    stat = float(np.sqrt(((gpu_frame_rows(lrows).astype(np.float64) - lframe[lrows]) ** 2).mean()))
    rmse = {"same_stream": exact, "same_stream_rows": len(rows),
            "libc_random_stream": stat, "libc_random_rows": len(lrows), "unit": "8-bit levels"}
    return out, rmse


def read_profiled_traffic(workload):
    """HBM bytes per launch from the committed rocprofv3 PMC summary, if any."""
    p = ROOT / "profiles" / "hbm_traffic.json"
    try:
        data = json.loads(p.read_text())
        return data.get(workload, {}).get("hbm_bytes_per_launch")
    except Exception:
        return None


def bvh_work_roofline(workload, rays_per_launch, kernel_seconds):
    """For a hierarchy scan the arithmetic actually asked of the device per ray
    (measured averages of a counting build, profiles/bvh_work.json) against the
    FP32 vector roof: the utilisation figure `achieved`/`frac` cannot give there."""
    p = ROOT / "profiles" / "bvh_work.json"
    try:
        w = json.loads(p.read_text())[workload]
    except Exception:
        return {}
    flop = (w["inner_node_steps_per_ray"] * w["flop_per_inner_step"] +
            w["objects_screened_per_ray"] * w["flop_per_screen"] +
            w["exact_tests_per_ray"] * w["flop_per_exact_sphere_test"] + w["flop_per_shade"])
    achieved = rays_per_launch * flop / kernel_seconds / 1e12
    return {"bvh_work": {"flop_per_ray": flop, "achieved": achieved, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_FP32_VECTOR_TFLOPS,
                         "note": "per-ray averages measured with tools/bvh_profile.py (profiles/bvh_work.json)"}}


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run "
                             "(one process per GPU)")
        args.gpus = world

    import torch
    import torch.distributed as dist

    import path_tracer_c_amd as pt
    from path_tracer_c_amd import tiles

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the render path has no CPU fallback")
    device = local_rank % torch.cuda.device_count() if args.backend == "gloo" else local_rank
    torch.cuda.set_device(device)
    pt._abi.check(pt.lib().ptgpu_set_device(device))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device))
        else:
            dist.init_process_group("gloo")

    cfg = dict(pt.CONFIGS[int(args.workload[1])])
    if args.spp:
        cfg["spp"] = args.spp
    w, h = cfg["width"], cfg["height"]
    progressive = args.workload == "c3"
    pass_spp = min(args.pass_spp, cfg["spp"]) if progressive else cfg["spp"]
    n_pass = cfg["spp"] // pass_spp if progressive else 1

    scene = pt.HostScene(cfg["scene"])
    n_sph, n_tri = scene.counts()
    dscene = pt.DeviceScene(scene)               # scene resident in HBM before timing
    stream = torch.cuda.current_stream().cuda_stream

    tile = args.tile_rows or tiles.default_tile_rows(h, world)
    shard = tiles.shard_rows(h, world, rank, tile)
    pad_rows = tiles.max_shard_rows(h, world, tile)
    # The timed headline scans the scene for EVERY counted ray: the library's
    # default shares the (identical) primary-ray scan between the samples of a
    # pixel, which lowers frame time but would count rays whose scan was not
    # executed.  That mode is measured separately below (N = 1) and reported
    # under config.primary_share, never as `value`.
    flags = args.flags | pt._abi.FLAG_NO_PRIMARY_SHARE | (pt._abi.FLAG_COMPACT_ROWS if world > 1 else 0)
    base = dict(max_bounces=cfg["max_bounces"], seed=cfg["seed"], row_begin=shard.row_begin,
                num_rows=shard.num_rows, tile_rows=shard.tile_rows if world > 1 else 0,
                tile_stride=shard.tile_stride, flags=flags, block_threads=args.block)
    local = torch.zeros((pad_rows, w, 3), dtype=torch.uint8, device="cuda")
    accum = torch.zeros((pad_rows, w, 4), dtype=torch.float32, device="cuda") if progressive else None
    frame = local
    gather_scratch = {}
    if world > 1:
        frame = None

    kernel_ms = []
    gather_ms = []
    # parameter blocks are built once, outside the timed loop
    frame_params = pt.make_params(spp=cfg["spp"], **base)
    pass_params = [pt.make_params(spp=pass_spp, sample_offset=k * pass_spp,
                                  **{**base, "flags": base["flags"] | pt._abi.FLAG_COMPACT_ROWS})
                   for k in range(n_pass)] if progressive else []

    def step(timed):
        """One frame: render this rank's rows (+ gather at N > 1)."""
        ev0 = torch.cuda.Event(enable_timing=True)
        ev1 = torch.cuda.Event(enable_timing=True)
        ev0.record()
        if progressive:
            accum.zero_()
            for p in pass_params:
                dscene.accumulate(accum.data_ptr(), w, h, p, stream)
            pt.host.resolve(accum.data_ptr(), local.data_ptr(), shard.num_rows * w, stream)
        else:
            dscene.render(local.data_ptr(), w, h, frame_params, stream)
        ev1.record()
        nonlocal frame
        if world > 1:
            if args.backend == "nccl":   # ONE gather over RCCL + one scatter kernel on rank 0
                got = tiles.gather_frame(local, h, w, tile, scratch=gather_scratch)
            else:                        # rehearsal: the same code path through host memory
                got = tiles.gather_frame(local.cpu(), h, w, tile, scratch=gather_scratch)
                got = got.cuda() if got is not None else None
            if rank == 0:
                frame = got
            # (the collective has been waited for on this stream: the event
            # closes the gather + un-interleave as this rank saw them)
            ev2 = torch.cuda.Event(enable_timing=True)
            ev2.record()
            if timed:
                gather_ms.append((ev1, ev2))
        if timed:
            kernel_ms.append((ev0, ev1))

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # which scan form / schedule / lanes per pixel suit this scene is measured
    # once, explicitly and untimed (ptgpu_scene_tune); the timed entry points
    # are asynchronous and never measure
    dscene.tune(w, h, pass_params[0] if progressive else frame_params)
    for _ in range(args.warmup):
        step(False)
    fence()
    pt.host.reset_stats()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    fence()
    elapsed = time.perf_counter() - t0

    st = pt.host.read_stats()
    red = "cuda" if args.backend == "nccl" else "cpu"
    counts = torch.tensor([st.rays, st.samples, st.wave_iters], dtype=torch.float64, device=red)
    tmax = torch.tensor([elapsed], dtype=torch.float64, device=red)
    kmax = torch.tensor([sum(a.elapsed_time(b) for a, b in kernel_ms) / max(1, len(kernel_ms))],
                        dtype=torch.float64, device=red)
    if world > 1:
        dist.all_reduce(counts, op=dist.ReduceOp.SUM)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(kmax, op=dist.ReduceOp.MAX)
    rays, samples, wave_iters = (float(x) for x in counts.tolist())
    elapsed = float(tmax.item())
    kernel_ms_avg = float(kmax.item())
    # what each rank did, so that a sub-linear point of a scaling curve can be
    # attributed: kernel time per rank, the gather as each rank saw it, imbalance
    per_rank = None
    if world > 1:
        my_kernel = sum(a.elapsed_time(b) for a, b in kernel_ms) / max(1, len(kernel_ms))
        my_gather = sum(a.elapsed_time(b) for a, b in gather_ms) / max(1, len(gather_ms))
        per_rank = tiles.scaling_report(my_kernel, my_gather, h, w, tile, device=red)

    if rank == 0:
        rays_per_step = rays / args.steps
        # megakernel launches per frame: 1 in pixel-owner mode, one per queue
        # pass in sample-queue mode, one per progressive pass for c3
        launches_per_step = max(1, round(st.launches / args.steps))
        fpr = flop_per_ray(n_sph, n_tri)
        # per-GPU kernel: this rank's share of the rays over its kernel time
        rays_per_launch = rays_per_step / world / launches_per_step
        k_s = kernel_ms_avg * 1e-3 / launches_per_step
        achieved_tflops = rays_per_launch * fpr / k_s / 1e12
        queue = dscene.schedule == "sample-queue"
        bvh = dscene.scan == "bvh"
        if progressive:   # read + write one float4 per pixel per pass
            hbm_alg = 32.0 * shard.num_rows * w
        elif queue:       # one float4 per sample stored, then read back in order
            hbm_alg = 32.0 * samples / args.steps / world / launches_per_step
        else:             # 3 bytes per pixel, once
            hbm_alg = 3.0 * shard.num_rows * w
        line = {
            "metric": METRIC,
            "value": rays / elapsed / 1e6,
            "unit": "Mray/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"BASELINE config {args.workload[1]}: scene {cfg['scene']} "
                            f"({n_sph} spheres, {n_tri} triangles), {w}x{h}, {cfg['spp']} spp, "
                            f"{cfg['max_bounces']} bounces, seed {cfg['seed']}"
                            + (f", {n_pass} progressive passes of {pass_spp} spp" if progressive else ""),
                "sharding": "single launch" if world == 1 else
                            f"{tile}-row tiles interleaved over {world} ranks + one RCCL gather",
                "rays_per_sample": rays / samples if samples else None,
                "msample_per_s": samples / elapsed / 1e6,
                "lane_utilisation": rays / (64.0 * wave_iters) if wave_iters else None,
                "schedule": dscene.schedule,
                "scan": dscene.scan,
            },
            "roofline": {
                "kernel": ("pt_streamkernel" if bvh else "pt_samplekernel") if queue else "pt_megakernel",
                "bound": "valu",
                "achieved": achieved_tflops,
                "peak": PEAK_FP32_VECTOR_TFLOPS,
                "unit": "TFLOP/s",
                "frac": achieved_tflops / PEAK_FP32_VECTOR_TFLOPS,
                "flop_per_ray": fpr,
                "rays_per_launch": rays_per_launch,
                "kernel_ms": kernel_ms_avg / launches_per_step,
                "note": "FP32 vector ALU roof (SURVEY.md 8(d)): no MFMA on this path and HBM "
                        "is not binding; peak = FP32 vector figure of "
                        "MI355X_MICROARCH.md (256 CUs x 4 SIMD-32 x 2 FLOP x 2.4 GHz)",
                **(bvh_work_roofline(args.workload, rays_per_launch, k_s) if bvh else {}),
                "traffic": read_profiled_traffic(args.workload),
                "hbm": {"algorithmic_bytes_per_launch": hbm_alg,
                        "achieved": hbm_alg / k_s / 1e9, "peak": PEAK_HBM_GBS,
                        "unit": "GB/s", "frac": hbm_alg / k_s / 1e9 / PEAK_HBM_GBS},
            },
        }
        if bvh and "bvh_work" in line["roofline"]:
            # A hierarchy scan does not execute the reference's linear scan, so
            # pricing its rays at 17 FLOP x every object would report a
            # "fraction" above 1.  The utilisation figure is the arithmetic the
            # walk really asks for (measured per-ray averages); the linear-scan
            # pricing is kept beside it as a speed-up, never as `frac`.
            rf = line["roofline"]
            work = rf.pop("bvh_work")
            rf["linear_scan_equivalent"] = {
                "flop_per_ray": rf["flop_per_ray"], "tflops": rf["achieved"],
                "note": "rays priced at the reference's linear scan (src/ray.c:12-65: every "
                        "object tested); ratio to the roof = speed-up over the reference's "
                        "algorithm run at the ALU roof, not a utilisation"}
            rf["achieved"], rf["frac"], rf["flop_per_ray"] = work["achieved"], work["frac"], work["flop_per_ray"]
            rf["note"] += "; flop_per_ray = " + work["note"]
        assert st.scans == st.rays, "headline must scan the scene for every counted ray"
        if per_rank is not None:
            per_rank["backend"] = "RCCL (torch nccl)" if args.backend == "nccl" else "gloo rehearsal (ranks share devices)"
            per_rank["note"] = ("kernel_ms / gather_ms: per rank, per frame, HIP events on the rank's stream; "
                                "imbalance = max / mean of kernel_ms; ms_per_step is the wall clock between "
                                "the barriers, maximum over ranks")
            line["ranks"] = per_rank
        if world == 1 and not progressive:
            # library default: primary-ray scan shared between a pixel's samples
            shared = pt.make_params(spp=cfg["spp"], **{**base, "flags": base["flags"] & ~pt._abi.FLAG_NO_PRIMARY_SHARE})
            dscene.render(local.data_ptr(), w, h, shared, stream)
            torch.cuda.synchronize()
            pt.host.reset_stats()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                dscene.render(local.data_ptr(), w, h, shared, stream)
            e1.record()
            torch.cuda.synchronize()
            s2 = pt.host.read_stats()
            ms = e0.elapsed_time(e1) / 3
            line["config"]["primary_share"] = {
                "ms_per_step": ms, "path_segments_per_s_M": s2.rays / 3 / ms / 1e3,
                "scene_scans_per_s_M": s2.scans / 3 / ms / 1e3,
                "scans_per_segment": s2.scans / s2.rays,
                "note": "library default; same image; not used for `value`"}
        if world == 1 and not progressive:
            # the opt-in tolerance build (PTGPU_FLAG_FAST_MATH): same frame, same
            # stream, every ray's scan executed; reported beside the exact
            # kernel with its own roofline and its distance from the exact
            # frame -- never as `value`
            exact_frame = local[:h].clone()
            fast = pt.make_params(spp=cfg["spp"], **{**base, "flags": base["flags"] | pt._abi.FLAG_FAST_MATH})
            dscene.render(local.data_ptr(), w, h, fast, stream)
            torch.cuda.synchronize()
            pt.host.reset_stats()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                dscene.render(local.data_ptr(), w, h, fast, stream)
            e1.record()
            torch.cuda.synchronize()
            s3 = pt.host.read_stats()
            ms = e0.elapsed_time(e1) / 3
            diff = local[:h].to(torch.int32) - exact_frame.to(torch.int32)
            tf = s3.rays / 3 * fpr / (ms * 1e-3) / 1e12
            line["fast_math"] = {
                "flag": "PTGPU_FLAG_FAST_MATH", "ms_per_step": ms, "value": s3.rays / 3 / ms / 1e3,
                "unit": "Mray/s", "speedup_vs_exact": line["ms_per_step"] / ms,
                "roofline": ({"bound": "valu", "achieved": tf, "peak": PEAK_FP32_VECTOR_TFLOPS,
                              "unit": "TFLOP/s", "frac": tf / PEAK_FP32_VECTOR_TFLOPS,
                              "flop_per_ray": fpr} if not bvh else
                             bvh_work_roofline(args.workload, s3.rays / 3 / launches_per_step,
                                               ms * 1e-3 / launches_per_step).get("bvh_work")),
                "vs_exact_frame": {
                    "bytes_within_1": float((diff.abs() <= 1).float().mean()),
                    "bytes_equal": float((diff == 0).float().mean()),
                    "rmse_8bit_levels": float(diff.float().pow(2).mean().sqrt()),
                    "mean_diff": float(diff.float().mean())},
                "note": "tolerance build: fused multiply-adds, v_sqrt/v_rcp/v_rsq as they come; "
                        "the default build stays bit-exact"}
            local[:h].copy_(exact_frame)
            del exact_frame, diff
        if world == 1 and args.workload == "c5":
            # BASELINE configs[4] asks for this comparison by name: the 1000-sphere scene
            # scanned linearly with the scene records read through the scalar data cache
            # (global memory: "HBM linear scan") against the per-workgroup LDS copy, both in
            # the sample-queue schedule (ballot + prefix-sum refill of finished lanes = the
            # live-ray compaction), beside the hierarchy walk the library picks.  16 spp each.
            A = pt._abi
            forms = [("screened linear scan, records from global memory (scalar loads)", A.FLAG_SCAN_SCREENED | A.FLAG_SAMPLE_QUEUE),
                     ("screened linear scan, records from an LDS tile", A.FLAG_SCAN_SCREENED | A.FLAG_SAMPLE_QUEUE | A.FLAG_SCENE_LDS),
                     ("plain linear scan, records from global memory", A.FLAG_SCAN_PLAIN | A.FLAG_SAMPLE_QUEUE),
                     ("plain linear scan, records from an LDS tile", A.FLAG_SCAN_PLAIN | A.FLAG_SAMPLE_QUEUE | A.FLAG_SCENE_LDS),
                     ("screened linear scan, pixel-owner schedule (no compaction)", A.FLAG_SCAN_SCREENED | A.FLAG_PIXEL_OWNER),
                     ("hierarchy, resumable walks (the library's choice)", A.FLAG_SCAN_BVH | A.FLAG_SAMPLE_QUEUE)]
            table = []
            keep = local.clone()
            for label, f in forms:
                q = pt.make_params(spp=16, **{**base, "flags": base["flags"] | f})
                dscene.render(local.data_ptr(), w, h, q, stream)
                torch.cuda.synchronize()
                pt.host.reset_stats()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                dscene.render(local.data_ptr(), w, h, q, stream)
                e1.record()
                torch.cuda.synchronize()
                sf = pt.host.read_stats()
                ms = e0.elapsed_time(e1)
                table.append({"form": label, "ms_per_16spp": ms, "Mray_per_s": sf.rays / ms / 1e3,
                              "lane_utilisation": sf.rays / (64.0 * sf.wave_iters) if sf.wave_iters else None})
            line["config"]["scan_forms_16spp"] = table
            local.copy_(keep)
            del keep
        if world == 1 and not progressive:
            # end to end through the host-buffer entry point (PCIe included):
            # scene upload + kernel + frame download; reported, never `value`
            import numpy as _np
            host = _np.zeros((h, w, 3), dtype=_np.uint8)
            kw = dict(spp=cfg["spp"], max_bounces=cfg["max_bounces"], seed=cfg["seed"])
            pt.render_frame(scene, w, h, **kw)
            t0 = time.perf_counter()
            host = pt.render_frame(scene, w, h, **kw)
            line["config"]["host_frame_ms"] = (time.perf_counter() - t0) * 1e3
            line["config"]["host_frame_note"] = ("ptgpu_render_frame: scene upload + kernel + "
                                                 f"{3 * w * h / 1e6:.1f} MB download, library defaults")
        if world == 1 and not progressive:
            # the batched trace_ray kernel (src/ray.c:4-73 per ray) on rays
            # resident in HBM: the eye rays of a 4096x4096 image through this
            # scene.  64 algorithmic bytes per ray (24 in, 40 out): HBM-bound
            # when the scene is small.  Reported beside the headline.
            n_rays = 1 << 24
            ix = torch.arange(n_rays, device="cuda")
            dx = (ix % 4096).float() / 4096 * 2 - 1
            dy = -((ix // 4096).float() / 4096 * 2 - 1)
            dirs = torch.stack([dx, dy, torch.ones_like(dx)], 1)
            dirs = (dirs / dirs.norm(dim=1, keepdim=True)).contiguous()
            orgs = torch.zeros((n_rays, 3), device="cuda")
            hits = torch.empty((n_rays, 10), dtype=torch.float32, device="cuda")
            scan_flag = pt._abi.FLAG_SCAN_BVH if dscene.scan == "bvh" else 0
            dscene.trace(orgs.data_ptr(), dirs.data_ptr(), n_rays, hits.data_ptr(), scan_flag, stream)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                dscene.trace(orgs.data_ptr(), dirs.data_ptr(), n_rays, hits.data_ptr(), scan_flag, stream)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 5
            gbs = 64.0 * n_rays / ms / 1e6
            line["trace_rays"] = {
                "kernel": "pt_trace_kernel", "rays": n_rays, "ms": ms, "value": n_rays / ms / 1e3,
                "unit": "Mray/s", "roofline": {"bound": "hbm", "achieved": gbs, "peak": PEAK_HBM_GBS,
                                               "unit": "GB/s", "frac": gbs / PEAK_HBM_GBS,
                                               "bytes_per_ray": 64,
                                               "frac_of_measured_copy_rate": gbs / 6290.0,
                                               "note": "MI355X_MICROARCH.md: 8.0 TB/s spec, 6.29 TB/s "
                                                       "measured for a float4 copy"},
                "note": "ptgpu_trace_rays_device, eye rays of a 4096x4096 image; not part of `value`"}
            del dirs, orgs, hits, ix, dx, dy
        if world == 1 and not args.no_cpu_baseline:
            host_frame = frame.cpu().numpy()
            base_line, rmse = cpu_baseline(pt, scene, cfg, lambda rows: host_frame[rows])
            line["cpu_baseline"] = base_line
            line["rmse_vs_cpu_ref"] = rmse
        if args.verify and frame is not None:
            # the assembled frame must equal a single-launch render on this GPU
            check = torch.zeros((h, w, 3), dtype=torch.uint8, device="cuda")
            dscene.render(check.data_ptr(), w, h,
                          pt.make_params(spp=cfg["spp"], max_bounces=cfg["max_bounces"],
                                         seed=cfg["seed"]), stream)
            torch.cuda.synchronize()
            line["verified_equal_to_single_launch"] = bool(torch.equal(check, frame[:h]))
        if args.out and frame is not None:
            pt.host.write_bitmap(args.out, frame.cpu().numpy())
        print(json.dumps(line), flush=True)

    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
