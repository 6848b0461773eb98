// pt_owner.h -- the per-lane loops of the pixel-owner schedule
// (pt_megakernel, pt_kernels.hip): a lane traces all samples of its pixel
// (run_paths), or k lanes share a pixel (run_paths_split).  kTree: the scene
// has a material of 0 < transparency < 1, incident_light is a binary tree
// (src/renderer.c:45,56,72) and a settled lane may go on with a transmitted
// ray instead of ending its sample (pt_shade.h, settle_path).
#pragma once

#include "pt_shade.h"

namespace PT_NS {

template <int kScan, bool kTree>
__device__ __forceinline__ void
run_paths(const SceneRecords &rec, const LaunchArgs &a, uint16_t *stack,
          uint32_t stack_stride, const ScreenQueue queue, V3 d0, uint32_t pixel,
          uint32_t spp, V3 sky, V3 &sum, uint32_t &rays, uint32_t &scans,
          uint32_t &iters) {
    uint32_t s = 0;
    PathState p;
    p.o = splat(0.0f);
    p.d = d0;
    p.r = splat(0.0f);
    p.depth = 0;
    p.rng = ptgpu_rng_seed(a.seed, pixel, a.sample_offset);
    p.live = s < spp;
    p.parked = false;
    Branches<kTree> br;

    // Every sample of a pixel starts with the same un-jittered primary ray
    // (src/renderer.c:96-100 computes x, y once per pixel), so its closest hit
    // is the same: scan for it once.  The stored hit is used only on passes
    // where every live lane of the wave is at a path's first segment -- a
    // wave-uniform test, true at the start of nearly every sample when the
    // wave's paths have equal lengths (closed scenes) -- otherwise the scan
    // runs anyway for the other lanes and sharing would save nothing.
    Hit h_primary = Hit{-1.0f, -1};
    if (a.share_primary && p.live) {
        h_primary = scan_scene<kScan>(rec, p.o, p.d, queue);
        ++scans;
    }

    PT_STAMP(begin);
    while (pt_any(p.live || p.parked)) {
        ++iters;
        const bool all_at_start = a.share_primary && !pt_any(p.live && p.depth != 0u);
        if (p.live) {
            ++rays;
            if (!all_at_start)
                ++scans;
            advance_path<kScan, kTree>(rec, a, stack, stack_stride, queue, sky, p, br,
                                       all_at_start, h_primary);
        }
        PT_STAMP(settle0);
        if (settle_now(p.live, p.parked, a.settle_lanes)) {
            V3 sample;
            if (p.parked && settle_path<kTree>(rec, a, stack, stack_stride, sky, p, br, sample)) {
                sum = sum + sample;
                ++s;
                p.depth = 0;
                p.o = splat(0.0f);
                p.d = d0;
                p.rng = ptgpu_rng_seed(a.seed, pixel, a.sample_offset + s);
                p.parked = false;
                p.live = s < spp;
            }
        }
        PT_STAMP(settle1);
        PT_STAMP_ADD(p.t_settle, settle1, settle0);
    }
#ifdef PT_STAMPS
    PT_STAMP(end);
    if ((threadIdx.x & 63u) == 0u) {
        atomicAdd(&g_stamps[0], p.t_scan);
        atomicAdd(&g_stamps[1], p.t_shade);
        atomicAdd(&g_stamps[2], p.t_settle);
        atomicAdd(&g_stamps[3], end - begin);
    }
#endif
}

// ---------------------------------------------------------------------------
// The same, with k = 2, 4, ... lanes per pixel (consecutive lanes of a wave).
//
// A row set that is too small to fill the device with one lane per pixel (a
// shard of a frame split over many GPUs, a small image) gets k lanes per
// pixel: in round r lane q of a pixel traces sample r*k + q.  The wave is
// sample-synchronous here: when every lane has parked, all unwind, and the
// pixel's first lane adds the k clamped samples IN SAMPLE ORDER, fetching its
// partners' values with cross-lane reads -- so the sum is the very same
// sequence of binary32 additions one lane would have made, and the image is
// unchanged.  No HBM traffic, k times the waves.
// ---------------------------------------------------------------------------
template <int kScan, bool kTree>
__device__ __forceinline__ void
run_paths_split(const SceneRecords &rec, const LaunchArgs &a, uint16_t *stack,
                uint32_t stack_stride, const ScreenQueue queue, V3 d0,
                uint32_t pixel, uint32_t spp, uint32_t k, uint32_t q, V3 sky,
                V3 &sum, uint32_t &rays, uint32_t &scans, uint32_t &iters,
                uint32_t &samples_done) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t group = lane - q; // first lane of this pixel's group
    const uint32_t rounds = (a.spp + k - 1u) / k;
    PathState p;
    p.r = splat(0.0f);
    p.live = p.parked = false;
    Branches<kTree> br;

    Hit h_primary = Hit{-1.0f, -1};
    if (a.share_primary && spp != 0u) {
        h_primary = scan_scene<kScan>(rec, splat(0.0f), d0, queue);
        ++scans;
    }

    for (uint32_t round = 0; round < rounds; ++round) { // wave-uniform
        const uint32_t s = round * k + q;
        p.o = splat(0.0f);
        p.d = d0;
        p.depth = 0;
        p.rng = ptgpu_rng_seed(a.seed, pixel, a.sample_offset + s);
        p.live = s < spp;
        p.parked = false;
        bool open = p.live; // this lane traces a sample in this round, not complete yet
        V3 sample = splat(0.0f);
        for (;;) {
            while (pt_any(p.live)) {
                ++iters;
                const bool all_at_start = a.share_primary && !pt_any(p.live && p.depth != 0u);
                if (p.live) {
                    ++rays;
                    if (!all_at_start)
                        ++scans;
                    advance_path<kScan, kTree>(rec, a, stack, stack_stride, queue, sky, p, br,
                                               all_at_start, h_primary);
                }
            }
            // every path has ended: unwind (with branching materials a lane may
            // have a transmitted ray to go on with, and the wave goes round again)
            if (open && settle_path<kTree>(rec, a, stack, stack_stride, sky, p, br, sample)) {
                open = false;
                ++samples_done;
            }
            if (!kTree || !pt_any(p.live))
                break;
        }
        // the pixel's first lane adds samples round*k .. round*k + k-1, in order
        for (uint32_t i = 0; i < k; ++i) {
            const V3 other = V3{__shfl(sample.x, (int)(group + i), 64),
                                __shfl(sample.y, (int)(group + i), 64),
                                __shfl(sample.z, (int)(group + i), 64)};
            if (round * k + i < spp)
                sum = sum + other;
        }
    }
}

} // namespace PT_NS
