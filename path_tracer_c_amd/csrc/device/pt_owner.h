// pt_owner.h -- the per-lane loops of the pixel-owner schedule
// (pt_megakernel, pt_kernels.hip): a lane traces all samples of its pixel
// (run_paths), k lanes share a pixel (run_paths_split), or -- scenes with a
// material of 0 < transparency < 1, where incident_light is a binary tree
// (src/renderer.c:45,56,72) -- a lane walks each sample's tree depth first
// with a private frame stack (run_tree).
#pragma once

#include "pt_shade.h"

namespace PT_NS {

template <int kScan>
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
    while (__any(p.live || p.parked)) {
        ++iters;
        const bool all_at_start = a.share_primary && !__any(p.live && p.depth != 0u);
        if (p.live) {
            ++rays;
            if (!all_at_start)
                ++scans;
            advance_path<kScan>(rec, a, stack, stack_stride, queue, sky, p,
                                all_at_start, h_primary);
        }
        PT_STAMP(settle0);
        if (settle_now(p.live, p.parked, a.settle_lanes)) {
            if (p.parked) {
                sum = sum + unwind_path(rec, stack, stack_stride, p);
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
template <int kScan>
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
        const bool mine = p.live; // this lane traces a sample in this round

        while (__any(p.live)) {
            ++iters;
            const bool all_at_start = a.share_primary && !__any(p.live && p.depth != 0u);
            if (p.live) {
                ++rays;
                if (!all_at_start)
                    ++scans;
                advance_path<kScan>(rec, a, stack, stack_stride, queue, sky, p,
                                    all_at_start, h_primary);
            }
        }
        V3 sample = splat(0.0f);
        if (mine) {
            sample = unwind_path(rec, stack, stack_stride, p);
            ++samples_done;
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

// ---------------------------------------------------------------------------
// Branching paths: explicit depth-first evaluation of the recursion tree.
// ---------------------------------------------------------------------------
constexpr int kTreeFrames = 16; // PTGPU_MAX_BOUNCES_TREE + 2

struct Frame {
    V3 d_in;   // direction of the ray that hit this surface
    V3 normal; // perturbed normal
    V3 point;
    V3 refl;   // reflected_Li once known
    uint32_t material;
    uint32_t awaiting_transmit; // 0: child in flight is the reflection
};

template <int kScan>
__device__ void run_tree(const SceneRecords &rec, const LaunchArgs &a, const ScreenQueue queue,
                         V3 d0, uint32_t pixel, uint32_t spp, V3 sky, V3 &sum,
                         uint32_t &rays, uint32_t &iters) {
    // Per lane: the recursion of incident_light as an explicit stack of frames,
    // one sample after the other.  A pass of the outer loop has two parts:
    //   bookkeeping  the lane returns values up its stack and picks the next
    //                child to evaluate, until it stands before a call that
    //                needs a ray traced (or has finished its last sample) --
    //                cheap, lane by lane;
    //   trace+shade  every lane that has such a call scans the scene and opens
    //                the frame -- the expensive part, run with all of them at
    //                once instead of whenever a lane happened to get there.
    Frame frames[kTreeFrames];
    uint32_t s = 0;
    uint32_t rng = ptgpu_rng_seed(a.seed, pixel, a.sample_offset);
    int sp = 0; // frames in use; a call at recursion depth k has sp == k
    V3 o = splat(0.0f), d = d0;
    V3 ret = splat(0.0f);
    // state: 0 = enter a call with (o, d, bounces = sp)
    //        1 = a call returned `ret` to the frame on top
    //        2 = top frame must now evaluate its transmission branch
    //        3 = all samples done
    int state = spp != 0u ? 0 : 3;
#ifdef PT_STAMPS
    unsigned long long t_book = 0, t_trace = 0;
    PT_STAMP(tree_begin);
#endif
    for (;;) {
        PT_STAMP(bk0);
        // ---- bookkeeping until a trace is due ----------------------------------
        while (state == 1 || state == 2 || (state == 0 && (uint32_t)sp > a.max_bounces)) {
            if (state == 0) { // the call is beyond the bounce limit: sky, no trace
                ret = sky;
                state = 1;
            } else if (state == 1) {
                if (sp == 0) { // the root call returned: the sample is complete
                    sum = sum + clamp01(ret);
                    ++s;
                    if (s < spp) {
                        rng = ptgpu_rng_seed(a.seed, pixel, a.sample_offset + s);
                        o = splat(0.0f);
                        d = d0;
                        state = 0;
                    } else {
                        state = 3;
                    }
                    continue;
                }
                Frame &f = frames[sp - 1];
                const F4 m0 = rec.mat[2 * f.material + 0];
                const F4 m1 = rec.mat[2 * f.material + 1];
                if (f.awaiting_transmit) {
                    // transmitted_Li arrived: finish this level
                    V3 li = lerp(f.refl, ret, m1.w);
                    ret = xyz(m1) + li * xyz(m0);
                    --sp; // state stays 1: return to the parent
                    continue;
                }
                // reflected_Li arrived: on to the transmission branch, with the
                // frame already in hand (one visit of the frame, not two)
                f.refl = ret;
                const float t = m1.w;
                bool has_child = false;
                if (t > 0.0f) {
                    Bounce b = refract_ray(f.d_in, f.normal, f.point);
                    has_child = b.has_child;
                    if (has_child) {
                        f.awaiting_transmit = 1;
                        o = b.origin;
                        d = b.direction;
                        state = 0;
                    }
                }
                if (!has_child) {
                    V3 li = lerp(f.refl, splat(0.0f), t);
                    ret = xyz(m1) + li * xyz(m0);
                    --sp; // state stays 1
                }
            } else { // state == 2: a hit on a material with t >= 1 (no reflection child)
                Frame &f = frames[sp - 1];
                const F4 m0 = rec.mat[2 * f.material + 0];
                const F4 m1 = rec.mat[2 * f.material + 1];
                const float t = m1.w;
                bool has_child = false;
                if (t > 0.0f) {
                    Bounce b = refract_ray(f.d_in, f.normal, f.point);
                    has_child = b.has_child;
                    if (has_child) {
                        f.awaiting_transmit = 1;
                        o = b.origin;
                        d = b.direction;
                        state = 0;
                    }
                }
                if (!has_child) {
                    V3 li = lerp(f.refl, splat(0.0f), t);
                    ret = xyz(m1) + li * xyz(m0);
                    --sp;
                    state = 1;
                }
            }
        }
        PT_STAMP(bk1);
        PT_STAMP_ADD(t_book, bk1, bk0);
        if (!__any(state == 0))
            break; // every lane of the wave has finished its samples
        // ---- trace + shade for the lanes that stand before a call -----------------
        ++iters;
        if (state == 0) {
            ++rays;
            const Hit h = scan_scene<kScan>(rec, o, d, queue);
            if (h.distance < 0.0f) {
                ret = sky;
                state = 1;
            } else {
                const Surface sf = surface_at(rec, o, d, h);
                const F4 m0 = rec.mat[2 * sf.material + 0];
                const float t = rec.mat[2 * sf.material + 1].w;
                V3 deviation = rand_unit_sphere(rng) * m0.w;
                Frame &f = frames[sp];
                f.d_in = d;
                f.normal = normalize(sf.normal + deviation);
                f.point = sf.point;
                f.refl = splat(0.0f);
                f.material = sf.material;
                f.awaiting_transmit = 0;
                ++sp;
                if (t < 1.0f) {
                    Bounce b = reflect_ray(d, f.normal, f.point);
                    o = b.origin;
                    d = b.direction;
                    state = 0;
                } else {
                    state = 2;
                }
            }
        }
        PT_STAMP(tr1);
        PT_STAMP_ADD(t_trace, tr1, bk1);
    }
#ifdef PT_STAMPS
    PT_STAMP(tree_end);
    if ((threadIdx.x & 63u) == 0u) {
        atomicAdd(&g_stamps[5], t_book);
        atomicAdd(&g_stamps[6], t_trace);
        atomicAdd(&g_stamps[7], tree_end - tree_begin);
    }
#endif
}

} // namespace PT_NS
