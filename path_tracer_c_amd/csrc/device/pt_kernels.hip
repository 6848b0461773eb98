// pt_kernels.hip -- the gfx950 kernels of the render path.
//
// What they compute is the reference's render_task (src/renderer.c:92-116)
// for every pixel of a call's row set: the primary ray of src/renderer.c:76-85
// (the same un-jittered ray for every sample of a pixel), `spp` samples of the
// recursive incident_light (src/renderer.c:21-74) over trace_ray
// (src/ray.c:4-73), each clamped, summed in sample order, scaled, truncated
// and stored as RGB8 (src/renderer.c:88,103-114).  One call replaces the
// whole threadpool_add_task / render_task / threadpool_wait_for_tasks loop.
//
//   pt_megakernel     pixel-owner schedule: one work-item per pixel, all its
//                     samples, one launch per frame
//   pt_samplekernel   sample-queue schedule: persistent waves pull
//   + pt_sum_kernel   (pixel, sample) paths from a queue; the ordered
//                     per-pixel sum runs afterwards (same bits)
//   pt_resolve_kernel float4 accumulator -> RGB8
//   pt_trace_kernel   batched trace_ray
//
// Common building blocks (pt_shade.h = one recursion level
// for a lane, pt_owner.h = the pixel-owner loops, pt_trace.h = the scans):
//   * the scene records are scanned with wave-uniform scalar loads (default)
//     or from a per-workgroup LDS copy; three scan forms (plain, screened,
//     BVH) in pt_trace.h;
//   * incident_light's recursion is a forward walk that records one 16-bit
//     material index per bounce in an LDS stack, and an unwind that re-applies
//     `Le + lerp(refl, trans, t) * albedo` from the deepest level up -- the
//     reference's own evaluation order, so results are bit-equal to a
//     sequential evaluation, which a throughput-product formulation is not;
//   * lanes whose path ended park, and settle (unwind, emit, take new work)
//     together when the wave's ballots say enough of them are waiting;
//   * per-lane xorshift32 in a register, re-seeded per (pixel, sample)
//     (include/ptgpu_rng.h), three draws per hit like rand_unit_sphere
//     (src/renderer.c:11-19);
//   * materials with 0 < transparency < 1 make incident_light a binary tree
//     (src/renderer.c:45,56,72); scenes that have one run the `kTree`
//     instantiations of the same kernels: a lane's path follows the reflection,
//     the transmitted ray is put aside and the unwind resumes there (pt_shade.h);
//   * RGB8 stores: a wave's 64 pixels are 192 contiguous bytes; lanes 0..47
//     each assemble one dword with two cross-lane reads and store it.
//
// No MFMA: there is no dense contraction anywhere on this path.
#include "pt_launch.h"
#include "pt_owner.h"
#include "pt_shade.h"
#include "pt_trace.h"
#include "pt_vec.h"
#include "ptgpu_rng.h"

namespace PT_NS {

__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
        v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ uint32_t wave_max(uint32_t v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        uint32_t other = __shfl_xor(v, off, 64);
        v = other > v ? other : v;
    }
    return v;
}

// The scene as uploaded, read from global memory (constant cache / L2).
__device__ __forceinline__ SceneRecords records_in_global_memory(const SceneDesc &scene) {
    SceneRecords rec;
    rec.geom = scene.geom;
    rec.tri = scene.tri;
    rec.mat = scene.mat;
    rec.meta = scene.meta;
    rec.link = scene.link;
    rec.n_obj = scene.n_obj;
    rec.bvh_nodes = scene.bvh_nodes;
    rec.bvh_order = scene.bvh_order;
    rec.bvh_leaf_geom = scene.bvh_leaf_geom;
    rec.bvh_unbounded = scene.bvh_unbounded;
    rec.n_nodes = scene.n_nodes;
    rec.n_unbounded = scene.n_unbounded;
    rec.bvh_root = scene.bvh_root;
    rec.last_sphere = scene.last_sphere;
    return rec;
}

// ---- pieces shared by the kernels ---------------------------------------------
struct WorkgroupScene {
    SceneRecords rec;
    uint16_t *stack;   // this lane's level-0 entry of the bounce stack
    ScreenQueue queue; // this lane's candidate queue (screened scan)
    uint4 *tally;      // one record per wave of the workgroup (count_block)
};

// Stage the scene blob into LDS (kLds) or point at global memory, and carve
// the per-lane bounce stack and candidate queue out of dynamic LDS.
template <bool kLds>
__device__ __forceinline__ WorkgroupScene
stage_scene(const LaunchArgs &a, uint4 *lds_dyn, uint32_t tid, uint32_t nthreads) {
    WorkgroupScene w;
    const uint32_t n_obj = a.scene.n_obj;
    const uint32_t n_tri = a.scene.n_tri;
    const uint32_t n_mat = a.scene.n_mat;
    w.rec.n_obj = n_obj;
    w.tally = lds_dyn;
    lds_dyn += nthreads / 64u; // everything else sits behind the tally records
    uint32_t lds_used16 = 0;
    if (kLds) {
        const uint32_t blob16 =
            (uint32_t)(scene_lds_bytes(n_obj, n_tri, n_mat) / 16u);
        const uint4 *src = reinterpret_cast<const uint4 *>(a.scene.geom);
        for (uint32_t i = tid; i < blob16; i += nthreads)
            lds_dyn[i] = src[i];
        const F4 *base = reinterpret_cast<const F4 *>(lds_dyn);
        w.rec.geom = base;
        w.rec.tri = base + n_obj;
        w.rec.mat = base + n_obj + 4u * n_tri;
        w.rec.meta = reinterpret_cast<const uint32_t *>(base + n_obj + 4u * n_tri +
                                                        2u * n_mat);
        w.rec.link = w.rec.meta + n_obj;
        lds_used16 = blob16;
        if (a.screened == 2u) { // the hierarchy's nodes ride along
            const uint32_t node16 = 2u * a.scene.n_nodes;
            const uint4 *nsrc = reinterpret_cast<const uint4 *>(a.scene.bvh_nodes);
            for (uint32_t i = tid; i < node16; i += nthreads)
                lds_dyn[lds_used16 + i] = nsrc[i];
            lds_used16 += node16;
        }
    } else {
        w.rec.geom = a.scene.geom;
        w.rec.tri = a.scene.tri;
        w.rec.mat = a.scene.mat;
        w.rec.meta = a.scene.meta;
        w.rec.link = a.scene.link;
    }
    w.rec.bvh_nodes = (kLds && a.screened == 2u)
                          ? reinterpret_cast<const F4 *>(lds_dyn) +
                                (scene_lds_bytes(n_obj, n_tri, n_mat) / 16u)
                          : a.scene.bvh_nodes;
    w.rec.bvh_order = a.scene.bvh_order;
    w.rec.bvh_leaf_geom = a.scene.bvh_leaf_geom;
    w.rec.bvh_unbounded = a.scene.bvh_unbounded;
    w.rec.n_nodes = a.scene.n_nodes;
    w.rec.n_unbounded = a.scene.n_unbounded;
    w.rec.bvh_root = a.scene.bvh_root;
    w.rec.last_sphere = a.scene.last_sphere;
    w.stack = reinterpret_cast<uint16_t *>(lds_dyn + lds_used16) + tid;
    // the candidate queues of the screened scan sit behind the bounce stack
    // (2 * nthreads * (max_bounces + 1) bytes, rounded up to 16)
    const uint32_t stack16 = (2u * nthreads * (a.max_bounces + 1u) + 15u) / 16u;
    w.queue.slot =
        reinterpret_cast<uint32_t *>(lds_dyn + lds_used16 + stack16) + tid;
    w.queue.stride = nthreads;
    // and behind them the traversal stacks of the BVH scan
    w.queue.stack = w.queue.slot + kScreenQueue * nthreads;
    __syncthreads();
    return w;
}

// Where local pixel j of the call's row set lives, and its primary ray
// (src/renderer.c:94-100, 76-85).
struct PixelSite {
    uint32_t pixel;     // global pixel index grow * W + sx: RNG stream + ray
    uint64_t out_pixel; // index in the destination buffer
    V3 d0;
};

// (j < num_rows * framew <= 2^32 - 1, checked by the host: 32-bit arithmetic)
__device__ __forceinline__ PixelSite pixel_site(const LaunchArgs &a, uint32_t j) {
    PixelSite ps;
    const uint32_t W = a.framew;
    const uint32_t lrow = j / W;
    const uint32_t sx = j - lrow * W;
    uint32_t grow = a.row_begin + lrow;
    if (a.tile_rows != 0u) {
        const uint32_t tile = lrow / a.tile_rows;
        grow = a.row_begin + tile * a.tile_rows * a.tile_stride +
               (lrow - tile * a.tile_rows);
    }
    ps.pixel = grow * W + sx;
    ps.out_pixel = (uint64_t)(a.compact ? lrow : grow) * W + sx;
    const float fw = (float)a.framew, fh = (float)a.frameh;
    const float aspect = fw / fh;
#if defined(PT_FUSE)
    const float x = mul_add((float)sx / fw, 2.0f, -1.0f);
    const float y = -mul_add((float)grow / fh, 2.0f, -1.0f);
#else
    const float x = (float)sx / fw * 2.0f - 1.0f;
    const float y = -((float)grow / fh * 2.0f - 1.0f);
#endif
    ps.d0 = normalize_dir(V3{x * a.tan_half_fov, y * a.tan_half_fov / aspect, 1.0f});
    return ps;
}

// result * (1.0f / NUM_SAMPLES * 255.0f), truncated (src/renderer.c:108-114).
__device__ __forceinline__ uint32_t pack_rgb8(V3 sum, float samples) {
    const float scale = 1.0f / samples * 255.0f;
    const V3 px = sum * scale;
    const uint32_t r8 = (uint32_t)(uint8_t)px.x;
    const uint32_t g8 = (uint32_t)(uint8_t)px.y;
    const uint32_t b8 = (uint32_t)(uint8_t)px.z;
    return r8 | (g8 << 8) | (b8 << 16);
}

__device__ __forceinline__ void store_rgb8_bytes(uint8_t *frame, uint32_t packed,
                                                 uint64_t out_pixel) {
    uint8_t *dst = frame + 3u * out_pixel;
    dst[0] = (uint8_t)packed;
    dst[1] = (uint8_t)(packed >> 8);
    dst[2] = (uint8_t)(packed >> 16);
}

// One pixel per lane: stored as coalesced dwords when the wave's 64 pixels
// are 192 contiguous, 4-byte aligned bytes, else as single bytes.
__device__ __forceinline__ void store_rgb8(uint8_t *frame, V3 sum, float samples,
                                           uint64_t out_pixel, bool valid,
                                           uint32_t lane) {
    const uint32_t packed = pack_rgb8(sum, samples);
    const uint64_t first = __shfl(out_pixel, 0, 64);
    const bool wave_dense =
        pt_all(valid && out_pixel == first + lane) &&
        ((reinterpret_cast<uintptr_t>(frame + 3u * first) & 3u) == 0u);
    // dword k of the wave's 192 bytes = bytes 4k..4k+3 = tail of pixel 4k/3
    // and head of the next one
    const uint32_t p0 = (4u * lane) / 3u;
    const uint32_t sh = 8u * (4u * lane - 3u * p0);
    const uint32_t lo = __shfl(packed, (int)(p0 & 63u), 64);
    const uint32_t hi = __shfl(packed, (int)((p0 + 1u) & 63u), 64);
    if (wave_dense) {
        if (lane < 48u) {
            const uint64_t both = (uint64_t)lo | ((uint64_t)hi << 24);
            uint32_t *dst = reinterpret_cast<uint32_t *>(frame + 3u * first);
            dst[lane] = (uint32_t)(both >> sh);
        }
    } else if (valid) {
        store_rgb8_bytes(frame, packed, out_pixel);
    }
}

// k lanes per pixel: a wave holds only 64 / k pixels (192 / k bytes), and a
// store of part of a cache line leaves the XCD's L2 as a masked write of its
// own (measured: 45 MB of write traffic for the 6.2 MB C2 frame).  So the
// workgroup (64 * k work-items when the host can arrange it) gathers its
// pixels in LDS -- each wave in its own 128 bytes of the bounce stack's first
// row, free by now -- and its first wave stores them as whole dwords, 192
// contiguous bytes per 64 pixels.  `owner`: this lane holds pixel
// block_first + (tid / k).  All work-items of the workgroup must call this.
__device__ __forceinline__ void
store_rgb8_block(const LaunchArgs &a, uint16_t *stack_row, V3 sum, float samples,
                 uint64_t out_pixel, bool owner, uint32_t tid, uint32_t nthreads,
                 uint32_t k, uint64_t first_local_pixel, uint64_t npix) {
    const uint32_t packed = pack_rgb8(sum, samples);
    const uint32_t block_pixels = nthreads / k;
    const uint32_t wave_pixels = 64u / k;
    // contiguous in the destination: rows follow each other (no tile gaps) and
    // the workgroup's pixels all exist
    const bool contiguous = (a.compact != 0u || a.tile_rows == 0u) &&
                            first_local_pixel + block_pixels <= npix;
    const uint64_t first_out = out_pixel - (tid / k); // of the workgroup, if contiguous
    const uint64_t block_first_out = __shfl(first_out, 0, 64);
    const bool dense = contiguous && (block_pixels & 3u) == 0u &&
                       ((reinterpret_cast<uintptr_t>(a.frame + 3u * block_first_out) & 3u) == 0u);
    if (!dense) { // wave-uniform (same for the whole workgroup, in fact)
        if (owner)
            store_rgb8_bytes(a.frame, packed, out_pixel);
        return;
    }
    uint32_t *stage = reinterpret_cast<uint32_t *>(stack_row - tid); // workgroup's row
    if ((tid & (k - 1u)) == 0u) {
        const uint32_t p = tid / k; // pixel of the workgroup
        stage[(p / wave_pixels) * 32u + p % wave_pixels] = packed;
    }
    __syncthreads();
    // 3 * block_pixels / 4 dwords, stored by the first work-items
    const uint32_t dwords = 3u * block_pixels / 4u;
    uint32_t *dst = reinterpret_cast<uint32_t *>(a.frame + 3u * block_first_out);
    for (uint32_t j = tid; j < dwords; j += nthreads) {
        const uint32_t p0 = (4u * j) / 3u;
        const uint32_t sh = 8u * (4u * j - 3u * p0);
        const uint32_t p1 = p0 + 1u < block_pixels ? p0 + 1u : p0;
        const uint32_t lo = stage[(p0 / wave_pixels) * 32u + p0 % wave_pixels];
        const uint32_t hi = stage[(p1 / wave_pixels) * 32u + p1 % wave_pixels];
        const uint64_t both = (uint64_t)lo | ((uint64_t)hi << 24);
        dst[j] = (uint32_t)(both >> sh);
    }
}

// Adds a workgroup's totals to the statistics.  Device-scope atomics are
// performed memory-side on this chip (the XCDs' L2s do not see each other) and
// cost a 32-byte fabric write each: four per wave were 33 MB per C2 frame,
// five times the frame itself.  So the waves of a workgroup first add up in
// LDS, and the four counters -- adjacent words of one 32-byte record -- are
// then updated by four lanes of ONE atomic instruction, which travels as one
// request.  All work-items of the workgroup must call this (it may barrier).
__device__ __forceinline__ void count_block(Counters *c, uint4 *tally, uint32_t rays,
                                            uint32_t scans, uint32_t samples,
                                            uint32_t iters, uint32_t tid,
                                            uint32_t nthreads) {
    if (c == nullptr) // (wave-uniform, in fact launch-uniform)
        return;
    c += blockIdx.x & (kCounterShards - 1u);
    const uint32_t lane = tid & 63u;
    uint4 mine = make_uint4(wave_sum(rays), wave_sum(scans), wave_sum(samples), wave_max(iters));
    if (nthreads > 64u) {
        if (lane == 0u)
            tally[tid / 64u] = mine;
        __syncthreads();
        if (tid >= 64u)
            return;
        mine = make_uint4(0u, 0u, 0u, 0u);
        for (uint32_t wv = 0; wv < nthreads / 64u; ++wv) {
            const uint4 t = tally[wv];
            mine.x += t.x;
            mine.y += t.y;
            mine.z += t.z;
            mine.w += t.w;
        }
    }
    // Counters = { rays, samples, wave_iters, scans }: lane i adds word i
    static_assert(offsetof(Counters, rays) == 0 && offsetof(Counters, samples) == 8 &&
                      offsetof(Counters, wave_iters) == 16 && offsetof(Counters, scans) == 24,
                  "count_block addresses the counters as four adjacent words");
    if (lane < 4u) {
        const uint32_t v = lane == 0u ? mine.x : lane == 1u ? mine.z : lane == 2u ? mine.w : mine.y;
        // rays, samples, wave_iters, scans
        atomicAdd(reinterpret_cast<unsigned long long *>(c) + lane, (unsigned long long)v);
    }
}

// ---------------------------------------------------------------------------
// Kernel 1: one work-item per pixel (pixel-owner mode).
// ---------------------------------------------------------------------------
// Waves per SIMD the register allocator must leave room for.  The exact build's
// path kernels fit 60 VGPRs unprompted (8 waves); the tolerance build's want 71
// (7 waves) and are held to 64: no spill, 43.9 -> 42.4 ms on configuration 2.
// The branching variants and the screened / BVH pixel-owner variants keep
// their registers (they would spill).
#ifndef PT_MEGA_WAVES
#define PT_MEGA_WAVES 1
#endif
template <bool kLds, bool kTree, int kScan>
__global__ void __launch_bounds__(1024)
    __attribute__((amdgpu_waves_per_eu((kTree || kScan != 0) ? 1 : PT_MEGA_WAVES, 8)))
    pt_megakernel(const LaunchArgs a) {
    extern __shared__ uint4 lds_dyn[]; // 16-byte aligned base
    const uint32_t tid = threadIdx.x;
    const uint32_t nthreads = blockDim.x;
    const WorkgroupScene w = stage_scene<kLds>(a, lds_dyn, tid, nthreads);

    const uint64_t npix = (uint64_t)a.num_rows * a.framew;
    const uint32_t k = a.lanes_per_pixel; // 1, or a power of two <= 64
    const uint64_t t = (uint64_t)blockIdx.x * nthreads + tid;
    const uint64_t j = t / k;             // local pixel
    const uint32_t q = (uint32_t)(t - j * k); // this lane's place among the pixel's k lanes
    const bool valid = j < npix;
    const bool owner = valid && q == 0u;  // holds the sum, reads and writes the pixel
    const PixelSite ps = pixel_site(a, valid ? (uint32_t)j : 0u);
    const V3 sky = V3{a.scene.sky[0], a.scene.sky[1], a.scene.sky[2]};

    // ---- all samples of this pixel -----------------------------------------
    V3 sum = splat(0.0f);
    float had = 0.0f;
    const uint64_t acc_at = a.accum_local ? j : ps.out_pixel;
    if (a.accum != nullptr && owner) {
        const float4 prev = reinterpret_cast<const float4 *>(a.accum)[acc_at];
        sum = V3{prev.x, prev.y, prev.z};
        had = prev.w;
    }
    uint32_t rays = 0, scans = 0, iters = 0;
    const uint32_t spp = valid ? a.spp : 0u;
    uint32_t samples_done = owner ? spp : 0u;
    if (k == 1u) {
        run_paths<kScan, kTree>(w.rec, a, w.stack, nthreads, w.queue, ps.d0, ps.pixel,
                                spp, sky, sum, rays, scans, iters);
    } else {
        samples_done = 0u;
        run_paths_split<kScan, kTree>(w.rec, a, w.stack, nthreads, w.queue, ps.d0, ps.pixel,
                                      spp, k, q, sky, sum, rays, scans, iters, samples_done);
    }

    // ---- write back ------------------------------------------------------------
    if (a.accum != nullptr) {
        if (owner)
            reinterpret_cast<float4 *>(a.accum)[acc_at] =
                make_float4(sum.x, sum.y, sum.z, had + (float)a.spp);
    } else {
        if (k == 1u)
            store_rgb8(a.frame, sum, (float)a.spp, ps.out_pixel, owner, tid & 63u);
        else
            store_rgb8_block(a, w.stack, sum, (float)a.spp, ps.out_pixel, owner, tid, nthreads, k,
                             (uint64_t)blockIdx.x * (nthreads / k), npix);
    }
    count_block(a.counters, w.tally, rays, scans, samples_done, iters, tid, nthreads);
}

// ---- task hand-out shared by the two sample-queue kernels ---------------------
// Wave-uniform cursor over the chunks (64 tasks each) this wave has reserved.
struct ChunkCursor {
    uint32_t pos = 64u;    // tasks of the current chunk already handed out
    uint32_t sample = 0u;  // the current chunk's sample index ...
    uint32_t j0 = 0u;      // ... and first local pixel
    uint32_t next = 0u, limit = 0u; // reserved chunks not yet started
    bool dry = false;      // the global pool has no chunk left
};

// One round of hand-out: the lanes of `free_mask` (this lane: `is_free`) get
// the next tasks of the wave's current chunk in lane order -- a lane's place
// among the free lanes is the popcount of the mask below it.  Returns true if
// this lane got a task: sample index and local pixel j (j >= npix is a padding
// slot of the last chunk of a sample: the lane stays free).  A wave reserves
// a.chunk_batch chunks with one atomic on the global counter.  Callers loop
// while free lanes remain and !dry.
__device__ __forceinline__ bool take_task(ChunkCursor &c, const LaunchArgs &a,
                                          uint32_t *chunk_counter,
                                          unsigned long long free_mask, bool is_free,
                                          uint32_t lane, uint32_t &sample, uint32_t &j) {
    if (c.pos >= 64u) {
        if (c.next >= c.limit) {
            const int leader = __ffsll((long long)free_mask) - 1;
            uint32_t first = 0;
            if ((int)lane == leader)
                first = atomicAdd(chunk_counter, a.chunk_batch);
            first = __shfl(first, leader, 64);
            if (first >= a.num_chunks) {
                c.dry = true;
                return false;
            }
            c.next = first;
            c.limit = first + a.chunk_batch < a.num_chunks ? first + a.chunk_batch : a.num_chunks;
        }
        const uint32_t chunk = c.next++;
        c.sample = chunk / a.chunks_per_sample;
        c.j0 = (chunk - c.sample * a.chunks_per_sample) * 64u;
        c.pos = 0u;
    }
    const uint32_t avail = 64u - c.pos;
    const uint32_t rank = (uint32_t)__popcll(free_mask & ((1ull << lane) - 1ull));
    const bool got = is_free && rank < avail;
    sample = c.sample;
    j = c.j0 + c.pos + rank;
    const uint32_t wanted = (uint32_t)__popcll(free_mask);
    c.pos += wanted < avail ? wanted : avail;
    return got;
}

// A fresh path for local pixel j, sample `sample` (src/renderer.c:76-87).
__device__ __forceinline__ void start_path(const LaunchArgs &a, uint32_t j, uint32_t sample,
                                           PathState &p, uint32_t &out_index) {
    const PixelSite ps = pixel_site(a, j);
    out_index = sample * a.npix_pad + j;
    p.o = splat(0.0f);
    p.d = ps.d0;
    p.r = splat(0.0f);
    p.depth = 0;
    p.rng = ptgpu_rng_seed(a.seed, ps.pixel, a.sample_offset + sample);
    p.live = true;
    p.parked = false;
}

// ---------------------------------------------------------------------------
// Kernel 2: sample-queue mode.  The unit of work is one (pixel, sample) path.
//
// In pixel-owner mode a wave runs until its most expensive pixel has finished
// all its samples; lanes whose pixels see the sky idle (configuration 5: 25 %
// of all lane-passes).  Here every wave is a persistent worker: tasks are
// numbered sample-major (t = sample * npix_pad + local pixel), handed out in
// chunks of 64 from one global counter; inside a wave, the ballot of free
// lanes and its prefix sum (popcount of the lower bits) give each free lane
// the next task of the wave's current chunk, so a lane whose path ends is
// refilled on the next pass and all 64 lanes stay live until the task pool is
// dry.  A finished path stores its clamped sample (float4, 16 bytes) to
// sample_buf[t]; pt_sum_kernel then adds each pixel's samples in sample order,
// which keeps the image bit-identical to pixel-owner mode.
// ---------------------------------------------------------------------------
constexpr uint32_t kChunkBatchMax = 32; // chunks of 64 tasks reserved per atomic, at most

// The scene arrays, the sample buffer and the chunk counter come in as separate
// __restrict__ kernel arguments: this kernel stores to global memory inside
// its loop, and without the no-alias guarantee the compiler must assume a
// store may change the scene, which turns the wave-uniform scalar loads of
// the object scan (s_load, constant cache) into 64-lane vector loads.
template <bool kLds, int kScan, bool kTree>
__global__ void __launch_bounds__(1024)
    pt_samplekernel(const LaunchArgs a, const F4 *__restrict__ scene_geom,
                    const F4 *__restrict__ scene_tri,
                    const F4 *__restrict__ scene_mat,
                    const uint32_t *__restrict__ scene_meta,
                    const uint32_t *__restrict__ scene_link,
                    float4 *__restrict__ out,
                    uint32_t *__restrict__ chunk_counter) {
    extern __shared__ uint4 lds_dyn[];
    const uint32_t tid = threadIdx.x;
    const uint32_t nthreads = blockDim.x;
    const uint32_t lane = tid & 63u;
    WorkgroupScene w = stage_scene<kLds>(a, lds_dyn, tid, nthreads);
    if (!kLds) {
        w.rec.geom = scene_geom;
        w.rec.tri = scene_tri;
        w.rec.mat = scene_mat;
        w.rec.meta = scene_meta;
        w.rec.link = scene_link;
    }
    const V3 sky = V3{a.scene.sky[0], a.scene.sky[1], a.scene.sky[2]};
    const uint32_t npix = a.num_rows * a.framew;

    // (one global counter sustains only ~90 atomics per microsecond chip-wide;
    // per-chunk atomics made the counter, not the ALUs, the bottleneck: hence
    // several chunks per reservation)
    ChunkCursor cursor;

    bool have = false; // this lane owns a task
    uint32_t out_index = 0;
    PathState p;
    p.o = p.d = p.r = splat(0.0f);
    p.depth = 0;
    p.rng = 1u;
    p.live = p.parked = false;
    Branches<kTree> br;
    uint32_t rays = 0, iters = 0, done = 0;

    for (;;) {
        // ---- refill free lanes from the wave's chunk (ballot + prefix sum) ----
        unsigned long long free_mask = pt_ballot(!have);
        while (free_mask != 0ull && !cursor.dry) {
            uint32_t sample, j;
            if (take_task(cursor, a, chunk_counter, free_mask, !have, lane, sample, j) && j < npix) {
                have = true;
                start_path(a, j, sample, p, out_index);
            }
            free_mask = pt_ballot(!have);
        }
        if (pt_ballot(have) == 0ull)
            break; // pool dry and every lane idle

        // ---- one recursion level for the live lanes -------------------------------
        ++iters;
        if (p.live) {
            ++rays;
            advance_path<kScan, kTree>(w.rec, a, w.stack, nthreads, w.queue, sky, p, br);
        }

        // ---- settle parked lanes: unwind, emit the sample, free the lane ----------
        // (kTree: or go on with a transmitted ray that was put aside)
        if (settle_now(p.live, p.parked, a.settle_lanes)) {
            V3 sample;
            if (p.parked && settle_path<kTree>(w.rec, a, w.stack, nthreads, sky, p, br, sample)) {
                out[out_index] = make_float4(sample.x, sample.y, sample.z, 0.0f);
                p.parked = false;
                have = false;
                ++done;
            }
        }
    }
    count_block(a.counters, w.tally, rays, rays, done, iters, tid, nthreads);
}

// ---------------------------------------------------------------------------
// Kernel 2b: sample-queue mode for the BVH scan, with resumable walks.
//
// In pt_samplekernel a pass of the loop traces one whole segment per lane, so
// a wave waits for its longest walk through the hierarchy, pass after pass.
// Here a lane is in one of four states and the wave alternates between two
// activities:
//   walk     every lane in kWalk advances its walk by one quantum
//            (bvh_walk_quantum, pt_trace.h);
//   service  when enough lanes wait for something else (or none walks):
//            lanes with a finished walk shade their hit, which yields the next
//            ray or ends the path; ended paths unwind and store their sample;
//            free lanes fetch a new (pixel, sample) task; every new ray starts
//            its walk.
// Tasks, output and the ordered sum (pt_sum_kernel) are those of
// pt_samplekernel; only the order in which the device gets the work done
// differs, so the image is the same.
// ---------------------------------------------------------------------------
// (five waves per SIMD: 96 VGPRs; the kernel wants ~100 and is measurably
// slower at the four waves that would give it)
#ifndef PT_STREAM_WAVES
#define PT_STREAM_WAVES 5
#endif
template <bool kLds, bool kTree>
__global__ void __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(PT_STREAM_WAVES, 8)))
    pt_streamkernel(const LaunchArgs a, const F4 *__restrict__ scene_geom,
                    const F4 *__restrict__ scene_tri,
                    const F4 *__restrict__ scene_mat,
                    const uint32_t *__restrict__ scene_meta,
                    const uint32_t *__restrict__ scene_link,
                    float4 *__restrict__ out,
                    uint32_t *__restrict__ chunk_counter) {
    extern __shared__ uint4 lds_dyn[];
    const uint32_t tid = threadIdx.x;
    const uint32_t nthreads = blockDim.x;
    const uint32_t lane = tid & 63u;
    WorkgroupScene w = stage_scene<kLds>(a, lds_dyn, tid, nthreads);
    if (!kLds) {
        w.rec.geom = scene_geom;
        w.rec.tri = scene_tri;
        w.rec.mat = scene_mat;
        w.rec.meta = scene_meta;
        w.rec.link = scene_link;
    }
    const V3 sky = V3{a.scene.sky[0], a.scene.sky[1], a.scene.sky[2]};
    const uint32_t npix = a.num_rows * a.framew;

    ChunkCursor cursor;

    // (numbered so that "waits for a service" is one comparison: >= kFree while the
    // task pool has chunks left, >= kHit once it is dry)
    enum : uint32_t { kWalk = 0, kFresh = 1, kFree = 2, kHit = 3, kParked = 4 };
    uint32_t st = kFree;
    uint32_t out_index = 0;
    PathState p;
    p.o = p.d = p.r = splat(0.0f);
    p.depth = 0;
    p.rng = 1u;
    p.live = p.parked = false;
    Branches<kTree> br;
    BvhWalk walk;
    walk.inv = walk.oi = splat(0.0f);
    walk.slack0 = 0.0f;
    walk.u = UnorderedBest{0.0f, -1, 0.0f, -1};
    walk.cur = kBvhDone;
    walk.sp = w.queue.stack;
    walk.wr = w.queue.slot;
    walk.min_index = 0u;
    Hit hit = Hit{-1.0f, -1};
    uint32_t rays = 0, iters = 0, done = 0;
    uint32_t shared = 0; // segments whose hit came from the primary-hit table
#ifdef PT_STAMPS
    unsigned long long t_walk = 0, t_shade = 0, t_settle = 0, t_refill = 0;
    PT_STAMP(k_begin);
#endif
#ifdef PT_BVH_PROFILE
    uint32_t prof[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#elif defined(PT_STAMPS)
    unsigned long long quantum_stamps[3] = {0, 0, 0}; // inner steps, leaf visits, exact tests
    uint32_t *const prof = reinterpret_cast<uint32_t *>(quantum_stamps);
#else
    uint32_t *const prof = nullptr;
#endif

    // every new ray starts its walk (a poisoned one gets the reference's
    // sequential scan at once, see closest_hit_bvh)
    auto begin_walks = [&]() {
        if (st == kFresh) {
            ++rays;
            if (ray_is_poisoned(p.o, p.d)) {
#ifdef PT_BVH_PROFILE
                atomicAdd(&g_bvh_profile[9], 1ull);
#endif
                hit = poisoned_hit(w.rec, p.o, p.d);
                st = kHit;
            } else {
                bvh_walk_first(w.rec, w.queue, p.o, p.d, walk);
                st = kWalk;
            }
        }
    };
    // one recursion level of incident_light for the lanes whose walk has ended
    auto shade_hits = [&]() {
        if (st == kHit) {
            advance_path<2, kTree>(w.rec, a, w.stack, nthreads, w.queue, sky, p, br, true, hit);
            st = p.live ? kFresh : kParked;
        }
    };

    for (;;) {
        // ---- walk quanta, until a shading stage or a service is due -----------------
        // (a loop of its own: a quantum changes the walk's fields only, and as one arm
        // of the outer loop it made the compiler carry the lane's WHOLE state -- ray,
        // stream state, 1 / d ... -- in two register sets and move one into the
        // other, ~54 v_mov, on every trip through a quantum)
        bool full_service;
        for (;;) {
            if (pt_any(st == kFresh))
                begin_walks();
            if (st == kWalk && bvh_walk_idle(w.queue, walk)) {
                if (bvh_walk_finish(w.rec, w.queue, p.o, p.d, walk, hit))
                    st = kHit;
#ifdef PT_BVH_PROFILE
                else
                    atomicAdd(&g_bvh_profile[8], 1ull);
#endif
            }
            const unsigned long long walking = pt_ballot(st == kWalk);
            const unsigned long long waiting = pt_ballot(st >= (cursor.dry ? kHit : kFree));
            full_service = walking == 0ull || (uint32_t)__popcll(waiting) >= a.settle_lanes;
            if (full_service || (uint32_t)__popcll(pt_ballot(st == kHit)) >= a.shade_lanes)
                break;
            ++iters;
            PT_STAMP(q0);
            bvh_walk_quantum(w.rec, w.queue, p.o, p.d, st == kWalk, a.walk_budget, a.leaf_lanes,
                             a.resolve_lanes, walk, prof);
            PT_STAMP(q1);
            PT_STAMP_ADD(t_walk, q1, q0);
        }
        if (!full_service) {
            // ---- quick stage: finished walks are shaded and their next rays start,
            // which is most of what waits (3.4 segments per path on C5); ended
            // paths and free lanes wait for a full service
            PT_STAMP(sq0);
            shade_hits();
            PT_STAMP(sq1);
            PT_STAMP_ADD(t_shade, sq1, sq0);
        } else {
            // ---- service -----------------------------------------------------------
            // (twice over when the refill has handed out enough tasks that
            // start at a known primary hit: they are shaded at once instead
            // of idling until the next service)
#pragma nounroll
            for (int round = 0; round < 2; ++round) {
                PT_BVH_COUNT(6, st == kHit);
                PT_STAMP(sv0);
                shade_hits();
                PT_STAMP(sv1);
                PT_STAMP_ADD(t_shade, sv1, sv0);
                if (st == kParked) {
                    V3 sample;
                    if (settle_path<kTree>(w.rec, a, w.stack, nthreads, sky, p, br, sample)) {
                        out[out_index] = make_float4(sample.x, sample.y, sample.z, 0.0f);
                        p.parked = false;
                        st = kFree;
                        ++done;
                    } else {
                        st = kFresh; // a transmitted ray that was put aside: its walk starts below
                    }
                }
                PT_STAMP(sv2);
                PT_STAMP_ADD(t_settle, sv2, sv1);
                // refill free lanes from the wave's chunk (as pt_samplekernel does)
                unsigned long long free_mask = pt_ballot(st == kFree);
                while (free_mask != 0ull && !cursor.dry) {
                    uint32_t sample, j;
                    if (take_task(cursor, a, chunk_counter, free_mask, st == kFree, lane, sample, j) &&
                        j < npix) {
                        st = kFresh;
                        start_path(a, j, sample, p, out_index);
                        if (a.primary_hits != nullptr) { // the first segment's hit is known
                            const float2 known = reinterpret_cast<const float2 *>(a.primary_hits)[j];
                            hit.distance = known.x;
                            hit.object = __builtin_bit_cast(int, known.y);
                            st = kHit;
                            ++rays;
                            ++shared;
                        }
                    }
                    free_mask = pt_ballot(st == kFree);
                }
                PT_STAMP(sv3);
                PT_STAMP_ADD(t_refill, sv3, sv2);
                if (a.primary_hits == nullptr ||
                    (uint32_t)__popcll(pt_ballot(st == kHit)) < a.reshade_lanes)
                    break;
            }
            if (pt_ballot(st != kFree) == 0ull)
                break; // pool dry and every lane idle
        }
    }
#ifdef PT_STAMPS
    {
        PT_STAMP(k_end);
        if (lane == 0u) {
            atomicAdd(&g_stamps[0], t_walk);
            atomicAdd(&g_stamps[1], t_shade);
            atomicAdd(&g_stamps[2], t_settle);
            atomicAdd(&g_stamps[3], k_end - k_begin);
            atomicAdd(&g_stamps[4], t_refill);
            atomicAdd(&g_stamps[5], quantum_stamps[0]);
            atomicAdd(&g_stamps[6], quantum_stamps[1]);
            atomicAdd(&g_stamps[7], quantum_stamps[2]);
        }
    }
#endif
#ifdef PT_BVH_PROFILE
    if (lane == 0u)
        for (int k = 0; k < 12; ++k)
            atomicAdd(&g_bvh_profile[k], (unsigned long long)prof[k]);
#endif
    count_block(a.counters, w.tally, rays, rays - shared, done, iters, tid, nthreads);
}

#include "pt_duo.h"

// The closest hit of every pixel's primary ray, once per frame.  The reference
// sends the same un-jittered ray for every sample of a pixel
// (src/renderer.c:96-107), so pt_streamkernel can start each (pixel, sample)
// task at that hit instead of walking the hierarchy again -- a third of
// configuration 5's segments.  Eye rays of neighbouring pixels walk alike, so
// the whole-walk form is at its best here.
__global__ void __launch_bounds__(256)
    pt_primary_kernel(const LaunchArgs a, float2 *__restrict__ primary_hits) {
    extern __shared__ uint4 lds_dyn[];
    const uint32_t npix = a.num_rows * a.framew;
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    const SceneRecords rec = records_in_global_memory(a.scene);
    ScreenQueue queue;
    queue.slot = reinterpret_cast<uint32_t *>(lds_dyn) + threadIdx.x;
    queue.stride = blockDim.x;
    queue.stack = queue.slot + kScreenQueue * blockDim.x;
    const PixelSite ps = pixel_site(a, j < npix ? j : 0u);
    const Hit h = closest_hit_bvh(rec, splat(0.0f), ps.d0, queue);
    if (j < npix)
        primary_hits[j] = make_float2(h.distance, __builtin_bit_cast(float, h.object));
}

// Adds the `spp` buffered samples of each pixel, in sample order, onto the
// pixel's running sum (src/renderer.c:103-107), then stores either the
// accumulator (float4: sums + sample count) or, on a frame's last pass, RGB8.
__global__ void __launch_bounds__(256) pt_sum_kernel(const LaunchArgs a) {
    const uint32_t npix = a.num_rows * a.framew;
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = j < npix;
    const PixelSite ps = pixel_site(a, valid ? (uint32_t)j : 0u);
    const float4 *buf = reinterpret_cast<const float4 *>(a.sample_buf);
    float4 *acc = reinterpret_cast<float4 *>(a.accum);
    // accum is addressed like the destination frame when it is the caller's
    // (out_pixel), and by local pixel when it is the library's scratch
    const uint64_t at = a.accum_local ? j : ps.out_pixel;

    V3 sum = splat(0.0f);
    float had = 0.0f;
    if (valid && !a.first_pass) {
        const float4 prev = acc[at];
        sum = V3{prev.x, prev.y, prev.z};
        had = prev.w;
    }
    if (valid) {
        for (uint32_t s = 0; s < a.spp; ++s) {
            const float4 v = buf[(uint64_t)s * a.npix_pad + j];
            sum = sum + V3{v.x, v.y, v.z};
        }
    }
    if (a.frame != nullptr) {
        store_rgb8(a.frame, sum, had + (float)a.spp, ps.out_pixel, valid,
                   threadIdx.x & 63u);
    } else if (valid) {
        acc[at] = make_float4(sum.x, sum.y, sum.z, had + (float)a.spp);
    }
}

// frame[i] = (uint8)(sum * (1.0f / samples * 255.0f)), 4 pixels (12 bytes =
// 3 dwords) per work-item so the stores are whole dwords.
__global__ void __launch_bounds__(256)
    pt_resolve_kernel(const float4 *__restrict__ accum,
                      uint8_t *__restrict__ frame, uint64_t npixels) {
    const uint64_t quad = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t first = quad * 4u;
    if (first >= npixels)
        return;
    uint8_t bytes[12];
    const uint32_t count = (npixels - first) < 4u ? (uint32_t)(npixels - first) : 4u;
#pragma unroll
    for (uint32_t k = 0; k < 4u; ++k) {
        if (k < count) {
            const float4 s = accum[first + k];
            const float scale = 1.0f / s.w * 255.0f;
            bytes[3 * k + 0] = (uint8_t)(s.x * scale);
            bytes[3 * k + 1] = (uint8_t)(s.y * scale);
            bytes[3 * k + 2] = (uint8_t)(s.z * scale);
        } else {
            bytes[3 * k + 0] = bytes[3 * k + 1] = bytes[3 * k + 2] = 0;
        }
    }
    uint8_t *dst = frame + 3u * first;
    if (count == 4u && ((reinterpret_cast<uintptr_t>(dst) & 3u) == 0u)) {
        uint32_t w0 = bytes[0] | (bytes[1] << 8) | (bytes[2] << 16) | ((uint32_t)bytes[3] << 24);
        uint32_t w1 = bytes[4] | (bytes[5] << 8) | (bytes[6] << 16) | ((uint32_t)bytes[7] << 24);
        uint32_t w2 = bytes[8] | (bytes[9] << 8) | (bytes[10] << 16) | ((uint32_t)bytes[11] << 24);
        uint32_t *d32 = reinterpret_cast<uint32_t *>(dst);
        d32[0] = w0;
        d32[1] = w1;
        d32[2] = w2;
    } else {
        for (uint32_t k = 0; k < 3u * count; ++k)
            dst[k] = bytes[k];
    }
}

// One work-item per query ray: the batched trace_ray of include/ray.h:14-15,
// through any of the three scan forms (kScan as in scan_scene).
template <int kScan>
__global__ void __launch_bounds__(256)
    pt_trace_kernel(const SceneDesc scene, const float *__restrict__ origins,
                    const float *__restrict__ directions, uint64_t n,
                    DeviceRayHit *__restrict__ hits) {
    extern __shared__ uint4 lds_dyn[];
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    // every lane walks the object loop (it is wave-uniform); lanes past the
    // end trace a dummy ray and store nothing
    const bool valid = i < n;
    const uint64_t at = valid ? i : 0;
    const SceneRecords rec = records_in_global_memory(scene);
    ScreenQueue queue;
    queue.slot = reinterpret_cast<uint32_t *>(lds_dyn) + threadIdx.x;
    queue.stride = blockDim.x;
    queue.stack = queue.slot + kScreenQueue * blockDim.x;
    // Rays in and hits out are arrays of 12- and 40-byte records.  A wave's 64
    // of them are 768 / 2560 contiguous bytes: moved as whole 16-byte words
    // (48 / 160 of them per wave) and re-dealt to the lanes through LDS,
    // because with a small scene this kernel is bound by HBM (64 bytes per
    // ray), and 4-byte accesses with a 12- or 40-byte stride waste it.
    // `staged` (wave-uniform): the wave is full and the arrays are 16-byte
    // aligned; otherwise every lane moves its own record word by word.
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t wave_first = i - lane;
    const bool staged = wave_first + 64u <= n &&
                        ((reinterpret_cast<uintptr_t>(origins) | reinterpret_cast<uintptr_t>(directions) |
                          reinterpret_cast<uintptr_t>(hits)) & 15u) == 0u;
    // Staging area, 640 words per wave.  Plain scan: a contiguous run.
    // Screened / BVH scan: the wave's share of the candidate queues and
    // stacks (idle before and after the scan) -- word j sits in row j / 64 of
    // the block-wide rows, at the wave's 64 columns -- so that staging costs
    // those kernels no LDS of its own (10 KB more per workgroup cost the BVH
    // form 15 %).
    uint32_t *const lds_words = reinterpret_cast<uint32_t *>(lds_dyn);
    const uint32_t wave_column = threadIdx.x & ~63u;
    auto stage_word = [&](uint32_t j) -> uint32_t * {
        if (kScan == 0)
            return lds_words + 10u * wave_column + j;
        return lds_words + (j >> 6) * blockDim.x + wave_column + (j & 63u);
    };
    V3 o, d;
    if (staged) {
        const uint4 *src_o = reinterpret_cast<const uint4 *>(origins + 3u * wave_first);
        const uint4 *src_d = reinterpret_cast<const uint4 *>(directions + 3u * wave_first);
        if (lane < 48u) {
            *reinterpret_cast<uint4 *>(stage_word(4u * lane)) = src_o[lane];
            *reinterpret_cast<uint4 *>(stage_word(192u + 4u * lane)) = src_d[lane];
        }
        __builtin_amdgcn_wave_barrier();
        auto word = [&](uint32_t j) { return __builtin_bit_cast(float, *stage_word(j)); };
        o = V3{word(3u * lane), word(3u * lane + 1u), word(3u * lane + 2u)};
        d = V3{word(192u + 3u * lane), word(192u + 3u * lane + 1u), word(192u + 3u * lane + 2u)};
        __builtin_amdgcn_wave_barrier();
    } else {
        o = V3{origins[3 * at], origins[3 * at + 1], origins[3 * at + 2]};
        d = V3{directions[3 * at], directions[3 * at + 1], directions[3 * at + 2]};
    }
    const Hit h = scan_scene<kScan>(rec, o, d, queue);
    DeviceRayHit out;
    out.distance = h.distance;
    out.point[0] = out.point[1] = out.point[2] = 0.0f;
    out.normal[0] = out.normal[1] = out.normal[2] = 0.0f;
    out.pad = 0;
    out.material = 0;
    if (h.object >= 0) {
        const Surface sf = surface_at(rec, o, d, h);
        out.point[0] = sf.point.x;
        out.point[1] = sf.point.y;
        out.point[2] = sf.point.z;
        out.normal[0] = sf.normal.x;
        out.normal[1] = sf.normal.y;
        out.normal[2] = sf.normal.z;
        out.material = sf.material;
    }
    if (staged) {
        static_assert(sizeof(DeviceRayHit) == 40, "ray_hit is 10 words");
        uint32_t words[10];
        __builtin_memcpy(words, &out, sizeof out);
#pragma unroll
        for (uint32_t k = 0; k < 10u; ++k)
            *stage_word(10u * lane + k) = words[k];
        __builtin_amdgcn_wave_barrier();
        uint4 *dst = reinterpret_cast<uint4 *>(hits + wave_first);
        dst[lane] = *reinterpret_cast<const uint4 *>(stage_word(4u * lane));
        dst[64u + lane] = *reinterpret_cast<const uint4 *>(stage_word(256u + 4u * lane));
        if (lane < 32u)
            dst[128u + lane] = *reinterpret_cast<const uint4 *>(stage_word(512u + 4u * lane));
    } else if (valid) {
        hits[i] = out;
    }
}

// ---------------------------------------------------------------------------
// Launchers (called from the C-ABI layer).
// ---------------------------------------------------------------------------
template <bool kLds, bool kTree, int kScan>
static hipError_t launch_variant(const LaunchArgs &a, uint32_t grid,
                                 uint32_t block, size_t lds_bytes,
                                 hipStream_t stream) {
    auto kernel = pt_megakernel<kLds, kTree, kScan>;
    if (lds_bytes > 64u * 1024u) {
        hipError_t e = hipFuncSetAttribute(
            reinterpret_cast<const void *>(kernel),
            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess)
            return e;
    }
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), lds_bytes, stream, a);
    return hipGetLastError();
}

size_t megakernel_lds_bytes(const LaunchArgs &a, uint32_t block, bool tree) {
    size_t bytes = (size_t)16 * (block / 64u); // tally records (count_block)
    if (a.scene_in_lds) {
        bytes += scene_lds_bytes(a.scene.n_obj, a.scene.n_tri, a.scene.n_mat);
        if (a.screened == 2u)
            bytes += (size_t)32 * a.scene.n_nodes;
    }
    (void)tree; // (same layout: what a branching level puts aside is in private memory)
    bytes += ((size_t)2 * block * ((size_t)a.max_bounces + 1u) + 15u) & ~(size_t)15u;
    if (a.screened)
        bytes += (size_t)4 * block * kScreenQueue;
    if (a.screened == 2u)
        bytes += (size_t)4 * block * (a.scene.bvh_depth + 2u);
    return (bytes + 15u) & ~(size_t)15u;
}

hipError_t launch_megakernel(const LaunchArgs &a, uint32_t block, bool tree,
                             hipStream_t stream) {
    const uint64_t npix = (uint64_t)a.num_rows * a.framew;
    if (npix == 0 || a.spp == 0)
        return hipSuccess;
    const uint64_t lanes = npix * (a.lanes_per_pixel == 0 ? 1u : a.lanes_per_pixel);
    const uint32_t grid = (uint32_t)((lanes + block - 1) / block);
    const size_t lds = megakernel_lds_bytes(a, block, tree);
    if (tree && a.screened == 2u)
        return a.scene_in_lds
                   ? launch_variant<true, true, 2>(a, grid, block, lds, stream)
                   : launch_variant<false, true, 2>(a, grid, block, lds, stream);
    if (tree && a.screened == 1u)
        return a.scene_in_lds
                   ? launch_variant<true, true, 1>(a, grid, block, lds, stream)
                   : launch_variant<false, true, 1>(a, grid, block, lds, stream);
    if (tree)
        return a.scene_in_lds
                   ? launch_variant<true, true, 0>(a, grid, block, lds, stream)
                   : launch_variant<false, true, 0>(a, grid, block, lds, stream);
    if (a.screened == 2u)
        return a.scene_in_lds
                   ? launch_variant<true, false, 2>(a, grid, block, lds, stream)
                   : launch_variant<false, false, 2>(a, grid, block, lds, stream);
    if (a.screened == 1u)
        return a.scene_in_lds
                   ? launch_variant<true, false, 1>(a, grid, block, lds, stream)
                   : launch_variant<false, false, 1>(a, grid, block, lds, stream);
    return a.scene_in_lds
               ? launch_variant<true, false, 0>(a, grid, block, lds, stream)
               : launch_variant<false, false, 0>(a, grid, block, lds, stream);
}

template <bool kLds, int kScan, bool kStream, bool kTree>
static hipError_t launch_sample_variant(const LaunchArgs &a, uint32_t grid,
                                        uint32_t block, size_t lds_bytes,
                                        hipStream_t stream) {
    auto kernel = kStream ? pt_streamkernel<kLds, kTree> : pt_samplekernel<kLds, kScan, kTree>;
    if (lds_bytes > 64u * 1024u) {
        hipError_t e = hipFuncSetAttribute(
            reinterpret_cast<const void *>(kernel),
            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess)
            return e;
    }
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), lds_bytes, stream, a,
                       a.scene.geom, a.scene.tri, a.scene.mat, a.scene.meta,
                       a.scene.link, reinterpret_cast<float4 *>(a.sample_buf),
                       a.chunk_counter);
    return hipGetLastError();
}

static hipError_t launch_sample_dispatch(const LaunchArgs &a, uint32_t grid, uint32_t block,
                                         size_t lds, hipStream_t stream);

hipError_t launch_samplekernel(const LaunchArgs &a, uint32_t block,
                               uint32_t max_waves, hipStream_t stream) {
    if (a.num_chunks == 0)
        return hipSuccess;
    // persistent workers: no more waves than the device keeps resident, and
    // no more than there are chunks
    const uint32_t waves_per_block = block / 64u;
    uint64_t waves = a.num_chunks < max_waves ? a.num_chunks : max_waves;
    uint32_t grid = (uint32_t)((waves + waves_per_block - 1) / waves_per_block);
    if (grid == 0)
        grid = 1;
    const size_t lds = megakernel_lds_bytes(a, block, false);
    // Chunks per reservation: 32 keep the one global counter out of the way
    // (~90 atomics per microsecond is all it sustains), but a pass of few
    // chunks must still be spread over all the waves, several reservations
    // each, or most of the device idles while a few waves work them off.
    LaunchArgs b = a;
    uint64_t batch = a.num_chunks / ((uint64_t)grid * waves_per_block * 8u);
    b.chunk_batch = (uint32_t)(batch < 1u ? 1u : (batch > kChunkBatchMax ? kChunkBatchMax : batch));
    return launch_sample_dispatch(b, grid, block, lds, stream);
}

// Two paths per lane (pt_duo.h): non-branching scenes, records in global memory,
// one wave per workgroup.
static hipError_t launch_duo(const LaunchArgs &a, uint32_t grid, hipStream_t stream) {
    const size_t lds_bytes = duo_lds_bytes(a.max_bounces, a.scene.bvh_depth);
    if (lds_bytes > 64u * 1024u) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(pt_duokernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess)
            return e;
    }
    hipLaunchKernelGGL(pt_duokernel, dim3(grid), dim3(64), lds_bytes, stream, a, a.scene.geom,
                       a.scene.tri, a.scene.mat, a.scene.meta, a.scene.link,
                       reinterpret_cast<float4 *>(a.sample_buf), a.chunk_counter);
    return hipGetLastError();
}

template <bool kTree>
static hipError_t launch_sample_scan(const LaunchArgs &a, uint32_t grid, uint32_t block,
                                     size_t lds, hipStream_t stream) {
    if (!kTree && a.screened == 2u && a.stream_walks == 2u && !a.scene_in_lds && block == 64u)
        return launch_duo(a, grid, stream);
    if (a.screened == 2u && a.stream_walks)
        return a.scene_in_lds
                   ? launch_sample_variant<true, 2, true, kTree>(a, grid, block, lds, stream)
                   : launch_sample_variant<false, 2, true, kTree>(a, grid, block, lds, stream);
    if (a.screened == 2u)
        return a.scene_in_lds
                   ? launch_sample_variant<true, 2, false, kTree>(a, grid, block, lds, stream)
                   : launch_sample_variant<false, 2, false, kTree>(a, grid, block, lds, stream);
    if (a.screened == 1u)
        return a.scene_in_lds
                   ? launch_sample_variant<true, 1, false, kTree>(a, grid, block, lds, stream)
                   : launch_sample_variant<false, 1, false, kTree>(a, grid, block, lds, stream);
    return a.scene_in_lds
               ? launch_sample_variant<true, 0, false, kTree>(a, grid, block, lds, stream)
               : launch_sample_variant<false, 0, false, kTree>(a, grid, block, lds, stream);
}

static hipError_t launch_sample_dispatch(const LaunchArgs &a, uint32_t grid, uint32_t block,
                                         size_t lds, hipStream_t stream) {
    return a.tree ? launch_sample_scan<true>(a, grid, block, lds, stream)
                  : launch_sample_scan<false>(a, grid, block, lds, stream);
}

hipError_t launch_primary(const LaunchArgs &a, float *primary_hits, hipStream_t stream) {
    const uint64_t npix = (uint64_t)a.num_rows * a.framew;
    if (npix == 0)
        return hipSuccess;
    const uint32_t grid = (uint32_t)((npix + 255u) / 256u);
    const size_t lds = (size_t)4 * 256u * (kScreenQueue + a.scene.bvh_depth + 2u);
    if (lds > 64u * 1024u) { // a hierarchy deeper than 50 levels
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(pt_primary_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess)
            return e;
    }
    hipLaunchKernelGGL(pt_primary_kernel, dim3(grid), dim3(256), lds, stream, a,
                       reinterpret_cast<float2 *>(primary_hits));
    return hipGetLastError();
}

hipError_t launch_sum(const LaunchArgs &a, hipStream_t stream) {
    const uint64_t npix = (uint64_t)a.num_rows * a.framew;
    if (npix == 0)
        return hipSuccess;
    const uint32_t grid = (uint32_t)((npix + 255u) / 256u);
    hipLaunchKernelGGL(pt_sum_kernel, dim3(grid), dim3(256), 0, stream, a);
    return hipGetLastError();
}

#if defined(PT_FARREJECT_COUNT) && !defined(PT_FAST_MATH)
extern "C" int ptgpu_debug_farreject(unsigned long long out[4], int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_farreject), sizeof(unsigned long long) * 4) != hipSuccess)
        return -1;
    if (reset) {
        unsigned long long zero[4] = {0, 0, 0, 0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_farreject), zero, sizeof zero) != hipSuccess)
            return -1;
    }
    return 0;
}
#endif

#if defined(PT_STAMPS) && !defined(PT_FAST_MATH)
extern "C" int ptgpu_debug_stamps(unsigned long long out[8], int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * 8) != hipSuccess)
        return -1;
    if (reset) {
        unsigned long long zero[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), zero, sizeof zero) != hipSuccess)
            return -1;
    }
    return 0;
}
#endif

#if defined(PT_SCAN_AUDIT) && !defined(PT_FAST_MATH)
extern "C" int ptgpu_debug_scan_audit(unsigned long long out[8], int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_scan_audit), sizeof(unsigned long long) * 8) != hipSuccess)
        return -1;
    if (reset) {
        unsigned long long zero[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_scan_audit), zero, sizeof zero) != hipSuccess)
            return -1;
    }
    return 0;
}
#endif

#if defined(PT_BVH_PROFILE) && !defined(PT_FAST_MATH)
extern "C" int ptgpu_debug_bvh_profile(unsigned long long out[12], int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_bvh_profile), sizeof(unsigned long long) * 12) != hipSuccess)
        return -1;
    if (reset) {
        unsigned long long zero[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_bvh_profile), zero, sizeof zero) != hipSuccess)
            return -1;
    }
    return 0;
}
#endif

hipError_t launch_resolve(const float *accum, uint8_t *frame, uint64_t npixels,
                          hipStream_t stream) {
    if (npixels == 0)
        return hipSuccess;
    const uint64_t quads = (npixels + 3u) / 4u;
    const uint32_t grid = (uint32_t)((quads + 255u) / 256u);
    hipLaunchKernelGGL(pt_resolve_kernel, dim3(grid), dim3(256), 0, stream,
                       reinterpret_cast<const float4 *>(accum), frame, npixels);
    return hipGetLastError();
}

#ifndef PT_FAST_MATH
// One workgroup per frame row of the call's row set: copies the row from its
// shard in the staging buffer (pt_launch.h, ScatterArgs).  16-byte moves when
// source and destination allow, bytes otherwise.  HBM-bound: 3*W bytes read +
// written per row.
__global__ void __launch_bounds__(256) pt_scatter_kernel(const ScatterArgs s) {
    const uint32_t i = blockIdx.x; // row of the set
    if (i >= s.num_rows)
        return;
    const uint32_t tile = i / s.tile_rows;
    const uint32_t shard = tile % s.shards;
    const uint32_t local = (tile / s.shards) * s.tile_rows + (i - tile * s.tile_rows);
    const uint8_t *src = s.stage + s.shard_offset[shard] + (uint64_t)local * s.row_bytes;
    uint8_t *dst = s.frame + (uint64_t)(s.row_begin + i) * s.row_bytes;
    const uintptr_t both = reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst);
    uint32_t done = 0;
    if ((both & 15u) == 0u) {
        const uint32_t n16 = s.row_bytes / 16u;
        const uint4 *s16 = reinterpret_cast<const uint4 *>(src);
        uint4 *d16 = reinterpret_cast<uint4 *>(dst);
        for (uint32_t k = threadIdx.x; k < n16; k += blockDim.x)
            d16[k] = s16[k];
        done = n16 * 16u;
    } else if ((both & 3u) == 0u) {
        const uint32_t n4 = s.row_bytes / 4u;
        const uint32_t *s4 = reinterpret_cast<const uint32_t *>(src);
        uint32_t *d4 = reinterpret_cast<uint32_t *>(dst);
        for (uint32_t k = threadIdx.x; k < n4; k += blockDim.x)
            d4[k] = s4[k];
        done = n4 * 4u;
    }
    for (uint32_t k = done + threadIdx.x; k < s.row_bytes; k += blockDim.x)
        dst[k] = src[k];
}

hipError_t launch_scatter_tiles(const ScatterArgs &s, hipStream_t stream) {
    if (s.num_rows == 0 || s.row_bytes == 0)
        return hipSuccess;
    hipLaunchKernelGGL(pt_scatter_kernel, dim3(s.num_rows), dim3(256), 0, stream, s);
    return hipGetLastError();
}
#endif

hipError_t launch_trace(const SceneDesc &scene, const float *origins,
                        const float *directions, uint64_t n,
                        DeviceRayHit *hits, int scan, hipStream_t stream) {
    if (n == 0)
        return hipSuccess;
    const uint32_t grid = (uint32_t)((n + 255u) / 256u);
    // rows of 256 words: candidate queues + traversal stacks (screened / BVH
    // scan), doubling as the waves' staging areas (10 rows)
    const uint32_t rows = scan == 0 ? 10u : kScreenQueue + scene.bvh_depth + 2u;
    const size_t lds = (size_t)4 * 256u * rows;
    if (lds > 64u * 1024u) { // a hierarchy deeper than 50 levels
        const void *kernel = scan == 2   ? reinterpret_cast<const void *>(pt_trace_kernel<2>)
                             : scan == 1 ? reinterpret_cast<const void *>(pt_trace_kernel<1>)
                                         : reinterpret_cast<const void *>(pt_trace_kernel<0>);
        hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess)
            return e;
    }
    if (scan == 2)
        hipLaunchKernelGGL(pt_trace_kernel<2>, dim3(grid), dim3(256), lds, stream,
                           scene, origins, directions, n, hits);
    else if (scan == 1)
        hipLaunchKernelGGL(pt_trace_kernel<1>, dim3(grid), dim3(256), lds, stream,
                           scene, origins, directions, n, hits);
    else
        hipLaunchKernelGGL(pt_trace_kernel<0>, dim3(grid), dim3(256), lds, stream,
                           scene, origins, directions, n, hits);
    return hipGetLastError();
}

} // namespace PT_NS
