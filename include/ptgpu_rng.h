/*
 * ptgpu_rng.h -- the sample stream of the MI355X render path (a specification,
 * written as code so that host tools can reproduce any pixel).
 *
 * The reference draws from the process-global glibc random() seeded with
 * time(NULL) (src/renderer.c:6,11-19; src/main.c:140): one serial stream whose
 * assignment to pixels depends on thread scheduling.  A GPU needs a stream
 * that is a pure function of (seed, pixel, sample), so that an image does not
 * depend on the launch geometry, the row-tile sharding or the number of GPUs:
 *
 *   state0(seed, pixel, sample) = mix(mix(seed ^ GOLD*pixel) ^ sample*K2), never 0
 *   draw: xorshift32 (13,17,5);  u = (state >> 8) * 2^-24   in [0, 1)
 *
 * `pixel` is the GLOBAL pixel index y*framew + x, `sample` the global sample
 * index (sample_offset + j).  Each path consumes three draws per surface hit,
 * in x, y, z order, exactly where the reference's rand_unit_sphere() does
 * (src/renderer.c:12-14), including hits on roughness-0 materials.
 *
 * The reference maps random() to [0,1] inclusive by /0x7FFFFFFF; this map is
 * [0,1) on a 2^-24 grid -- a statistically immaterial difference that the
 * statistical parity test (tests/test_parity_gpu.py) bounds.
 */
#ifndef PTGPU_RNG_H
#define PTGPU_RNG_H

#include <stdint.h>

#if defined(__HIPCC__)
#define PTGPU_HD __host__ __device__ __forceinline__
#else
#define PTGPU_HD static inline
#endif

/* 32-bit finaliser (murmur3 fmix32): full avalanche, bijective. */
PTGPU_HD uint32_t ptgpu_mix32(uint32_t h) {
    h ^= h >> 16;
    h *= 0x85EBCA6Bu;
    h ^= h >> 13;
    h *= 0xC2B2AE35u;
    h ^= h >> 16;
    return h;
}

/* Start state of the stream of one (pixel, sample) pair. */
PTGPU_HD uint32_t ptgpu_rng_seed(uint32_t seed, uint32_t pixel,
                                 uint32_t sample) {
    uint32_t h = ptgpu_mix32(seed ^ (pixel * 0x9E3779B1u));
    h = ptgpu_mix32(h ^ (sample * 0x85EBCA77u) ^ 0x5bd1e995u);
    return h ? h : 0x6D2B79F5u; /* xorshift must not start at 0 */
}

/* One xorshift32 step; returns the new state (which is also the output). */
PTGPU_HD uint32_t ptgpu_xorshift32(uint32_t x) {
    x ^= x << 13;
    x ^= x >> 17;
    x ^= x << 5;
    return x;
}

/* u32 -> float in [0,1): top 24 bits, exact in binary32. */
PTGPU_HD float ptgpu_u32_to_unit(uint32_t x) {
    return (float)(x >> 8) * 0x1.0p-24f;
}

/* ---- the OpenGL renderer's stream (PTGPU_FLAG_GLSL_SEMANTICS only) ----------
 * shaders/rtx_frag.glsl:64-75: PCG hash; one uint state per PIXEL, started at
 * pixel_idx + random_seed, bumped by the sample index at the start of every
 * sample and carried from sample to sample (`rng_state += sample_id`,
 * :277-278,289) -- so a pixel's samples must be traced in order. */
PTGPU_HD uint32_t ptgpu_pcg_next(uint32_t *state) {
    *state = *state * 747796405u + 2891336453u;
    uint32_t result = ((*state >> ((*state >> 28) + 4u)) ^ *state) * 277803737u;
    result = (result >> 22) ^ result;
    return result;
}

/* `next_random(state) / 4294967295.0` (:73-75): uint -> float (to nearest), divided
 * by the float nearest to 2^32 - 1, which is 2^32; can be exactly 1. */
PTGPU_HD float ptgpu_pcg_unit(uint32_t *state) {
    return (float)ptgpu_pcg_next(state) / 4294967296.0f;
}

#endif /* PTGPU_RNG_H */
