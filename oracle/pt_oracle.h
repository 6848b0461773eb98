/*
 * oracle/pt_oracle.h -- TEST INFRASTRUCTURE: interface of the CPU oracle
 * (see pt_oracle.c).  PARITY UNPINNED: nothing in the reference pins it.
 * Loaded through ctypes by tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg only.
 */
#ifndef ORACLE_PT_ORACLE_H
#define ORACLE_PT_ORACLE_H

#include "path_tracer_api.h"

#define ORACLE_RNG_XORSHIFT 0u /* include/ptgpu_rng.h stream (what the GPU uses) */
#define ORACLE_RNG_LIBC 1u     /* glibc random(), the reference's generator */

typedef struct {
    uint32_t spp;           /* NUM_SAMPLES, src/renderer.c:9 */
    uint32_t max_bounces;   /* MAX_BOUNCES, src/renderer.c:8 */
    float fov_rad;          /* FOV, src/renderer.c:7 */
    uint32_t seed;          /* xorshift stream seed (ignored by RNG_LIBC) */
    uint32_t sample_offset; /* index of the first sample (xorshift stream) */
    uint32_t rng_mode;      /* ORACLE_RNG_* */
} oracle_params;

/* src/ray.c:4-73 */
ray_hit oracle_trace_ray(const vec3s origin, const vec3s direction,
                         const scene *world);
void oracle_trace_rays(const scene *world, const float *origins,
                       const float *directions, size_t n, ray_hit *hits);

/* src/renderer.c:92-116 for rows [row_begin, row_end), ascending, one thread.
 * Returns the number of trace_ray calls made. */
uint64_t oracle_render_rows(const scene *world, uint8_t *frame, size_t w,
                            size_t h, size_t row_begin, size_t row_end,
                            const oracle_params *p);

/* The sample sum of src/renderer.c:103-107 continued across calls: accum is
 * one float4 (sum rgb, sample count) per pixel of the w x h frame. */
uint64_t oracle_accumulate_rows(const scene *world, float *accum, size_t w,
                                size_t h, size_t row_begin, size_t row_end,
                                const oracle_params *p);

/* One clamped sample (src/renderer.c:76-90) of pixel (sx, sy). */
void oracle_sample(const scene *world, size_t w, size_t h, int sx, int sy,
                   uint32_t sample, const oracle_params *p, float out_rgb[3]);

/* The rays (origin xyz, direction xyz) of every trace_ray call that sample
 * makes, in call order; returns their number (debugging aid for the tests). */
size_t oracle_sample_rays(const scene *world, size_t w, size_t h, int sx, int sy,
                          uint32_t sample, const oracle_params *p, float *rays6,
                          size_t max_rays);

/* srandom(), for RNG_LIBC runs (src/main.c:140 seeds with time(NULL)). */
void oracle_srandom(unsigned int seed);

/* The same rows as one task each on a corrected row-task pool of
 * `num_threads` workers (the reference's scheduler, src/threadpool.c, minus
 * its lost-wakeup and shutdown defects).  Every `row_step`-th row from
 * row_begin is rendered.  Returns the trace_ray calls made. */
uint64_t oracle_render_mt(const scene *world, uint8_t *frame, size_t w,
                          size_t h, size_t row_begin, size_t row_end,
                          size_t row_step, const oracle_params *p,
                          unsigned int num_threads);

/* The reference's OTHER renderer, shaders/rtx_frag.glsl, restated (pt_glsl.c):
 * the checker of PTGPU_FLAG_GLSL_SEMANTICS.  Rows [row_begin, row_end), row 0 =
 * top; `mean` (may be NULL): unclamped per-pixel means, 3 floats per pixel. */
uint64_t oracle_glsl_render_rows(const scene *world, uint8_t *frame, float *mean, size_t w, size_t h,
                                 size_t row_begin, size_t row_end, uint32_t spp, uint32_t max_bounces,
                                 float fov_rad, uint32_t random_seed);

#endif /* ORACLE_PT_ORACLE_H */
