/*
 * ptgpu.h -- explicit C-ABI of the MI355X (gfx950) render path.
 *
 * The reference's CPU render path is
 *     threadpool_add_task(pool, render_task, &args[y]) for every row y
 *     threadpool_wait_for_tasks(pool)
 * (include/threadpool.h:15-20, include/renderer.h:6-13; the task body is
 * src/renderer.c:92-116 calling src/ray.c:4-73).  That surface is kept, see
 * path_tracer_api.h.  It cannot express samples per pixel, bounce limit, seed
 * or field of view (compile-time macros in src/renderer.c:6-9) nor device
 * residency, so this header adds the entry points a binding would use:
 *
 *   reference interface replaced            entry point here
 *   --------------------------------------  ---------------------------------
 *   scene* argument of render_task          ptgpu_scene_upload / _free
 *     (include/renderer.h:10)
 *   one frame of render_task rows           ptgpu_render_frame  (host frame)
 *     (src/renderer.c:92-116)               ptgpu_render_device (HBM frame)
 *   the per-pixel sample sum                ptgpu_accumulate_device +
 *     (src/renderer.c:103-108)              ptgpu_resolve_device (progressive)
 *   trace_ray (include/ray.h:14-15)         ptgpu_trace_rays (batched)
 *   NUM_SAMPLES/MAX_BOUNCES/FOV macros      ptgpu_params
 *     (src/renderer.c:7-9)
 *
 * Plain C types only; no exceptions cross the boundary; every function
 * returns 0 or a negative PTGPU_ERR_* and leaves a message retrievable with
 * ptgpu_last_error().  There is no CPU fallback anywhere behind this header:
 * without a usable HIP device every compute call fails with
 * PTGPU_ERR_NO_DEVICE.
 */
#ifndef PTGPU_H
#define PTGPU_H

#include "path_tracer_api.h"

#ifdef __cplusplus
extern "C" {
#endif

#define PTGPU_OK 0
#define PTGPU_ERR_INVALID (-1)   /* bad argument */
#define PTGPU_ERR_NO_DEVICE (-2) /* no HIP device / runtime unusable */
#define PTGPU_ERR_HIP (-3)       /* a HIP call failed; see ptgpu_last_error */
#define PTGPU_ERR_NOMEM (-4)
#define PTGPU_ERR_LIMIT (-5)     /* scene or parameters exceed a kernel limit */

/* flags */
#define PTGPU_FLAG_COMPACT_ROWS 0x1u  /* output row i of the call's row set is
                                         stored at row i of the destination
                                         (a shard buffer) instead of at its
                                         global row (a full frame) */
#define PTGPU_FLAG_SCENE_GLOBAL 0x2u  /* scheduling only: scan the scene records
                                         from global memory with wave-uniform
                                         scalar loads (constant cache / L2).
                                         This is what is chosen when neither
                                         SCENE flag is given: it measured
                                         faster than the LDS copy on every
                                         configuration */
#define PTGPU_FLAG_SCENE_LDS 0x40u    /* scheduling only: stage the scene into
                                         LDS once per workgroup and scan it
                                         with broadcast LDS reads (falls back
                                         to the global scan if it cannot fit) */
#define PTGPU_FLAG_FORCE_TREE 0x4u    /* use the kernels' branching (tree)
                                         instantiations even if no material
                                         needs them */
#define PTGPU_FLAG_NO_RAY_COUNT 0x8u  /* skip the ray/sample counters */
#define PTGPU_FLAG_SCAN_PLAIN 0x10u   /* scheduling only: test every object in
                                         full (no culling of any kind: the
                                         reference's scan, object for object).
                                         The one form that is exact whatever the
                                         ray.  IN-PLANE RAYS: the screened and
                                         hierarchy scans decide by geometry which
                                         triangles get the reference's test, and
                                         for a ray lying in a triangle's plane
                                         that test is rounding noise -- it can
                                         "hit" beside the triangle, at any
                                         distance along the plane (error ~6e-8 *
                                         distance / |n.d| in the barycentric
                                         coordinates), so no finite margin holds
                                         it.  Measured (round 3, profiles/r03):
                                         in full frames of a floor of 3200
                                         coplanar triangles and of height-field
                                         meshes, every ray audited against this
                                         scan: 0 of 17 million rays answered
                                         differently, 0 of 5.2e10 culled (ray,
                                         triangle) pairs where the reference's
                                         test reports a hit; rays CONSTRUCTED in
                                         a triangle's plane: 0.3-0.8 % differ.
                                         Boxing triangles by 3 bounding radii
                                         changes that count by < 5 % and costs
                                         3-3.6x on meshes: not done.  Use this
                                         flag where that regime matters */
#define PTGPU_FLAG_SCAN_SCREENED 0x20u /* scheduling only: conservative screen
                                         + per-lane queues.  Same image either
                                         way (one stated limit of the screen
                                         and the hierarchy, which decide by
                                         geometry: a ray lying in a triangle's
                                         plane to ~2e-4 rad, where the
                                         reference's triangle test is rounding
                                         noise -- DESIGN.md section 4.1).
                                         Without a SCAN flag the form is
                                         timed on a scene's first large render
                                         (8 samples per candidate form) and
                                         remembered; smaller renders go by
                                         object count: plain < 32 <= screened
                                         < 96 <= BVH */

#define PTGPU_FLAG_PIXEL_OWNER 0x80u   /* scheduling only: one work-item owns a
                                         pixel for all its samples (one launch
                                         per frame; best when every path has
                                         about the same length) */
#define PTGPU_FLAG_SAMPLE_QUEUE 0x100u /* scheduling only: persistent waves pull
                                         (pixel, sample) paths from a queue and
                                         a second kernel sums each pixel's
                                         samples in order (best when path
                                         lengths differ between pixels).  With
                                         neither flag the library measures lane
                                         utilisation on a scene's first large
                                         render (that call synchronises the
                                         stream once) and remembers the
                                         choice in the ptgpu_scene. */

#define PTGPU_FLAG_SCAN_BVH 0x400u     /* scheduling only: walk a bounding-volume
                                         hierarchy (built at upload for >= 64
                                         objects; default for >= 96) instead
                                         of scanning every object.  Candidates
                                         still get the reference's exact tests
                                         and accept rule: same image */
#define PTGPU_FLAG_WHOLE_WALKS 0x800u   /* scheduling only: in the sample-queue
                                         schedule with the BVH scan, let each
                                         wave finish its 64 walks together (the
                                         older form) instead of handing a lane
                                         its next ray as soon as its walk ends */
#define PTGPU_FLAG_NO_PRIMARY_SHARE 0x200u /* scheduling only: scan the scene
                                         for the primary ray of every sample,
                                         as the reference does, instead of once
                                         per pixel (the ray is the same) */

#define PTGPU_FLAG_FAST_MATH 0x1000u    /* NOT an exactness-preserving flag -- the
                                         opt-in tolerance mode.  The same kernels
                                         built with fused multiply-adds and the
                                         hardware's approximate sqrt / reciprocal /
                                         rsqrt (1 ulp) instead of the reference's
                                         sequentially rounded arithmetic.  Same
                                         algorithm, same random stream; the frame
                                         agrees with the exact one to a stated
                                         tolerance (>= 99.9 % of bytes within +-1
                                         at 256 spp, RMSE <= 0.5 of an 8-bit level;
                                         measured 99.993 % / 0.11 on configuration
                                         2, 99.956 % / 0.39 on configuration 5 --
                                         1000 small spheres, where every bounce
                                         multiplies a last-bit difference by ~100;
                                         16-spp frames, where one sample is 16
                                         levels: >= 99.5 %), not to the bit;
                                         within the mode the scheduling flags
                                         change nothing.  Also accepted in the
                                         scan_flags of the trace_rays calls */
#define PTGPU_FLAG_GLSL_SEMANTICS 0x2000u /* NOT scheduling only -- render by the
                                         rules of the reference's OTHER renderer,
                                         its OpenGL fragment shader
                                         (shaders/rtx_frag.glsl), where they differ
                                         from src/renderer.c: PCG stream carried
                                         across a pixel's samples (`seed` = the
                                         shader's random_seed uniform), Box-Muller
                                         rand_unit_sphere, roughness not halved,
                                         the material's refractive_index, total
                                         internal reflection drops the transmitted
                                         ray only, rays through pixel centres, no
                                         per-sample clamp, round-to-nearest RGB8.
                                         ptgpu_render_device / _frame only (no
                                         accumulator, sample_offset 0), at most
                                         PTGPU_MAX_BOUNCES_TREE bounces.  Checked
                                         against a CPU restatement of the shader
                                         (oracle/pt_glsl.c): bit-equal, through a
                                         shared specification of log / cos
                                         (ptgpu_glsl_math.h).  Scenes of >= 96
                                         objects find their closest hits through
                                         the hierarchy (boxes padded for the
                                         shader's own, noisier sphere test;
                                         configuration 5 at 32 spp 337 -> 40 ms);
                                         PTGPU_FLAG_SCAN_PLAIN keeps the shader's
                                         scan of every object, _SCAN_BVH forces
                                         the hierarchy from 64 objects: same
                                         frame */

/* limits */
#define PTGPU_MAX_BOUNCES_PATH 62 /* non-branching kernel (LDS material stack) */
#define PTGPU_MAX_BOUNCES_TREE 14 /* scenes with a branching material (6 words put aside per level) */

typedef struct ptgpu_params {
    uint32_t spp;           /* samples per pixel rendered by this call
                               (NUM_SAMPLES, src/renderer.c:9); default 16 */
    uint32_t max_bounces;   /* MAX_BOUNCES, src/renderer.c:8; default 4.
                               A non-branching path traces <= max_bounces+1
                               rays. */
    float fov_rad;          /* FOV, src/renderer.c:7; default pi/2 */
    uint32_t seed;          /* stream seed (ptgpu_rng.h); default 1 */
    uint32_t sample_offset; /* global index of this call's first sample */
    /* Row set of this call.  Local row i (0 <= i < num_rows) is global row
     *   row_begin + i                                        (tile_rows == 0)
     *   row_begin + (i / tile_rows) * tile_rows * tile_stride + i % tile_rows
     * num_rows == 0 means "every row from row_begin to the bottom". */
    uint32_t row_begin;
    uint32_t num_rows;
    uint32_t tile_rows;
    uint32_t tile_stride;
    uint32_t flags;         /* PTGPU_FLAG_* */
    uint32_t block_threads; /* work-items per workgroup: 0 (= 64 for the
                               global scan, 256 with the LDS copy), or a
                               multiple of 64 up to 1024 */
    uint32_t settle_lanes;  /* scheduling only, never changes the image: how
                               many lanes of a wavefront must have finished a
                               sample before they settle it together.  0 =
                               chosen from the scene size; 1 = per-lane path
                               regeneration; 64 = sample-synchronous waves.
                               In the sample queue with resumable BVH walks:
                               lanes that must wait before the wave serves
                               them (0 = 32) */
    uint32_t lanes_per_pixel; /* scheduling only, never changes the image:
                               lanes of a wavefront sharing one pixel's
                               samples in the pixel-owner schedule (the sum
                               stays in sample order).  0 = chosen from the
                               row set's size (more than 1 only when it is
                               too small to fill the device); else a power
                               of two, 1..64 */
} ptgpu_params;

typedef struct ptgpu_stats {
    uint64_t rays;        /* trace_ray-equivalent path segments traced */
    uint64_t samples;     /* per_pixel-equivalent samples finished */
    uint64_t wave_iters;  /* sum over wavefronts of bounce-loop iterations */
    uint64_t launches;    /* megakernel launches */
    uint64_t scans;       /* closest-hit scans of the scene actually executed.
                             Equals `rays` unless the primary segment's scan
                             was shared between the samples of a pixel */
    double kernel_ms;     /* device time of the render / accumulate calls
                             (first to last kernel of each call, HIP events
                             on the call's stream), summed over calls */
    uint64_t hbm_bytes;   /* algorithmic HBM bytes those calls moved: 3 per
                             pixel of a fused render, 32 per pixel of an
                             accumulate pass, 32 per sample of a sample-queue
                             pass (+ its accumulator traffic) */
} ptgpu_stats;

/* Opaque device-resident scene (geometry + materials in HBM, laid out for the
 * kernel).  Immutable after upload; belongs to the device current at upload. */
typedef struct ptgpu_scene ptgpu_scene;

/* -- parameters ------------------------------------------------------------ */
/* The reference's compile-time settings: 16 spp, 4 bounces, fov pi/2. */
int ptgpu_params_default(ptgpu_params *p);
/* Defaults overridden by PTGPU_SPP, PTGPU_BOUNCES, PTGPU_SEED, PTGPU_FOV_DEG,
 * PTGPU_BLOCK, PTGPU_FLAGS, PTGPU_SETTLE, PTGPU_LANES_PER_PIXEL; this is what
 * the legacy entry points use (they also honour PTGPU_DEVICES, see
 * ptgpu_render_frame_multi). */
int ptgpu_params_from_env(ptgpu_params *p);

/* -- devices --------------------------------------------------------------- */
int ptgpu_device_count(void); /* >= 0, or a negative error */
int ptgpu_set_device(int device);

/* -- scene residency ------------------------------------------------------- */
int ptgpu_scene_upload(const scene *host_scene, ptgpu_scene **out);
void ptgpu_scene_free(ptgpu_scene *s);
/* 1 if some material has 0 < transparency < 1 (needs the tree kernel). */
int ptgpu_scene_is_branching(const ptgpu_scene *s);
/* Schedule of this scene's latest render (before any render: the remembered
 * choice): 0 not measured yet, 1 pixel-owner, 2 sample-queue (see
 * PTGPU_FLAG_PIXEL_OWNER / _SAMPLE_QUEUE). */
int ptgpu_scene_schedule(const ptgpu_scene *s);
/* Scan form remembered for this scene (measured on the same occasion):
 * 0 not measured yet, 1 plain, 2 screened linear, 3 BVH. */
int ptgpu_scene_scan(const ptgpu_scene *s);
/* Size of the bounding-volume hierarchy built at upload (scenes of >= 64
 * objects; not in the reference, see PTGPU_FLAG_SCAN_BVH): inner nodes and the
 * longest chain of them.  Both 0 when the scene has none. */
int ptgpu_scene_hierarchy(const ptgpu_scene *s, uint32_t *nodes, uint32_t *depth);

/* Measures, once, which scan form (plain / screened / BVH), which schedule
 * (pixel owner / sample queue) and how many lanes per pixel suit this scene on
 * this device: timed pilot renders of a framew x frameh frame with `p` (NULL =
 * defaults at 64 spp) into a scratch frame.  SYNCHRONOUS.  The result is
 * remembered in the scene (and, by content, for later uploads of the same
 * scene) and never changes an image.  The asynchronous entry points below do
 * NOT measure: an untuned scene goes by object count (plain < 32 <= screened
 * < 96 <= BVH + sample queue), which is right for most and up to ~7 % off the
 * measured choice on configuration 2.  The synchronous ones (ptgpu_render_frame,
 * the legacy shims) tune on a scene's first large frame by themselves.  Too
 * small a frame (< 2^22 samples, < 16 spp) measures nothing and returns OK. */
int ptgpu_scene_tune(const ptgpu_scene *s, size_t framew, size_t frameh, const ptgpu_params *p);

/* -- rendering ------------------------------------------------------------- */
/* Asynchronous: renders the row set of `p` of a framew x frameh image into
 * `d_frame` (device memory, RGB8, 3*framew bytes per row) on `hip_stream`
 * (a hipStream_t, NULL = default stream) and returns without waiting.  It
 * never measures (see ptgpu_scene_tune).  It does not synchronise either,
 * with one exception: a sample-queue pass that finds the per-device scratch
 * too small -- first use, a larger frame than any before, after
 * ptgpu_release_scratch -- grows it, and hipFree / hipMalloc wait for the
 * device.  ptgpu_scene_tune with the same frame size and parameters leaves the
 * scratch sized for them, so a tuned scene's calls do not allocate.
 * Bytes written equal, bit for bit, what the reference's render_task would
 * store for the same stream of random numbers (src/renderer.c:92-116) -- with
 * the one stated limit of the two scan forms that cull objects by geometry
 * (PTGPU_FLAG_SCAN_SCREENED / _BVH, the default from 32 objects on) on scenes
 * with TRIANGLES: see "in-plane rays" under PTGPU_FLAG_SCAN_PLAIN above. */
int ptgpu_render_device(const ptgpu_scene *s, uint8_t *d_frame, size_t framew,
                        size_t frameh, const ptgpu_params *p,
                        void *hip_stream);

/* Synchronous convenience: upload, render, copy the rows back into the
 * caller's host frame, free.  Always full-frame addressing: row y of the row
 * set lands at row y of `frame` (PTGPU_FLAG_COMPACT_ROWS has no effect here). */
int ptgpu_render_frame(const scene *host_scene, uint8_t *frame, size_t framew,
                       size_t frameh, const ptgpu_params *p);

/* Progressive accumulation.  `d_accum` holds one float4 per pixel of the
 * destination (sum_r, sum_g, sum_b, samples); zero it before the first pass.
 * Each call adds p->spp clamped samples per pixel, in sample order, so k
 * passes of n samples leave exactly the sum one pass of k*n samples leaves. */
int ptgpu_accumulate_device(const ptgpu_scene *s, float *d_accum,
                            size_t framew, size_t frameh,
                            const ptgpu_params *p, void *hip_stream);
/* frame[i] = (uint8)(sum * (1.0f / samples * 255.0f)) for `npixels` pixels
 * (src/renderer.c:108-114). */
int ptgpu_resolve_device(const float *d_accum, uint8_t *d_frame,
                         size_t npixels, void *hip_stream);

/* Synchronous, single process, several GPUs: the frame's rows are dealt to
 * `n_devices` devices as interleaved tiles of `tile_rows` rows (0 = 8), each
 * device renders its tiles on its own stream, and the rows are copied
 * straight into the caller's host frame -- the host buffer is the gather
 * target, so no collective is needed.  `devices[i]` are HIP ordinals (an
 * ordinal may repeat).  The frame does not depend on n_devices or tile_rows.
 * The legacy entry points do the same when PTGPU_DEVICES=n is set. */
int ptgpu_render_frame_multi(const scene *host_scene, uint8_t *frame,
                             size_t framew, size_t frameh,
                             const ptgpu_params *p, const int *devices,
                             int n_devices, uint32_t tile_rows);

/* -- one process, several GPUs, frame assembled IN HBM (RCCL over xGMI) ------- */
/* The same row-tile deal as above, but the frame never leaves the devices:
 * every device renders its tiles into a compact shard in its own HBM, ONE
 * grouped ncclSend/ncclRecv (rccl.h:700,722) brings the shards of devices
 * 1..n-1 to device 0 -- each peer over its own xGMI link -- and a small
 * kernel on device 0 puts the rows in place.  A plain C host
 * gets its frame in HBM on devices[0], ready for a resolve / encode / display
 * step there.  librccl is loaded on first use (dlopen), so hosts that never
 * call this pay nothing for it.
 *
 *   ptgpu_multi_create   communicators (ncclCommInitAll, rccl.h:236), one
 *                        stream per device; `devices` are HIP ordinals, all
 *                        different, n_devices >= 2 -- else PTGPU_ERR_LIMIT.
 *                        flags = PTGPU_MULTI_PEER_COPY: no RCCL; shards travel
 *                        by hipMemcpyPeerAsync instead, ordinals may repeat
 *                        (a rehearsal of everything but the collective on a
 *                        box with one GPU, and a fallback where librccl is
 *                        missing)
 *   ptgpu_multi_upload   replicates the scene on every device (<= 56 KB)
 *   ptgpu_multi_render   one frame into `d_frame` (device memory of
 *                        devices[0], full-frame addressing, RGB8); returns
 *                        when the frame is complete.  `p` must not carry a
 *                        tile deal of its own; tile_rows 0 = 8.  The frame is
 *                        the single-device frame, bit for bit.
 *   ptgpu_render_device_multi   create + upload + render + destroy in one call
 */
#define PTGPU_MULTI_PEER_COPY 0x1u
#define PTGPU_MULTI_RCCL_LOOPBACK 0x2u /* test aid: RCCL, but the ordinals may repeat (even
                                         all the same).  The communicator spans the distinct
                                         devices; shards that share devices[0] travel by
                                         ncclSend / ncclRecv to self.  Runs the library calls
                                         -- dlopen, ncclCommInitAll, the grouped exchange,
                                         the stream ordering -- on a box with one GPU */
typedef struct ptgpu_multi ptgpu_multi;
int ptgpu_multi_create(const int *devices, int n_devices, uint32_t flags, ptgpu_multi **out);
int ptgpu_multi_upload(ptgpu_multi *m, const scene *host_scene);
int ptgpu_multi_render(ptgpu_multi *m, uint8_t *d_frame, size_t framew, size_t frameh,
                       const ptgpu_params *p, uint32_t tile_rows);
/* "rccl" or "peer-copy" */
const char *ptgpu_multi_transport(const ptgpu_multi *m);
void ptgpu_multi_destroy(ptgpu_multi *m);
int ptgpu_render_device_multi(const scene *host_scene, uint8_t *d_frame_on_device0,
                              size_t framew, size_t frameh, const ptgpu_params *p,
                              const int *devices, int n_devices, uint32_t tile_rows);

/* -- device memory helpers for hosts without a HIP binding (C, cgo, ctypes) -- */
int ptgpu_device_alloc(size_t bytes, void **out); /* hipMalloc, zero-filled */
int ptgpu_device_free(void *ptr);
int ptgpu_device_upload(void *dst_device, const void *src_host, size_t bytes);
int ptgpu_device_download(void *dst_host, const void *src_device, size_t bytes);
int ptgpu_device_synchronize(void);

/* -- batched closest-hit queries (host buffers, synchronous) ---------------- */
/* origins/directions: n x 3 floats; hits: n ray_hit records. */
int ptgpu_trace_rays(const scene *host_scene, const float *origins,
                     const float *directions, size_t n, ray_hit *hits);
/* The same query through a chosen scan form: scan_flags = 0 (plain, what
 * ptgpu_trace_rays uses), PTGPU_FLAG_SCAN_SCREENED or PTGPU_FLAG_SCAN_BVH.
 * All three return the same hits. */
int ptgpu_trace_rays_scan(const scene *host_scene, const float *origins,
                          const float *directions, size_t n, ray_hit *hits,
                          uint32_t scan_flags);

/* The same for rays that already live in HBM: asynchronous on `hip_stream`;
 * d_origins / d_directions: n x 3 floats, d_hits: n ray_hit records (40 bytes
 * each), all device pointers.  64 bytes of HBM traffic per ray: with a small
 * scene this kernel is bound by HBM, not by the ALUs. */
int ptgpu_trace_rays_device(const ptgpu_scene *s, const float *d_origins,
                            const float *d_directions, size_t n, ray_hit *d_hits,
                            uint32_t scan_flags, void *hip_stream);

/* -- bookkeeping ------------------------------------------------------------ */
int ptgpu_stats_reset(void);
int ptgpu_stats_read(ptgpu_stats *out); /* synchronises the device */
/* Frees the per-device sample/accumulator scratch of the sample-queue
 * schedule (up to PTGPU_SCRATCH_MB, default 2048 MiB; regrown on demand). */
int ptgpu_release_scratch(void);
const char *ptgpu_last_error(void);     /* thread-local, never NULL */
const char *ptgpu_version(void);

/* -- the legacy entry points' two policies (environment, read per call) ------
 * render_task / threadpool_* / trace_ray return void in the reference
 * (include/renderer.h:13, include/threadpool.h:15-20) and cannot report a
 * failure.
 *   PTGPU_ON_ERROR=abort (default)  a failed legacy call prints the reason and
 *                    abort()s: there is no CPU renderer to fall back to, and a
 *                    frame that silently stays black is worse than none.
 *   PTGPU_ON_ERROR=zero   the call writes nothing (trace_ray returns the
 *                    reference's "no hit", distance -1), prints the reason on
 *                    stderr once per process and returns; ptgpu_last_error()
 *                    has it.  For a host program that must outlive a
 *                    transient HIP failure.
 *   PTGPU_FRAME_SEED=auto  every legacy frame draws from the stream PTGPU_SEED
 *                    + (frames rendered before it), the analogue of the fresh
 *                    random_seed the reference's real-time loop takes per frame
 *                    (src/main.c:276).  A frame is one threadpool_wait_for_tasks
 *                    that rendered rows or, for render_task called directly, the
 *                    calls up to the first row asked for a second time.
 *                    Default: PTGPU_SEED for every frame (the same picture for
 *                    the same scene).
 * Number of legacy frames counted so far under PTGPU_FRAME_SEED=auto: */
uint32_t ptgpu_legacy_frames(void);

/* -- synthetic scenes of BASELINE.json's five configurations --------------- */
/* Fills an initialised-or-not `scene` (scene_init is called inside) with
 * configuration `config` in 1..5, or 0 for the reference's demo scene
 * (src/main.c:202-230).  Free with scene_destroy. */
int ptgpu_build_config_scene(int config, scene *out);

#ifdef __cplusplus
}
#endif
#endif /* PTGPU_H */
