/*
 * oracle/pt_oracle.c -- TEST INFRASTRUCTURE: a CPU restatement of the
 * reference's render hot path.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this; the product library never does.
 *
 * PARITY UNPINNED (see pt_math.h): the reference has no tests or fixtures for
 * this path and cannot be compiled here (cglm is absent), so this file is a
 * line-by-line restatement of src/renderer.c:6-116 and src/ray.c:4-73 that
 * nothing from the reference can confirm; SURVEY.md section 8(c)'s known
 * answers are asserted in tests/test_oracle.py.
 *
 * What is kept from the reference, on purpose:
 *   - the depth rule `bounces > MAX_BOUNCES -> sky` (src/renderer.c:24-26)
 *   - three RNG draws per hit, in x,y,z order, even when roughness is 0
 *     (src/renderer.c:11-19,40)
 *   - double-typed sub-expressions rounded to float where the reference's
 *     types do so: roughness*0.5, 0.001, 1.5, 1/1.5, 1.0/ri
 *     (src/renderer.c:40,49,58,63,66)
 *   - the hard-coded index of refraction 1.5 (src/renderer.c:58)
 *   - lerp(reflected, transmitted, transparency) with both branches recursed
 *     when 0 < transparency < 1 (src/renderer.c:45,56,72)
 *   - per-sample clamp, then the sum, then scale, then the truncating uint8
 *     store (src/renderer.c:88,103-114)
 *   - the closest-hit rule `!(dist >= closest && closest > 0)` in object
 *     order (src/ray.c:25,45)
 * What is made a run-time parameter: NUM_SAMPLES, MAX_BOUNCES, FOV
 * (src/renderer.c:7-9) and the random stream:
 *   ORACLE_RNG_LIBC     glibc random()/0x7FFFFFFF, the reference's generator
 *   ORACLE_RNG_XORSHIFT the per-(pixel,sample) stream of include/ptgpu_rng.h,
 *                       i.e. the stream the GPU path consumes
 */
#include "pt_oracle.h"
#include "pt_math.h"
#include "ptgpu_rng.h"

#include <pthread.h>
#include <stdio.h>
#include <string.h>

#define RANDOM_MAX 0x7FFFFFFF /* src/renderer.c:6 */

typedef struct {
    uint32_t mode;
    uint32_t state;      /* xorshift state of the current sample */
    uint32_t max_bounces;
    uint64_t rays;       /* trace_ray calls made through this context */
    float *record;       /* optional: 6 floats (origin, direction) per trace_ray call */
    size_t record_cap, record_n;
} path_ctx;

/* ---- src/ray.c:4-73 ------------------------------------------------------- */
static ray_hit trace_closest(vec3s origin, vec3s direction, const scene *world) {
    float closest_dist = -1.0f;
    vec3s closest_point = om_zero();
    vec3s closest_normal = om_zero();
    size_t closest_material = 0;

    for (size_t i = 0; i < world->num_objects; i++) {
        const shape *obj = &world->objects[i];
        if (obj->tag == SPHERE) {
            float t1, t2;
            if (!om_ray_sphere(origin, direction, obj->sphere.center,
                               obj->sphere.radius, &t1, &t2))
                continue;
            float dist = t1 > 0.0f ? t1 : t2; /* src/ray.c:24 */
            if (dist >= closest_dist && closest_dist > 0.0f)
                continue;
            vec3s at = om_add(om_scale(direction, dist), origin);
            closest_dist = dist;
            closest_point = at;
            closest_normal = om_normalize(om_sub(at, obj->sphere.center));
            closest_material = obj->material;
        } else if (obj->tag == TRIANGLE) {
            float dist;
            if (!om_ray_triangle(origin, direction, obj->triangle.v0,
                                 obj->triangle.v1, obj->triangle.v2, &dist))
                continue;
            if (dist >= closest_dist && closest_dist > 0.0f)
                continue;
            vec3s at = om_add(om_scale(direction, dist), origin);
            vec3s n = om_crossn(om_sub(obj->triangle.v0, obj->triangle.v1),
                                om_sub(obj->triangle.v0, obj->triangle.v2));
            if (om_dot(n, direction) > 0.0f) /* face the ray, src/ray.c:55 */
                n = om_negate(n);
            closest_dist = dist;
            closest_point = at;
            closest_normal = n;
            closest_material = obj->material;
        }
    }

    ray_hit hit;
    memset(&hit, 0, sizeof hit);
    hit.distance = closest_dist;
    hit.point = closest_point;
    hit.normal = closest_normal;
    hit.material = closest_material;
    return hit;
}

ray_hit oracle_trace_ray(const vec3s origin, const vec3s direction,
                         const scene *world) {
    return trace_closest(origin, direction, world);
}

void oracle_trace_rays(const scene *world, const float *origins,
                       const float *directions, size_t n, ray_hit *hits) {
    for (size_t i = 0; i < n; i++) {
        vec3s o = om_make(origins[3 * i], origins[3 * i + 1], origins[3 * i + 2]);
        vec3s d = om_make(directions[3 * i], directions[3 * i + 1],
                          directions[3 * i + 2]);
        hits[i] = trace_closest(o, d, world);
    }
}

/* ---- src/renderer.c:11-19 -------------------------------------------------- */
static vec3s rand_unit_sphere(path_ctx *c) {
    float rx, ry, rz;
    if (c->mode == ORACLE_RNG_LIBC) {
        /* gcc evaluates the reference's brace initialiser left to right; the
         * draws are named here so the order is not left to the compiler. */
        double a = (double)random() / (double)RANDOM_MAX;
        double b = (double)random() / (double)RANDOM_MAX;
        double g = (double)random() / (double)RANDOM_MAX;
        rx = (float)a;
        ry = (float)b;
        rz = (float)g;
    } else {
        c->state = ptgpu_xorshift32(c->state);
        rx = ptgpu_u32_to_unit(c->state);
        c->state = ptgpu_xorshift32(c->state);
        ry = ptgpu_u32_to_unit(c->state);
        c->state = ptgpu_xorshift32(c->state);
        rz = ptgpu_u32_to_unit(c->state);
    }
    vec3s cube = om_sub(om_scale(om_make(rx, ry, rz), 2.0f), om_one());
    return om_normalize(cube);
}

/* ---- src/renderer.c:21-74 -------------------------------------------------- */
static vec3s incident_light(path_ctx *c, vec3s origin, vec3s direction,
                            const scene *world, size_t bounces) {
    if (bounces > c->max_bounces)
        return world->sky_color;
    bounces++;

    c->rays++;
    if (c->record != NULL && c->record_n < c->record_cap) {
        float *r = c->record + 6 * c->record_n++;
        r[0] = origin.x; r[1] = origin.y; r[2] = origin.z;
        r[3] = direction.x; r[4] = direction.y; r[5] = direction.z;
    }
    ray_hit hit = trace_closest(origin, direction, world);
    if (hit.distance < 0.0f)
        return world->sky_color;

    shape_material mat = world->materials[hit.material];

    vec3s Le = om_scale(mat.emission_color, mat.emission_strength);

    /* `mat.roughness * 0.5` is a double product narrowed to the float
     * parameter of glms_vec3_scale: exact, equals roughness * 0.5f. */
    float half_rough = (float)((double)mat.roughness * 0.5);
    vec3s deviation = om_scale(rand_unit_sphere(c), half_rough);
    vec3s normal = om_normalize(om_add(hit.normal, deviation));

    vec3s reflected = om_zero();
    if ((double)mat.transparency < 1.0) {
        vec3s rdir = om_normalize(om_reflect(direction, normal));
        vec3s rorg = om_add(hit.point, om_scale(rdir, (float)0.001));
        reflected = incident_light(c, rorg, rdir, world, bounces);
    }

    vec3s transmitted = om_zero();
    if ((double)mat.transparency > 0.0) {
        float facing = om_dot(direction, normal);
        float ior = facing < 0.0f ? (float)1.5 : (float)(1 / 1.5);
        vec3s rn = facing < 0.0f ? normal : om_negate(normal);
        float eta = (float)(1.0 / (double)ior);
        vec3s tdir;
        if (om_refract(direction, rn, eta, &tdir)) {
            tdir = om_normalize(tdir);
            vec3s torg = om_add(hit.point, om_scale(tdir, (float)0.001));
            transmitted = incident_light(c, torg, tdir, world, bounces);
        }
    }

    vec3s Li = om_lerp(reflected, transmitted, mat.transparency);
    return om_add(Le, om_mul(Li, mat.albedo));
}

/* ---- src/renderer.c:76-90 -------------------------------------------------- */
static vec3s per_pixel(path_ctx *c, float x, float y, float aspect_ratio,
                       float tangent_fov_2, const scene *world) {
    vec3s origin = om_zero();
    vec3s direction = om_normalize(
        om_make(x * tangent_fov_2, y * tangent_fov_2 / aspect_ratio, 1.0f));
    vec3s result = incident_light(c, origin, direction, world, 0);
    return om_clamp(result, 0.0f, 1.0f);
}

static float tan_half_fov(const oracle_params *p) {
    /* tanf(FOV / 2.0f), src/renderer.c:77; halving a float is exact. */
    return tanf(p->fov_rad / 2.0f);
}

/* Sum of `spp` clamped samples of pixel (sx, sy), added onto `start`. */
static vec3s pixel_sum(path_ctx *c, const scene *world, size_t w, size_t h,
                       int sx, int sy, const oracle_params *p, vec3s start) {
    float aspect_ratio = (float)w / (float)h;
    float x = ((float)sx / (float)w * 2.0f - 1.0f);
    float y = -((float)sy / (float)h * 2.0f - 1.0f);
    float t = tan_half_fov(p);
    uint32_t pixel = (uint32_t)((size_t)sy * w + (size_t)sx);
    vec3s result = start;
    for (uint32_t j = 0; j < p->spp; j++) {
        if (c->mode != ORACLE_RNG_LIBC)
            c->state = ptgpu_rng_seed(p->seed, pixel, p->sample_offset + j);
        vec3s sample = per_pixel(c, x, y, aspect_ratio, t, world);
        result = om_add(result, sample);
    }
    return result;
}

/* ---- src/renderer.c:92-116: one image row ---------------------------------- */
static uint64_t render_row(const scene *world, uint8_t *frame, size_t w,
                           size_t h, int sy, const oracle_params *p) {
    path_ctx c = {p->rng_mode, 0u, p->max_bounces, 0u, NULL, 0, 0};
    for (int sx = 0; sx < (int)w; sx++) {
        vec3s result = pixel_sum(&c, world, w, h, sx, sy, p, om_zero());
        result = om_scale(result, 1.0f / (float)p->spp * 255.0f);
        size_t idx = (size_t)sy * w + (size_t)sx;
        frame[3 * idx + 0] = (uint8_t)result.x;
        frame[3 * idx + 1] = (uint8_t)result.y;
        frame[3 * idx + 2] = (uint8_t)result.z;
    }
    return c.rays;
}

uint64_t oracle_render_rows(const scene *world, uint8_t *frame, size_t w,
                            size_t h, size_t row_begin, size_t row_end,
                            const oracle_params *p) {
    uint64_t rays = 0;
    for (size_t y = row_begin; y < row_end && y < h; y++)
        rays += render_row(world, frame, w, h, (int)y, p);
    return rays;
}

uint64_t oracle_accumulate_rows(const scene *world, float *accum, size_t w,
                                size_t h, size_t row_begin, size_t row_end,
                                const oracle_params *p) {
    path_ctx c = {p->rng_mode, 0u, p->max_bounces, 0u, NULL, 0, 0};
    for (size_t y = row_begin; y < row_end && y < h; y++) {
        for (size_t sx = 0; sx < w; sx++) {
            float *px = accum + 4 * (y * w + sx);
            vec3s sum = pixel_sum(&c, world, w, h, (int)sx, (int)y, p,
                                  om_make(px[0], px[1], px[2]));
            px[0] = sum.x;
            px[1] = sum.y;
            px[2] = sum.z;
            px[3] = px[3] + (float)p->spp;
        }
    }
    return c.rays;
}

void oracle_sample(const scene *world, size_t w, size_t h, int sx, int sy,
                   uint32_t sample, const oracle_params *p, float out_rgb[3]) {
    path_ctx c = {p->rng_mode, 0u, p->max_bounces, 0u, NULL, 0, 0};
    oracle_params one = *p;
    one.spp = 1;
    one.sample_offset = sample;
    vec3s s = pixel_sum(&c, world, w, h, sx, sy, &one, om_zero());
    out_rgb[0] = s.x;
    out_rgb[1] = s.y;
    out_rgb[2] = s.z;
}

size_t oracle_sample_rays(const scene *world, size_t w, size_t h, int sx, int sy,
                          uint32_t sample, const oracle_params *p, float *rays6,
                          size_t max_rays) {
    path_ctx c = {p->rng_mode, 0u, p->max_bounces, 0u, rays6, max_rays, 0};
    oracle_params one = *p;
    one.spp = 1;
    one.sample_offset = sample;
    (void)pixel_sum(&c, world, w, h, sx, sy, &one, om_zero());
    return c.record_n;
}

void oracle_srandom(unsigned int seed) { srandom(seed); }

/* ---- a correct row-task pool, for the timed CPU baseline ------------------- */
/* The reference's pool (src/threadpool.c:14-101) returns from
 * threadpool_wait_for_tasks before the last rows finish and hangs in
 * threadpool_destroy with more than one idle worker.
 * The baseline keeps its shape -- a mutex-guarded LIFO of row tasks, N
 * workers, one condition for "work available" and one for "all done" -- with
 * predicate-guarded waits, an in-flight count and a broadcast on shutdown. */
typedef struct {
    const scene *world;
    uint8_t *frame;
    size_t w, h;
    const oracle_params *params;
    int *rows;          /* LIFO of pending row numbers */
    size_t pending;
    size_t in_flight;
    bool closing;
    uint64_t rays;
    pthread_mutex_t lock;
    pthread_cond_t work_ready;
    pthread_cond_t all_done;
} row_pool;

static void *row_worker(void *arg) {
    row_pool *pool = arg;
    uint64_t my_rays = 0;
    pthread_mutex_lock(&pool->lock);
    for (;;) {
        while (pool->pending == 0 && !pool->closing)
            pthread_cond_wait(&pool->work_ready, &pool->lock);
        if (pool->pending == 0 && pool->closing)
            break;
        int y = pool->rows[--pool->pending];
        pool->in_flight++;
        pthread_mutex_unlock(&pool->lock);

        my_rays += render_row(pool->world, pool->frame, pool->w, pool->h, y,
                              pool->params);

        pthread_mutex_lock(&pool->lock);
        pool->in_flight--;
        if (pool->pending == 0 && pool->in_flight == 0)
            pthread_cond_broadcast(&pool->all_done);
    }
    pool->rays += my_rays;
    pthread_mutex_unlock(&pool->lock);
    return NULL;
}

uint64_t oracle_render_mt(const scene *world, uint8_t *frame, size_t w,
                          size_t h, size_t row_begin, size_t row_end,
                          size_t row_step, const oracle_params *p,
                          unsigned int num_threads) {
    if (row_step == 0)
        row_step = 1;
    if (num_threads == 0)
        num_threads = 1;
    if (row_end > h)
        row_end = h;

    row_pool pool;
    memset(&pool, 0, sizeof pool);
    pool.world = world;
    pool.frame = frame;
    pool.w = w;
    pool.h = h;
    pool.params = p;
    pool.rows = malloc(sizeof(int) * (row_end > row_begin ? row_end - row_begin : 1));
    pthread_mutex_init(&pool.lock, NULL);
    pthread_cond_init(&pool.work_ready, NULL);
    pthread_cond_init(&pool.all_done, NULL);

    pthread_t *workers = malloc(sizeof(pthread_t) * num_threads);
    for (unsigned int i = 0; i < num_threads; i++)
        pthread_create(&workers[i], NULL, row_worker, &pool);

    /* submit one task per row, as the reference's driver would */
    for (size_t y = row_begin; y < row_end; y += row_step) {
        pthread_mutex_lock(&pool.lock);
        pool.rows[pool.pending++] = (int)y;
        pthread_cond_signal(&pool.work_ready);
        pthread_mutex_unlock(&pool.lock);
    }

    /* wait for all tasks */
    pthread_mutex_lock(&pool.lock);
    while (pool.pending != 0 || pool.in_flight != 0)
        pthread_cond_wait(&pool.all_done, &pool.lock);
    pool.closing = true;
    pthread_cond_broadcast(&pool.work_ready);
    pthread_mutex_unlock(&pool.lock);

    for (unsigned int i = 0; i < num_threads; i++)
        pthread_join(workers[i], NULL);

    uint64_t rays = pool.rays;
    free(workers);
    free(pool.rows);
    pthread_mutex_destroy(&pool.lock);
    pthread_cond_destroy(&pool.work_ready);
    pthread_cond_destroy(&pool.all_done);
    return rays;
}
