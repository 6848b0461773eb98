/*
 * thread_stress.c -- four host threads hammering the C-ABI at once, for a
 * ThreadSanitizer build of the library's HOST code (make tsan-driver; the GPU
 * pool has no device sanitizer, and the kernels are not instrumented).  Each
 * thread runs a fixed pseudo-random mix of: ptgpu_render_frame on shared and on
 * freshly built scenes (every configuration, both schedules), ptgpu_trace_rays,
 * the legacy pool path, direct render_task calls, ptgpu_scene_upload / _free,
 * ptgpu_stats_read and ptgpu_release_scratch.  Frames are compared with the
 * first frame of their kind (the stream is deterministic), so a race that
 * corrupts a picture fails the run even where TSan sees nothing.
 * Run on a GPU box: tools/tsan_driver.sh
 */
#include "ptgpu.h"

#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

enum { THREADS = 4, CONFIGS = 4, W = 40, H = 24 };
static const int config_of[CONFIGS] = {0, 1, 2, 5};
static scene shared[CONFIGS];
static uint8_t want[CONFIGS][3 * W * H];
static int failures = 0;
static pthread_mutex_t fail_lock = PTHREAD_MUTEX_INITIALIZER;

static void fail_now(const char *what, int t, int it) {
    pthread_mutex_lock(&fail_lock);
    failures++;
    fprintf(stderr, "thread %d iteration %d: %s (%s)\n", t, it, what, ptgpu_last_error());
    pthread_mutex_unlock(&fail_lock);
}

static uint32_t next(uint32_t *s) {
    *s ^= *s << 13;
    *s ^= *s >> 17;
    *s ^= *s << 5;
    return *s;
}

static void params(ptgpu_params *p, uint32_t flags) {
    ptgpu_params_default(p);
    p->spp = 3;
    p->max_bounces = 4;
    p->seed = 9;
    p->flags = flags;
}

typedef struct {
    int t, rounds;
} job;

static void *worker(void *arg) {
    const job *j = (const job *)arg;
    uint32_t rng = 1234567u + 977u * (uint32_t)j->t;
    static const uint32_t scheds[3] = {0, PTGPU_FLAG_PIXEL_OWNER, PTGPU_FLAG_SAMPLE_QUEUE};
    uint8_t *frame = malloc(3 * W * H);
    for (int it = 0; it < j->rounds; it++) {
        const int k = (int)(next(&rng) % CONFIGS);
        const uint32_t op = next(&rng) % 7u;
        ptgpu_params p;
        params(&p, scheds[next(&rng) % 3u]);
        memset(frame, 0, 3 * W * H);
        if (op == 0) {
            if (ptgpu_render_frame(&shared[k], frame, W, H, &p) != PTGPU_OK || memcmp(frame, want[k], 3 * W * H))
                fail_now("shared scene frame", j->t, it);
        } else if (op == 1) { /* a scene of this thread's own, same content */
            scene mine;
            if (ptgpu_build_config_scene(config_of[k], &mine) != PTGPU_OK) {
                fail_now("scene build", j->t, it);
                continue;
            }
            if (ptgpu_render_frame(&mine, frame, W, H, &p) != PTGPU_OK || memcmp(frame, want[k], 3 * W * H))
                fail_now("own scene frame", j->t, it);
            scene_destroy(&mine);
        } else if (op == 2) { /* the reference's driver loop */
            threadpool pool;
            render_args rows[H];
            threadpool_init(&pool, 2);
            for (int y = 0; y < H; y++) {
                rows[y] = (render_args){.screen_y = y, .framew = W, .frameh = H, .frame = frame, .world = &shared[k]};
                threadpool_add_task(&pool, (void (*)(void *))render_task, &rows[y]);
            }
            threadpool_wait_for_tasks(&pool);
            threadpool_destroy(&pool);
            if (memcmp(frame, want[k], 3 * W * H))
                fail_now("pool frame", j->t, it);
        } else if (op == 3) { /* direct calls, a few rows */
            render_args a = {.framew = W, .frameh = H, .frame = frame, .world = &shared[k]};
            for (int n = 0; n < 5; n++) {
                a.screen_y = (int)(next(&rng) % H);
                render_task(&a);
                if (memcmp(frame + 3 * W * a.screen_y, want[k] + 3 * W * a.screen_y, 3 * W))
                    fail_now("direct row", j->t, it);
            }
        } else if (op == 4) {
            float o[3 * 64], d[3 * 64];
            ray_hit hits[64];
            for (int n = 0; n < 64; n++) {
                o[3 * n] = o[3 * n + 1] = o[3 * n + 2] = 0.0f;
                d[3 * n] = 0.01f * (float)(n - 32);
                d[3 * n + 1] = 0.0f;
                d[3 * n + 2] = 1.0f; /* (not unit: the plain scan serves it) */
            }
            if (ptgpu_trace_rays(&shared[k], o, d, 64, hits) != PTGPU_OK)
                fail_now("trace_rays", j->t, it);
        } else if (op == 5) {
            ptgpu_scene *dev = NULL;
            uint32_t nodes = 0, depth = 0;
            if (ptgpu_scene_upload(&shared[k], &dev) != PTGPU_OK || ptgpu_scene_hierarchy(dev, &nodes, &depth) != PTGPU_OK)
                fail_now("upload", j->t, it);
            ptgpu_scene_free(dev);
        } else {
            ptgpu_stats st;
            if ((next(&rng) & 3u) == 0u)
                ptgpu_release_scratch();
            else if (ptgpu_stats_read(&st) != PTGPU_OK)
                fail_now("stats", j->t, it);
        }
    }
    free(frame);
    return NULL;
}

int main(int argc, char **argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 200;
    setenv("PTGPU_SPP", "3", 1);
    setenv("PTGPU_BOUNCES", "4", 1);
    setenv("PTGPU_SEED", "9", 1);
    for (int k = 0; k < CONFIGS; k++) {
        ptgpu_params p;
        params(&p, 0);
        if (ptgpu_build_config_scene(config_of[k], &shared[k]) != PTGPU_OK ||
            ptgpu_render_frame(&shared[k], want[k], W, H, &p) != PTGPU_OK) {
            fprintf(stderr, "setup failed: %s\n", ptgpu_last_error());
            return 1;
        }
    }
    pthread_t th[THREADS];
    job jobs[THREADS];
    for (int t = 0; t < THREADS; t++) {
        jobs[t] = (job){t, rounds};
        pthread_create(&th[t], NULL, worker, &jobs[t]);
    }
    for (int t = 0; t < THREADS; t++)
        pthread_join(th[t], NULL);
    for (int k = 0; k < CONFIGS; k++)
        scene_destroy(&shared[k]);
    if (failures) {
        fprintf(stderr, "thread_stress: %d failure(s)\n", failures);
        return 1;
    }
    printf("thread_stress: ok (%d threads x %d calls)\n", THREADS, rounds);
    return 0;
}
