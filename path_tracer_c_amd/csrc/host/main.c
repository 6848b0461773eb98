/*
 * main.c -- host driver of the render path, in the shape the reference's
 * (missing) CPU driver had: build a scene, queue one render_task per image
 * row on a threadpool, wait, write output.bmp.  The reference's HEAD only
 * builds its OpenGL driver (CMakeLists.txt:7-8 -- the CPU file set is
 * commented out); the calls below are the ones that file set implies
 * (SURVEY.md section 3.1): scene_* (include/scene.h:17-30), threadpool_*
 * (include/threadpool.h:15-20), render_task (include/renderer.h:13) and
 * write_bitmap(..., y_inverted = true) (include/bitmap.h:4-5).
 *
 * usage: ptrender [config 0..5] [width] [height] [out.bmp]
 * Samples per pixel, bounce limit and seed come from PTGPU_SPP,
 * PTGPU_BOUNCES and PTGPU_SEED (defaults: the reference's 16 / 4);
 * PTGPU_DEVICES=n spreads the rows over n GPUs; PTGPU_FRAMES=n renders the
 * frame n times and reports the last.
 *
 * PTGPU_MULTI=n renders through ptgpu_multi_* (include/ptgpu.h): n devices,
 * each its row tiles, ONE grouped RCCL send/recv of the shards to device 0,
 * the frame assembled in HBM there and downloaded once.  With fewer than n
 * GPUs in the machine the ordinals repeat and the shards travel by peer copies
 * instead (PTGPU_MULTI_PEER_COPY): the same code but for the collective.
 *
 * PTGPU_PROGRESSIVE=k switches to the explicit API (include/ptgpu.h): the
 * frame is accumulated k samples at a time into a float accumulator in HBM
 * and out.bmp is rewritten after every pass -- the windowless stand-in for
 * the reference's real-time loop, which re-renders with a fresh seed each
 * frame (src/main.c:250-286).  With PTGPU_RESUME=file the accumulator is
 * loaded from / saved to that file, so an interrupted render continues at
 * the sample it stopped at and ends with the very same image.
 */
#include "ptgpu.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#define WIDTH 1280 /* src/main.c:17-18 */
#define HEIGHT 800

static double seconds_now(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static uint32_t env_number(const char *name, uint32_t fallback) {
    const char *v = getenv(name);
    return (v && *v) ? (uint32_t)strtoul(v, NULL, 0) : fallback;
}

/* What a snapshot belongs to: a resumed render must continue THE SAME render.
 * (FNV-1a over the scene's arrays: any edit of an object or material shows.) */
static uint64_t fnv1a(const void *data, size_t n, uint64_t h) {
    const unsigned char *p = data;
    for (size_t i = 0; i < n; i++) {
        h ^= p[i];
        h *= 0x100000001B3ull;
    }
    return h;
}

typedef struct {
    uint64_t magic, width, height, samples_done;
    uint64_t scene_hash;
    uint32_t seed, max_bounces, fov_bits, flags;
} snapshot_head;
#define SNAPSHOT_MAGIC 0x32544E5350475450ull /* "PTGPSNT2" */

static snapshot_head snapshot_of(const scene *world, size_t width, size_t height,
                                 const ptgpu_params *p, uint32_t done) {
    snapshot_head h;
    memset(&h, 0, sizeof h);
    h.magic = SNAPSHOT_MAGIC;
    h.width = width;
    h.height = height;
    h.samples_done = done;
    uint64_t x = 0xCBF29CE484222325ull;
    x = fnv1a(&world->num_objects, sizeof world->num_objects, x);
    x = fnv1a(&world->num_materials, sizeof world->num_materials, x);
    x = fnv1a(&world->sky_color, sizeof world->sky_color, x);
    if (world->num_objects)
        x = fnv1a(world->objects, world->num_objects * sizeof(shape), x);
    if (world->num_materials)
        x = fnv1a(world->materials, world->num_materials * sizeof(shape_material), x);
    h.scene_hash = x;
    h.seed = p->seed;
    h.max_bounces = p->max_bounces;
    memcpy(&h.fov_bits, &p->fov_rad, sizeof h.fov_bits);
    h.flags = p->flags & (PTGPU_FLAG_FAST_MATH | PTGPU_FLAG_GLSL_SEMANTICS); /* what changes a sample */
    return h;
}

static int progressive(scene *world, size_t width, size_t height, char *path,
                       uint32_t per_pass) {
    ptgpu_params p;
    ptgpu_params_from_env(&p);
    const uint32_t total = p.spp;
    const size_t npix = width * height;
    ptgpu_scene *dev = NULL;
    void *d_accum = NULL, *d_frame = NULL;
    uint8_t *frame = malloc(3 * npix);
    float *accum = malloc(16 * npix);
    int rc = ptgpu_scene_upload(world, &dev);
    if (rc == PTGPU_OK) rc = ptgpu_device_alloc(16 * npix, &d_accum);
    if (rc == PTGPU_OK) rc = ptgpu_device_alloc(3 * npix, &d_frame);
    int status = 1;
    if (rc != PTGPU_OK || frame == NULL || accum == NULL) {
        fprintf(stderr, "progressive setup failed: %s\n", ptgpu_last_error());
        goto cleanup;
    }

    /* resume: a header that says WHICH render the sums belong to (frame size,
     * scene content, seed, bounce limit, field of view, arithmetic mode) +
     * the float4 accumulator.  A snapshot of another render is refused, not
     * added to. */
    uint32_t done = 0;
    const char *resume = getenv("PTGPU_RESUME");
    if (resume && *resume) {
        FILE *in = fopen(resume, "rb");
        if (in) {
            snapshot_head have, want = snapshot_of(world, width, height, &p, 0);
            if (fread(&have, sizeof have, 1, in) != 1 || have.magic != SNAPSHOT_MAGIC) {
                fprintf(stderr, "%s is not a snapshot of this program: starting over\n", resume);
            } else {
                want.samples_done = have.samples_done;
                if (memcmp(&have, &want, sizeof have) != 0 || have.samples_done > total) {
                    fprintf(stderr, "%s belongs to another render (frame size, scene, seed, "
                                    "bounces, field of view or mode differ): refusing to "
                                    "continue it\n", resume);
                    fclose(in);
                    goto cleanup;
                }
                if (fread(accum, 16, npix, in) == npix &&
                    ptgpu_device_upload(d_accum, accum, 16 * npix) == PTGPU_OK) {
                    done = (uint32_t)have.samples_done;
                    printf("resumed %s at %u of %u spp\n", resume, done, total);
                } else {
                    fprintf(stderr, "%s is truncated: starting over\n", resume);
                }
            }
            fclose(in);
        }
    }
    uint32_t stop_after = env_number("PTGPU_STOP_AFTER", total); /* for tests */

    double t0 = seconds_now();
    while (done < total && done < stop_after) {
        p.spp = total - done < per_pass ? total - done : per_pass;
        p.sample_offset = done;
        rc = ptgpu_accumulate_device(dev, d_accum, width, height, &p, NULL);
        if (rc == PTGPU_OK) rc = ptgpu_resolve_device(d_accum, d_frame, npix, NULL);
        if (rc == PTGPU_OK) rc = ptgpu_device_download(frame, d_frame, 3 * npix);
        if (rc != PTGPU_OK) {
            fprintf(stderr, "pass failed: %s\n", ptgpu_last_error());
            goto cleanup;
        }
        done += p.spp;
        write_bitmap(path, (unsigned int)width, (unsigned int)height, frame, true);
        printf("%u/%u spp  %.3f s\n", done, total, seconds_now() - t0);
    }
    if (resume && *resume && ptgpu_device_download(accum, d_accum, 16 * npix) == PTGPU_OK) {
        /* written beside the old snapshot and renamed over it: a crash while
         * writing leaves the previous one intact */
        size_t n = strlen(resume) + 5;
        char *tmp = malloc(n);
        if (tmp != NULL) {
            snprintf(tmp, n, "%s.tmp", resume);
            FILE *out = fopen(tmp, "wb");
            const snapshot_head head = snapshot_of(world, width, height, &p, done);
            int ok = out != NULL && fwrite(&head, sizeof head, 1, out) == 1 &&
                     fwrite(accum, 16, npix, out) == npix;
            if (out != NULL && fclose(out) != 0)
                ok = 0;
            if (!ok || rename(tmp, resume) != 0) {
                fprintf(stderr, "could not write snapshot %s\n", resume);
                remove(tmp);
            }
            free(tmp);
        }
    }
    status = 0;
cleanup:
    ptgpu_device_free(d_accum);
    ptgpu_device_free(d_frame);
    ptgpu_scene_free(dev);
    free(frame);
    free(accum);
    return status;
}

static int multi_device(scene *world, size_t width, size_t height, char *path, uint32_t n) {
    ptgpu_params p;
    ptgpu_params_from_env(&p);
    int have = ptgpu_device_count();
    if (have < 1 || n < 1 || n > 64) {
        fprintf(stderr, "PTGPU_MULTI: %s\n", have < 1 ? ptgpu_last_error() : "1..64 devices");
        return 1;
    }
    int devices[64];
    for (uint32_t i = 0; i < n; i++)
        devices[i] = (int)(i % (uint32_t)have);
    const uint32_t flags = ((uint32_t)have >= n && n >= 2) ? 0u : PTGPU_MULTI_PEER_COPY;
    const size_t bytes = 3 * width * height;
    ptgpu_multi *m = NULL;
    void *d_frame = NULL;
    uint8_t *frame = malloc(bytes);
    int status = 1;
    int rc = ptgpu_set_device(devices[0]);
    if (rc == PTGPU_OK) rc = ptgpu_multi_create(devices, (int)n, flags, &m);
    if (rc == PTGPU_OK) rc = ptgpu_multi_upload(m, world);
    if (rc == PTGPU_OK) rc = ptgpu_device_alloc(bytes, &d_frame);
    double t0 = seconds_now();
    if (rc == PTGPU_OK) rc = ptgpu_multi_render(m, d_frame, width, height, &p, 0);
    double t1 = seconds_now();
    if (rc == PTGPU_OK && frame != NULL) rc = ptgpu_device_download(frame, d_frame, bytes);
    if (rc != PTGPU_OK || frame == NULL) {
        fprintf(stderr, "multi-device render failed: %s\n", ptgpu_last_error());
    } else {
        write_bitmap(path, (unsigned int)width, (unsigned int)height, frame, true);
        printf("%zux%zu over %u devices (%s): %.3f ms -> %s\n", width, height, n,
               ptgpu_multi_transport(m), (t1 - t0) * 1e3, path);
        status = 0;
    }
    ptgpu_device_free(d_frame);
    ptgpu_multi_destroy(m);
    free(frame);
    return status;
}

int main(int argc, char **argv) {
    int config = argc > 1 ? atoi(argv[1]) : 0;
    size_t width = argc > 2 ? (size_t)atol(argv[2]) : WIDTH;
    size_t height = argc > 3 ? (size_t)atol(argv[3]) : HEIGHT;
    char *path = argc > 4 ? argv[4] : "output.bmp";
    if ((argc > 2 && atol(argv[2]) <= 0) || (argc > 3 && atol(argv[3]) <= 0) ||
        width > 1048560 || height > 1048560) { /* the library's own limit: 65535 * 16 */
        fprintf(stderr, "usage: %s [config 0..5] [width] [height] [out.bmp]  (sizes 1..1048560)\n", argv[0]);
        return 2;
    }

    scene world;
    if (ptgpu_build_config_scene(config, &world) != PTGPU_OK) {
        fprintf(stderr, "unknown scene configuration %d\n", config);
        return 1;
    }

    uint32_t multi = env_number("PTGPU_MULTI", 0);
    if (multi > 0) {
        int status = multi_device(&world, width, height, path, multi);
        scene_destroy(&world);
        return status;
    }
    uint32_t per_pass = env_number("PTGPU_PROGRESSIVE", 0);
    if (per_pass > 0) {
        int status = progressive(&world, width, height, path, per_pass);
        scene_destroy(&world);
        return status;
    }

    uint8_t *frame = calloc(3 * width * height, 1);
    render_args *rows = calloc(height, sizeof *rows);
    if (frame == NULL || rows == NULL) {
        fprintf(stderr, "out of memory\n");
        return 1;
    }

    /* PTGPU_FRAMES=n renders the frame n times over (the reference's window
     * loop re-renders continuously, src/main.c:250-286) and reports the last
     * one: the steady-state cost of the legacy call sequence, first-use
     * initialisation excluded. */
    uint32_t frames = env_number("PTGPU_FRAMES", 1);
    if (frames == 0)
        frames = 1;
    double t0 = 0.0, t1 = 0.0;
    threadpool pool;
    threadpool_init(&pool, 8);
    for (uint32_t f = 0; f < frames; f++) {
        if (f + 1 == frames)
            ptgpu_stats_reset();
        t0 = seconds_now();
        for (size_t y = 0; y < height; y++) {
            rows[y].screen_y = (int)y;
            rows[y].framew = width;
            rows[y].frameh = height;
            rows[y].frame = frame;
            rows[y].world = &world;
            threadpool_add_task(&pool, (void (*)(void *))render_task, &rows[y]);
        }
        threadpool_wait_for_tasks(&pool);
        t1 = seconds_now();
    }
    threadpool_destroy(&pool);

    write_bitmap(path, (unsigned int)width, (unsigned int)height, frame, true);

    ptgpu_stats st;
    if (ptgpu_stats_read(&st) == PTGPU_OK && t1 > t0)
        printf("%zux%zu config %d: %.3f ms, %llu rays (%.1f Mray/s), %llu samples -> %s\n",
               width, height, config, (t1 - t0) * 1e3, (unsigned long long)st.rays,
               (double)st.rays / (t1 - t0) * 1e-6,
               (unsigned long long)st.samples, path);

    free(rows);
    free(frame);
    scene_destroy(&world);
    return 0;
}
