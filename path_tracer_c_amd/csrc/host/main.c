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
 * PTGPU_BOUNCES and PTGPU_SEED (defaults: the reference's 16 / 4).
 */
#include "ptgpu.h"

#include <stdio.h>
#include <stdlib.h>
#include <time.h>

#define WIDTH 1280 /* src/main.c:17-18 */
#define HEIGHT 800

static double seconds_now(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

int main(int argc, char **argv) {
    int config = argc > 1 ? atoi(argv[1]) : 0;
    size_t width = argc > 2 ? (size_t)atol(argv[2]) : WIDTH;
    size_t height = argc > 3 ? (size_t)atol(argv[3]) : HEIGHT;
    char *path = argc > 4 ? argv[4] : "output.bmp";

    scene world;
    if (ptgpu_build_config_scene(config, &world) != PTGPU_OK) {
        fprintf(stderr, "unknown scene configuration %d\n", config);
        return 1;
    }

    uint8_t *frame = calloc(3 * width * height, 1);
    render_args *rows = calloc(height, sizeof *rows);
    if (frame == NULL || rows == NULL) {
        fprintf(stderr, "out of memory\n");
        return 1;
    }

    double t0 = seconds_now();
    threadpool pool;
    threadpool_init(&pool, 8);
    for (size_t y = 0; y < height; y++) {
        rows[y].screen_y = (int)y;
        rows[y].framew = width;
        rows[y].frameh = height;
        rows[y].frame = frame;
        rows[y].world = &world;
        threadpool_add_task(&pool, (void (*)(void *))render_task, &rows[y]);
    }
    threadpool_wait_for_tasks(&pool);
    threadpool_destroy(&pool);
    double t1 = seconds_now();

    write_bitmap(path, (unsigned int)width, (unsigned int)height, frame, true);

    ptgpu_stats st;
    if (ptgpu_stats_read(&st) == PTGPU_OK && t1 > t0)
        printf("%zux%zu config %d: %.3f s, %llu rays (%.1f Mray/s), %llu samples -> %s\n",
               width, height, config, t1 - t0, (unsigned long long)st.rays,
               (double)st.rays / (t1 - t0) * 1e-6,
               (unsigned long long)st.samples, path);

    free(rows);
    free(frame);
    scene_destroy(&world);
    return 0;
}
