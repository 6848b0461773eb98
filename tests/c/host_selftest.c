/*
 * host_selftest.c -- the host-C substrate and the CPU oracle under
 * AddressSanitizer + UndefinedBehaviorSanitizer (the GPU pool has no device
 * sanitizer; host code is where one can run).  Built and run by
 * tests/test_sanitizers.py:
 *   gcc -fsanitize=address,undefined -fno-omit-frame-pointer -Iinclude -Ioracle
 *       tests/c/host_selftest.c csrc/host/{scene,vector,bitmap,scenes}.c
 *       oracle/pt_oracle.c -lm -lpthread
 * Exit status 0 = every check passed and no sanitizer report.
 */
#include "ptgpu.h"
#include "pt_oracle.h"

#include <stdio.h>
#include <string.h>

static int failures = 0;
#define CHECK(cond)                                                            \
    do {                                                                       \
        if (!(cond)) {                                                         \
            fprintf(stderr, "CHECK failed at line %d: %s\n", __LINE__, #cond); \
            failures++;                                                        \
        }                                                                      \
    } while (0)

static vec3s v(float x, float y, float z) {
    vec3s r;
    r.x = x; r.y = y; r.z = z;
    return r;
}

int main(int argc, char **argv) {
    const char *bmp = argc > 1 ? argv[1] : "/tmp/host_selftest.bmp";

    /* scene growth far past both initial capacities */
    scene s;
    scene_init(&s);
    s.sky_color = v(0.2f, 0.3f, 0.4f);
    for (int i = 0; i < 100; i++)
        CHECK(scene_add_material(&s, v(0.5f, 0.5f, 0.5f), 0.1f * (float)(i % 10), 0, v(0, 0, 0),
                                 0, 0, 1) == (size_t)i);
    for (int i = 0; i < 1000; i++) {
        if (i & 1)
            scene_add_sphere(&s, v((float)(i % 13) - 6, (float)(i % 7) - 3, 5 + (float)(i % 11)), 0.4f, (size_t)(i % 100));
        else
            scene_add_triangle(&s, v((float)(i % 9) - 4, -1, 3), v((float)(i % 9) - 3, -1, 3),
                               v((float)(i % 9) - 4, 0, 4), (size_t)(i % 100));
    }
    CHECK(s.num_objects == 1000 && s.max_objects == 1024);
    CHECK(s.num_materials == 100 && s.max_materials == 128);

    /* oracle over it: rows, accumulate, multi-threaded pool, both streams */
    enum { W = 40, H = 24 };
    static uint8_t frame[3 * W * H], frame_mt[3 * W * H];
    static float accum[4 * W * H];
    oracle_params p = {4, 4, 1.5707964f, 7, 0, ORACLE_RNG_XORSHIFT};
    uint64_t rays = oracle_render_rows(&s, frame, W, H, 0, H, &p);
    uint64_t rays_mt = oracle_render_mt(&s, frame_mt, W, H, 0, H, 1, &p, 5);
    CHECK(rays == rays_mt && memcmp(frame, frame_mt, sizeof frame) == 0);
    memset(accum, 0, sizeof accum);
    CHECK(oracle_accumulate_rows(&s, accum, W, H, 0, H, &p) == rays);
    p.rng_mode = ORACLE_RNG_LIBC;
    oracle_srandom(1);
    CHECK(oracle_render_mt(&s, frame_mt, W, H, 0, H, 3, &p, 4) > 0);
    float rgb[3];
    p.rng_mode = ORACLE_RNG_XORSHIFT;
    oracle_sample(&s, W, H, 3, 5, 2, &p, rgb);
    CHECK(rgb[0] >= 0 && rgb[0] <= 1);
    ray_hit h = oracle_trace_ray(v(0, 0, 0), v(0, 0, 1), &s);
    CHECK(h.distance > 0 || h.distance == -1.0f);

    /* BMP writer, padded and unpadded widths, both orientations */
    write_bitmap((char *)bmp, W, H, frame, true);
    static uint8_t odd[3 * 5 * 3];
    write_bitmap((char *)bmp, 5, 3, odd, false);
    write_bitmap((char *)bmp, 1, 1, odd, true);

    /* pointer vector */
    vector vec;
    vector_init_with_capacity(&vec, 0);
    for (size_t i = 1; i <= 100; i++)
        vector_push(&vec, (void *)i);
    for (size_t i = 100; i >= 1; i--)
        CHECK(vector_pop(&vec) == (void *)i);
    CHECK(vector_pop(&vec) == NULL);
    vector_free(&vec);

    /* the five configuration scenes */
    for (int c = 0; c <= 5; c++) {
        scene w;
        CHECK(ptgpu_build_config_scene(c, &w) == PTGPU_OK);
        CHECK(w.num_objects > 0);
        oracle_params q = {1, 2, 1.5707964f, 1, 0, ORACLE_RNG_XORSHIFT};
        static uint8_t small[3 * 8 * 6];
        CHECK(oracle_render_rows(&w, small, 8, 6, 0, 6, &q) >= 48);
        scene_destroy(&w);
    }
    CHECK(ptgpu_build_config_scene(6, &s) == PTGPU_ERR_INVALID);

    scene_destroy(&s);
    if (failures == 0)
        puts("host_selftest: ok");
    return failures ? 1 : 0;
}
