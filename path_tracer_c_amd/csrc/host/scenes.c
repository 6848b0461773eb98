/*
 * scenes.c -- the synthetic scenes of BASELINE.json's five configurations,
 * built through the legacy scene_* API so CPU and GPU runs consume the very
 * same `scene` struct.  The reference ships exactly one scene, inside its GL
 * driver (src/main.c:202-230); it is config 0 here.  Camera conventions are
 * the renderer's: eye at the origin looking down +z, +y up, fov 90 degrees
 * (src/renderer.c:76-85).
 *
 *   0  reference demo: 4 spheres + 2 ground triangles, black sky
 *   1  one red sphere on a ground sphere under a blue sky        (C1)
 *   2  "Cornell-style" room of exactly 8 spheres, front open     (C2, C3, C4)
 *   5  1000 random spheres over a ground sphere, blue sky         (C5)
 *   3 and 4 are config 2's scene (they differ in resolution / spp only).
 */
#include "ptgpu.h"

#include <string.h>

static vec3s v3(double x, double y, double z) {
    /* double -> float per component, like the reference's brace initialisers
     * with double literals (src/main.c:209-219). */
    vec3s v;
    v.x = (float)x;
    v.y = (float)y;
    v.z = (float)z;
    return v;
}

static const vec3s BLACK = {{0.0f, 0.0f, 0.0f}};

/* Matte (perturbed-mirror "diffuse") material: no emission, opaque. */
static size_t matte(scene *s, double r, double g, double b, double roughness) {
    return scene_add_material(s, v3(r, g, b), (float)roughness, 0.0f, BLACK,
                              0.0f, 0.0f, 1.0f);
}

static void build_demo(scene *w) {
    /* Values are the reference scene's data (src/main.c:206-230). */
    w->sky_color = v3(0.0, 0.0, 0.0);
    size_t sun = scene_add_material(w, v3(0.9372, 0.7490, 0.0157), 0.3f, 1.0f,
                                    v3(0.9372, 0.7490, 0.0157), 10.0f, 0.0f,
                                    1.0f);
    size_t red_plastic = scene_add_material(w, v3(1, 0, 0), 0.85f, 0.5f,
                                            v3(1, 0, 0), 0.0f, 0.0f, 1.0f);
    size_t green_grass = scene_add_material(
        w, v3(65.0 / 255.0, 152.0 / 255.0, 10.0 / 255.0), 1.0f, 0.1f,
        v3(65.0 / 255.0, 152.0 / 255.0, 10.0 / 255.0), 0.0f, 0.0f, 1.0f);
    size_t mirror = scene_add_material(w, v3(1, 1, 1), 0.0f, 1.0f, BLACK, 0.0f,
                                       0.0f, 1.0f);
    size_t glass = scene_add_material(w, v3(1, 1, 1), 0.0f, 0.0f, v3(1, 1, 1),
                                      0.0f, 1.0f, 1.52f);

    scene_add_sphere(w, v3(80.0, 50.0, 100.0), 40.0f, sun);
    scene_add_sphere(w, v3(-2.0, 0.0, 4.0), 1.0f, red_plastic);
    scene_add_sphere(w, v3(2.5, -0.2, 5.0), 1.0f, glass);
    scene_add_sphere(w, v3(0.0, 1.5, 10.0), 2.5f, mirror);

    scene_add_triangle(w, v3(-50, -1, -50), v3(50, -1, -50), v3(50, -1, 50),
                       green_grass);
    scene_add_triangle(w, v3(-50, -1, -50), v3(-50, -1, 50), v3(50, -1, 50),
                       green_grass);
}

static void build_c1(scene *w) {
    /* sky (135,206,235)/255, a red rough sphere
     * and the ground sphere the reference left commented out
     * (src/main.c:224). */
    w->sky_color = v3(135.0 / 255.0, 206.0 / 255.0, 235.0 / 255.0);
    size_t red = matte(w, 1.0, 0.0, 0.0, 0.85);
    size_t grass = matte(w, 65.0 / 255.0, 152.0 / 255.0, 10.0 / 255.0, 1.0);
    scene_add_sphere(w, v3(0.0, 0.0, 4.0), 1.0f, red);
    scene_add_sphere(w, v3(0.0, -101.0, 5.0), 100.0f, grass);
}

static void build_cornell(scene *w) {
    /* Room x in [-3,3], y in [-2,2], back wall at z = 8; the five walls are
     * spheres of radius 1000 seen from inside; the front (behind the eye) is
     * OPEN and the sky is black, so a path that leaves through the front
     * ends.  Exactly 8 spheres, as BASELINE.json's config names. */
    const double R = 1000.0;
    w->sky_color = v3(0.0, 0.0, 0.0);
    size_t white = matte(w, 0.73, 0.73, 0.73, 1.0);
    size_t red = matte(w, 0.65, 0.05, 0.05, 1.0);
    size_t green = matte(w, 0.12, 0.45, 0.15, 1.0);
    size_t lamp = scene_add_material(w, v3(1.0, 1.0, 1.0), 1.0f, 0.0f,
                                     v3(1.0, 0.93, 0.82), 10.0f, 0.0f, 1.0f);
    size_t mirror = scene_add_material(w, v3(0.95, 0.95, 0.95), 0.0f, 1.0f,
                                       BLACK, 0.0f, 0.0f, 1.0f);
    size_t glass = scene_add_material(w, v3(1.0, 1.0, 1.0), 0.0f, 0.0f, BLACK,
                                      0.0f, 1.0f, 1.5f);

    scene_add_sphere(w, v3(-(R + 3.0), 0.0, 5.0), (float)R, red);   /* left */
    scene_add_sphere(w, v3(R + 3.0, 0.0, 5.0), (float)R, green);    /* right */
    scene_add_sphere(w, v3(0.0, 0.0, R + 8.0), (float)R, white);    /* back */
    scene_add_sphere(w, v3(0.0, -(R + 2.0), 5.0), (float)R, white); /* floor */
    scene_add_sphere(w, v3(0.0, R + 2.0, 5.0), (float)R, white);    /* ceiling */
    scene_add_sphere(w, v3(0.0, 4.4, 5.0), 3.0f, lamp);   /* cap below y = 2 */
    scene_add_sphere(w, v3(-1.3, -1.0, 5.8), 1.0f, mirror);
    scene_add_sphere(w, v3(1.3, -1.2, 4.4), 0.8f, glass);
}

/* Scene-generation stream: xorshift32 seeded with 42, 24-bit uniform floats.
 * (Independent of the render stream in ptgpu_rng.h.)
 * The state is the caller's: two threads may build configuration 5 at the
 * same time (a file-level state let them scramble each other's scene -- found
 * by the host-thread stress test of round 2). */
static double gen_uniform_from(uint32_t *state, double lo, double hi) {
    uint32_t x = *state;
    x ^= x << 13;
    x ^= x >> 17;
    x ^= x << 5;
    *state = x;
    double u = (double)(x >> 8) / 16777216.0;
    return lo + (hi - lo) * u;
}

static void build_c5(scene *w) {
    /* 1000 spheres, centres uniform in
     * x [-20,20], y [-1,10], z [3,60], radius uniform [0.2,0.8], material
     * uniform over 16; overlapping spheres are NOT rejected.  A ground sphere
     * is object 1001.  The 16 materials: 10 matte colours, 2 lamps, 2
     * mirrors, 2 glasses. */
    uint32_t gen_state = 42u;
#define gen_uniform(lo, hi) gen_uniform_from(&gen_state, (lo), (hi))
    w->sky_color = v3(135.0 / 255.0, 206.0 / 255.0, 235.0 / 255.0);
    size_t first = 0;
    for (int i = 0; i < 16; i++) {
        double r = gen_uniform(0.1, 0.95), g = gen_uniform(0.1, 0.95),
               b = gen_uniform(0.1, 0.95), rough = gen_uniform(0.05, 1.0);
        size_t id;
        if (i < 10)
            id = matte(w, r, g, b, rough);
        else if (i < 12)
            id = scene_add_material(w, v3(r, g, b), (float)rough, 0.0f,
                                    v3(r, g, b), 6.0f, 0.0f, 1.0f);
        else if (i < 14)
            id = scene_add_material(w, v3(0.9, 0.9, 0.9), 0.0f, 1.0f, BLACK,
                                    0.0f, 0.0f, 1.0f);
        else
            id = scene_add_material(w, v3(1.0, 1.0, 1.0), 0.0f, 0.0f, BLACK,
                                    0.0f, 1.0f, 1.5f);
        if (i == 0)
            first = id;
    }
    for (int i = 0; i < 1000; i++) {
        double x = gen_uniform(-20.0, 20.0), y = gen_uniform(-1.0, 10.0),
               z = gen_uniform(3.0, 60.0), rad = gen_uniform(0.2, 0.8);
        size_t m = first + (size_t)gen_uniform(0.0, 16.0);
        if (m > first + 15)
            m = first + 15;
        scene_add_sphere(w, v3(x, y, z), (float)rad, m);
    }
    size_t ground = matte(w, 0.45, 0.5, 0.4, 1.0);
    scene_add_sphere(w, v3(0.0, -1001.8, 30.0), 1000.0f, ground);
#undef gen_uniform
}

int ptgpu_build_config_scene(int config, scene *out) {
    if (out == NULL || config < 0 || config > 5)
        return PTGPU_ERR_INVALID;
    memset(out, 0, sizeof *out);
    scene_init(out);
    switch (config) {
    case 0: build_demo(out); break;
    case 1: build_c1(out); break;
    case 2:
    case 3:
    case 4: build_cornell(out); break;
    default: build_c5(out); break;
    }
    return PTGPU_OK;
}
