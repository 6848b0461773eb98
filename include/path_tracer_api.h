/*
 * path_tracer_api.h -- the legacy host-C surface of path_tracer_c, restated.
 *
 * Every type and prototype in this file keeps the name, field order, size and
 * alignment of the reference project's public headers, so a host program
 * written against the reference (scene construction, the row-task pool, the
 * BMP writer) recompiles against this tree unchanged.  Nothing here is new
 * API; the new, explicit MI355X entry points live in ptgpu.h.
 *
 * Mirrors (reference file:line):
 *   vec3s / vec4s         cglm types-struct.h (absent from the reference tree;
 *                         layout from the reference's uses: src/ray.c:18,
 *                         src/renderer.c:112-114, src/main.c:36)
 *   sphere, triangle,
 *   shape_material, shape include/shapes.h:6-36
 *   scene, scene_*        include/scene.h:7-30
 *   ray_hit, trace_ray    include/ray.h:7-15
 *   render_args,
 *   render_task           include/renderer.h:6-13
 *   vector, vector_*      include/vector.h:5-16
 *   threadpool,
 *   threadpool_*          include/threadpool.h:6-20
 *   write_bitmap          include/bitmap.h:4-5
 *
 * The thin headers scene.h / shapes.h / ray.h / renderer.h / vector.h /
 * threadpool.h / bitmap.h next to this file only include it, so the
 * reference's `#include "scene.h"` etc. keep working.
 */
#ifndef PATH_TRACER_API_H
#define PATH_TRACER_API_H

#include <pthread.h>
#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- 3- and 4-float vectors --------------------------------------------- */
/* A caller that has the real cglm on its include path keeps using it; the
 * fallback below is layout-identical (12 B / align 4 and 16 B / align 16). */
#if defined(__has_include)
#if __has_include("cglm/types-struct.h")
#include "cglm/types-struct.h"
#define PT_API_HAVE_CGLM 1
#endif
#endif

#ifndef PT_API_HAVE_CGLM
typedef union vec3s {
    float raw[3];
    struct {
        float x, y, z;
    };
    struct {
        float r, g, b;
    };
} vec3s;

typedef union
#if defined(__GNUC__)
    __attribute__((aligned(16)))
#endif
    vec4s {
    float raw[4];
    struct {
        float x, y, z, w;
    };
    struct {
        float r, g, b, a;
    };
} vec4s;
#endif /* !PT_API_HAVE_CGLM */

/* ---- shapes and materials (the scene's element types) ------------------- */
typedef struct {
    vec3s center;
    float radius;
} sphere; /* 16 B */

typedef struct {
    vec3s v0;
    vec3s v1;
    vec3s v2;
} triangle; /* 36 B */

typedef struct {
    vec3s albedo;
    float roughness;
    float metallicity; /* stored, never read by the renderer */

    vec3s emission_color;
    float emission_strength;

    float transparency;
    float refractive_index; /* stored; the CPU renderer hard-codes 1.5 */
} shape_material; /* 44 B */

typedef struct {
    enum { SPHERE, TRIANGLE } tag;
    size_t material;
    union {
        sphere sphere;
        triangle triangle;
    };
} shape; /* 56 B: tag@0, material@8, payload@16 */

/* ---- scene: two append-only arrays + sky colour -------------------------- */
typedef struct {
    shape *objects;
    size_t num_objects;
    size_t max_objects;
    shape_material *materials;
    size_t num_materials;
    size_t max_materials;
    vec3s sky_color; /* set by the caller, not by scene_init */
} scene; /* 64 B */

void scene_init(scene *s);
size_t scene_add_material(scene *s, const vec3s albedo, const float roughness,
                          const float metallicity, const vec3s emission_color,
                          const float emission_strength,
                          const float transparency,
                          const float refractive_index);
void scene_add_sphere(scene *s, const vec3s center, const float radius,
                      const size_t material);
void scene_add_triangle(scene *s, const vec3s v0, const vec3s v1,
                        const vec3s v2, const size_t material);
void scene_destroy(scene *s);

/* ---- closest-hit query ---------------------------------------------------- */
typedef struct {
    float distance; /* < 0: miss (other fields then unspecified) */
    vec3s point;
    vec3s normal;
    size_t material;
} ray_hit; /* 40 B */

/* One ray against the whole scene.  In this tree the query runs on the GPU
 * (a one-ray launch of the same device function the megakernel uses); batch
 * callers should use ptgpu_trace_rays(). */
ray_hit trace_ray(const vec3s origin, const vec3s direction,
                  const scene *world);

/* ---- the row task ---------------------------------------------------------- */
typedef struct {
    int screen_y;
    size_t framew, frameh;
    uint8_t *frame; /* 3*framew*frameh bytes, RGB8, row 0 = top */
    scene *world;
} render_args; /* 40 B */

/* Renders image row `screen_y` into `frame`.  Called directly it is a one-row
 * GPU launch; handed to threadpool_add_task it is recognised by address and
 * all queued rows of a frame are rendered by one megakernel launch. */
void render_task(render_args *arg);

/* ---- growable pointer array (the pool's queue type) ----------------------- */
typedef struct {
    void **data;
    size_t size;
    size_t capacity;
} vector;

void vector_init(vector *v);
void vector_init_with_capacity(vector *v, size_t initial_capacity);
void vector_free(vector *v);
void vector_push(vector *v, void *element);
void *vector_pop(vector *v);

/* ---- the task pool (caller-allocated, fields public, 192 B) -------------- */
typedef struct {
    vector threads;
    vector tasks;
    pthread_mutex_t tasks_mutex;
    pthread_cond_t tasks_available_cond;
    pthread_cond_t tasks_exhausted_cond;
    bool threads_running;
} threadpool;

/* Tasks whose function is render_task are queued for the GPU (rendered by
 * threadpool_wait_for_tasks); any other task is run by one of `num_threads`
 * worker pthreads at once, newest first, as src/threadpool.c:14-41 does. */
void threadpool_init(threadpool *pool, size_t num_threads);
/* Running tasks finish; queued ones are dropped (src/threadpool.c:60-82). */
void threadpool_destroy(threadpool *pool);
/* A render_task row is COPIED from *arg here, when it is queued (the reference
 * copies it when a worker starts the task, src/renderer.c:93: a caller that
 * patches or reuses the struct between queuing and waiting gets the values of
 * queuing time here, a race there).  Any other task gets `arg` itself. */
void threadpool_add_task(threadpool *pool, void (*f)(void *), void *arg);
/* Returns only when every task added so far has completed. */
void threadpool_wait_for_tasks(threadpool *pool);

/* ---- 24-bit BMP writer ----------------------------------------------------- */
void write_bitmap(char *filename, unsigned int imgwidth, unsigned int imgheight,
                  uint8_t *pixels, bool y_inverted);

/* ---- layout contract (x86-64 SysV), checked at compile time -------------- */
#if defined(__x86_64__) && !defined(PT_API_NO_LAYOUT_ASSERTS)
#define PT_API_ASSERT(c, m) _Static_assert(c, m)
#ifdef __cplusplus
#undef PT_API_ASSERT
#define PT_API_ASSERT(c, m) static_assert(c, m)
#endif
PT_API_ASSERT(sizeof(vec3s) == 12, "vec3s is 3 packed floats");
PT_API_ASSERT(sizeof(sphere) == 16, "sphere");
PT_API_ASSERT(sizeof(triangle) == 36, "triangle");
PT_API_ASSERT(sizeof(shape_material) == 44, "shape_material");
PT_API_ASSERT(offsetof(shape_material, emission_color) == 20, "material.Le");
PT_API_ASSERT(offsetof(shape_material, transparency) == 36, "material.t");
PT_API_ASSERT(sizeof(shape) == 56, "shape");
PT_API_ASSERT(offsetof(shape, material) == 8, "shape.material");
PT_API_ASSERT(offsetof(shape, sphere) == 16, "shape payload");
PT_API_ASSERT(sizeof(scene) == 64, "scene");
PT_API_ASSERT(offsetof(scene, sky_color) == 48, "scene.sky_color");
PT_API_ASSERT(sizeof(ray_hit) == 40, "ray_hit");
PT_API_ASSERT(offsetof(ray_hit, material) == 32, "ray_hit.material");
PT_API_ASSERT(sizeof(render_args) == 40, "render_args");
PT_API_ASSERT(offsetof(render_args, frame) == 24, "render_args.frame");
PT_API_ASSERT(sizeof(vector) == 24, "vector");
PT_API_ASSERT(sizeof(threadpool) == 192, "threadpool");
#endif

#ifdef __cplusplus
}
#endif
#endif /* PATH_TRACER_API_H */
