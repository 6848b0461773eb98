/*
 * scene.c -- host-side scene builder: two append-only arrays that double when
 * full.  Behaviour follows the reference's src/scene.c:19-76 (initial
 * capacity 64 objects / 16 materials, zero-filled; sky_color is left for the
 * caller to set; material ids are the insertion index), with the allocation
 * results checked, which the reference does not do (src/scene.c:9,14,21,25).
 */
#include "path_tracer_api.h"

#include <stdio.h>
#include <string.h>

enum { OBJECTS_AT_START = 64, MATERIALS_AT_START = 16 };

static void *must_alloc(void *p, const char *what) {
    if (p == NULL) {
        fprintf(stderr, "path_tracer: out of memory growing %s\n", what);
        abort();
    }
    return p;
}

/* Make room for one more element in a {ptr, count, capacity} triple. */
static void reserve_one(void **items, size_t count, size_t *capacity,
                        size_t item_size, const char *what) {
    if (count < *capacity)
        return;
    size_t wider = *capacity ? *capacity * 2 : 1;
    *items = must_alloc(realloc(*items, wider * item_size), what);
    *capacity = wider;
}

void scene_init(scene *s) {
    s->objects = must_alloc(calloc(OBJECTS_AT_START, sizeof(shape)), "objects");
    s->num_objects = 0;
    s->max_objects = OBJECTS_AT_START;

    s->materials = must_alloc(
        calloc(MATERIALS_AT_START, sizeof(shape_material)), "materials");
    s->num_materials = 0;
    s->max_materials = MATERIALS_AT_START;
}

size_t scene_add_material(scene *s, const vec3s albedo, const float roughness,
                          const float metallicity, const vec3s emission_color,
                          const float emission_strength,
                          const float transparency,
                          const float refractive_index) {
    reserve_one((void **)&s->materials, s->num_materials, &s->max_materials,
                sizeof(shape_material), "materials");

    shape_material *m = &s->materials[s->num_materials];
    memset(m, 0, sizeof *m);
    m->albedo = albedo;
    m->roughness = roughness;
    m->metallicity = metallicity;
    m->emission_color = emission_color;
    m->emission_strength = emission_strength;
    m->transparency = transparency;
    m->refractive_index = refractive_index;
    return s->num_materials++;
}

static shape *next_object(scene *s) {
    reserve_one((void **)&s->objects, s->num_objects, &s->max_objects,
                sizeof(shape), "objects");
    shape *o = &s->objects[s->num_objects++];
    memset(o, 0, sizeof *o);
    return o;
}

void scene_add_sphere(scene *s, const vec3s center, const float radius,
                      const size_t material) {
    shape *o = next_object(s);
    o->tag = SPHERE;
    o->material = material;
    o->sphere.center = center;
    o->sphere.radius = radius;
}

void scene_add_triangle(scene *s, const vec3s v0, const vec3s v1,
                        const vec3s v2, const size_t material) {
    shape *o = next_object(s);
    o->tag = TRIANGLE;
    o->material = material;
    o->triangle.v0 = v0;
    o->triangle.v1 = v1;
    o->triangle.v2 = v2;
}

void scene_destroy(scene *s) {
    free(s->objects);
    free(s->materials);
    s->objects = NULL;
    s->materials = NULL;
    s->num_objects = s->max_objects = 0;
    s->num_materials = s->max_materials = 0;
}
