/*
 * vector.c -- the growable void* array the task pool's public struct embeds
 * (reference: include/vector.h:5-16, src/vector.c:3-31: default capacity 8,
 * doubling, pop from the back, NULL when empty).
 */
#include "path_tracer_api.h"

void vector_init_with_capacity(vector *v, size_t initial_capacity) {
    v->size = 0;
    v->capacity = initial_capacity;
    v->data = malloc((initial_capacity ? initial_capacity : 1) * sizeof(void *));
}

void vector_init(vector *v) { vector_init_with_capacity(v, 8); }

void vector_free(vector *v) {
    free(v->data);
    v->data = NULL;
    v->size = v->capacity = 0;
}

void vector_push(vector *v, void *element) {
    if (v->size >= v->capacity) {
        size_t wider = v->capacity ? 2 * v->capacity : 8;
        void **grown = realloc(v->data, wider * sizeof(void *));
        if (grown == NULL)
            abort();
        v->data = grown;
        v->capacity = wider;
    }
    v->data[v->size] = element;
    v->size += 1;
}

void *vector_pop(vector *v) {
    if (v->size == 0)
        return NULL;
    v->size -= 1;
    return v->data[v->size];
}
