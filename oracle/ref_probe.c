/*
 * ref_probe.c -- TEST INFRASTRUCTURE.  Compiled by oracle/ref.mk against the
 * reference's OWN headers (-I/root/reference/include), never against ours:
 * reports what include/threadpool.h:6-13 and include/vector.h:5-9 compile to
 * on this ABI, so tests/test_ref_pins.py can hold include/path_tracer_api.h
 * to it.  (scene.h, shapes.h, ray.h and renderer.h include cglm, which is
 * absent: their layouts cannot be probed and stay pinned by SURVEY 2.2 only.)
 */
#include "bitmap.h"
#include "threadpool.h"
#include "vector.h"

#include <stddef.h>

#define FIELD(T, f) { #T "." #f, offsetof(T, f), sizeof(((T *)0)->f) }

struct ref_field {
    const char *name;
    size_t offset, size;
};

static const struct ref_field fields[] = {
    FIELD(vector, data),
    FIELD(vector, size),
    FIELD(vector, capacity),
    FIELD(threadpool, threads),
    FIELD(threadpool, tasks),
    FIELD(threadpool, tasks_mutex),
    FIELD(threadpool, tasks_available_cond),
    FIELD(threadpool, tasks_exhausted_cond),
    FIELD(threadpool, threads_running),
};

size_t ref_sizeof_vector(void) { return sizeof(vector); }
size_t ref_sizeof_threadpool(void) { return sizeof(threadpool); }
size_t ref_alignof_threadpool(void) { return _Alignof(threadpool); }
size_t ref_num_fields(void) { return sizeof fields / sizeof fields[0]; }
const char *ref_field_name(size_t i) { return fields[i].name; }
size_t ref_field_offset(size_t i) { return fields[i].offset; }
size_t ref_field_size(size_t i) { return fields[i].size; }
