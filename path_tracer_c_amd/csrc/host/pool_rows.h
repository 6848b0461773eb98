/* Internal to libptgpu.so: the one call from the pool shim (host/threadpool.c,
 * plain C) into the GPU path (ptgpu_abi.hip).  Not part of the C-ABI. */
#ifndef PTGPU_POOL_ROWS_H
#define PTGPU_POOL_ROWS_H

#include "path_tracer_api.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Renders the rows the descriptors name (any order, any mix of frames, sizes
 * and scenes): grouped by (frame, size, scene) and rendered as contiguous row
 * runs, one launch per run, rows copied into each `frame`.  Returns when every
 * row is in place.  No HIP device: prints the HIP error and aborts, as the
 * other legacy entry points do. */
__attribute__((visibility("hidden"))) void ptgpu_pool_render_rows(const render_args *rows, size_t n);

#ifdef __cplusplus
}
#endif

#endif
