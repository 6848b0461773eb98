// pt_launch.h -- internal C++ interface between the C-ABI layer
// (ptgpu_abi.hip) and the kernels (pt_kernels.hip).
#pragma once

#include <hip/hip_runtime.h>

#include "pt_layout.h"

namespace pt {

// Byte-for-byte the legacy `ray_hit` (include/ray.h:7-12): 40 bytes.
struct DeviceRayHit {
    float distance;
    float point[3];
    float normal[3];
    uint32_t pad;
    uint64_t material;
};
static_assert(sizeof(DeviceRayHit) == 40, "must match ray_hit");

} // namespace pt

// The launchers exist twice: `pt` = the bit-exact build of the device code,
// `pt_fast` = the same sources compiled as the tolerance mode (pt_vec.h).
#define PT_DECLARE_LAUNCHERS                                                                       \
    /* Dynamic LDS one workgroup of `block` work-items needs for this launch. */                   \
    size_t megakernel_lds_bytes(const pt::LaunchArgs &a, uint32_t block, bool tree);               \
    hipError_t launch_megakernel(const pt::LaunchArgs &a, uint32_t block, bool tree,               \
                                 hipStream_t stream);                                              \
    /* Sample-queue mode: persistent waves pulling (pixel, sample) tasks, then the ordered         \
     * per-pixel sum.  `max_waves` = waves the device keeps resident. */                           \
    hipError_t launch_samplekernel(const pt::LaunchArgs &a, uint32_t block, uint32_t max_waves,    \
                                   hipStream_t stream);                                            \
    hipError_t launch_sum(const pt::LaunchArgs &a, hipStream_t stream);                            \
    /* Closest hit of each local pixel's primary ray (BVH scan): 8 bytes per pixel. */             \
    hipError_t launch_primary(const pt::LaunchArgs &a, float *primary_hits, hipStream_t stream);   \
    hipError_t launch_resolve(const float *accum, uint8_t *frame, uint64_t npixels,                \
                              hipStream_t stream);                                                 \
    /* scan: 0 plain, 1 screened linear, 2 BVH (the scene must have a hierarchy) */                \
    hipError_t launch_trace(const pt::SceneDesc &scene, const float *origins,                      \
                            const float *directions, uint64_t n, pt::DeviceRayHit *hits, int scan, \
                            hipStream_t stream);

#ifndef PT_FAST_MATH // (hidden from the pt_fast build itself: its unqualified calls would find
                     // these too, through the argument types' namespace)
namespace pt {
PT_DECLARE_LAUNCHERS
}
#endif
namespace pt_fast {
PT_DECLARE_LAUNCHERS
}

namespace pt {
// Multi-device gather (ptgpu_multi_render): `stage` holds G compact shards --
// shard r at byte offset shard_offset[r], its local rows back to back -- and
// every row goes to its place in the frame: tile t of `tile_rows` rows from
// `row_begin` on belongs to shard t mod G.
constexpr uint32_t kMaxShards = 64;
struct ScatterArgs {
    const uint8_t *stage;
    uint8_t *frame;
    uint32_t row_bytes;  // 3 * framew
    uint32_t row_begin, num_rows, tile_rows, shards;
    uint64_t shard_offset[kMaxShards];
};
hipError_t launch_scatter_tiles(const ScatterArgs &s, hipStream_t stream);
// PTGPU_FLAG_GLSL_SEMANTICS: one frame's rows by the shader's rules (pt_glsl.hip)
hipError_t launch_glsl(const LaunchArgs &a, hipStream_t stream);
} // namespace pt
