// pt_layout.h -- how a scene lies in HBM / LDS for the megakernel, and the
// argument block of one launch.  Shared by the uploader (host) and the
// kernels (device).
//
// The reference scans an array of 56-byte tagged unions by value
// (include/shapes.h:29-36, src/ray.c:12-13) and fetches a 44-byte material per
// hit (src/renderer.c:34).  Device layout instead, all 16-byte records:
//
//   geom[i]  float4   sphere  : centre.xyz, radius*radius        (w >= 0)
//                     triangle: centre and -(radius^2) of a bounding sphere,
//                               slightly inflated; the sign bit of w is the
//                               record tag (spheres have it clear, also r = 0)
//   tri[4k..4k+3]     float4  v0 | edge1 = v1-v0 | edge2 = v2-v0 | unit normal
//                     crossn(v0-v1, v0-v2) (src/ray.c:50-52), w unused
//   meta[i]  uint32   material index
//   link[i]  uint32   triangle ordinal k (0 for spheres)
//   mat[2m], mat[2m+1] float4 albedo.rgb, roughness*0.5 |
//                     emission_color*emission_strength, transparency
//
// Object ORDER is the scene's: the closest-hit rule of src/ray.c:25,45 is
// order dependent in its degenerate cases (a hit at distance <= 0 is replaced
// by any later hit), so no reordering by type.  Everything precomputed here
// is the value the per-ray code would compute with the same binary32
// operations (r*r, v1-v0, the unit normal, Le), so hoisting it changes no bit.
#pragma once

#include <cstddef>
#include <cstdint>

#if defined(__HIPCC__)
#define PT_LAYOUT_HD __host__ __device__ inline
#else
#define PT_LAYOUT_HD inline
#endif

namespace pt {

constexpr uint32_t kMetaMaterialMask = 0x7FFFFFFFu;
constexpr uint32_t kGlslMaxBounces = 14; // pt_glsl.hip keeps a ray stack of this + 2 per lane

struct F4 { // plain 16-byte record usable on host and device
    float x, y, z, w;
};

// What the shader-rules mode (PTGPU_FLAG_GLSL_SEMANTICS, pt_glsl.hip) needs to cull with
// the hierarchy: its sphere test is a plain quadratic whose rounding error grows with the
// distance to the sphere (|error of r^2 - perp^2| <~ 12 eps (D^2 + r^2)), so the walk pads
// the boxes by what that error can reach for the given ray -- which takes the extent and
// the radii of the scene WITHOUT the few spheres that dwarf it (a radius-1000 ground
// under unit-size objects); those are outside the hierarchy and tested for every ray
// in every mode (bvh_build.hpp, `apart`).
constexpr uint32_t kGlslBigMax = 8;
struct GlslCull {
    float lo[3], hi[3];       // box around every object of the hierarchy
    float r2_min, r2_max;     // squared radii of its spheres (0, 0: none)
    uint32_t last_scanned;    // the object the shader's two-array scan visits last
                              // (last triangle, else last sphere; 0xFFFFFFFF: empty scene)
};

// One device-resident scene: a single allocation, records at these offsets.
struct SceneDesc {
    const F4 *geom;       // n_obj
    const F4 *tri;        // 4 * n_tri
    const F4 *mat;        // 2 * n_mat
    const uint32_t *meta; // n_obj
    const uint32_t *link; // n_obj
    const F4 *mat_extra;  // n_mat: roughness (not halved), refractive_index, 0, 0 --
                          // read by the GLSL-semantics kernel only (pt_glsl.hip);
                          // behind the blob the LDS copy takes
    // optional hierarchy over the objects (csrc/bvh_build.hpp); n_nodes == 0
    // when it was not built
    const F4 *bvh_nodes;           // 2 * n_nodes: both child boxes in binary16, rounded
                                   // outward, + the child refs (csrc/bvh_build.hpp)
    const uint32_t *bvh_order;     // object indices, leaf runs
    const F4 *bvh_leaf_geom;       // geom[bvh_order[j]] copied into leaf order: a leaf's
                                   // records lie side by side and are fetched together
                                   // (both arrays are padded by a leaf's worth of slots)
    const uint32_t *bvh_unbounded; // objects without a finite box
    uint32_t n_nodes;
    uint32_t n_unbounded;
    uint32_t bvh_root;             // ref to start from (leaf bit = 0x80000000)
    uint32_t bvh_depth;            // deepest chain of inner nodes
    uint32_t n_obj;
    uint32_t n_tri;
    uint32_t n_mat;
    uint32_t branching; // some material has 0 < transparency < 1
    float sky[3];
    uint32_t last_sphere; // highest index of a sphere, 0xFFFFFFFF: no sphere
    GlslCull glsl;
};

// Bytes of the scene blob = LDS bytes its copy needs (16-byte records first,
// then meta, then link).
PT_LAYOUT_HD size_t scene_lds_bytes(uint32_t n_obj, uint32_t n_tri, uint32_t n_mat) {
    size_t b = (size_t)16 * ((size_t)n_obj + 4u * (size_t)n_tri + 2u * (size_t)n_mat);
    b += (size_t)8 * n_obj; // meta + link
    return (b + 15u) & ~(size_t)15u;
}

// Device-side statistics.  Every wave adds its totals once, at its end -- and
// at the end of a launch thousands of waves do so within microseconds.  One
// shared record made that 1.5 ms per launch (4 x 32400 atomics through one
// cache line at ~90 atomics per microsecond), so the record is sharded:
// kCounterShards of them, one 64-byte line each, picked by workgroup index;
// the host adds the shards up when asked.
struct alignas(64) Counters {
    unsigned long long rays;
    unsigned long long samples;
    unsigned long long wave_iters;
    unsigned long long scans;    // closest-hit scans executed (<= rays)
    // (the four above are one 32-byte record: count_block updates them with
    // one atomic instruction)
    unsigned long long launches; // unused on the device (counted on the host)
};
constexpr uint32_t kCounterShards = 256;

struct LaunchArgs {
    SceneDesc scene;
    uint8_t *frame;      // RGB8 destination, or nullptr in accumulate mode
    float *accum;        // float4 per destination pixel, or nullptr
    Counters *counters;  // may be nullptr
    uint32_t framew, frameh;
    uint32_t num_rows;   // rows in this call's row set
    uint32_t row_begin, tile_rows, tile_stride;
    uint32_t compact;    // 1: destination row = local row, 0: = global row
    uint32_t spp, sample_offset, max_bounces, seed;
    float tan_half_fov;
    uint32_t scene_in_lds;
    uint32_t screened;     // scan form: 0 plain, 1 screened linear, 2 BVH (pt_trace.h)
    uint32_t settle_lanes; // parked lanes that trigger a settle stage (1..64)
    uint32_t share_primary; // pixel-owner kernel: scan the primary ray once per pixel
    uint32_t chunk_batch;   // sample queue: chunks a wave reserves per atomic
    uint32_t stream_walks;  // sample queue + BVH scan: resumable walks (pt_streamkernel)
    uint32_t walk_budget;   // ... inner-node steps per quantum
    uint32_t leaf_lanes;    // ... lanes standing on a leaf before leaves are visited
    uint32_t resolve_lanes; // ... lanes with queued candidates before the exact tests run
    uint32_t shade_lanes;   // ... lanes with a finished walk that trigger a shading-only stage
    uint32_t reshade_lanes; // ... second shading round of a service stage from this many lanes
    uint32_t lanes_per_pixel; // pixel-owner kernel: 1, or 2..64 (power of two) lanes
                              // sharing a pixel's samples (small row sets)
    uint32_t tree;            // the scene has a branching material: kTree instantiations
    // sample-queue mode (pt_samplekernel / pt_sum_kernel)
    float *sample_buf;       // float4 per task: [sample][npix_pad]
    uint32_t *chunk_counter; // next chunk of 64 tasks; zeroed before a launch
    const float *primary_hits; // (distance, object) per local pixel of the row set:
                               // the closest hit of its primary ray, or null
    uint32_t npix_pad;       // pixels of the row set, rounded up to 64
    uint32_t chunks_per_sample; // npix_pad / 64
    uint32_t num_chunks;     // spp * chunks_per_sample
    uint32_t first_pass;     // sum kernel: start from zero instead of accum
    uint32_t accum_local;    // sum kernel: accum indexed by local pixel
};

} // namespace pt
