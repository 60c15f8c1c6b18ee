// ptrt_internal.h — structs shared by the host API (api.cpp) and the kernel launchers (kernels.hip).
#pragma once
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include "../../include/ptrt.h"

namespace ptrt {

constexpr uint32_t kTileShift = 6;                 // 64x64 tiles (docs/SPEC.md §6)
constexpr uint32_t kTile = 1u << kTileShift;
constexpr uint32_t kTilePixels = kTile * kTile;
constexpr uint32_t kBlock = 256;                   // threads per workgroup = 4 wavefronts
constexpr uint32_t kStackLds = 24;                 // traversal-stack entries kept in LDS per lane
constexpr uint32_t kMaxSpheres = 64;

// counters block in device memory (uint32 unless noted)
enum Counter : uint32_t {
    C_EXT0 = 0, C_EXT1 = 1,          // extend-queue sizes, by iteration parity
    C_BUCKET0 = 2,                   // 2 parities x 4 buckets: C_BUCKET0 + parity*4 + bucket
    C_ERROR = 10,                    // sticky device-side error flag
    C_RAYS_LO = 12, C_RAYS_HI = 13,  // u64 total extend-queue entries processed
    C_NODES_LO = 14, C_TRIS_LO = 16, C_SPH_LO = 18, // u64 visit counters (COUNT builds)
    C_COUNT = 32
};
enum Bucket : uint32_t { B_MISS = 0, B_LAMBERT = 1, B_METAL = 2, B_DIELECTRIC = 3, B_COUNT = 4 };

struct DeviceScene {
    const float4 *nodes;   // BVH-N: node i slot c = rows (i*N + c)*2 + {0: lo.xyz|ref, 1: hi.xyz|0}
    const float4 *tris;    // 3 rows per triangle: v0|orig_id, e1|material, e2|0
    const float4 *spheres; // cx,cy,cz,r
    const uint32_t *sph_mat;
    const float4 *mats;    // 3 rows per material (48 B pt_material)
    uint32_t n_nodes, n_tris, n_spheres, n_mats;
    float sky[3];
    uint32_t bvh_width;
    pt_camera cam;
};

struct PathState {           // SoA over slots
    float4 *ray_o;           // o.xyz, -
    float4 *ray_d;           // d.xyz, -
    float2 *hit;             // t, ref bits
    float4 *thr;             // T.rgb, key bits
    uint32_t *sd;            // sample << 8 | depth
    float4 *acc;             // radiance sum rgb, path count
    uint32_t *q_ext[2];      // extend queues (slot ids), by parity
    uint32_t *q_bucket[B_COUNT];
    uint32_t *counters;
    int32_t *stack_ovf;      // traversal stack overflow, [entry][slot]
    uint32_t stack_ovf_entries;
    uint32_t n_slots;
};

struct FrameParams {
    uint32_t width, height, spp, max_depth, rr_start, seed_hashed, sample_offset;
    float ray_eps;
    uint32_t rank, nranks, tiles_x, n_tiles;
};

// kernel launchers (kernels.hip). All enqueue on `s` and return the launch error.
hipError_t launch_reference_sphere(hipStream_t s, uint32_t w, uint32_t h, float4 *out_f, uint32_t *out_rgba8);
hipError_t launch_generate(hipStream_t s, const DeviceScene &sc, const PathState &ps, const FrameParams &fp);
hipError_t launch_extend(hipStream_t s, const DeviceScene &sc, const PathState &ps, uint32_t parity, uint32_t n_bound, bool count);
hipError_t launch_shade(hipStream_t s, const DeviceScene &sc, const PathState &ps, const FrameParams &fp, uint32_t parity, uint32_t n_bound);
hipError_t launch_assemble(hipStream_t s, const float4 *gathered, uint32_t nranks, uint32_t slots_per_rank,
                           uint32_t width, uint32_t height, uint32_t tiles_x, uint32_t n_tiles, float inv_spp,
                           float4 *fb, uint32_t *fb8);

} // namespace ptrt
