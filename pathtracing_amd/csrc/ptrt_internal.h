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
#ifndef PT_SHARD_GROUP_SHIFT
#define PT_SHARD_GROUP_SHIFT 6
#endif
// Consecutive slots that share a shard: 2^this. 64 = one wavefront: consecutive wavefronts (the sample streams of one 8x8 pixel
// block) go to consecutive shards, i.e. to different XCDs, and every shard gets an even cut of every tile. Measured, ms per frame
// with groups of 64 / 256 / 512 / 1024 / 4096 slots: headline 17.91 / 18.01 / 18.29 / 18.18 / 18.70; a rank's 1/8 of it 2.65 / 2.72 /
// 2.71 / 2.62 / 3.16; 512 spp 135.9 / 137.1 / 138.5 / 139.0 / 142.2; glass 256 spp 38.9 / 39.1 / 39.5 / 39.8 / 40.7; soup ±0.5 %.
constexpr uint32_t kShardGroupShift = PT_SHARD_GROUP_SHIFT;
#ifndef PT_STACK_LDS
#define PT_STACK_LDS 12
#endif
#ifndef PT_EXT_BLOCK
#define PT_EXT_BLOCK 64
#endif
constexpr uint32_t kStackLds = PT_STACK_LDS;       // traversal-stack entries kept in LDS per lane
constexpr uint32_t kExtBlock = PT_EXT_BLOCK;       // workgroup size of k_extend (a wave retires on its own when it is 64)
constexpr uint32_t kMaxSpheres = 64;
#ifndef PT_POOL
#define PT_POOL 128
#endif
constexpr uint32_t kPool = PT_POOL;                // queue entries per wavefront of k_extend_pool (64 P)
enum ExtendKernel : int { EXT_SIMPLE = 1, EXT_PACKED = 2, EXT_POOL = 3 }; // pt_stats.reserved[0]

// Queue sharding. A slot belongs to shard (slot >> kShardGroupShift) % kShards for the whole frame, every queue is kShards
// independent regions of `shard_cap` entries with one counter line each, and a workgroup works on exactly one shard
// (kernels.hip block_pos: 1-D grids with the shard as the fastest index).
// Why: a queue push is one returning atomic per wavefront; on ONE address that saturates at ~88 atomics/us
// (MI355X_MICROARCH.md "dequeue"), which made both wavefront kernels atomic-bound (~230 us per launch regardless of
// the scene). 64 counters on 64 separate lines take the same pushes at ~64x the rate.
constexpr uint32_t kShards = 64;
constexpr uint32_t kCounterStride = 16;            // u32 words between counters: one 64-B line each

enum Bucket : uint32_t { B_MISS = 0, B_LAMBERT = 1, B_METAL = 2, B_DIELECTRIC = 3, B_COUNT = 4 };

// counters block (u32 words). ext[iteration % 3][shard], bucket[parity][bucket][shard], rays[shard] (u64), then globals.
constexpr uint32_t kCntExt = 0;
constexpr uint32_t kCntBucket = kCntExt + 3 * kShards * kCounterStride;
constexpr uint32_t kCntRays = kCntBucket + 2 * B_COUNT * kShards * kCounterStride;
constexpr uint32_t kCntGlobals = kCntRays + kShards * kCounterStride;
constexpr uint32_t kCntError = kCntGlobals + 0;
constexpr uint32_t kCntNodes = kCntGlobals + 2, kCntTris = kCntGlobals + 4, kCntSph = kCntGlobals + 6; // u64 each
constexpr uint32_t kCntCompactions = kCntGlobals + 8; // (shard, iteration) pairs that re-packed their queue
constexpr uint32_t kCntWaveNodeIters = kCntGlobals + 10; // u64, PT_FLAG_COUNT_VISITS: iterations of the wave-level node loop of k_extend
// u64 x 3, PT_FLAG_COUNT_VISITS, one-ray-per-lane kernel: lane-slots of the node loop spent (a) waiting at a leaf for the wave's node loop,
// (b) with the ray already finished, (c) without a ray at all (hole, ended stream); the rest of 64 x iterations made a node visit
constexpr uint32_t kCntIdleLeaf = kCntGlobals + 16, kCntIdleDone = kCntGlobals + 18, kCntIdleDead = kCntGlobals + 20;
constexpr uint32_t kCntTotalWords = kCntGlobals + 24;

// An extend queue has LEN entries of which ALIVE hold a slot; the rest are kInvalidSlot holes left by paths that ended
// while the queue was carried over in place (k_shade, compact == 0). Both counts share the shard's 64-B line.
constexpr uint32_t kInvalidSlot = 0xFFFFFFFFu;
inline __host__ __device__ uint32_t cnt_ext_index(uint32_t parity, uint32_t shard) { return kCntExt + (parity * kShards + shard) * kCounterStride; }
inline __host__ __device__ uint32_t cnt_alive_index(uint32_t parity, uint32_t shard) { return cnt_ext_index(parity, shard) + 1u; }
// third word of the line: rays traced by the iteration that FILLED this queue; the next iteration folds it into rays[shard]
// word 4 of the line: alive entries at the start of the launch that FILLED this queue (written by that launch's first thread)
inline __host__ __device__ uint32_t cnt_prev_alive_index(uint32_t parity, uint32_t shard) { return cnt_ext_index(parity, shard) + 4u; }
inline __host__ __device__ uint32_t cnt_traced_index(uint32_t parity, uint32_t shard) { return cnt_ext_index(parity, shard) + 2u; }
inline __host__ __device__ uint32_t cnt_bucket_index(uint32_t parity, uint32_t bucket, uint32_t shard)
{
    return kCntBucket + ((parity * B_COUNT + bucket) * kShards + shard) * kCounterStride;
}
inline __host__ __device__ uint32_t cnt_rays_index(uint32_t shard) { return kCntRays + shard * kCounterStride; }

struct DeviceScene {
    const float4 *nodes;   // BVH-N: node i slot c = rows (i*N + c)*2 + {0: lo.xyz|ref, 1: hi.xyz|0}
    const float4 *tris;    // 4 rows (one 64-B line) per triangle, blob order: v0|orig_id, e1|material, e2|0, then the
                           // shading row normalize(cross(e1,e2))|material
    const float4 *spheres; // cx,cy,cz,r
    const uint2 *sph_mat;  // per sphere: material id, bits of 1.0f / r (the IEEE quotient, made once at commit: the shading step's own division gone)
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
    uint32_t *q_ext[2];      // extend queues (slot ids) [shard][shard_cap], by iteration parity
    uint32_t *q_bucket[B_COUNT]; // shade queues per material kind, [shard][shard_cap]
    uint32_t *counters;
    int32_t *stack_ovf;      // traversal stack overflow, [entry][kShards * shard_cap]
    uint32_t stack_ovf_entries;
    uint32_t n_slots;
    uint32_t shard_cap;      // entries per shard region = slots owned by a shard (multiple of 2^kShardGroupShift)
    uint32_t shard_base, shard_count; // the shards this launch covers: shard_base .. + shard_count (groups of shards run as
                                      // independent wavefront loops on their own streams, see api.cpp)
    // queue policy, decided on the device from the shard's own counters (no host lag):
    float compact_below;     // re-pack a shard's queue when the alive/length ratio it would leave is below this (> 1: always, 0: never)
    uint32_t repack_sticky;  // this frame has few samples per stream: a shard that has re-packed once re-packs in every launch (want_compact)
    float sparse_below;      // fused one-ray-per-lane kernel: a launch that starts with alive < sparse_below * length advances one
                             // vertex only (and re-packs), instead of running `bounces` vertices on mostly idle wavefronts
    uint32_t finish_below;   // fused kernel: once a shard has no more alive paths than this, a launch runs them to their end
    // Queue sizes for the host, without a copy dispatch between two launches: the first thread of shard s in launch `it` stores the line
    // the PREVIOUS launch left behind (queue length, alive entries, rays it traced) to host_ring[((it - 1) % ring_slots) * kShards + s]
    // — host-mapped pinned memory; the host reads it after the event that follows launch `it`. NULL: the host copies the lines itself.
    uint4 *host_ring;
    uint32_t ring_slots;
};

struct FrameParams {
    uint32_t width, height, spp, max_depth, rr_start, seed_hashed, sample_offset;
    float ray_eps;
    uint32_t rank, nranks, tiles_x, n_tiles;
    uint32_t streams;          // K sample streams per pixel (docs/SPEC.md §5); slot <-> (pixel slot, stream): kernels.hip slot_of()
    uint32_t slots_per_stream; // pixel slots of this rank (tiles_per_rank * 4096)
    uint32_t accumulate;       // PT_FLAG_ACCUMULATE: keep the partial sums of the previous frame(s)
    // Division by `streams` and by `tiles_x` (slot -> pixel, once per regenerated path) as multiply-high + shift: a 32-bit division by a
    // run-time value is ~22 instructions, five of them quarter-rate integer multiplies, and the regeneration code runs in practically
    // every bounce of every wave. q = umulhi(n, magic) >> shift is exact for n < 2^31 (magic = ceil(2^(31 + l) / d), shift = l - 1,
    // l = ceil(log2 d)); magic 0 stands for d == 1. div_magic() below makes the pair on the host.
    uint32_t streams_magic, streams_shift, tiles_x_magic, tiles_x_shift;
    uint32_t offset_mod;       // sample_offset % streams
};
inline void div_magic(uint32_t d, uint32_t &magic, uint32_t &shift)
{
    if (d <= 1u) { magic = 0u; shift = 0u; return; }
    uint32_t l = 0; while ((1ull << l) < d) ++l;
    magic = (uint32_t)(((1ull << (31u + l)) + d - 1u) / d); shift = l - 1u;
}
inline __host__ __device__ uint32_t div_by(uint32_t n, uint32_t magic, uint32_t shift)
{
    return magic ? (uint32_t)(((uint64_t)n * magic) >> 32) >> shift : n; // the high half of a 32 x 32 multiply: v_mul_hi_u32 on the device
}

// kernel launchers (kernels.hip). All enqueue on `s` and return the launch error.
// `shard_bound` = upper bound of any shard's queue length for this launch.
hipError_t launch_reference_sphere(hipStream_t s, uint32_t w, uint32_t h, float4 *out_f, uint32_t *out_rgba8);
// mode 1: write every slot's initial path state (k_shade and k_extend_pool read it back); 0: queue and counters only — the frame's first
// launch is a fused one-ray-per-lane or lane-packing kernel, which builds the state in registers; 2: like 0 with the queue dense (only
// slots that hold a path: frames in which whole sample streams are empty)
hipError_t launch_generate(hipStream_t s, const DeviceScene &sc, const PathState &ps, const FrameParams &fp, uint32_t mode);
// `it` = iteration index of the wavefront loop: queues alternate by it & 1, queue counters rotate by it % 3.
// kernel: ExtendKernel. fuse: -1 = extend only (k_shade follows); 0 / 2 = the kernel also shades (Lambert-only / all kinds) and
// queues the next iteration, honouring `compact` like launch_shade. packed_chunk: queue entries per wavefront of EXT_PACKED.
// bounces (fused only): path vertices a lane advances per launch.
hipError_t launch_extend(hipStream_t s, const DeviceScene &sc, const PathState &ps, const FrameParams &fp, uint32_t it, uint32_t shard_bound, bool count,
                         int kernel, uint32_t packed_chunk, int fuse, bool compact, uint32_t bounces);
// mode: 0 = queue order, specular kinds deferred to buckets; 1 = the specular buckets; 2 = queue order, everything shaded in place
// compact: 1 = survivors are appended densely to the next queue (ballot + one returning atomic per wavefront);
//          0 = every lane writes its own position of the next queue (slot or kInvalidSlot): no returning atomics, and the
//              queue keeps its slot order, which is what keeps the slot-indexed path state coalesced (modes 0 and 2 only)
hipError_t launch_shade(hipStream_t s, const DeviceScene &sc, const PathState &ps, const FrameParams &fp, uint32_t it, uint32_t shard_bound, int mode, bool compact);
hipError_t launch_reduce_streams(hipStream_t s, const float4 *acc, float4 *tiles, uint32_t slots_per_stream, uint32_t streams);
hipError_t launch_assemble(hipStream_t s, const float4 *gathered, uint32_t nranks, uint32_t slots_per_rank,
                           uint32_t width, uint32_t height, uint32_t tiles_x, uint32_t n_tiles, float inv_spp,
                           float4 *fb, uint32_t *fb8);

// api.cpp: what comm.cpp needs to know about a context
int context_device(const pt_context *c);
hipStream_t context_stream(const pt_context *c);
void context_set_error(pt_context *c, const char *msg); // c == NULL: the calling thread's last error

// lbvh.hip: Morton sort + Karras hierarchy + refit on the device; returns the binary tree on the host
struct BinaryBvh;
struct DeviceBlob4Q;
hipError_t build_lbvh_device(hipStream_t s, const float *verts9, uint32_t n_tris, BinaryBvh &out);
// the same, packed into the BVH4Q blob + triangle records on the device (only the few-thousand-box top storey visits the host)
hipError_t build_lbvh_blob4q_device(hipStream_t s, const float *verts9, const uint32_t *mats, uint32_t n_tris, DeviceBlob4Q &out);

} // namespace ptrt
