/*
 * ptrt.h — C ABI of libptrt.so, the MI355X (gfx950) wavefront path tracer that stands where the
 * reference's Vulkan compute path stands.
 *
 * The reference (chairclr/PathTracing) has no plugin/FFI interface; its de-facto seam is the private trio
 *   Renderer.CreateResources      RayTracing/Graphics/Renderer.cs:105-196   (W x H RGBA image)
 *   Renderer.CreateComputePipeline RayTracing/Graphics/Renderer.cs:293-403  (load Test.spirv, bind image)
 *   Renderer.ComputeFrame(delta)  RayTracing/Graphics/Renderer.cs:1006-1040 (CmdDispatch + QueueSubmit)
 * plus the fence wait in Renderer.Render (Renderer.cs:970-972). Contract of the seam: "after ComputeFrame
 * and the fence wait, a W x H RGBA image owned by the renderer holds the frame". Each entry point below
 * names the reference code it replaces. The P/Invoke stub a maintainer would add is in INTEGRATION.md.
 *
 * Conventions (mirroring the reference's behaviour):
 *  - every call returns pt_status (0 = ok); nothing throws or aborts across the ABI. The C#/C++/Python host
 *    wrappers turn non-zero into an exception, as every Vulkan failure does in the reference
 *    (e.g. Renderer.cs:1022-1025, 1036-1039).
 *  - a pt_context is single-caller; pt_render is synchronous (returns after the stream is idle), like the
 *    host-side fence wait at Renderer.cs:972.
 *  - host arrays passed in are borrowed for the call only; output buffers are caller-allocated.
 *  - all structs are plain little-endian f32/u32, blittable.
 *  - there is NO CPU backend: without a gfx950 device pt_context_create fails with PT_ERR_NO_DEVICE.
 */
#ifndef PTRT_H
#define PTRT_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* 2: pt_tuning grew to 40 bytes (extend_kernel, readback); pt_comm_*, pt_framebuffer_read_srgb8, PT_FLAG_EXTEND_POOL, pt_bvh_info.stack_need and
 * the BVH2 default for small scenes had arrived under version 1. Hosts compare pt_abi_version() with the header they were built against. */
#define PTRT_ABI_VERSION 2

typedef int32_t pt_status;
enum {
    PT_OK = 0,
    PT_ERR_INVALID_ARGUMENT = 1,
    PT_ERR_NO_DEVICE = 2,      /* no HIP device / not gfx950 */
    PT_ERR_HIP = 3,            /* a HIP runtime call failed; text in pt_last_error */
    PT_ERR_OUT_OF_MEMORY = 4,
    PT_ERR_NOT_COMMITTED = 5,  /* scene used before pt_scene_commit */
    PT_ERR_UNSUPPORTED = 6,
    PT_ERR_INTERNAL = 7
};

typedef struct pt_context pt_context; /* opaque; owns the device, stream, path-state queues, framebuffer */
typedef struct pt_scene pt_scene;     /* opaque; owns host copies, the BVH and the device scene */

/* modes of pt_render */
enum {
    PT_REFERENCE_SPHERE = 0, /* exact Test.hlsl:1-40 semantics (docs/SPEC.md §1) */
    PT_PATH_TRACE = 1        /* wavefront path tracer (docs/SPEC.md §2-§6) */
};
/* material kinds */
enum { PT_LAMBERT = 0, PT_METAL = 1, PT_DIELECTRIC = 2 };
/* pt_render_params.flags */
enum {
    PT_FLAG_PROFILE_KERNELS = 1u, /* bracket every kernel with HIP events on the context's stream; fills pt_stats.*_ms */
    PT_FLAG_COUNT_VISITS = 2u,    /* count BVH node visits / triangle / sphere tests on the device (slower build of extend) */
    PT_FLAG_EXTEND_PACKED = 4u,   /* force the lane-packing extend kernel (a wavefront owns a chunk of the queue and refills idle lanes by ballot) */
    PT_FLAG_EXTEND_SIMPLE = 8u,   /* force the one-ray-per-lane extend kernel. Neither flag: probed per scene (pt_stats.reserved[0]) */
    PT_FLAG_BUCKET_SPECULAR = 32u, /* shade metal / dielectric hits from per-kind bucket queues (wave-uniform BSDF code) instead of
                                     in queue order. Slower on MI355X: re-queued slots scramble the queue order and state access
                                     loses its coalescing; kept for comparison (implies PT_FLAG_SPLIT_KERNELS) */
    PT_FLAG_SPLIT_KERNELS = 64u,  /* run extend and shade as two kernels per iteration (hit records through HBM) instead of shading
                                     inside the extend kernel; same frame, lets PT_FLAG_PROFILE_KERNELS time the two separately */
    PT_FLAG_EXTEND_POOL = 128u,   /* force the pooled extend kernel (a wavefront owns 128 queue entries, refills idle lanes during
                                     traversal and shades the whole pool at full width) */
    PT_FLAG_ACCUMULATE = 16u     /* progressive rendering: keep the sums of the previous call(s) (same size/rank/streams/seed,
                                     sample_offset = samples so far) and show the mean over all samples — the converging analogue
                                     of the reference's render-every-frame loop (App.cs:39-42) */
};
/* pt_scene_commit options */
enum {
    PT_BVH_WIDTH_DEFAULT = 0, /* PT_BVH_WIDTH_4Q; PT_BVH_WIDTH_2 for scenes of at most 192 triangles (pt_bvh_info.width tells) */
    PT_BVH_WIDTH_2 = 2,       /* binary, 64-B nodes, f32 child boxes */
    PT_BVH_WIDTH_4 = 4,       /* 4-wide, 128-B nodes, f32 child boxes */
    PT_BVH_WIDTH_4Q = 68,     /* 4-wide, 64-B nodes, child boxes quantised to 8 bits on a per-node power-of-two grid */
    PT_BVH_WIDTH_8Q = 72,     /* 8-wide, one 128-B cache line per node (96 B used), the same quantisation: a node visit costs the
                                 memory system one line either way, and there are fewer visits */
    PT_BVH_WIDTH_8O = 73,     /* the 128-B node of PT_BVH_WIDTH_8Q with the children placed in slots by the octant of the node they lie in:
                                 visited in the order slot ^ (sign bits of the ray direction), no distance sort (docs/SPEC.md §4.1) */
    PT_BVH_BUILD_LBVH = 0x100 /* OR-ed onto a layout: build the hierarchy on the GPU (Morton sort + Karras + refit) instead of the
                                 host's binned-SAH builder. Faster to build, slower to trace; the picture is identical either way. */
};

typedef struct {
    int32_t device_ordinal; /* hipSetDevice argument; the reference picks its device in GraphicsDevice.cs:38-43 */
    void *stream;           /* an existing hipStream_t to run on, or NULL to create one */
    uint32_t flags;         /* reserved, 0 */
    uint32_t reserved;
} pt_device_desc;

typedef struct { uint32_t kind; float albedo[3]; float emission[3]; float roughness; float ior; uint32_t pad[3]; } pt_material; /* 48 B */

/* pinhole camera, docs/SPEC.md §3. The reference's literal camera (Test.hlsl:6-10) is
 * origin (0,0,1), forward (0,0,-1), right (1,0,0), up (0,1,0), scale 2/1080, cx = cy = 1, jitter 0
 * up to the +0.5 pixel centre that Test.hlsl omits. */
typedef struct { float origin[3], forward[3], right[3], up[3]; float scale, cx, cy; uint32_t jitter; } pt_camera; /* 64 B */

typedef struct {
    uint32_t width, height;  /* frame size; the reference hard-codes 1920x1080 (App.cs:27, Renderer.cs:1020) */
    uint32_t spp;            /* samples per pixel (reference: 1 ray per pixel, Test.hlsl) */
    uint32_t max_depth;      /* max path segments */
    uint32_t rr_start;       /* Russian roulette from this depth on */
    uint32_t seed;
    uint32_t sample_offset;  /* first sample index (progressive accumulation across calls) */
    uint32_t mode;           /* PT_REFERENCE_SPHERE | PT_PATH_TRACE */
    float ray_eps;           /* origin offset along the normal for continuation rays */
    uint32_t rank, nranks;   /* image-space partition: this process renders tiles t with t % nranks == rank */
    uint32_t tile_size;      /* 0 = 64 */
    uint32_t flags;          /* PT_FLAG_* */
    uint32_t streams;        /* sample streams per pixel kept in flight at once (docs/SPEC.md §5): sample s goes to stream
                                s mod streams, each stream has its own partial sum, the pixel is their fixed-order sum.
                                0 = 1; at most 64. More streams = more rays per wavefront iteration, same picture definition. */
    uint32_t pad[2];
} pt_render_params; /* 64 B */

typedef struct {
    uint64_t rays;          /* ray-scene intersection queries = path segments (the benchmark's unit) */
    uint64_t paths;
    uint64_t node_visits, tri_tests, sphere_tests; /* only with PT_FLAG_COUNT_VISITS, else 0 */
    uint32_t iterations;    /* wavefront iterations = launches of the extend kernel; the default (fused one-ray-per-lane) kernel advances
                               every path by up to 3/4 max_depth - 2 clamped to [4, 12] vertices per iteration, the lane-packing one by up
                               to 64 (pt_tuning.bounces overrides) */
    uint32_t extend_launches;
    double gpu_ms;          /* hipEvent start->stop around all kernels of the frame */
    double extend_ms;       /* sum of extend-kernel durations (PT_FLAG_PROFILE_KERNELS); includes shading when fused */
    double shade_ms;        /* sum of shade-kernel durations  (PT_FLAG_PROFILE_KERNELS); ~0 when fused */
    double other_ms;        /* generate / resolve kernels */
    uint64_t reserved[4];   /* diagnostics: [0] extend kernel in use (1 one ray per lane, 2 lane-packing, 3 pooled, 0 unprobed),
                               [1] iterations that re-packed their queues, [2] path states loaded+stored by the loop
                               (sum over iterations of the paths alive at its start), [3] with PT_FLAG_COUNT_VISITS: wave-level
                               iterations of k_extend's node loop in bits 0-39 (node_visits / (64 * that) = lane utilisation of the
                               loop), those after the wave's first leaf phase from bit 40 up */
} pt_stats;

typedef struct {
    uint32_t width;          /* the layout id the scene was committed with: PT_BVH_WIDTH_2, _4, _4Q, _8Q or _8O */
    uint32_t n_nodes;
    uint32_t n_tris;
    uint32_t max_depth;
    uint64_t node_bytes;     /* n_nodes * 64 (layouts 2 and 4Q) or n_nodes * 128 (layouts 4, 8Q and 8O) */
    uint64_t tri_bytes;      /* n_tris * 48: the blob's triangle records as pt_scene_bvh_read copies them out (docs/SPEC.md §4.1). On the device
                                every record is padded to one 64-byte line (+ a shading row): n_tris * 64 bytes of HBM */
    double build_ms;
    float sah_cost;
    uint32_t stack_need;     /* worst-case traversal-stack depth of this tree (entries); the kernels keep 12 per lane in LDS and
                                size a global overflow area from this */
} pt_bvh_info;

/* Scheduling knobs of a context (none changes a pixel; tests/test_gpu_parity.py holds every setting to the same frame). Read the
 * current values with pt_context_get_tuning, change what you want, write them back. The library reads no environment variables for
 * these (developer aids aside: the stderr diagnostics PTRT_TRACE / PTRT_TIMING, and PTRT_NODE_ORDER / PTRT_UNIFIED / PTRT_COLLAPSE, which give
 * the same tree another memory order, or collapse it to wide nodes by the old rule, for the builder experiments of DESIGN.md). */
typedef struct {
    uint32_t bounces;       /* path vertices a lane advances per launch of the fused extend kernels, 1..64; 0 (default) = 3/4 max_depth - 2
                               clamped to [4, 12] for the one-ray-per-lane kernel, 64 for the lane-packing one */
    uint32_t loops;         /* independent shard-group wavefront loops per frame, each on its own stream: 1, 2 or 4; 0 (default) = two
                               (frames with PT_FLAG_PROFILE_KERNELS / PT_FLAG_COUNT_VISITS always run one: their kernels are timed alone) */
    uint32_t finish_below;  /* a shard with at most this many live paths runs them to their end in one launch (default 4096; 0 = never) */
    uint32_t packed_chunk;  /* queue entries per wavefront of the lane-packing kernel, >= 64; 0 (default) = 256 with >= 4 streams, else 128 */
    float compact_below;    /* a shard re-packs its queue in a launch that would otherwise leave alive < this * length behind (default 0.9;
                               > 1 every launch; 0 never) */
    float sparse_below;     /* one-ray-per-lane kernel: a launch that starts with alive < this * length advances one vertex only (default 0 = off) */
    uint32_t sticky_samples; /* frames with at most this many samples per stream (spp / streams): a shard that has re-packed once re-packs
                               in every launch, and with at most 2 samples per stream every launch re-packs (default 32; 0 = off) */
    uint32_t lag;           /* wavefront iterations the host runs ahead of the queue sizes it reads back: 2..5; 0 (default) = 3 for frames
                               of at most 2 samples per stream, else 4 (one less each with readback = 1) */
    uint32_t extend_kernel; /* 0 (default) = the extend kernel is probed per scene (timed, remembered in the scene; pt_stats.reserved[0]
                               tells which one ran); 1 / 2 / 3 = every frame uses the one-ray-per-lane / lane-packing / pooled kernel, so
                               that runs are reproducible in speed as well as in pixels. A PT_FLAG_EXTEND_* of a frame overrides it */
    uint32_t readback;      /* how a launch's queue sizes reach the host: 0 (default) = the next launch's first thread per shard stores them
                               to host-mapped pinned memory; 1 = a 2-4 KB device-to-host copy behind every launch (a copy dispatch
                               that sits in order between two launches: rounds 1-2) */
} pt_tuning; /* 40 B */

/* ---- context: replaces GraphicsDevice.Init (GraphicsDevice.cs:38-43) + Renderer.CreateResources (Renderer.cs:105-196) */
pt_status pt_context_create(const pt_device_desc *desc, pt_context **out);
void pt_context_destroy(pt_context *ctx);                 /* Renderer.Dispose, Renderer.cs:1192-1216 */
const char *pt_last_error(const pt_context *ctx);         /* ctx may be NULL: last error of the calling thread */
pt_status pt_context_get_tuning(const pt_context *ctx, pt_tuning *out);
pt_status pt_context_set_tuning(pt_context *ctx, const pt_tuning *tuning);
uint32_t pt_abi_version(void);

/* ---- scene: the reference has only shader literals (Test.hlsl:6,8,12,13); this is the API that replaces them */
pt_status pt_scene_create(pt_context *ctx, pt_scene **out);
void pt_scene_destroy(pt_scene *scene);
pt_status pt_scene_set_triangles(pt_scene *s, const float *verts9, const uint32_t *material_ids, uint64_t count);
pt_status pt_scene_set_spheres(pt_scene *s, const float *cxyzr, const uint32_t *material_ids, uint64_t count);
pt_status pt_scene_set_materials(pt_scene *s, const pt_material *mats, uint64_t count);
pt_status pt_scene_set_camera(pt_scene *s, const pt_camera *cam);
pt_status pt_scene_set_sky(pt_scene *s, const float rgb[3]);
/* builds the BVH on the host (binned SAH) and uploads everything; replaces CreateComputePipeline's one-time
 * descriptor/pipeline set-up (Renderer.cs:293-403). bvh_width: PT_BVH_WIDTH_* */
pt_status pt_scene_commit(pt_scene *s, uint32_t bvh_width);
pt_status pt_scene_bvh_info(const pt_scene *s, pt_bvh_info *out);
/* copies out the acceleration-structure blob (docs/SPEC.md §4.1) so that a checker can traverse the same bytes */
pt_status pt_scene_bvh_read(const pt_scene *s, void *nodes, uint64_t node_bytes, void *tris48, uint64_t tri_bytes);

/* ---- frame: replaces Renderer.ComputeFrame (Renderer.cs:1006-1040) + the fence wait (Renderer.cs:970-972) */
pt_status pt_render(pt_context *ctx, const pt_scene *scene, const pt_render_params *params, pt_stats *stats);

/* ---- results: the reference never reads its image back (it is sampled by the display pass,
 *      Renderer.cs:1042-1121); these replace that consumer. float4 linear radiance, row-major. */
pt_status pt_framebuffer_read(pt_context *ctx, float *rgba, uint64_t n_floats);
pt_status pt_framebuffer_read_rgba8(pt_context *ctx, uint8_t *rgba8, uint64_t n_bytes); /* R8G8B8A8Unorm, Renderer.cs:124 */
pt_status pt_framebuffer_device_ptr(pt_context *ctx, void **dptr, uint64_t *n_floats);   /* row-major float4 on the device */
/* What the reference's window shows: its display pass samples the R8G8B8A8Unorm image with a nearest sampler (Renderer.cs:184-185)
 * and writes it to a B8G8R8A8Srgb swapchain (SwapChain.cs:157-158), i.e. every 8-bit linear value q becomes
 * round(255 * srgb_oetf(q / 255)). No tone mapping: radiance above 1 clips, as it does in the UNORM image. RGBA order. */
pt_status pt_framebuffer_read_srgb8(pt_context *ctx, uint8_t *rgba8, uint64_t n_bytes);

/* ---- image-space partition (docs/SPEC.md §6). After pt_render with nranks > 1 the rank's tiles sit
 * tile-major in a device staging buffer; the host gathers them (RCCL via torch.distributed or
 * ncclGather) and rank 0 calls pt_assemble_tiles on the gathered buffer. */
typedef struct {
    uint32_t tile_size, tiles_x, tiles_y, n_tiles;
    uint32_t tiles_mine;      /* tiles owned by params.rank */
    uint32_t tiles_per_rank;  /* max over ranks = stride of one rank's block in the gathered buffer, in tiles */
    uint64_t floats_per_tile; /* tile_size^2 * 4 */
} pt_tile_layout;
pt_status pt_tile_layout_query(const pt_render_params *params, pt_tile_layout *out);
/* tiles_per_rank*floats_per_tile floats. Ownership: the buffer is rewritten by the context's next pt_render, on the context's
 * own streams — a caller that hands it to an asynchronous collective on another stream must wait for that collective before it
 * renders again (bench.py does; pt_comm orders the two on one stream). */
pt_status pt_tiles_device_ptr(pt_context *ctx, void **dptr, uint64_t *n_floats);
/* gathered = nranks blocks of tiles_per_rank*floats_per_tile floats (device pointer on ctx's device) */
pt_status pt_assemble_tiles(pt_context *ctx, const pt_render_params *params, const void *gathered_dptr, uint64_t n_floats);

/* ---- several GPUs of one node behind one call (docs/SPEC.md §6). The reference is single-device (GraphicsDevice.cs:176-183) and its
 * host is one process (Program.cs:3-7), so a C# caller cannot use a process-per-GPU launcher: a pt_comm holds one pt_context per
 * rank inside the calling process. Ranks on distinct devices exchange their tiles with ONE ncclGather per frame over an RCCL
 * communicator made by ncclCommInitAll (librccl.so.1 is loaded on first use; libptrt does not link it). Ranks that share one
 * context ("virtual ranks": rehearsal of the partition on a single GPU) are rendered one after the other and staged by device
 * copies. A rank's tile buffer (pt_tiles_device_ptr) belongs to the library between pt_render and the end of the exchange. */
typedef struct pt_comm pt_comm;
enum {
    PT_COMM_FORCE_RCCL = 1u,    /* pt_comm_create flags: use the RCCL path even for a single rank (plumbing check on one GPU) */
    PT_COMM_COPY_EXCHANGE = 2u  /* no RCCL: every rank's tile block reaches the root by a device copy on the rank's own stream (a peer copy over
                                   xGMI between devices). Also lifts the one-context-per-device rule: N contexts on ONE device render
                                   concurrently, one host thread each — the multi-context code path as far as a single GPU can take it */
};
/* ctxs[i] renders rank i of n_ranks; the assembled frame lands in ctxs[root] (pt_framebuffer_read*). Either every rank has its own
 * context on its own device, or all ranks share one context, or (PT_COMM_COPY_EXCHANGE) every rank has its own context anywhere. */
pt_status pt_comm_create(pt_context *const *ctxs, uint32_t n_ranks, uint32_t root, uint32_t flags, pt_comm **out);
void pt_comm_destroy(pt_comm *comm); /* before or after its contexts: every pt_comm call returns with nothing in flight */
/* One frame: rank i renders its tiles of `params` (rank / nranks are filled in) on scenes[i] — the same scene committed on every
 * context — concurrently (one host thread per context), then the tiles are gathered to the root and assembled there.
 * stats: n_ranks entries or NULL. Synchronous, like pt_render. */
pt_status pt_comm_render(pt_comm *comm, const pt_scene *const *scenes, const pt_render_params *params, pt_stats *stats);
/* The two halves of the exchange for callers that drive pt_render themselves (rank = the rank just rendered on its context):
 * pt_comm_stage_tiles copies that rank's tile block to where the exchange wants it; pt_comm_assemble runs the exchange (if any
 * rank lives on another device) and un-tiles the gathered blocks into the root's framebuffer. */
pt_status pt_comm_stage_tiles(pt_comm *comm, uint32_t rank);
pt_status pt_comm_assemble(pt_comm *comm, const pt_render_params *params);

/* ---- deterministic synthetic scenes (C1..C5); host only, no device needed.
 * Two-call pattern: pass NULL arrays to get counts, then buffers of those sizes. */
enum {
    PT_SCENE_CORNELL = 0,        /* C1/C2: 5 walls (10 tris) + emissive ceiling quad (2 tris) + 4 Lambert spheres */
    PT_SCENE_CORNELL_GLASS = 1,  /* C4: same box, spheres = dielectric, metal (rough), 2 Lambert */
    PT_SCENE_TRIANGLE_SOUP = 2,  /* C3: `detail` random triangles in [-1,1]^3, sky emitter */
    PT_SCENE_CORNELL_TESS = 3    /* C5: Cornell with walls tessellated to ~`detail` triangles */
};
typedef struct { uint64_t n_tris, n_spheres, n_mats; } pt_scene_counts;
pt_status pt_scenegen(uint32_t kind, uint32_t detail, uint32_t seed, uint32_t width, uint32_t height,
                      pt_scene_counts *counts, float *verts9, uint32_t *tri_mat, float *spheres, uint32_t *sph_mat,
                      pt_material *mats, pt_camera *cam, float sky[3]);

#ifdef __cplusplus
}
#endif
#endif
