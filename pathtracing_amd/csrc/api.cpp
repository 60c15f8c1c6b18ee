// api.cpp — C ABI of libptrt.so (include/ptrt.h): context, scene, wavefront frame loop.
// Stands where Renderer.CreateResources / CreateComputePipeline / ComputeFrame + the compute-fence wait stand in
// the reference (RayTracing/Graphics/Renderer.cs:105-196, 293-403, 1006-1040, 970-972).
// HIP only: there is no CPU fallback anywhere in this library.
#include "ptrt_internal.h"
#include "bvh_build.h"
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

using namespace ptrt;

namespace {

thread_local std::string g_err;

constexpr uint32_t kLag = 5; // most wavefront iterations kept in flight before the host looks at a queue size (ring sizes; pt_tuning.lag)
constexpr uint32_t kRingWords = kShards * kCounterStride; // one iteration's readback: (up to) the kShards extend-queue sizes
constexpr uint32_t kMaxGroups = 4;  // independent wavefront loops (shard groups) per frame, each on its own stream
constexpr size_t kFinalOffset = (size_t)kMaxGroups * kLag * kRingWords; // where the frame-end copy of all counters lands in h_counts

uint32_t host_pcg(uint32_t x)
{
    uint32_t s = x * 747796405u + 2891336453u;
    uint32_t w = ((s >> ((s >> 28) + 4u)) ^ s) * 277803737u;
    return (w >> 22) ^ w;
}

thread_local uint64_t g_device_allocs = 0; // counts DevBuf allocations: a frame that had to allocate is a cold frame (its rate is not a measurement)

template <typename T> struct DevBuf {
    T *p = nullptr; size_t n = 0;
    hipError_t ensure(size_t count)
    {
        if (p && count <= n && (n <= (1u << 20) || count >= n / 4)) return hipSuccess; // big enough, and not more than 4x too big
        ++g_device_allocs;
        if (p) { (void)hipFree(p); p = nullptr; n = 0; }
        hipError_t e = hipMalloc((void **)&p, (count ? count : 1) * sizeof(T));
        if (e == hipSuccess) n = count ? count : 1;
        return e;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; n = 0; }
    void adopt(void *q, size_t count) { release(); p = (T *)q; n = count; } // take over a hipMalloc-ed array
};

} // namespace

struct pt_context {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    std::string err;
    // path state
    DevBuf<float4> ray_o, ray_d, thr, acc, tiles, fb;
    // what the partial sums in `acc` currently hold (PT_FLAG_ACCUMULATE continues them): frame geometry and samples so far
    uint32_t acc_w = 0, acc_h = 0, acc_rank = 0, acc_nranks = 0, acc_streams = 0, acc_seed = 0;
    uint64_t acc_spp = 0;
    DevBuf<float2> hit;
    DevBuf<uint32_t> sd, q_ext0, q_ext1, q_b[B_COUNT], counters, fb8;
    DevBuf<int32_t> stack_ovf;
    // What k_generate would write at the start of every frame of a fused pipeline, kept from the first frame of its kind: the first
    // extend queue (every shard's slots in slot order, holes for off-image pixels and sample-less streams) and the counter block
    // that goes with it. A frame then starts with one 2.4 KB device copy instead of a kernel over every slot; the first extend
    // launch reads q_init in place of q_ext[0] and zeroes the radiance sums of the slots it starts (kernels.hip, it == 0).
    DevBuf<uint32_t> q_init, cnt_init;
    struct InitKey { uint32_t w, h, rank, nranks, streams, first_spp, offset, n_slots, shard_cap; const void *q, *acc;
                     bool operator==(const InitKey &o) const { return std::memcmp(this, &o, sizeof *this) == 0; } } init_key{};
    bool init_valid = false;
    uint32_t init_bound = 0; // longest shard queue of the template: the first launch's grid bound
    uint32_t *h_counts = nullptr; // pinned: kLag readbacks of the per-shard queue sizes (pt_tuning.readback = 1) + one copy of all counters
    uint4 *h_ring = nullptr, *d_ring = nullptr; // mapped pinned memory the extend kernels report their queue sizes to, kLag x kShards lines
                                                // (host address, device address); PathState::host_ring
    uint32_t readback = 0;                      // pt_tuning.readback: 0 = the kernels store the sizes to h_ring, 1 = one 2-4 KB copy per launch
    uint32_t extend_kernel = 0;                 // pt_tuning.extend_kernel: 0 = probed per scene, else the ExtendKernel every scene uses
    hipEvent_t ev_lag[kMaxGroups][kLag] = {};
    hipStream_t group_stream[kMaxGroups] = {}; // group 0 runs on `stream` when there is one group only
    hipEvent_t ev_fork = nullptr, ev_join[kMaxGroups] = {};
    uint32_t groups = 0;                        // pt_tuning.loops (1, 2, 4) overrides; 0 = two loops, whose launch tails overlap. Measured
                                                // (tools/exp_loops.py, ms per frame with 1 / 2 / 4 loops): 1M-tri Cornell 1080p/64spp 18.42 /
                                                // 18.08 / 18.77, a rank's 1/8 of it 4.11 / 3.81 / -, soup 76.2 / 73.3 / 72.3, glass 256 spp
                                                // 37.8 / 37.0 / 36.6, 4K/1024 spp 1062 / 1051 / 1046. Frames that time single kernels
                                                // (PT_FLAG_PROFILE_KERNELS, visit counting, the extend-kernel probe) run one loop, so that
                                                // a timed launch has the GPU to itself.
    uint32_t bounces = 0;                       // pt_tuning.bounces (1..64): path vertices per launch of the fused kernel (state in registers);
                                                // 0 = 3/4 max_depth - 2 clamped to [4, 12]: depth 8 -> 4, depth 16 -> 10 (ms per frame with 2 / 3 /
                                                // 4 / 6 / 8 / 12 vertices: 1M-triangle Cornell, depth 8: 18.57 / 17.80 / 17.52 / 17.50 / 17.68 / 17.87;
                                                // Cornell+glass+metal, depth 16: 47.5 / 41.0 / 38.3 / 35.3 / 34.3 / 33.5)
    double compact_below = 0.9;                // pt_tuning.compact_below: a shard re-packs its queue in a launch that would leave alive/length
                                                // below this (>1 = every launch, 0 = never); else carried in place (want_compact, kernels.hip).
    uint32_t lag = 0;                           // pt_tuning.lag (2..4; 0 = by frame length, see render_frame)
    uint32_t sticky_samples = 32;               // pt_tuning.sticky_samples. Measured, 1M-tri Cornell 1080p, ms per frame by spp (8 streams),
                                                // start-of-launch ratio (round 1) / predicted ratio / sticky / every launch:
                                                //   8: 4.13/4.20/4.07/3.01  32: 10.95/10.93/9.70/9.61  64: 19.28/18.82/18.45/18.36
                                                //   128: 38.15/36.12/36.04/36.06  256: 73.06/71.95/71.99/72.15  512: 142.8/141.5/143.0/143.7
                                                //   1024: 284.5/283.7/289.9/292.5; 4K/1024: 1067.7/1064.1/1097.0/1108 (tools/exp_compact.py)
    double sparse_below = 0.0;                  // pt_tuning.sparse_below (0 = off, the default: measured ±0): see PathState::sparse_below
    uint32_t finish_below = 4096;             // pt_tuning.finish_below: a shard with no more alive paths than this runs them to
                                                // their end in one launch of the fused kernel (0 = never)
    uint32_t packed_chunk = 0;                  // pt_tuning.packed_chunk: queue entries per wavefront of the lane-packing kernel (0 = by stream count)
    hipEvent_t ev_start = nullptr, ev_stop = nullptr;
    hipEvent_t ev_probe[4] = {}; // brackets of the two probe iterations that pick the extend kernel
    std::vector<hipEvent_t> ev_pool;
    uint32_t fb_w = 0, fb_h = 0;
    uint32_t n_slots = 0; // slots of the last path-traced frame (acc layout)
    bool fb_valid = false;
};

struct pt_scene {
    pt_context *ctx = nullptr;
    std::vector<float> verts; std::vector<uint32_t> tri_mat;
    std::vector<float> spheres; std::vector<uint32_t> sph_mat;
    std::vector<pt_material> mats;
    pt_camera cam{};
    float sky[3] = { 0.f, 0.f, 0.f };
    bool have_cam = false, committed = false;
    mutable BvhBlob bvh;                 // (mutable: pt_scene_bvh_read fills the host copy of a device-packed blob on first use)
    uint32_t layout = 0;                 // PT_BVH_WIDTH_* the scene was committed with
    mutable std::vector<uint8_t> packed_nodes; // layouts PT_BVH_WIDTH_4Q / _8Q: the 64- / 128-byte nodes that are uploaded / read back
    bool device_packed = false;          // the blob was packed on the device (lbvh.hip build_lbvh_blob4q_device): the host copies below
    mutable bool host_mirror = true;     // (packed_nodes, bvh.tris) are fetched from the device the first time pt_scene_bvh_read wants them
    bool quantised() const { return layout == PT_BVH_WIDTH_4Q || layout == PT_BVH_WIDTH_8Q || layout == PT_BVH_WIDTH_8O; }
    const void *node_data() const { return quantised() ? (const void *)packed_nodes.data() : (const void *)bvh.slots.data(); }
    uint64_t node_bytes() const { return device_packed ? (uint64_t)bvh.n_nodes * 64u : quantised() ? packed_nodes.size() : bvh.slots.size() * sizeof(BvhSlot); }
    uint64_t n_blob_tris() const { return device_packed ? tri_mat.size() : bvh.tris.size(); }
    DevBuf<float4> d_nodes, d_tris, d_spheres, d_mats;
    bool has_specular = false;
    mutable uint32_t ext_choice = 0;     // cache, not scene content: the extend kernel an earlier frame's probe picked (0 = none yet, ExtendKernel otherwise)
    mutable double rate_simple = 0.0, rate_packed = 0.0; // rays per ms of whole frames run on one kernel (frames too short to probe inside)
    mutable uint32_t probe_misses = 0;   // warm frames that neither decided nor fed the decision (too few rays to time): after three the scene
                                         // settles on the one-ray-per-lane kernel instead of staying in probe mode (one loop, no finish mode) for good
    DevBuf<uint2> d_sph_mat;
    DeviceScene ds{};
};

namespace {

pt_status fail(pt_context *ctx, pt_status code, const char *fmt, ...)
{
    char buf[512];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    g_err = buf;
    if (ctx) ctx->err = buf;
    return code;
}
#define HIP_TRY(ctx, expr)                                                                          \
    do { hipError_t _e = (expr);                                                                    \
         if (_e != hipSuccess)                                                                      \
             return fail(ctx, _e == hipErrorOutOfMemory ? PT_ERR_OUT_OF_MEMORY : PT_ERR_HIP,        \
                         "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)

bool finite3(const float *p) { return std::isfinite(p[0]) && std::isfinite(p[1]) && std::isfinite(p[2]); }

pt_status layout_of(const pt_render_params *p, pt_tile_layout *o)
{
    if (!p || !o) return PT_ERR_INVALID_ARGUMENT;
    if (p->width == 0 || p->height == 0 || p->width > 32768 || p->height > 32768) return PT_ERR_INVALID_ARGUMENT;
    if (p->tile_size != 0 && p->tile_size != kTile) return PT_ERR_UNSUPPORTED;
    const uint32_t nr = p->nranks ? p->nranks : 1u;
    if (p->rank >= nr) return PT_ERR_INVALID_ARGUMENT;
    o->tile_size = kTile;
    o->tiles_x = (p->width + kTile - 1) / kTile;
    o->tiles_y = (p->height + kTile - 1) / kTile;
    o->n_tiles = o->tiles_x * o->tiles_y;
    o->tiles_mine = (o->n_tiles > p->rank) ? (o->n_tiles - p->rank + nr - 1) / nr : 0u;
    o->tiles_per_rank = (o->n_tiles + nr - 1) / nr;
    o->floats_per_tile = (uint64_t)kTilePixels * 4u;
    return PT_OK;
}

hipEvent_t pool_event(pt_context *c, size_t i)
{
    while (c->ev_pool.size() <= i) {
        hipEvent_t e = nullptr;
        if (hipEventCreate(&e) != hipSuccess) return nullptr;
        c->ev_pool.push_back(e);
    }
    return c->ev_pool[i];
}

} // namespace

namespace ptrt {
int context_device(const pt_context *c) { return c->device; }
hipStream_t context_stream(const pt_context *c) { return c->stream; }
void context_set_error(pt_context *c, const char *msg) { g_err = msg; if (c) c->err = msg; }
} // namespace ptrt

extern "C" {

uint32_t pt_abi_version(void) { return PTRT_ABI_VERSION; }

const char *pt_last_error(const pt_context *ctx) { return ctx ? ctx->err.c_str() : g_err.c_str(); }

pt_status pt_context_create(const pt_device_desc *desc, pt_context **out)
{
    if (!out) return fail(nullptr, PT_ERR_INVALID_ARGUMENT, "pt_context_create: out is NULL");
    *out = nullptr;
    const int dev = desc ? desc->device_ordinal : 0;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return fail(nullptr, PT_ERR_NO_DEVICE, "no HIP device visible (%s); libptrt has no CPU backend",
                    e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    if (dev < 0 || dev >= ndev) return fail(nullptr, PT_ERR_INVALID_ARGUMENT, "device ordinal %d out of range [0,%d)", dev, ndev);
    hipDeviceProp_t prop;
    HIP_TRY(nullptr, hipGetDeviceProperties(&prop, dev));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(nullptr, PT_ERR_NO_DEVICE, "device %d is %s; libptrt ships gfx950 (MI355X) code objects only", dev, prop.gcnArchName);
    HIP_TRY(nullptr, hipSetDevice(dev));
    pt_context *c = new (std::nothrow) pt_context();
    if (!c) return fail(nullptr, PT_ERR_OUT_OF_MEMORY, "host allocation failed");
    c->device = dev;
    if (desc && desc->stream) { c->stream = (hipStream_t)desc->stream; c->own_stream = false; }
    else {
        e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
        if (e != hipSuccess) { delete c; return fail(nullptr, PT_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(e)); }
        c->own_stream = true;
    }
    bool ok = hipHostMalloc((void **)&c->h_counts, sizeof(uint32_t) * (kFinalOffset + kCntTotalWords), hipHostMallocDefault) == hipSuccess;
    if (ok) { // host-mapped ring for the kernels' own size reports; a platform without mapped pinned memory falls back to a copy per launch
        if (hipHostMalloc((void **)&c->h_ring, sizeof(uint4) * kLag * kShards, hipHostMallocMapped) == hipSuccess &&
            hipHostGetDevicePointer((void **)&c->d_ring, c->h_ring, 0) == hipSuccess && c->d_ring)
            std::memset(c->h_ring, 0, sizeof(uint4) * kLag * kShards);
        else {
            if (c->h_ring) (void)hipHostFree(c->h_ring);
            c->h_ring = c->d_ring = nullptr; c->readback = 1u;
            (void)hipGetLastError();
        }
    }
    for (uint32_t g = 0; ok && g < kMaxGroups; ++g) {
        ok = hipStreamCreateWithFlags(&c->group_stream[g], hipStreamNonBlocking) == hipSuccess &&
             hipEventCreateWithFlags(&c->ev_join[g], hipEventDisableTiming) == hipSuccess;
    }
    ok = ok && hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) == hipSuccess;
    ok = ok && hipEventCreate(&c->ev_start) == hipSuccess && hipEventCreate(&c->ev_stop) == hipSuccess;
    for (uint32_t g = 0; g < kMaxGroups; ++g)
        for (uint32_t i = 0; ok && i < kLag; ++i) ok = hipEventCreateWithFlags(&c->ev_lag[g][i], hipEventDisableTiming) == hipSuccess;
    for (uint32_t i = 0; ok && i < 4; ++i) ok = hipEventCreate(&c->ev_probe[i]) == hipSuccess;
    ok = ok && c->counters.ensure(kCntTotalWords) == hipSuccess;
    if (!ok) { pt_context_destroy(c); return fail(nullptr, PT_ERR_HIP, "context resource creation failed"); }
    *out = c;
    return PT_OK;
}

pt_status pt_context_get_tuning(const pt_context *c, pt_tuning *o)
{
    if (!c || !o) return fail(nullptr, PT_ERR_INVALID_ARGUMENT, "pt_context_get_tuning: NULL argument");
    std::memset(o, 0, sizeof *o);
    o->bounces = c->bounces; o->loops = c->groups; o->finish_below = c->finish_below; o->packed_chunk = c->packed_chunk;
    o->compact_below = (float)c->compact_below; o->sparse_below = (float)c->sparse_below; o->sticky_samples = c->sticky_samples; o->lag = c->lag;
    o->extend_kernel = c->extend_kernel; o->readback = c->readback;
    return PT_OK;
}

pt_status pt_context_set_tuning(pt_context *c, const pt_tuning *t)
{
    if (!c || !t) return fail(c, PT_ERR_INVALID_ARGUMENT, "pt_context_set_tuning: NULL argument");
    if (t->bounces > 64) return fail(c, PT_ERR_INVALID_ARGUMENT, "tuning: bounces must be 0 (default) or 1..64");
    if (t->loops != 0 && t->loops != 1 && t->loops != 2 && t->loops != 4) return fail(c, PT_ERR_INVALID_ARGUMENT, "tuning: loops must be 0 (default), 1, 2 or 4");
    if (t->packed_chunk != 0 && (t->packed_chunk < 64 || t->packed_chunk > (1u << 20))) return fail(c, PT_ERR_INVALID_ARGUMENT, "tuning: packed_chunk must be 0 (default) or 64..2^20");
    if (t->lag != 0 && (t->lag < 2 || t->lag > kLag)) return fail(c, PT_ERR_INVALID_ARGUMENT, "tuning: lag must be 0 (default) or 2..%u", kLag);
    if (!(t->compact_below >= 0.f && t->compact_below <= 2.f) || !(t->sparse_below >= 0.f && t->sparse_below <= 1.f))
        return fail(c, PT_ERR_INVALID_ARGUMENT, "tuning: compact_below must be in [0,2], sparse_below in [0,1]");
    c->bounces = t->bounces; c->groups = t->loops; c->finish_below = t->finish_below; c->packed_chunk = t->packed_chunk;
    if (t->extend_kernel > (uint32_t)EXT_POOL) return fail(c, PT_ERR_INVALID_ARGUMENT, "tuning: extend_kernel must be 0 (probed), 1 (one ray per lane), 2 (lane-packing) or 3 (pooled)");
    if (t->readback > 1) return fail(c, PT_ERR_INVALID_ARGUMENT, "tuning: readback must be 0 (mapped store) or 1 (copy per launch)");
    if (t->readback == 0 && !c->d_ring) return fail(c, PT_ERR_UNSUPPORTED, "tuning: readback 0 needs host-mapped pinned memory, which this platform did not provide");
    c->compact_below = t->compact_below; c->sparse_below = t->sparse_below; c->sticky_samples = t->sticky_samples; c->lag = t->lag;
    c->extend_kernel = t->extend_kernel; c->readback = t->readback;
    return PT_OK;
}

void pt_context_destroy(pt_context *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    for (auto &gs : c->group_stream) if (gs) (void)hipStreamSynchronize(gs); // nothing may still run on the buffers released below
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    c->ray_o.release(); c->ray_d.release(); c->thr.release(); c->acc.release(); c->tiles.release(); c->fb.release(); c->hit.release();
    c->sd.release(); c->q_ext0.release(); c->q_ext1.release(); c->counters.release(); c->fb8.release(); c->stack_ovf.release();
    c->q_init.release(); c->cnt_init.release();
    for (auto &q : c->q_b) q.release();
    if (c->h_counts) (void)hipHostFree(c->h_counts);
    if (c->h_ring) (void)hipHostFree(c->h_ring);
    for (auto &row : c->ev_lag) for (auto &e : row) if (e) (void)hipEventDestroy(e);
    for (auto &e : c->ev_join) if (e) (void)hipEventDestroy(e);
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    for (auto &gs : c->group_stream) if (gs) { (void)hipStreamSynchronize(gs); (void)hipStreamDestroy(gs); }
    for (auto &e : c->ev_probe) if (e) (void)hipEventDestroy(e);
    if (c->ev_start) (void)hipEventDestroy(c->ev_start);
    if (c->ev_stop) (void)hipEventDestroy(c->ev_stop);
    for (auto e : c->ev_pool) (void)hipEventDestroy(e);
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

// ------------------------------------------------------------------------------------------------ scene

pt_status pt_scene_create(pt_context *ctx, pt_scene **out)
{
    // ctx == NULL makes a detached (host-only) scene: commit builds the BVH blob for pt_scene_bvh_read/info,
    // nothing is uploaded and pt_render rejects it. Used to check the builder where no device exists.
    if (!out) return fail(ctx, PT_ERR_INVALID_ARGUMENT, "pt_scene_create: NULL argument");
    pt_scene *s = new (std::nothrow) pt_scene();
    if (!s) return fail(ctx, PT_ERR_OUT_OF_MEMORY, "host allocation failed");
    s->ctx = ctx;
    *out = s;
    return PT_OK;
}

void pt_scene_destroy(pt_scene *s)
{
    if (!s) return;
    if (s->ctx) { (void)hipSetDevice(s->ctx->device); (void)hipStreamSynchronize(s->ctx->stream); }
    s->d_nodes.release(); s->d_tris.release();s->d_spheres.release(); s->d_mats.release(); s->d_sph_mat.release();
    delete s;
}

pt_status pt_scene_set_triangles(pt_scene *s, const float *verts9, const uint32_t *material_ids, uint64_t count)
{
    if (!s) return fail(nullptr, PT_ERR_INVALID_ARGUMENT, "scene is NULL");
    if (count && !verts9) return fail(s->ctx, PT_ERR_INVALID_ARGUMENT, "verts9 is NULL");
    if (count >= (1ull << 28)) return fail(s->ctx, PT_ERR_UNSUPPORTED, "more than 2^28 triangles");
    for (uint64_t i = 0; i < count * 9; ++i)
        if (!std::isfinite(verts9[i])) return fail(s->ctx, PT_ERR_INVALID_ARGUMENT, "non-finite vertex coordinate at float %llu", (unsigned long long)i);
    s->verts.assign(verts9, verts9 + count * 9);
    if (material_ids) s->tri_mat.assign(material_ids, material_ids + count); else s->tri_mat.assign(count, 0u);
    s->committed = false;
    return PT_OK;
}

pt_status pt_scene_set_spheres(pt_scene *s, const float *cxyzr, const uint32_t *material_ids, uint64_t count)
{
    if (!s) return fail(nullptr, PT_ERR_INVALID_ARGUMENT, "scene is NULL");
    if (count && !cxyzr) return fail(s->ctx, PT_ERR_INVALID_ARGUMENT, "cxyzr is NULL");
    if (count > kMaxSpheres) return fail(s->ctx, PT_ERR_UNSUPPORTED, "more than %u spheres (they are a flat list)", kMaxSpheres);
    for (uint64_t i = 0; i < count; ++i)
        if (!finite3(cxyzr + i * 4) || !(cxyzr[i * 4 + 3] > 0.f) || !std::isfinite(cxyzr[i * 4 + 3]))
            return fail(s->ctx, PT_ERR_INVALID_ARGUMENT, "sphere %llu: non-finite centre or radius <= 0", (unsigned long long)i);
    s->spheres.assign(cxyzr, cxyzr + count * 4);
    if (material_ids) s->sph_mat.assign(material_ids, material_ids + count); else s->sph_mat.assign(count, 0u);
    s->committed = false;
    return PT_OK;
}

pt_status pt_scene_set_materials(pt_scene *s, const pt_material *mats, uint64_t count)
{
    if (!s) return fail(nullptr, PT_ERR_INVALID_ARGUMENT, "scene is NULL");
    if (count && !mats) return fail(s->ctx, PT_ERR_INVALID_ARGUMENT, "mats is NULL");
    for (uint64_t i = 0; i < count; ++i) {
        if (mats[i].kind > PT_DIELECTRIC) return fail(s->ctx, PT_ERR_INVALID_ARGUMENT, "material %llu: unknown kind %u", (unsigned long long)i, mats[i].kind);
        if (!finite3(mats[i].albedo) || !finite3(mats[i].emission) || !std::isfinite(mats[i].roughness) || !std::isfinite(mats[i].ior))
            return fail(s->ctx, PT_ERR_INVALID_ARGUMENT, "material %llu: non-finite field", (unsigned long long)i);
        if (mats[i].roughness < 0.f || mats[i].roughness > 1.f) return fail(s->ctx, PT_ERR_INVALID_ARGUMENT, "material %llu: roughness outside [0,1]", (unsigned long long)i);
        if (mats[i].kind == PT_DIELECTRIC && !(mats[i].ior > 0.f)) return fail(s->ctx, PT_ERR_INVALID_ARGUMENT, "material %llu: ior <= 0", (unsigned long long)i);
    }
    s->mats.assign(mats, mats + count);
    s->committed = false;
    return PT_OK;
}

pt_status pt_scene_set_camera(pt_scene *s, const pt_camera *cam)
{
    if (!s || !cam) return fail(s ? s->ctx : nullptr, PT_ERR_INVALID_ARGUMENT, "NULL argument");
    if (!finite3(cam->origin) || !finite3(cam->forward) || !finite3(cam->right) || !finite3(cam->up) ||
        !std::isfinite(cam->scale) || !std::isfinite(cam->cx) || !std::isfinite(cam->cy))
        return fail(s->ctx, PT_ERR_INVALID_ARGUMENT, "camera has a non-finite field");
    s->cam = *cam; s->have_cam = true;
    if (s->committed) s->ds.cam = *cam; // camera changes do not need a re-commit
    return PT_OK;
}

pt_status pt_scene_set_sky(pt_scene *s, const float rgb[3])
{
    if (!s || !rgb) return fail(s ? s->ctx : nullptr, PT_ERR_INVALID_ARGUMENT, "NULL argument");
    if (!finite3(rgb)) return fail(s->ctx, PT_ERR_INVALID_ARGUMENT, "sky is not finite");
    for (int k = 0; k < 3; ++k) { s->sky[k] = rgb[k]; s->ds.sky[k] = rgb[k]; }
    return PT_OK;
}

pt_status pt_scene_commit(pt_scene *s, uint32_t bvh_width)
{
    if (!s) return fail(nullptr, PT_ERR_INVALID_ARGUMENT, "scene is NULL");
    pt_context *c = s->ctx;
    const bool lbvh = (bvh_width & PT_BVH_BUILD_LBVH) != 0; // hierarchy built on the GPU instead of the host SAH builder
    bvh_width &= ~(uint32_t)PT_BVH_BUILD_LBVH;
    if (lbvh && !c) return fail(c, PT_ERR_UNSUPPORTED, "PT_BVH_BUILD_LBVH needs a device context (detached scenes use the host builder)");
    // default layout: BVH4Q; scenes of up to ~200 triangles get BVH2 with float boxes — their whole tree is a handful of L1-resident
    // lines, memory does not count and the 2-wide visit is the cheapest in ALU. ms per 1080p / 64 spp frame, BVH8Q | BVH4Q | BVH4 | BVH2
    // (tools/exp_layouts.py): Cornell (12 triangles) 8.30 | 9.07 | 8.60 | 8.32, Cornell+glass+metal 9.82 | 11.34 | 9.82 | 9.18, walls of
    // 42 triangles 17.7 | 12.2 | 11.4 | 11.4, of 162: 18.7 | 14.2 | 13.4 | 12.7, of 252: - | 13.0 | 13.7 | 13.4, of 1002: 20.2 | 14.2 | 16.5 |
    // 15.0; soups of 100 / 400: - | 2.11 / 2.68 | 2.24 / 2.82 | 2.21 / 2.87. (Round 1 gave everything up to 256 triangles BVH8Q, on the
    // strength of the 12-triangle box alone, where it is one node.)
    if (bvh_width == PT_BVH_WIDTH_DEFAULT) bvh_width = s->tri_mat.size() <= 192 ? PT_BVH_WIDTH_2 : PT_BVH_WIDTH_4Q;
    if (bvh_width != PT_BVH_WIDTH_2 && bvh_width != PT_BVH_WIDTH_4 && bvh_width != PT_BVH_WIDTH_4Q && bvh_width != PT_BVH_WIDTH_8Q && bvh_width != PT_BVH_WIDTH_8O)
        return fail(c, PT_ERR_INVALID_ARGUMENT, "bvh_width must be one of PT_BVH_WIDTH_* (0, 2, 4, 68, 72, 73)");
    if (!s->have_cam) return fail(c, PT_ERR_INVALID_ARGUMENT, "no camera set");
    const uint32_t nt = (uint32_t)s->tri_mat.size(), ns = (uint32_t)s->sph_mat.size(), nm = (uint32_t)s->mats.size();
    if ((nt || ns) && nm == 0) return fail(c, PT_ERR_INVALID_ARGUMENT, "primitives but no materials");
    for (uint32_t i = 0; i < nt; ++i) if (s->tri_mat[i] >= nm) return fail(c, PT_ERR_INVALID_ARGUMENT, "triangle %u: material id %u >= %u", i, s->tri_mat[i], nm);
    for (uint32_t i = 0; i < ns; ++i) if (s->sph_mat[i] >= nm) return fail(c, PT_ERR_INVALID_ARGUMENT, "sphere %u: material id %u >= %u", i, s->sph_mat[i], nm);

    const bool oct = bvh_width == PT_BVH_WIDTH_8O;
    const uint32_t fan = bvh_width == PT_BVH_WIDTH_2 ? 2u : (bvh_width == PT_BVH_WIDTH_8Q || oct) ? 8u : 4u;
    static const bool timing = getenv("PTRT_TIMING") != nullptr; // developer aid: where a commit's time goes, on stderr
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms_since = [&](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double, std::milli>(now() - t).count(); };
    auto t_phase = now();
    auto lap = [&](const char *what) { if (timing) fprintf(stderr, "ptrt commit: %-28s %8.2f ms\n", what, ms_since(t_phase)); t_phase = now(); };
    s->device_packed = false; s->host_mirror = true;
    uint32_t unified_units = 0;
    if (lbvh && nt >= 2 && bvh_width == PT_BVH_WIDTH_4Q) {
        // the default layout is also packed on the device: nodes and triangle records are born in device memory
        HIP_TRY(c, hipSetDevice(c->device));
        DeviceBlob4Q db;
        HIP_TRY(c, build_lbvh_blob4q_device(c->stream, s->verts.data(), s->tri_mat.data(), nt, db));
        s->d_nodes.adopt(db.nodes, (size_t)db.n_nodes * 4);
        s->d_tris.adopt(db.tris, (size_t)nt * 4);
        s->bvh = BvhBlob{};
        s->bvh.width = 4; s->bvh.n_nodes = db.n_nodes; s->bvh.max_depth = db.max_depth; s->bvh.stack_need = db.stack_need;
        s->bvh.sah_cost = db.sah_cost; s->bvh.build_ms = db.device_ms;
        s->device_packed = true; s->host_mirror = false;
    } else if (lbvh && nt >= 2) {
        HIP_TRY(c, hipSetDevice(c->device));
        BinaryBvh bt;
        HIP_TRY(c, build_lbvh_device(c->stream, s->verts.data(), nt, bt));
        build_bvh_from_binary(bt, s->verts.data(), s->tri_mat.data(), nt, fan, s->bvh, oct);
    } else build_bvh(s->verts.data(), s->tri_mat.data(), nt, fan, s->bvh, oct);
    lap("hierarchy + blob");
    if (s->bvh.max_depth > 90) return fail(c, PT_ERR_INTERNAL, "BVH depth %u exceeds the supported 90", s->bvh.max_depth);
    s->layout = bvh_width;
    s->packed_nodes.clear();
    if (bvh_width == PT_BVH_WIDTH_4Q && !s->device_packed) quantize_bvh4(s->bvh, s->packed_nodes);
    if (bvh_width == PT_BVH_WIDTH_8Q || oct) quantize_bvh8(s->bvh, s->packed_nodes);
    lap("quantise");
    if (!c) { s->committed = true; return PT_OK; } // detached scene: host-side blob only

    HIP_TRY(c, hipSetDevice(c->device));
    static_assert(sizeof(BvhSlot) == 32 && sizeof(BvhTri) == 48 && sizeof(pt_material) == 48, "blob layout");
    if (!s->device_packed) {
    HIP_TRY(c, s->d_nodes.ensure((size_t)(s->node_bytes() / 16)));
    HIP_TRY(c, s->d_tris.ensure(s->bvh.tris.size() * 4));
    {   // Device triangle record = one 64-byte line: the blob's three rows (docs/SPEC.md §4.1) + a shading row. A 48-byte
        // record straddles two cache lines 3 times out of 4 when k_extend fetches it; a padded one never does, and the
        // row that pads it is the one k_shade wants next. Shading row: ng = normalize(cross(e1,e2)) in exactly the op order
        // of docs/SPEC.md §0 (fma, IEEE sqrt and divide), so the bits equal what the kernel would compute from e1,e2.
        std::vector<float> rec(s->bvh.tris.size() * 16);
        for (size_t i = 0; i < s->bvh.tris.size(); ++i) {
            const BvhTri &t = s->bvh.tris[i];
            std::memcpy(&rec[i * 16], &t, sizeof(BvhTri));
            const float *a = t.e1, *b = t.e2;
            const float cx = std::fmaf(a[1], b[2], -(a[2] * b[1])), cy = std::fmaf(a[2], b[0], -(a[0] * b[2])), cz = std::fmaf(a[0], b[1], -(a[1] * b[0]));
            const float inv = 1.0f / std::sqrt(std::fmaf(cz, cz, std::fmaf(cy, cy, cx * cx)));
            rec[i * 16 + 12] = cx * inv; rec[i * 16 + 13] = cy * inv; rec[i * 16 + 14] = cz * inv;
            std::memcpy(&rec[i * 16 + 15], &t.mat, 4);
        }
        lap("triangle records");
        if (bvh_width == PT_BVH_WIDTH_4Q && getenv("PTRT_UNIFIED") && !rec.empty()) {
            // EXPERIMENT (layout rows; tools/exp_order.py): nodes and triangle records in ONE array of 64-byte units, every
            // node followed by the triangles of its leaf children, so that a bottom-level node and the first of its triangles share a
            // 128-byte line. Refs count units; the kernels are unchanged (nodes and tris are the same base pointer). The device
            // structure is then a renaming of the blob pt_scene_bvh_read hands out — same tree, same pictures.
            const uint32_t nn = s->bvh.n_nodes, ntb = (uint32_t)s->bvh.tris.size();
            std::vector<uint32_t> node_unit(nn), tri_unit(ntb);
            uint32_t u = 0;
            for (uint32_t i = 0; i < nn; ++i) {
                node_unit[i] = u++;
                for (int k = 0; k < 4; ++k) {
                    int32_t r; std::memcpy(&r, &s->packed_nodes[(size_t)i * 64 + 16 + 4 * k], 4);
                    if (r >= 0) continue;
                    const uint32_t enc = (uint32_t)~r, first = enc >> 3, cnt = (enc & 7u) + 1u;
                    for (uint32_t j = 0; j < cnt; ++j) tri_unit[first + j] = u++;
                }
            }
            std::vector<uint8_t> uni((size_t)u * 64);
            for (uint32_t i = 0; i < nn; ++i) {
                uint8_t *nd = &uni[(size_t)node_unit[i] * 64];
                std::memcpy(nd, &s->packed_nodes[(size_t)i * 64], 64);
                for (int k = 0; k < 4; ++k) {
                    int32_t r; std::memcpy(&r, nd + 16 + 4 * k, 4);
                    if (r == 0x7fffffff) continue;
                    if (r >= 0) r = (int32_t)node_unit[(uint32_t)r];
                    else { const uint32_t enc = (uint32_t)~r; r = (int32_t)~((tri_unit[enc >> 3] << 3) | (enc & 7u)); }
                    std::memcpy(nd + 16 + 4 * k, &r, 4);
                }
            }
            for (uint32_t j = 0; j < ntb; ++j) std::memcpy(&uni[(size_t)tri_unit[j] * 64], &rec[(size_t)j * 16], 64);
            HIP_TRY(c, s->d_nodes.ensure((size_t)u * 4));
            HIP_TRY(c, hipMemcpy(s->d_nodes.p, uni.data(), uni.size(), hipMemcpyHostToDevice));
            unified_units = u;
        } else {
        if (!rec.empty()) HIP_TRY(c, hipMemcpy(s->d_tris.p, rec.data(), rec.size() * sizeof(float), hipMemcpyHostToDevice));
        lap("upload triangles");
        }
    }
    if (s->node_bytes() && !unified_units) HIP_TRY(c, hipMemcpy(s->d_nodes.p, s->node_data(), s->node_bytes(), hipMemcpyHostToDevice));
    }
    HIP_TRY(c, s->d_spheres.ensure((ns + 3u) & ~3u)); // the kernels read the list four spheres (one 64-byte scalar load) at a time
    HIP_TRY(c, s->d_sph_mat.ensure(ns));
    HIP_TRY(c, s->d_mats.ensure((size_t)nm * 3));
    if (ns) {
        HIP_TRY(c, hipMemcpy(s->d_spheres.p, s->spheres.data(), (size_t)ns * 16, hipMemcpyHostToDevice));
        std::vector<uint2> mi(ns);
        for (uint32_t i = 0; i < ns; ++i) { // material id + 1.0f / r (IEEE single division: the value docs/SPEC.md §5 has the shading step compute)
            const float inv_r = 1.0f / s->spheres[(size_t)i * 4 + 3];
            mi[i].x = s->sph_mat[i]; std::memcpy(&mi[i].y, &inv_r, 4);
        }
        HIP_TRY(c, hipMemcpy(s->d_sph_mat.p, mi.data(), (size_t)ns * sizeof(uint2), hipMemcpyHostToDevice));
    }
    if (nm) HIP_TRY(c, hipMemcpy(s->d_mats.p, s->mats.data(), (size_t)nm * sizeof(pt_material), hipMemcpyHostToDevice));
    lap("upload nodes + rest");

    DeviceScene &d = s->ds;
    d.nodes = s->d_nodes.p; d.tris = s->d_tris.p; d.spheres = s->d_spheres.p; d.sph_mat = s->d_sph_mat.p; d.mats = s->d_mats.p;
    d.n_nodes = s->bvh.n_nodes; d.n_tris = nt; d.n_spheres = ns; d.n_mats = nm;
    if (unified_units) { d.tris = d.nodes; d.n_tris = unified_units; } // (experiment) triangle refs count units of the one array; sphere refs follow them
    for (int k = 0; k < 3; ++k) d.sky[k] = s->sky[k];
    d.bvh_width = bvh_width;
    d.cam = s->cam;
    s->ext_choice = 0; s->rate_simple = s->rate_packed = 0.0; s->probe_misses = 0;
    s->has_specular = false;
    for (const pt_material &m : s->mats) if (m.kind != PT_LAMBERT) s->has_specular = true;
    s->committed = true;
    return PT_OK;
}

pt_status pt_scene_bvh_info(const pt_scene *s, pt_bvh_info *o)
{
    if (!s || !o) return fail(s ? s->ctx : nullptr, PT_ERR_INVALID_ARGUMENT, "NULL argument");
    if (!s->committed) return fail(s->ctx, PT_ERR_NOT_COMMITTED, "scene not committed");
    std::memset(o, 0, sizeof *o);
    o->width = s->layout; o->n_nodes = s->bvh.n_nodes; o->n_tris = (uint32_t)s->n_blob_tris();
    o->max_depth = s->bvh.max_depth;
    o->node_bytes = s->node_bytes();
    o->tri_bytes = s->n_blob_tris() * sizeof(BvhTri);
    o->build_ms = s->bvh.build_ms; o->sah_cost = s->bvh.sah_cost;
    o->stack_need = s->bvh.stack_need;
    return PT_OK;
}

pt_status pt_scene_bvh_read(const pt_scene *s, void *nodes, uint64_t node_bytes, void *tris48, uint64_t tri_bytes)
{
    if (!s) return fail(nullptr, PT_ERR_INVALID_ARGUMENT, "scene is NULL");
    if (!s->committed) return fail(s->ctx, PT_ERR_NOT_COMMITTED, "scene not committed");
    if (!s->host_mirror) { // packed on the device: fetch the blob now (nodes as they are, triangles = rows 0-2 of the 64-byte records)
        const pt_scene *m = s; // fills the (mutable) host copies the scene owns; device data and results are untouched
        pt_context *c = s->ctx;
        HIP_TRY(c, hipSetDevice(c->device));
        m->packed_nodes.resize((size_t)s->bvh.n_nodes * 64);
        std::vector<float> rec(s->tri_mat.size() * 16);
        if (!m->packed_nodes.empty()) HIP_TRY(c, hipMemcpy(m->packed_nodes.data(), s->d_nodes.p, m->packed_nodes.size(), hipMemcpyDeviceToHost));
        if (!rec.empty()) HIP_TRY(c, hipMemcpy(rec.data(), s->d_tris.p, rec.size() * sizeof(float), hipMemcpyDeviceToHost));
        m->bvh.tris.resize(s->tri_mat.size());
        for (size_t i = 0; i < m->bvh.tris.size(); ++i) std::memcpy(&m->bvh.tris[i], &rec[i * 16], sizeof(BvhTri));
        s->host_mirror = true;
    }
    const uint64_t nb = s->node_bytes(), tb = (uint64_t)s->bvh.tris.size() * sizeof(BvhTri);
    if (node_bytes < nb || tri_bytes < tb || (nb && !nodes) || (tb && !tris48)) return fail(s->ctx, PT_ERR_INVALID_ARGUMENT, "buffers too small: need %llu + %llu bytes", (unsigned long long)nb, (unsigned long long)tb);
    if (nb) std::memcpy(nodes, s->node_data(), nb);
    if (tb) std::memcpy(tris48, s->bvh.tris.data(), tb);
    return PT_OK;
}

// ------------------------------------------------------------------------------------------------ frame

pt_status pt_tile_layout_query(const pt_render_params *p, pt_tile_layout *o)
{
    pt_status st = layout_of(p, o);
    if (st != PT_OK) return fail(nullptr, st, "invalid render params for tile layout");
    return PT_OK;
}

static pt_status ensure_frame(pt_context *c, uint32_t w, uint32_t h)
{
    const size_t n = (size_t)w * h;
    HIP_TRY(c, c->fb.ensure(n));
    HIP_TRY(c, c->fb8.ensure(n));
    c->fb_w = w; c->fb_h = h;
    return PT_OK;
}

static pt_status render_frame(pt_context *c, const pt_scene *s, const pt_render_params *p, pt_stats *stats);

pt_status pt_render(pt_context *c, const pt_scene *s, const pt_render_params *p, pt_stats *stats)
{
    const pt_status st = render_frame(c, s, p, stats);
    if (st != PT_OK && c) { // an error exit may leave kernels in flight on the loop streams: nothing of this frame survives the call
        (void)hipSetDevice(c->device);
        for (auto &gs : c->group_stream) if (gs) (void)hipStreamSynchronize(gs);
        if (c->stream) (void)hipStreamSynchronize(c->stream);
        c->acc_spp = 0; c->fb_valid = false;
    }
    return st;
}

static pt_status render_frame(pt_context *c, const pt_scene *s, const pt_render_params *p, pt_stats *stats)
{
    if (!c || !p) return fail(c, PT_ERR_INVALID_ARGUMENT, "pt_render: NULL argument");
    pt_tile_layout lay;
    pt_status st = layout_of(p, &lay);
    if (st != PT_OK) return fail(c, st, "pt_render: bad width/height/rank/nranks/tile_size");
    HIP_TRY(c, hipSetDevice(c->device));
    pt_stats out; std::memset(&out, 0, sizeof out);
    c->fb_valid = false;
    const bool profile = (p->flags & PT_FLAG_PROFILE_KERNELS) != 0, count = (p->flags & PT_FLAG_COUNT_VISITS) != 0;
    // rays per wavefront of the lane-packing kernel: 256 once several sample streams keep the queues long, else 128 (measured)
    const uint32_t packed_chunk = c->packed_chunk >= 64u ? c->packed_chunk : ((p->streams >= 4u) ? 256u : 128u);
    const bool bucket_specular = (p->flags & PT_FLAG_BUCKET_SPECULAR) != 0;
    const bool split_kernels = bucket_specular || (p->flags & PT_FLAG_SPLIT_KERNELS) != 0;
    const uint32_t forced_choice = (p->flags & PT_FLAG_EXTEND_POOL) ? (uint32_t)EXT_POOL : (p->flags & PT_FLAG_EXTEND_PACKED) ? (uint32_t)EXT_PACKED
                                   : (p->flags & PT_FLAG_EXTEND_SIMPLE) ? (uint32_t)EXT_SIMPLE : c->extend_kernel; // a flag beats pt_tuning.extend_kernel

    if (p->mode == PT_REFERENCE_SPHERE) {
        // Renderer.ComputeFrame: one dispatch, then the host blocks on the fence (Renderer.cs:1020,1036,972)
        if ((st = ensure_frame(c, p->width, p->height)) != PT_OK) return st;
        HIP_TRY(c, hipEventRecord(c->ev_start, c->stream));
        HIP_TRY(c, launch_reference_sphere(c->stream, p->width, p->height, c->fb.p, c->fb8.p));
        HIP_TRY(c, hipEventRecord(c->ev_stop, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        float ms = 0.f; HIP_TRY(c, hipEventElapsedTime(&ms, c->ev_start, c->ev_stop));
        out.gpu_ms = ms; out.other_ms = ms;
        out.rays = out.paths = (uint64_t)p->width * p->height;
        out.iterations = 1;
        c->fb_valid = true;
        if (stats) *stats = out;
        return PT_OK;
    }
    if (p->mode != PT_PATH_TRACE) return fail(c, PT_ERR_INVALID_ARGUMENT, "unknown mode %u", p->mode);
    if (!s) return fail(c, PT_ERR_INVALID_ARGUMENT, "pt_render: scene is NULL");
    if (s->ctx != c) return fail(c, PT_ERR_INVALID_ARGUMENT, "scene belongs to another context");
    if (!s->committed) return fail(c, PT_ERR_NOT_COMMITTED, "scene not committed");
    if (p->spp == 0 || p->spp >= (1u << 24)) return fail(c, PT_ERR_INVALID_ARGUMENT, "spp must be in [1, 2^24)");
    if (p->max_depth == 0 || p->max_depth > 255) return fail(c, PT_ERR_INVALID_ARGUMENT, "max_depth must be in [1,255]");
    if (!std::isfinite(p->ray_eps) || p->ray_eps < 0.f) return fail(c, PT_ERR_INVALID_ARGUMENT, "ray_eps must be finite and >= 0");

    if (p->streams > 64) return fail(c, PT_ERR_INVALID_ARGUMENT, "streams must be in [0,64]");
    const uint32_t nranks = p->nranks ? p->nranks : 1u;
    const uint32_t streams = p->streams ? p->streams : 1u;
    const uint32_t pixel_slots = lay.tiles_per_rank * kTilePixels;              // one slot per owned pixel ...
    const uint64_t slots64 = (uint64_t)pixel_slots * streams;                   // ... per sample stream
    if (slots64 >= (1ull << 28)) return fail(c, PT_ERR_UNSUPPORTED, "frame too large: %llu slots (pixels of this rank x streams), limit 2^28", (unsigned long long)slots64); // kernels.hip at(): 32-bit byte offsets
    const uint32_t n_slots = (uint32_t)slots64;

    const uint64_t allocs_before = g_device_allocs;
    HIP_TRY(c, c->ray_o.ensure(n_slots)); HIP_TRY(c, c->ray_d.ensure(n_slots)); HIP_TRY(c, c->thr.ensure(n_slots));
    HIP_TRY(c, c->acc.ensure(n_slots)); HIP_TRY(c, c->tiles.ensure(pixel_slots)); HIP_TRY(c, c->hit.ensure(n_slots)); HIP_TRY(c, c->sd.ensure(n_slots));
    // every queue = kShards regions of shard_cap entries; shard s owns the 2^kShardGroupShift-slot groups g with g % kShards == s
    const uint32_t groups = (n_slots + (1u << kShardGroupShift) - 1u) >> kShardGroupShift, shard_cap = ((groups + kShards - 1) / kShards) << kShardGroupShift;
    const size_t q_entries = (size_t)kShards * shard_cap;
    HIP_TRY(c, c->q_ext0.ensure(q_entries)); HIP_TRY(c, c->q_ext1.ensure(q_entries));
    for (auto &q : c->q_b) HIP_TRY(c, q.ensure(q_entries));
    const uint32_t ovf = s->bvh.stack_need > kStackLds ? s->bvh.stack_need - kStackLds : 0u;
    if (ovf) HIP_TRY(c, c->stack_ovf.ensure((size_t)ovf * q_entries));
    if (nranks == 1) { if ((st = ensure_frame(c, p->width, p->height)) != PT_OK) return st; }

    PathState ps{};
    ps.ray_o = c->ray_o.p; ps.ray_d = c->ray_d.p; ps.hit = c->hit.p; ps.thr = c->thr.p; ps.sd = c->sd.p; ps.acc = c->acc.p;
    ps.q_ext[0] = c->q_ext0.p; ps.q_ext[1] = c->q_ext1.p;
    for (uint32_t b = 0; b < B_COUNT; ++b) ps.q_bucket[b] = c->q_b[b].p;
    ps.counters = c->counters.p; ps.stack_ovf = c->stack_ovf.p; ps.stack_ovf_entries = ovf; ps.n_slots = n_slots; ps.shard_cap = shard_cap;
    ps.shard_base = 0; ps.shard_count = kShards;
    ps.compact_below = (float)c->compact_below; ps.finish_below = c->finish_below; ps.sparse_below = (float)c->sparse_below;
    const uint32_t samples_per_stream = (p->spp + (p->streams ? p->streams : 1u) - 1u) / (p->streams ? p->streams : 1u);
    ps.repack_sticky = (samples_per_stream <= c->sticky_samples && c->compact_below > 0.0) ? 1u : 0u;
    const bool mapped = c->readback == 0u; // queue sizes reach the host by the kernels' own stores (fold_traced) instead of a copy per launch
    ps.host_ring = mapped ? c->d_ring : nullptr; ps.ring_slots = kLag;
    const bool repack_always = ps.repack_sticky && samples_per_stream <= 2u; // nothing (or next to nothing) regenerates: every launch leaves holes

    FrameParams fp{};
    fp.width = p->width; fp.height = p->height; fp.spp = p->spp; fp.max_depth = p->max_depth; fp.rr_start = p->rr_start;
    fp.seed_hashed = host_pcg(p->seed); fp.sample_offset = p->sample_offset; fp.ray_eps = p->ray_eps;
    fp.rank = p->rank; fp.nranks = nranks; fp.tiles_x = lay.tiles_x; fp.n_tiles = lay.n_tiles;
    fp.streams = streams; fp.slots_per_stream = pixel_slots;
    div_magic(streams, fp.streams_magic, fp.streams_shift); div_magic(lay.tiles_x, fp.tiles_x_magic, fp.tiles_x_shift);
    fp.offset_mod = p->sample_offset % streams;
    // progressive accumulation (the reference re-renders every frame, App.cs:39-42; this is its converging analogue):
    // keep the stream partials of the previous call(s) and divide by the total number of samples at the end
    const bool accumulate = (p->flags & PT_FLAG_ACCUMULATE) != 0;
    if (accumulate) {
        if (c->acc_spp == 0 || c->acc_w != p->width || c->acc_h != p->height || c->acc_rank != p->rank || c->acc_nranks != nranks ||
            c->acc_streams != streams || c->acc_seed != p->seed)
            return fail(c, PT_ERR_INVALID_ARGUMENT, "PT_FLAG_ACCUMULATE needs a previous frame with the same size, rank, nranks, streams and seed");
        if (p->sample_offset != c->acc_spp) return fail(c, PT_ERR_INVALID_ARGUMENT, "PT_FLAG_ACCUMULATE: sample_offset must be %llu (samples so far)", (unsigned long long)c->acc_spp);
    }
    fp.accumulate = accumulate ? 1u : 0u;
    const uint64_t total_spp = (accumulate ? c->acc_spp : 0u) + p->spp;
    c->acc_spp = 0; // invalid until this frame completes

    const DeviceScene &sc = s->ds;
    hipStream_t q = c->stream;
    // the fused one-ray-per-lane and lane-packing kernels build a slot's initial state in registers in their first launch; k_shade
    // (split pipelines) and the pooled kernel read it from memory
    const bool full_state = split_kernels || forced_choice == (uint32_t)EXT_POOL;
    if (full_state) {
        HIP_TRY(c, hipMemsetAsync(c->counters.p, 0, sizeof(uint32_t) * kCntTotalWords, q));
        HIP_TRY(c, hipEventRecord(c->ev_start, q));
        HIP_TRY(c, launch_generate(q, sc, ps, fp, 1u));
    } else {
        // k_generate's output depends on the frame's geometry only (which slots exist: size, rank, streams, whether every stream has
        // a sample): made once per geometry, then a frame starts with a copy of the counter block (pt_context::q_init)
        HIP_TRY(c, c->q_init.ensure(q_entries)); HIP_TRY(c, c->cnt_init.ensure(kCntTotalWords));
        pt_context::InitKey key;
        std::memset(&key, 0, sizeof key);
        key.w = p->width; key.h = p->height; key.rank = p->rank; key.nranks = nranks; key.streams = streams; key.first_spp = std::min(p->spp, streams);
        key.offset = p->sample_offset % streams; key.n_slots = n_slots; key.shard_cap = shard_cap; key.q = c->q_init.p; key.acc = c->acc.p;
        if (!c->init_valid || !(key == c->init_key)) {
            c->init_valid = false;
            HIP_TRY(c, hipMemsetAsync(c->cnt_init.p, 0, sizeof(uint32_t) * kCntTotalWords, q));
            PathState pt = ps;
            pt.counters = c->cnt_init.p; pt.q_ext[0] = c->q_init.p;
            // whole streams without a sample (spp < streams): the first queue holds the live slots only, and the first launch is sized by it
            const bool dense = key.first_spp < streams;
            HIP_TRY(c, launch_generate(q, sc, pt, fp, dense ? 2u : 0u)); // also zeroes every slot's sum (slots that never hold a path stay zero from here on)
            c->init_bound = shard_cap;
            if (dense) {
                HIP_TRY(c, hipMemcpyAsync(c->h_counts + kFinalOffset, c->cnt_init.p, sizeof(uint32_t) * kShards * kCounterStride, hipMemcpyDeviceToHost, q));
                HIP_TRY(c, hipStreamSynchronize(q));
                c->init_bound = 0;
                for (uint32_t sh = 0; sh < kShards; ++sh) c->init_bound = std::max(c->init_bound, c->h_counts[kFinalOffset + cnt_ext_index(0, sh)]);
            }
            c->init_key = key; c->init_valid = true;
        }
        HIP_TRY(c, hipEventRecord(c->ev_start, q));
        HIP_TRY(c, hipMemcpyAsync(c->counters.p, c->cnt_init.p, sizeof(uint32_t) * kCntTotalWords, hipMemcpyDeviceToDevice, q));
    }

    // Wavefront loops. Shards never exchange slots, so the 64 shards are split into `n_loops` independent loops, each on
    // its own HIP stream: the tail of one group's launch (its last wavefronts draining) is filled by the other's launch
    // (pt_context::groups has the measurements). Inside a loop a shard's queue can only shrink (slots die, none are born),
    // so the queue sizes read back `lag` iterations ago are valid launch bounds: the host never stalls the GPU to size a grid.
    // Per-kernel timing, visit counting and the extend-kernel probe (events around single iterations) want kernels alone on
    // the GPU: one loop.
    // Which extend kernel (when no flag forces one): measured, per scene, and remembered in the scene.
    //   inside a frame : iteration 2 of group 0 runs the one-ray-per-lane kernel, iteration 3 the lane-packing one (bit-identical
    //                    results), each bracketed by events; accepted only if both traced a real share of the frame's slots;
    //   across frames  : a frame too short for that (few samples per stream: it is over in two iterations) runs whole on one
    //                    kernel — the first on the one-ray-per-lane kernel, the next on the lane-packing one — and the rays per
    //                    millisecond of the two frames decide.
    // The faster per ray wins (packed needs +10 %). Deep incoherent traversals (1M-triangle soup) gain ~2.5x from packing, shallow
    // ones (walls of a box) lose ~35 %, and no static property of the tree tells them apart. Counting / profiling
    // frames neither probe nor feed the decision: their kernels are instrumented builds.
    const bool undecided = forced_choice == 0u && s->ext_choice == 0u && !count && !profile;
    const uint32_t frame_kernel = (undecided && s->rate_simple > 0.0 && s->rate_packed == 0.0) ? (uint32_t)EXT_PACKED : (uint32_t)EXT_SIMPLE;
    const bool will_probe = undecided && frame_kernel == (uint32_t)EXT_SIMPLE;
    const uint32_t n_loops = (profile || count || will_probe) ? 1u : c->groups ? c->groups : 2u;
    const uint32_t per_group = kShards / n_loops;
    //
    // Queues are carried over IN PLACE from one iteration to the next: a lane writes its own queue position, dead paths
    // leave holes, and lane <-> slot stays the generation order, so the slot-indexed state keeps its coalescing and no
    // returning atomic is needed. A shard re-packs its survivors (ballot + atomic append) in the iteration in which its
    // alive/length ratio is below `compact_below`, and runs its last `finish_below` paths to their end in one launch;
    // both are decided by the kernels from the shard's counters, the host only sizes grids and notices the end.
    struct Loop { hipStream_t stream; uint32_t base, bound, iters; bool done; };
    Loop loops[kMaxGroups];
    HIP_TRY(c, hipEventRecord(c->ev_fork, q));
    for (uint32_t g = 0; g < n_loops; ++g) {
        loops[g] = Loop{ n_loops == 1 ? q : c->group_stream[g], g * per_group, full_state ? shard_cap : c->init_bound, 0u, false }; // no shard's queue can outgrow its first one
        if (loops[g].stream != q) HIP_TRY(c, hipStreamWaitEvent(loops[g].stream, c->ev_fork, 0));
    }
    const uint64_t max_iters = (uint64_t)p->spp * p->max_depth + kLag + 2;
    // Iterations the host runs ahead of the queue sizes it reads back (pt_tuning.lag). The frame ends `lag` launches after its last
    // path, on grids sized `lag` iterations ago: short frames feel that (ms per 1080p frame with lag 4 / 3 / 2, tools/exp_lag.py:
    // 1 spp 0.567 / 0.537 / 0.529, 8 spp 2.79 / 2.73 / 2.70, glass 8 spp 1.57 / 1.52 / 1.48), long ones not (64 spp 17.73 / 17.68 /
    // 17.73; a rank's 1/8 2.63 / 2.59 / 2.61), and the lane-packing kernel's short tail launches want the host further ahead (soup
    // 72.3 / 72.4 / 73.1). At least 2: the launch after the last one that had paths clears that one's counter line.
    // With the sizes stored by the kernels themselves (pt_tuning.readback = 0) iteration j's line is written by launch j + 1, so the
    // same run-ahead of the GPU takes one more iteration of lag than with a copy behind every launch.
    const uint32_t lag = c->lag ? c->lag : (samples_per_stream <= 2u ? 2u : 3u) + (mapped ? 1u : 0u);
    size_t nev = 0;
    uint32_t iters_max = 0;
    // path vertices per launch of the one-ray-per-lane kernel: 3/4 max_depth - 2 (saturating), clamped to [4, 12]
    const uint32_t v34 = p->max_depth * 3u / 4u, default_bounces = std::min(12u, std::max(4u, v34 > 2u ? v34 - 2u : 0u));
    // 0 = probing inside this frame, else the ExtendKernel every iteration uses
    uint32_t ext_choice = forced_choice ? forced_choice : s->ext_choice ? s->ext_choice : will_probe ? 0u : frame_kernel;
    bool mixed = false; // this frame ran probe iterations on both kernels: its overall rate says nothing about either
    uint64_t probe_n[2] = { 0, 0 }, slot_launches = 0;
    const bool trace = profile && getenv("PTRT_TRACE") != nullptr; // developer aid: per-iteration table on stderr
    std::vector<uint64_t> trace_alive, trace_rays;
    for (uint32_t live = n_loops; live > 0;) {
        for (uint32_t g = 0; g < n_loops; ++g) {
            Loop &L = loops[g];
            if (L.done) continue;
            if (L.iters >= max_iters) return fail(c, PT_ERR_INTERNAL, "wavefront loop did not drain after %u iterations", L.iters);
            const uint32_t it = L.iters;
            PathState pg = ps;
            pg.shard_base = L.base; pg.shard_count = per_group;
            if (it == 0u && !full_state) pg.q_ext[0] = c->q_init.p; // the frame's first queue is the same every frame: read, never written
            if (ext_choice == 0u) pg.finish_below = 0u; // while the extend kernel is still being probed, iterations stay comparable
            hipEvent_t e0 = nullptr, e1 = nullptr, e2 = nullptr;
            if (profile) {
                e0 = pool_event(c, nev++); e1 = pool_event(c, nev++); e2 = pool_event(c, nev++);
                if (!e0 || !e1 || !e2) return fail(c, PT_ERR_HIP, "hipEventCreate failed");
                HIP_TRY(c, hipEventRecord(e0, L.stream));
            }
            const bool probing = g == 0u && ext_choice == 0u && (L.iters == 2u || L.iters == 3u);
            const bool use_packed = ext_choice == (uint32_t)EXT_PACKED || (probing && L.iters == 3u);
            const int kernel = use_packed ? EXT_PACKED : (ext_choice == (uint32_t)EXT_POOL && !split_kernels) ? EXT_POOL : EXT_SIMPLE;
            const bool compact = bucket_specular || repack_always; // forced: buckets re-append, there are no fixed positions
            // One kernel per iteration by default: every extend kernel (one ray per lane, lane-packing, pooled) shades its own hits
            // (mode 0: Lambert-only scene, lean code; 2: all kinds). PT_FLAG_SPLIT_KERNELS / _BUCKET_SPECULAR run k_shade as a second kernel.
            const int shade_mode = s->has_specular ? 2 : 0;
            const bool fused = !split_kernels;
            if (probing) HIP_TRY(c, hipEventRecord(c->ev_probe[(L.iters - 2u) * 2u], L.stream));
            HIP_TRY(c, launch_extend(L.stream, sc, pg, fp, it, L.bound, count, kernel, packed_chunk, fused ? shade_mode : -1, compact,
                                     c->bounces ? c->bounces : use_packed ? (probing ? 8u : 64u) : default_bounces));
                                     // lane-packing: a lane pulls a new entry whenever its budget ends, so a long budget costs nothing and
                                     // saves launches (ms per frame with 8 / 16 / 32 / 64 vertices, tools/exp_packed.py: 1M soup 72.2 / 71.5 /
                                     // 70.5 / 67.4, at 256 spp 277.6 / 271.8 / 268.3 / 266.1, 5k soup 7.99 / 7.43 / 7.35 / 7.39); the probe
                                     // iteration keeps 8 so that it stays comparable with the one before it
            if (profile) HIP_TRY(c, hipEventRecord(e1, L.stream));
            if (!fused && !bucket_specular) HIP_TRY(c, launch_shade(L.stream, sc, pg, fp, it, L.bound, shade_mode, compact));
            else if (!fused) {
                HIP_TRY(c, launch_shade(L.stream, sc, pg, fp, it, L.bound, 0, true));
                HIP_TRY(c, launch_shade(L.stream, sc, pg, fp, it, L.bound, 1, true)); // metal + dielectric buckets
            }
            if (probing) HIP_TRY(c, hipEventRecord(c->ev_probe[(L.iters - 2u) * 2u + 1u], L.stream)); // the whole iteration, either way
            if (profile) HIP_TRY(c, hipEventRecord(e2, L.stream));
            const uint32_t ring = L.iters % kLag;
            if (!mapped) {
                uint32_t *h_ring = c->h_counts + ((size_t)g * kLag + ring) * kRingWords;
                HIP_TRY(c, hipMemcpyAsync(h_ring, c->counters.p + cnt_ext_index((it + 1u) % 3u, L.base), sizeof(uint32_t) * per_group * kCounterStride,
                                          hipMemcpyDeviceToHost, L.stream));
            }
            HIP_TRY(c, hipEventRecord(c->ev_lag[g][ring], L.stream));
            ++L.iters;
            iters_max = std::max(iters_max, L.iters);
            if (L.iters >= lag) {
                const uint32_t old_iter = L.iters - lag; // iteration old_iter traced `traced` rays and left `total` paths alive: its survivors bound every later queue
                // mapped: launch old_iter + 1 stored old_iter's lines (fold_traced); it is at most the launch just enqueued since lag >= 2
                HIP_TRY(c, hipEventSynchronize(c->ev_lag[g][(mapped ? old_iter + 1u : old_iter) % kLag]));
                const volatile uint32_t *h_old = mapped ? (const volatile uint32_t *)(c->h_ring + (size_t)(old_iter % kLag) * kShards + L.base)
                                               : c->h_counts + ((size_t)g * kLag + old_iter % kLag) * kRingWords;
                const uint32_t line = mapped ? 4u : kCounterStride;
                uint32_t mx = 0;
                // a shard's line: word 0 = queue length (holes included), word 1 = alive entries, words 2-3 = rays the iteration traced
                uint64_t total = 0, traced = 0;
                for (uint32_t sh = 0; sh < per_group; ++sh) {
                    mx = std::max(mx, (uint32_t)h_old[sh * line]);
                    total += h_old[sh * line + 1];
                    traced += (uint64_t)h_old[sh * line + 2] | ((uint64_t)h_old[sh * line + 3] << 32);
                }
                L.bound = mx;
                if (trace) {
                    trace_alive.resize(std::max<size_t>(trace_alive.size(), old_iter + 1), 0); trace_rays.resize(trace_alive.size(), 0);
                    trace_alive[old_iter] += total; trace_rays[old_iter] += traced;
                }
                slot_launches += total; // = paths alive at the start of iteration old_iter + 1 (those read after the loop ended are all 0)
                if (total == 0) { L.done = true; --live; }
                if (g == 0u && ext_choice == 0u && (old_iter == 2u || old_iter == 3u)) probe_n[old_iter - 2u] = traced;
                if (g == 0u && ext_choice == 0u && old_iter == 3u) { // iterations 2 and 3 (and their events) are complete by now
                    float ms_simple = 0.f, ms_packed = 0.f;
                    HIP_TRY(c, hipEventElapsedTime(&ms_simple, c->ev_probe[0], c->ev_probe[1]));
                    HIP_TRY(c, hipEventElapsedTime(&ms_packed, c->ev_probe[2], c->ev_probe[3]));
                    const double r_simple = probe_n[0] / std::max((double)ms_simple, 1e-6), r_packed = probe_n[1] / std::max((double)ms_packed, 1e-6);
                    const uint64_t enough = (uint64_t)(n_slots / n_loops) / 8u; // each probe iteration must have traced a real share of the slots
                    if (probe_n[0] >= enough && probe_n[1] >= enough) {
                        ext_choice = r_packed > 1.10 * r_simple ? (uint32_t)EXT_PACKED : (uint32_t)EXT_SIMPLE;
                        s->ext_choice = ext_choice;
                    } else { // inconclusive (the frame was all but over): finish on the default and let whole frames decide
                        ext_choice = (uint32_t)EXT_SIMPLE;
                        mixed = probe_n[1] >= enough / 8u; // did the lane-packing iteration trace enough to colour this frame's rate?
                    }
                }
            }
        }
    }
    for (uint32_t g = 0; g < n_loops; ++g) // join: the main stream continues after every group's last kernel
        if (loops[g].stream != q) {
            HIP_TRY(c, hipEventRecord(c->ev_join[g], loops[g].stream));
            HIP_TRY(c, hipStreamWaitEvent(q, c->ev_join[g], 0));
        }
    const uint32_t iters = iters_max;
    HIP_TRY(c, launch_reduce_streams(q, c->acc.p, c->tiles.p, pixel_slots, streams)); // tiles = the pixel sums = the gather payload
    if (nranks == 1)
        HIP_TRY(c, launch_assemble(q, c->tiles.p, 1, pixel_slots, p->width, p->height, lay.tiles_x, lay.n_tiles, 1.0f / (float)total_spp, c->fb.p, c->fb8.p));
    HIP_TRY(c, hipEventRecord(c->ev_stop, q));
    HIP_TRY(c, hipMemcpyAsync(c->h_counts + kFinalOffset, c->counters.p, sizeof(uint32_t) * kCntTotalWords, hipMemcpyDeviceToHost, q));
    HIP_TRY(c, hipStreamSynchronize(q));

    const uint32_t *hc = c->h_counts + kFinalOffset;
    if (hc[kCntError]) return fail(c, PT_ERR_INTERNAL, "device error flag 0x%x (1 = traversal stack overflow, 2 = step limit)", hc[kCntError]);
    auto u64_at = [&](uint32_t w) { return (uint64_t)hc[w] | ((uint64_t)hc[w + 1] << 32); };
    for (uint32_t sh = 0; sh < kShards; ++sh) {
        if (hc[cnt_alive_index(0, sh)] || hc[cnt_alive_index(1, sh)] || hc[cnt_alive_index(2, sh)]) return fail(c, PT_ERR_INTERNAL, "extend queue of shard %u not empty at frame end", sh);
        out.rays += u64_at(cnt_rays_index(sh));
    }
    float ms = 0.f; HIP_TRY(c, hipEventElapsedTime(&ms, c->ev_start, c->ev_stop));
    out.gpu_ms = ms;
    out.node_visits = u64_at(kCntNodes); out.tri_tests = u64_at(kCntTris); out.sphere_tests = u64_at(kCntSph);
    // PT_FLAG_COUNT_VISITS + one-ray-per-lane kernel: wave-level node-loop iterations (bits 0-39) and, from bit 40 up, how many
    // of them came after the wave's first leaf phase of the ray (diagnostic for tools/exp_util.py)
    out.reserved[3] = (u64_at(kCntWaveNodeIters) & 0xFFFFFFFFFFull) | (u64_at(kCntWaveNodeIters + 2) << 40);
    if (count && getenv("PTRT_TRACE")) { // developer aid: where the node loop's lane-slots go (one-ray-per-lane kernel)
        const double slots = 64.0 * (double)u64_at(kCntWaveNodeIters), v = (double)out.node_visits, lf = (double)u64_at(kCntIdleLeaf), dn = (double)u64_at(kCntIdleDone);
        if (slots > 0) fprintf(stderr, "ptrt: node-loop lane-slots %.3g: visiting %.1f %%, waiting at a leaf %.1f %%, ray finished %.1f %%, no ray %.1f %%\n", slots,
                               100 * v / slots, 100 * lf / slots, 100 * dn / slots, 100 * (slots - v - lf - dn) / slots);
    }
    out.iterations = iters; out.extend_launches = iters;
    const bool cold_frame = g_device_allocs != allocs_before; // first touch of fresh allocations: 30 % slower, not a measurement
    if (undecided && s->ext_choice == 0u && !mixed && !cold_frame && out.rays >= (1u << 20) && out.gpu_ms > 0.0) { // a whole frame on one kernel: remember its rate
        (frame_kernel == (uint32_t)EXT_PACKED ? s->rate_packed : s->rate_simple) = (double)out.rays / out.gpu_ms;
        if (s->rate_simple > 0.0 && s->rate_packed > 0.0) s->ext_choice = s->rate_packed > 1.10 * s->rate_simple ? (uint32_t)EXT_PACKED : (uint32_t)EXT_SIMPLE;
    } else if (undecided && s->ext_choice == 0u && !cold_frame && ++s->probe_misses >= 3u) s->ext_choice = (uint32_t)EXT_SIMPLE; // frames too small to time
    out.reserved[0] = ext_choice ? ext_choice : (uint32_t)EXT_SIMPLE; // extend kernel in use at frame end (ExtendKernel)
    out.reserved[1] = hc[kCntCompactions]; // (shard, iteration) pairs that re-packed their queue (the others carried it over in place)
    {   // paths = owned in-image pixels x spp
        uint64_t px = 0;
        for (uint32_t t = p->rank; t < lay.n_tiles; t += nranks) {
            const uint32_t tx = t % lay.tiles_x, ty = t / lay.tiles_x;
            const uint32_t w = std::min(kTile, p->width - tx * kTile), h = std::min(kTile, p->height - ty * kTile);
            px += (uint64_t)w * h;
        }
        out.paths = px * p->spp;
        // path states read + written by the wavefront loop = sum over launches of the paths alive at launch start
        // (iteration 0 starts every (pixel, stream) pair that has a sample)
        out.reserved[2] = slot_launches + px * std::min(streams, p->spp);
    }
    if (profile) {
        for (size_t i = 0; i + 2 < nev; i += 3) { // three events per iteration: before extend, between, after shade
            float a = 0.f, b = 0.f;
            HIP_TRY(c, hipEventElapsedTime(&a, c->ev_pool[i], c->ev_pool[i + 1]));
            HIP_TRY(c, hipEventElapsedTime(&b, c->ev_pool[i + 1], c->ev_pool[i + 2]));
            out.extend_ms += a; out.shade_ms += b;
            if (trace) fprintf(stderr, "ptrt: iteration %3zu  rays %10llu  alive after %10llu  extend %8.3f ms  shade %8.3f ms\n", i / 3,
                               (unsigned long long)(i / 3 < trace_rays.size() ? trace_rays[i / 3] : 0),
                               (unsigned long long)(i / 3 < trace_alive.size() ? trace_alive[i / 3] : 0), a, b);
        }
        out.other_ms = out.gpu_ms - out.extend_ms - out.shade_ms;
    }
    c->n_slots = pixel_slots;
    c->acc_w = p->width; c->acc_h = p->height; c->acc_rank = p->rank; c->acc_nranks = nranks; c->acc_streams = streams; c->acc_seed = p->seed;
    c->acc_spp = total_spp;
    c->fb_valid = (nranks == 1);
    if (stats) *stats = out;
    return PT_OK;
}

pt_status pt_framebuffer_read(pt_context *c, float *rgba, uint64_t n_floats)
{
    if (!c || !rgba) return fail(c, PT_ERR_INVALID_ARGUMENT, "NULL argument");
    if (!c->fb_valid) return fail(c, PT_ERR_NOT_COMMITTED, "no assembled frame (render with nranks == 1 or call pt_assemble_tiles)");
    const uint64_t need = (uint64_t)c->fb_w * c->fb_h * 4;
    if (n_floats < need) return fail(c, PT_ERR_INVALID_ARGUMENT, "buffer too small: need %llu floats", (unsigned long long)need);
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipMemcpyAsync(rgba, c->fb.p, need * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return PT_OK;
}

pt_status pt_framebuffer_read_rgba8(pt_context *c, uint8_t *rgba8, uint64_t n_bytes)
{
    if (!c || !rgba8) return fail(c, PT_ERR_INVALID_ARGUMENT, "NULL argument");
    if (!c->fb_valid) return fail(c, PT_ERR_NOT_COMMITTED, "no assembled frame");
    const uint64_t need = (uint64_t)c->fb_w * c->fb_h * 4;
    if (n_bytes < need) return fail(c, PT_ERR_INVALID_ARGUMENT, "buffer too small: need %llu bytes", (unsigned long long)need);
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipMemcpyAsync(rgba8, c->fb8.p, need, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return PT_OK;
}

pt_status pt_framebuffer_read_srgb8(pt_context *c, uint8_t *rgba8, uint64_t n_bytes)
{
    // display transform of the reference (SwapChain.cs:157-158 B8G8R8A8Srgb target, nearest-sampled UNORM8 source): a function of
    // the 8-bit value, so it is a 256-entry table over the UNORM8 read-back; alpha is linear in sRGB formats
    pt_status st = pt_framebuffer_read_rgba8(c, rgba8, n_bytes);
    if (st != PT_OK) return st;
    uint8_t lut[256];
    for (int q = 0; q < 256; ++q) {
        const double l = q / 255.0, e = l <= 0.0031308 ? 12.92 * l : 1.055 * std::pow(l, 1.0 / 2.4) - 0.055;
        lut[q] = (uint8_t)std::floor(255.0 * e + 0.5);
    }
    const uint64_t n = (uint64_t)c->fb_w * c->fb_h * 4;
    for (uint64_t i = 0; i < n; ++i) if ((i & 3u) != 3u) rgba8[i] = lut[rgba8[i]];
    return PT_OK;
}

pt_status pt_framebuffer_device_ptr(pt_context *c, void **dptr, uint64_t *n_floats)
{
    if (!c || !dptr) return fail(c, PT_ERR_INVALID_ARGUMENT, "NULL argument");
    if (!c->fb_valid) return fail(c, PT_ERR_NOT_COMMITTED, "no assembled frame");
    *dptr = c->fb.p;
    if (n_floats) *n_floats = (uint64_t)c->fb_w * c->fb_h * 4;
    return PT_OK;
}

pt_status pt_tiles_device_ptr(pt_context *c, void **dptr, uint64_t *n_floats)
{
    if (!c || !dptr) return fail(c, PT_ERR_INVALID_ARGUMENT, "NULL argument");
    if (!c->n_slots) return fail(c, PT_ERR_NOT_COMMITTED, "no path-traced frame yet");
    *dptr = c->tiles.p;
    if (n_floats) *n_floats = (uint64_t)c->n_slots * 4;
    return PT_OK;
}

pt_status pt_assemble_tiles(pt_context *c, const pt_render_params *p, const void *gathered, uint64_t n_floats)
{
    if (!c || !p || !gathered) return fail(c, PT_ERR_INVALID_ARGUMENT, "NULL argument");
    pt_tile_layout lay;
    pt_status st = layout_of(p, &lay);
    if (st != PT_OK) return fail(c, st, "pt_assemble_tiles: bad params");
    if (p->spp == 0) return fail(c, PT_ERR_INVALID_ARGUMENT, "spp == 0");
    const uint32_t nranks = p->nranks ? p->nranks : 1u;
    const uint64_t per_rank = (uint64_t)lay.tiles_per_rank * kTilePixels;
    if (n_floats < per_rank * nranks * 4) return fail(c, PT_ERR_INVALID_ARGUMENT, "gathered buffer too small: need %llu floats", (unsigned long long)(per_rank * nranks * 4));
    HIP_TRY(c, hipSetDevice(c->device));
    if ((st = ensure_frame(c, p->width, p->height)) != PT_OK) return st;
    HIP_TRY(c, launch_assemble(c->stream, (const float4 *)gathered, nranks, (uint32_t)per_rank, p->width, p->height, lay.tiles_x, lay.n_tiles,
                               1.0f / (float)(((p->flags & PT_FLAG_ACCUMULATE) ? (uint64_t)p->sample_offset : 0u) + p->spp), c->fb.p, c->fb8.p));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->fb_valid = true;
    return PT_OK;
}

} // extern "C"
