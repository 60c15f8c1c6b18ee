// comm.cpp — several GPUs of one node behind the C ABI (include/ptrt.h "pt_comm"; docs/SPEC.md §6).
// One process, one pt_context per rank, tiles dealt round-robin to ranks, scene replicated, ONE exchange per frame:
//   ranks on distinct devices : ncclGather (rccl.h:745) on a communicator from ncclCommInitAll (rccl.h:236), one call per rank inside
//                               ncclGroupStart/End, each on its own context's stream, so the exchange is ordered after that rank's
//                               kernels and before its next frame without any host wait except the root's;
//   ranks sharing one context : rendered one after the other, blocks staged by device-to-device copies (single-GPU rehearsal).
// RCCL is loaded with dlopen on first use: libptrt.so has no link-time dependency on it, and inside a PyTorch process the copy
// that torch already mapped (same SONAME) is the one that is used.
// The reference has nothing like this (one logical device, GraphicsDevice.cs:176-183).
#include "ptrt_internal.h"
#include <hip/hip_runtime.h>
#include <rccl/rccl.h> // types and prototypes only; the functions are resolved at run time
#include <dlfcn.h>
#include <mutex>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <thread>
#include <vector>

using namespace ptrt;

namespace {

struct Rccl {
    void *handle = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGather) Gather = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string error;
    bool loaded = false; // set only once all six symbols are resolved: a half-loaded library is never called
    std::mutex mu;       // contexts (and their comms) may be created from several host threads
    bool load()
    {
        std::lock_guard<std::mutex> lock(mu);
        if (loaded) return true;
        if (!handle)
            for (const char *name : { "librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so" }) {
                handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
                if (handle) break;
            }
        if (!handle) { error = std::string("cannot load librccl.so.1: ") + dlerror(); return false; }
        error.clear();
        auto sym = [&](const char *n) { void *p = dlsym(handle, n); if (!p) error = std::string("librccl lacks ") + n; return p; };
        CommInitAll = (decltype(CommInitAll))sym("ncclCommInitAll");
        CommDestroy = (decltype(CommDestroy))sym("ncclCommDestroy");
        GroupStart = (decltype(GroupStart))sym("ncclGroupStart");
        GroupEnd = (decltype(GroupEnd))sym("ncclGroupEnd");
        Gather = (decltype(Gather))sym("ncclGather");
        GetErrorString = (decltype(GetErrorString))sym("ncclGetErrorString");
        loaded = CommInitAll && CommDestroy && GroupStart && GroupEnd && Gather && GetErrorString;
        return loaded;
    }
};
Rccl g_rccl;

} // namespace

struct pt_comm {
    std::vector<pt_context *> ctx; // per rank
    std::vector<int> dev;          // per rank: the context's device (pt_comm_destroy does not touch the contexts)
    uint32_t root = 0;
    bool shared = false;           // every rank on one context (virtual ranks)
    bool use_rccl = false;
    std::vector<ncclComm_t> comms; // per rank (use_rccl)
    float *gathered = nullptr;     // root device: n_ranks blocks of tiles_per_rank * 4096 float4
    uint64_t gathered_floats = 0;
    std::vector<uint8_t> staged;   // shared contexts: which blocks pt_comm_stage_tiles has filled
    std::string err;
};

namespace {

pt_status cfail(pt_comm *c, pt_status code, const char *fmt, ...)
{
    char buf[512];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    if (c) { c->err = buf; context_set_error(c->ctx.empty() ? nullptr : c->ctx[c->root], buf); }
    else context_set_error(nullptr, buf);
    return code;
}
#define C_HIP(c, expr) do { hipError_t _e = (expr); if (_e != hipSuccess) return cfail(c, PT_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(_e)); } while (0)
#define C_NCCL(c, expr) do { ncclResult_t _r = (expr); if (_r != ncclSuccess) return cfail(c, PT_ERR_HIP, "%s failed: %s", #expr, g_rccl.GetErrorString(_r)); } while (0)

// the gather buffer on the root's device, sized for `params`
pt_status ensure_gathered(pt_comm *c, const pt_render_params *p, uint64_t &per_rank_floats)
{
    pt_render_params q = *p;
    q.rank = 0; q.nranks = (uint32_t)c->ctx.size();
    pt_tile_layout lay;
    pt_status st = pt_tile_layout_query(&q, &lay);
    if (st != PT_OK) return cfail(c, st, "pt_comm: bad render params");
    per_rank_floats = (uint64_t)lay.tiles_per_rank * lay.floats_per_tile;
    const uint64_t need = per_rank_floats * c->ctx.size();
    if (need > c->gathered_floats) {
        C_HIP(c, hipSetDevice(context_device(c->ctx[c->root])));
        if (c->gathered) (void)hipFree(c->gathered);
        c->gathered = nullptr; c->gathered_floats = 0;
        C_HIP(c, hipMalloc((void **)&c->gathered, need * sizeof(float)));
        c->gathered_floats = need;
    }
    return PT_OK;
}

} // namespace

extern "C" {

pt_status pt_comm_create(pt_context *const *ctxs, uint32_t n_ranks, uint32_t root, uint32_t flags, pt_comm **out)
{
    if (!out) return cfail(nullptr, PT_ERR_INVALID_ARGUMENT, "pt_comm_create: out is NULL");
    *out = nullptr;
    if (!ctxs || n_ranks == 0 || n_ranks > 64 || root >= n_ranks) return cfail(nullptr, PT_ERR_INVALID_ARGUMENT, "pt_comm_create: need 1..64 contexts and root < n_ranks");
    for (uint32_t i = 0; i < n_ranks; ++i) if (!ctxs[i]) return cfail(nullptr, PT_ERR_INVALID_ARGUMENT, "pt_comm_create: context %u is NULL", i);
    bool all_same = true, all_distinct = true;
    for (uint32_t i = 0; i < n_ranks; ++i)
        for (uint32_t j = i + 1; j < n_ranks; ++j) {
            if (ctxs[i] != ctxs[j]) all_same = false;
            if (ctxs[i] == ctxs[j] || context_device(ctxs[i]) == context_device(ctxs[j])) all_distinct = false;
        }
    bool ctx_distinct = true; // every rank its own context (devices may repeat)
    for (uint32_t i = 0; i < n_ranks; ++i) for (uint32_t j = i + 1; j < n_ranks; ++j) if (ctxs[i] == ctxs[j]) ctx_distinct = false;
    const bool copies = (flags & PT_COMM_COPY_EXCHANGE) != 0;
    if (n_ranks > 1 && !all_same && !all_distinct && !(copies && ctx_distinct))
        return cfail(nullptr, PT_ERR_UNSUPPORTED, "pt_comm_create: ranks must either all have their own context on their own device, or all share one context, "
                                                  "or (PT_COMM_COPY_EXCHANGE) all have their own context on any devices");
    pt_comm *c = new (std::nothrow) pt_comm();
    if (!c) return cfail(nullptr, PT_ERR_OUT_OF_MEMORY, "host allocation failed");
    c->ctx.assign(ctxs, ctxs + n_ranks);
    for (uint32_t i = 0; i < n_ranks; ++i) c->dev.push_back(context_device(ctxs[i]));
    c->root = root;
    c->shared = n_ranks > 1 && all_same;
    c->staged.assign(n_ranks, 0);
    c->use_rccl = !c->shared && !copies && (n_ranks > 1 || (flags & PT_COMM_FORCE_RCCL));
    if (c->use_rccl) {
        if (!g_rccl.load()) { const std::string e = g_rccl.error; delete c; return cfail(nullptr, PT_ERR_UNSUPPORTED, "%s", e.c_str()); }
        std::vector<int> devs(n_ranks);
        for (uint32_t i = 0; i < n_ranks; ++i) devs[i] = context_device(ctxs[i]);
        c->comms.assign(n_ranks, nullptr);
        const ncclResult_t r = g_rccl.CommInitAll(c->comms.data(), (int)n_ranks, devs.data());
        if (r != ncclSuccess) { const char *e = g_rccl.GetErrorString(r); delete c; return cfail(nullptr, PT_ERR_HIP, "ncclCommInitAll failed: %s", e); }
    }
    *out = c;
    return PT_OK;
}

void pt_comm_destroy(pt_comm *c)
{
    if (!c) return;
    // Every pt_comm call returns with nothing of its own in flight (pt_comm_assemble waits for every rank's stream), so the
    // contexts are not needed here and may already be gone: the order of pt_comm_destroy and pt_context_destroy is free.
    for (size_t i = 0; i < c->comms.size(); ++i) if (c->comms[i]) { (void)hipSetDevice(c->dev[i]); (void)g_rccl.CommDestroy(c->comms[i]); }
    if (c->gathered) { (void)hipSetDevice(c->dev[c->root]); (void)hipFree(c->gathered); }
    delete c;
}

pt_status pt_comm_stage_tiles(pt_comm *c, uint32_t rank)
{
    if (!c || rank >= c->ctx.size()) return cfail(c, PT_ERR_INVALID_ARGUMENT, "pt_comm_stage_tiles: bad rank");
    if (!c->shared) return PT_OK; // own context: the block stays in that context's tile buffer until the exchange reads it
    void *tiles = nullptr; uint64_t nf = 0;
    pt_status st = pt_tiles_device_ptr(c->ctx[rank], &tiles, &nf);
    if (st != PT_OK) return cfail(c, st, "pt_comm_stage_tiles: rank %u has no rendered tiles", rank);
    if (nf * c->ctx.size() > c->gathered_floats) { // first frame of this size: allocate from the block size the context reports
        C_HIP(c, hipSetDevice(context_device(c->ctx[c->root])));
        if (c->gathered) (void)hipFree(c->gathered);
        c->gathered = nullptr; c->gathered_floats = 0;
        C_HIP(c, hipMalloc((void **)&c->gathered, nf * c->ctx.size() * sizeof(float)));
        c->gathered_floats = nf * c->ctx.size();
        std::fill(c->staged.begin(), c->staged.end(), 0);
    }
    hipStream_t s = context_stream(c->ctx[rank]);
    C_HIP(c, hipMemcpyAsync(c->gathered + nf * rank, tiles, nf * sizeof(float), hipMemcpyDeviceToDevice, s));
    C_HIP(c, hipStreamSynchronize(s)); // the context's next pt_render rewrites its tile buffer
    c->staged[rank] = 1;
    return PT_OK;
}

pt_status pt_comm_assemble(pt_comm *c, const pt_render_params *p)
{
    if (!c || !p) return cfail(c, PT_ERR_INVALID_ARGUMENT, "pt_comm_assemble: NULL argument");
    const uint32_t n = (uint32_t)c->ctx.size();
    uint64_t per_rank = 0;
    pt_render_params q = *p;
    q.rank = 0; q.nranks = n;
    pt_context *root = c->ctx[c->root];
    if (c->shared) {
        pt_tile_layout lay;
        pt_status st = pt_tile_layout_query(&q, &lay);
        if (st != PT_OK) return cfail(c, st, "pt_comm_assemble: bad render params");
        per_rank = (uint64_t)lay.tiles_per_rank * lay.floats_per_tile;
        for (uint32_t i = 0; i < n; ++i) if (!c->staged[i]) return cfail(c, PT_ERR_NOT_COMMITTED, "pt_comm_assemble: rank %u was not staged (pt_comm_stage_tiles)", i);
        if (per_rank * n > c->gathered_floats) return cfail(c, PT_ERR_INVALID_ARGUMENT, "pt_comm_assemble: params do not match the staged blocks");
        std::fill(c->staged.begin(), c->staged.end(), 0);
    } else {
        pt_status st = ensure_gathered(c, p, per_rank);
        if (st != PT_OK) return st;
        std::vector<void *> tiles(n);
        for (uint32_t i = 0; i < n; ++i) {
            uint64_t nf = 0;
            st = pt_tiles_device_ptr(c->ctx[i], &tiles[i], &nf);
            if (st != PT_OK || nf != per_rank) return cfail(c, st != PT_OK ? st : PT_ERR_INVALID_ARGUMENT, "pt_comm_assemble: rank %u has no tiles of this frame", i);
        }
        if (c->use_rccl) {
            // one collective per frame; every rank's call sits on its own context's stream (after its kernels)
            // Whatever fails between GroupStart and GroupEnd, the group is closed and the ranks already posted are drained before
            // the error goes back: no open RCCL group and no collective in flight survive the call.
            C_NCCL(c, g_rccl.GroupStart());
            pt_status posted = PT_OK;
            uint32_t n_posted = 0;
            for (uint32_t i = 0; i < n && posted == PT_OK; ++i) {
                const hipError_t he = hipSetDevice(context_device(c->ctx[i]));
                if (he != hipSuccess) { posted = cfail(c, PT_ERR_HIP, "hipSetDevice (rank %u) failed: %s", i, hipGetErrorString(he)); break; }
                const ncclResult_t r = g_rccl.Gather(tiles[i], c->gathered, per_rank, ncclFloat, (int)c->root, c->comms[i], context_stream(c->ctx[i]));
                if (r != ncclSuccess) { posted = cfail(c, PT_ERR_HIP, "ncclGather (rank %u) failed: %s", i, g_rccl.GetErrorString(r)); break; }
                ++n_posted;
            }
            const ncclResult_t ge = g_rccl.GroupEnd();
            if (posted != PT_OK || ge != ncclSuccess) {
                const std::string msg = posted != PT_OK ? c->err : std::string("ncclGroupEnd failed: ") + g_rccl.GetErrorString(ge);
                for (uint32_t i = 0; i < n_posted; ++i) { (void)hipSetDevice(context_device(c->ctx[i])); (void)hipStreamSynchronize(context_stream(c->ctx[i])); }
                return cfail(c, posted != PT_OK ? posted : PT_ERR_HIP, "%s", msg.c_str());
            }
        } else { // without RCCL (a single rank, or PT_COMM_COPY_EXCHANGE): every rank's block is copied to the root's buffer on that rank's own
                 // stream — after its kernels by stream order; a peer copy over xGMI when the rank lives on another device — then waited for
            const int rdev = context_device(root);
            for (uint32_t i = 0; i < n; ++i) {
                const int dev = context_device(c->ctx[i]);
                C_HIP(c, hipSetDevice(dev));
                float *dst = c->gathered + (size_t)per_rank * i;
                if (dev == rdev) C_HIP(c, hipMemcpyAsync(dst, tiles[i], per_rank * sizeof(float), hipMemcpyDeviceToDevice, context_stream(c->ctx[i])));
                else C_HIP(c, hipMemcpyPeerAsync(dst, rdev, tiles[i], dev, per_rank * sizeof(float), context_stream(c->ctx[i])));
            }
            for (uint32_t i = 0; i < n; ++i)
                if (c->ctx[i] != root) { C_HIP(c, hipSetDevice(context_device(c->ctx[i]))); C_HIP(c, hipStreamSynchronize(context_stream(c->ctx[i]))); }
        }
    }
    // un-tile on the root: same stream as the root's receive, then the host waits (pt_assemble_tiles is synchronous)
    pt_status st = pt_assemble_tiles(root, &q, c->gathered, per_rank * n);
    if (c->use_rccl) // the senders' halves of the gather: done once the root has received, waited for so that nothing outlives the call
        for (uint32_t i = 0; i < n; ++i)
            if (i != c->root) { (void)hipSetDevice(c->dev[i]); (void)hipStreamSynchronize(context_stream(c->ctx[i])); }
    if (st != PT_OK) return cfail(c, st, "pt_comm_assemble: %s", pt_last_error(root));
    return PT_OK;
}

pt_status pt_comm_render(pt_comm *c, const pt_scene *const *scenes, const pt_render_params *p, pt_stats *stats)
{
    if (!c || !scenes || !p) return cfail(c, PT_ERR_INVALID_ARGUMENT, "pt_comm_render: NULL argument");
    const uint32_t n = (uint32_t)c->ctx.size();
    std::vector<pt_status> rc(n, PT_OK);
    std::vector<pt_stats> local(n);
    auto render_rank = [&](uint32_t i) {
        pt_render_params q = *p;
        q.rank = i; q.nranks = n;
        rc[i] = pt_render(c->ctx[i], scenes[i], &q, &local[i]);
    };
    if (c->shared) { // virtual ranks: one context, one after the other, each block staged before the next render overwrites it
        for (uint32_t i = 0; i < n; ++i) {
            render_rank(i);
            if (rc[i] != PT_OK) return cfail(c, rc[i], "pt_comm_render: rank %u: %s", i, pt_last_error(c->ctx[i]));
            pt_status st = pt_comm_stage_tiles(c, i);
            if (st != PT_OK) return st;
        }
    } else { // one host thread per context: pt_render is synchronous and its frame loop is host-driven
        std::vector<std::thread> th;
        for (uint32_t i = 1; i < n; ++i) th.emplace_back(render_rank, i);
        render_rank(0);
        for (auto &t : th) t.join();
        for (uint32_t i = 0; i < n; ++i)
            if (rc[i] != PT_OK) return cfail(c, rc[i], "pt_comm_render: rank %u: %s", i, pt_last_error(c->ctx[i]));
    }
    if (stats) std::memcpy(stats, local.data(), sizeof(pt_stats) * n);
    return pt_comm_assemble(c, p);
}

} // extern "C"
