// C++ host above the C ABI (include/ptrt.h), mirroring the reference's App / Renderer shape
// (RayTracing/App.cs:7-68, RayTracing/Graphics/Renderer.cs:59-89, 933-1004, 1006-1040): throw-on-failure,
// RAII instead of IDisposable, synchronous Render(). Header-only; links against libptrt.so.
#pragma once
#include "../../include/ptrt.h"
#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace ptrt_host {

struct Error : std::runtime_error { pt_status status; Error(pt_status s, const std::string &m) : std::runtime_error(m), status(s) {} };
inline void check(pt_status s, const pt_context *c = nullptr)
{
    if (s != PT_OK) throw Error(s, std::string("ptrt status ") + std::to_string(s) + ": " + pt_last_error(c));
}

class Renderer {
public:
    pt_render_params Params{};
    pt_stats LastStats{};

    Renderer(uint32_t width = 1920, uint32_t height = 1080) // App.cs:27
    {
        Params.width = width; Params.height = height; Params.spp = 1; Params.max_depth = 8; Params.rr_start = 3;
        Params.seed = 0x5EED0001u; Params.mode = PT_REFERENCE_SPHERE; Params.ray_eps = 1e-4f; Params.nranks = 1; Params.streams = 8;
    }
    Renderer(const Renderer &) = delete;
    ~Renderer() { Dispose(); }

    void Init(int device = 0) // Renderer.Init, Renderer.cs:66-84
    {
        if (pt_abi_version() != PTRT_ABI_VERSION) // the library found at run time is not the one this header describes
            throw Error(PT_ERR_UNSUPPORTED, "libptrt has ABI version " + std::to_string(pt_abi_version()) + ", built against " + std::to_string(PTRT_ABI_VERSION));
        pt_device_desc d{}; d.device_ordinal = device;
        check(pt_context_create(&d, &ctx_));
    }
    void LoadSyntheticScene(uint32_t kind, uint32_t detail = 0, uint32_t seed = 0x5EED0001u, uint32_t bvh_width = 0)
    {
        pt_scene_counts n{}; pt_camera cam{}; float sky[3];
        check(pt_scenegen(kind, detail, seed, Params.width, Params.height, &n, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr));
        std::vector<float> verts(n.n_tris * 9 + 1), sph(n.n_spheres * 4 + 1);
        std::vector<uint32_t> tm(n.n_tris + 1), sm(n.n_spheres + 1);
        std::vector<pt_material> mats(n.n_mats + 1);
        check(pt_scenegen(kind, detail, seed, Params.width, Params.height, &n, verts.data(), tm.data(), sph.data(), sm.data(), mats.data(), &cam, sky));
        if (scene_) { pt_scene_destroy(scene_); scene_ = nullptr; }
        check(pt_scene_create(ctx_, &scene_), ctx_);
        check(pt_scene_set_triangles(scene_, verts.data(), tm.data(), n.n_tris), ctx_);
        check(pt_scene_set_spheres(scene_, sph.data(), sm.data(), n.n_spheres), ctx_);
        check(pt_scene_set_materials(scene_, mats.data(), n.n_mats), ctx_);
        check(pt_scene_set_camera(scene_, &cam), ctx_);
        check(pt_scene_set_sky(scene_, sky), ctx_);
        check(pt_scene_commit(scene_, bvh_width), ctx_);
        Params.mode = PT_PATH_TRACE;
    }
    void Update(float) {}                               // Renderer.cs:86-89: empty
    void Render(float delta) { ComputeFrame(delta); }   // Renderer.cs:933-1004 without acquire/draw/present
    std::vector<float> ReadFramebuffer()
    {
        std::vector<float> px((size_t)Params.width * Params.height * 4);
        check(pt_framebuffer_read(ctx_, px.data(), px.size()), ctx_);
        return px;
    }
    std::vector<uint8_t> ReadFramebufferRgba8()
    {
        std::vector<uint8_t> px((size_t)Params.width * Params.height * 4);
        check(pt_framebuffer_read_rgba8(ctx_, px.data(), px.size()), ctx_);
        return px;
    }
    void Dispose()
    {
        if (scene_) pt_scene_destroy(scene_);
        if (ctx_) pt_context_destroy(ctx_);
        scene_ = nullptr; ctx_ = nullptr;
    }
    pt_context *Context() const { return ctx_; }
    pt_scene *Scene() const { return scene_; }

private:
    void ComputeFrame(float) // Renderer.cs:1006-1040 + fence wait :972
    {
        check(pt_render(ctx_, Params.mode == PT_PATH_TRACE ? scene_ : nullptr, &Params, &LastStats), ctx_);
    }
    pt_context *ctx_ = nullptr;
    pt_scene *scene_ = nullptr;
};

// One frame over several GPUs of the node (include/ptrt.h pt_comm): a Renderer per device, the same scene committed on each, tiles
// dealt round-robin, one ncclGather per frame inside libptrt. `virtual_ranks`: every rank on device 0 through one Renderer
// (rehearsal on a single GPU). The reference is single-device; this is what stands above its Renderer when a node has 8 GPUs.
class MultiRenderer {
public:
    pt_render_params Params{};
    std::vector<pt_stats> LastStats;

    // mode 0: one device per rank, one RCCL gather per frame; 1: virtual ranks (one context renders the ranks one after the other);
    // 2: one context per rank, all on device 0, rendered concurrently, tiles exchanged by device copies (PT_COMM_COPY_EXCHANGE)
    MultiRenderer(uint32_t n_ranks, uint32_t width, uint32_t height, int mode = 0) : n_(n_ranks), virtual_(mode == 1), one_device_(mode == 2)
    {
        const uint32_t n_ctx = virtual_ ? 1u : n_;
        for (uint32_t i = 0; i < n_ctx; ++i) r_.emplace_back(new Renderer(width, height));
        Params = r_[0]->Params;
    }
    MultiRenderer(const MultiRenderer &) = delete;
    ~MultiRenderer() { Dispose(); }
    void Init()
    {
        for (size_t i = 0; i < r_.size(); ++i) r_[i]->Init(one_device_ ? 0 : (int)i);
    }
    void LoadSyntheticScene(uint32_t kind, uint32_t detail = 0, uint32_t seed = 0x5EED0001u, uint32_t bvh_width = 0)
    {
        for (auto &r : r_) r->LoadSyntheticScene(kind, detail, seed, bvh_width); // replicated scene
        Params.mode = PT_PATH_TRACE;
        std::vector<pt_context *> ctxs(n_);
        for (uint32_t i = 0; i < n_; ++i) ctxs[i] = r_[virtual_ ? 0 : i]->Context();
        if (comm_) pt_comm_destroy(comm_);
        comm_ = nullptr;
        check(pt_comm_create(ctxs.data(), n_, 0, one_device_ ? PT_COMM_COPY_EXCHANGE : 0u, &comm_));
    }
    void Render(float)
    {
        std::vector<const pt_scene *> scenes(n_);
        for (uint32_t i = 0; i < n_; ++i) scenes[i] = r_[virtual_ ? 0 : i]->Scene();
        LastStats.assign(n_, pt_stats{});
        check(pt_comm_render(comm_, scenes.data(), &Params, LastStats.data()), r_[0]->Context());
        r_[0]->Params = Params;
    }
    Renderer &Root() { return *r_[0]; }
    void Dispose()
    {
        if (comm_) pt_comm_destroy(comm_);
        comm_ = nullptr;
        r_.clear();
    }

private:
    uint32_t n_;
    bool virtual_, one_device_;
    std::vector<std::unique_ptr<Renderer>> r_;
    pt_comm *comm_ = nullptr;
};

} // namespace ptrt_host
