// ptrt_cli — headless counterpart of Program.cs / App.Run (RayTracing/Program.cs:1-9, App.cs:15-21):
// build a renderer, render N frames, write the image the reference would have shown in its window.
//   ptrt_cli [--scene reference|cornell|glass|soup|tess] [--detail N] [--size WxH] [--spp N] [--depth N]
//            [--frames N] [--ppm out.ppm] [--pfm out.pfm] [--gpus N] [--virtual 0|1|2]
// --gpus N: the frame's tiles over N devices of this node, one RCCL gather per frame (pt_comm); with --virtual 1 the N ranks are
// rendered one after the other on device 0 (rehearsal of the partition on a single GPU); with --virtual 2 every rank has its own
// context on device 0, the ranks render concurrently (one host thread each) and exchange their tiles by device copies.
#include "ptrt_host.hpp"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>

int main(int argc, char **argv)
{
    std::string scene = "reference", ppm = "frame.ppm", pfm;
    uint32_t w = 1920, h = 1080, detail = 0, spp = 64, depth = 8, frames = 1, gpus = 1, virt = 0;
    for (int i = 1; i + 1 < argc; i += 2) {
        const std::string a = argv[i];
        if (a == "--scene") scene = argv[i + 1];
        else if (a == "--detail") detail = (uint32_t)std::strtoul(argv[i + 1], nullptr, 0);
        else if (a == "--size") std::sscanf(argv[i + 1], "%ux%u", &w, &h);
        else if (a == "--spp") spp = (uint32_t)std::atoi(argv[i + 1]);
        else if (a == "--depth") depth = (uint32_t)std::atoi(argv[i + 1]);
        else if (a == "--frames") frames = (uint32_t)std::atoi(argv[i + 1]);
        else if (a == "--gpus") gpus = (uint32_t)std::atoi(argv[i + 1]);
        else if (a == "--virtual") virt = (uint32_t)std::atoi(argv[i + 1]);
        else if (a == "--ppm") ppm = argv[i + 1];
        else if (a == "--pfm") pfm = argv[i + 1];
        else { std::fprintf(stderr, "unknown option %s\n", a.c_str()); return 2; }
    }
    try {
        std::unique_ptr<ptrt_host::Renderer> single;
        std::unique_ptr<ptrt_host::MultiRenderer> multi;
        const uint32_t kind = scene == "cornell" ? PT_SCENE_CORNELL : scene == "glass" ? PT_SCENE_CORNELL_GLASS
                            : scene == "soup" ? PT_SCENE_TRIANGLE_SOUP : PT_SCENE_CORNELL_TESS;
        if (gpus > 1 && scene != "reference") {
            multi.reset(new ptrt_host::MultiRenderer(gpus, w, h, (int)virt));
            multi->Init();
            multi->LoadSyntheticScene(kind, detail);
            multi->Params.spp = spp; multi->Params.max_depth = depth;
            for (uint32_t f = 0; f < frames; ++f) multi->Render(0.f);
            unsigned long long rays = 0; double ms = 0;
            for (const pt_stats &s : multi->LastStats) { rays += s.rays; ms = s.gpu_ms > ms ? s.gpu_ms : ms; }
            std::printf("%llu rays over %u ranks, slowest rank %.3f ms\n", rays, gpus, ms);
        } else {
            single.reset(new ptrt_host::Renderer(w, h));
            single->Init();
            if (scene != "reference") {
                single->LoadSyntheticScene(kind, detail);
                single->Params.spp = spp; single->Params.max_depth = depth;
            }
            for (uint32_t f = 0; f < frames; ++f) single->Render(0.f);
            const pt_stats &s = single->LastStats;
            std::printf("%llu rays, %.3f ms, %.1f Mrays/s\n", (unsigned long long)s.rays, s.gpu_ms, s.rays / s.gpu_ms / 1e3);
        }
        ptrt_host::Renderer &r = multi ? multi->Root() : *single;
        if (!ppm.empty()) { // 8-bit image = the reference's R8G8B8A8Unorm storage image (Renderer.cs:124)
            const auto px = r.ReadFramebufferRgba8();
            FILE *f = std::fopen(ppm.c_str(), "wb");
            if (!f) throw std::runtime_error("cannot open " + ppm);
            std::fprintf(f, "P6\n%u %u\n255\n", w, h);
            for (size_t i = 0; i < px.size(); i += 4) std::fwrite(&px[i], 1, 3, f);
            std::fclose(f);
        }
        if (!pfm.empty()) { // linear float radiance, bottom-up rows, little endian
            const auto px = r.ReadFramebuffer();
            FILE *f = std::fopen(pfm.c_str(), "wb");
            if (!f) throw std::runtime_error("cannot open " + pfm);
            std::fprintf(f, "PF\n%u %u\n-1.0\n", w, h);
            for (uint32_t y = h; y-- > 0;)
                for (uint32_t x = 0; x < w; ++x) std::fwrite(&px[((size_t)y * w + x) * 4], 4, 3, f);
            std::fclose(f);
        }
    } catch (const std::exception &e) {
        std::fprintf(stderr, "ptrt_cli: %s\n", e.what());
        return 1;
    }
    return 0;
}
