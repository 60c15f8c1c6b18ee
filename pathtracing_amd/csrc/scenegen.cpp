// scenegen.cpp — deterministic synthetic scenes C1..C5 of BASELINE.md §3 (host only).
// The reference's whole "scene" is four shader literals (Test.hlsl:6,8,12,13); these generators stand
// where a host's scene set-up would (App.cs in north_star's wording).
#include "../../include/ptrt.h"
#include <cmath>
#include <cstring>
#include <vector>

namespace {

struct Gen {
    std::vector<float> verts; std::vector<uint32_t> tmat;
    std::vector<float> sph; std::vector<uint32_t> smat;
    std::vector<pt_material> mats;
    void tri(const float a[3], const float b[3], const float c[3], uint32_t m)
    {
        verts.insert(verts.end(), a, a + 3); verts.insert(verts.end(), b, b + 3); verts.insert(verts.end(), c, c + 3);
        tmat.push_back(m);
    }
    // quad p(s,t) = o + s*u + t*v tessellated k x k, two triangles per cell
    void quad(const float o[3], const float u[3], const float v[3], uint32_t k, uint32_t m)
    {
        const float ik = 1.0f / (float)k;
        for (uint32_t j = 0; j < k; ++j)
            for (uint32_t i = 0; i < k; ++i) {
                float p[4][3];
                for (int c = 0; c < 4; ++c) {
                    const float s = (float)(i + (c & 1)) * ik, t = (float)(j + (c >> 1)) * ik;
                    for (int a = 0; a < 3; ++a) p[c][a] = o[a] + s * u[a] + t * v[a];
                }
                tri(p[0], p[1], p[3], m);
                tri(p[0], p[3], p[2], m);
            }
    }
    uint32_t mat(uint32_t kind, float r, float g, float b, float er = 0.f, float eg = 0.f, float eb = 0.f, float rough = 0.f, float ior = 1.f)
    {
        pt_material m; std::memset(&m, 0, sizeof m);
        m.kind = kind; m.albedo[0] = r; m.albedo[1] = g; m.albedo[2] = b;
        m.emission[0] = er; m.emission[1] = eg; m.emission[2] = eb; m.roughness = rough; m.ior = ior;
        mats.push_back(m);
        return (uint32_t)mats.size() - 1;
    }
    void sphere(float x, float y, float z, float r, uint32_t m) { sph.insert(sph.end(), { x, y, z, r }); smat.push_back(m); }
};

uint32_t pcg(uint32_t x)
{
    uint32_t s = x * 747796405u + 2891336453u;
    uint32_t w = ((s >> ((s >> 28) + 4u)) ^ s) * 277803737u;
    return (w >> 22) ^ w;
}
struct Rng {
    uint32_t s;
    float next() { s = pcg(s + 0x9E3779B9u); return (float)(s >> 8) * (1.0f / 16777216.0f); }
};

void cornell(Gen &g, uint32_t k, bool glass)
{
    const uint32_t white = g.mat(PT_LAMBERT, 0.73f, 0.73f, 0.73f), red = g.mat(PT_LAMBERT, 0.65f, 0.05f, 0.05f),
                   green = g.mat(PT_LAMBERT, 0.12f, 0.45f, 0.15f), light = g.mat(PT_LAMBERT, 0.f, 0.f, 0.f, 15.f, 15.f, 15.f);
    // box [-1,1]^3, open towards +z
    const float fo[3] = { -1, -1, 1 }, fu[3] = { 2, 0, 0 }, fv[3] = { 0, 0, -2 };  g.quad(fo, fu, fv, k, white); // floor  y=-1
    const float co[3] = { -1, 1, -1 }, cu[3] = { 2, 0, 0 }, cv[3] = { 0, 0, 2 };   g.quad(co, cu, cv, k, white); // ceiling y=+1
    const float bo[3] = { -1, -1, -1 }, bu[3] = { 2, 0, 0 }, bv[3] = { 0, 2, 0 };  g.quad(bo, bu, bv, k, white); // back   z=-1
    const float lo[3] = { -1, -1, 1 }, lu[3] = { 0, 0, -2 }, lv[3] = { 0, 2, 0 };  g.quad(lo, lu, lv, k, red);   // left   x=-1
    const float ro[3] = { 1, -1, -1 }, ru[3] = { 0, 0, 2 }, rv[3] = { 0, 2, 0 };   g.quad(ro, ru, rv, k, green); // right  x=+1
    const float eo[3] = { -0.3f, 0.998f, -0.3f }, eu[3] = { 0.6f, 0, 0 }, ev[3] = { 0, 0, 0.6f }; g.quad(eo, eu, ev, 1, light);
    if (glass) {
        const uint32_t gl = g.mat(PT_DIELECTRIC, 1.f, 1.f, 1.f, 0, 0, 0, 0.f, 1.5f);
        const uint32_t me = g.mat(PT_METAL, 0.95f, 0.78f, 0.35f, 0, 0, 0, 0.15f, 1.f);
        const uint32_t mi = g.mat(PT_METAL, 0.9f, 0.9f, 0.9f, 0, 0, 0, 0.0f, 1.f);
        const uint32_t bl = g.mat(PT_LAMBERT, 0.2f, 0.3f, 0.7f);
        g.sphere(-0.45f, -0.65f, 0.25f, 0.35f, gl);
        g.sphere(0.5f, -0.65f, -0.3f, 0.35f, me);
        g.sphere(-0.35f, -0.75f, -0.5f, 0.25f, bl);
        g.sphere(0.1f, -0.8f, 0.55f, 0.2f, mi);
    } else {
        const uint32_t bl = g.mat(PT_LAMBERT, 0.2f, 0.3f, 0.7f), ye = g.mat(PT_LAMBERT, 0.7f, 0.6f, 0.2f);
        g.sphere(-0.45f, -0.65f, 0.25f, 0.35f, white);
        g.sphere(0.5f, -0.65f, -0.3f, 0.35f, bl);
        g.sphere(-0.35f, -0.75f, -0.5f, 0.25f, ye);
        g.sphere(0.1f, -0.8f, 0.55f, 0.2f, white);
    }
}

void soup(Gen &g, uint32_t n, uint32_t seed)
{
    const uint32_t grey = g.mat(PT_LAMBERT, 0.7f, 0.7f, 0.7f);
    Rng r{ pcg(seed) };
    for (uint32_t i = 0; i < n; ++i) {
        float c[3], p[3][3];
        for (int a = 0; a < 3; ++a) c[a] = r.next() * 2.0f - 1.0f;
        for (int v = 0; v < 3; ++v)
            for (int a = 0; a < 3; ++a) p[v][a] = c[a] + (r.next() * 2.0f - 1.0f) * 0.01f;
        g.tri(p[0], p[1], p[2], grey);
    }
}

} // namespace

extern "C" pt_status pt_scenegen(uint32_t kind, uint32_t detail, uint32_t seed, uint32_t width, uint32_t height,
                                 pt_scene_counts *counts, float *verts9, uint32_t *tri_mat, float *spheres, uint32_t *sph_mat,
                                 pt_material *mats, pt_camera *cam, float sky[3])
{
    if (!counts || width == 0 || height == 0) return PT_ERR_INVALID_ARGUMENT;
    Gen g;
    float sk[3] = { 0.f, 0.f, 0.f };
    switch (kind) {
    case PT_SCENE_CORNELL: cornell(g, 1, false); break;
    case PT_SCENE_CORNELL_GLASS: cornell(g, 1, true); break;
    case PT_SCENE_CORNELL_TESS: {
        uint32_t k = (uint32_t)std::floor(std::sqrt((double)(detail ? detail : 1u) / 10.0));
        cornell(g, k ? k : 1u, false);
        break;
    }
    case PT_SCENE_TRIANGLE_SOUP:
        soup(g, detail ? detail : 1u, seed);
        sk[0] = sk[1] = sk[2] = 1.0f;
        break;
    default: return PT_ERR_INVALID_ARGUMENT;
    }
    counts->n_tris = g.tmat.size(); counts->n_spheres = g.smat.size(); counts->n_mats = g.mats.size();
    if (verts9) std::memcpy(verts9, g.verts.data(), g.verts.size() * sizeof(float));
    if (tri_mat) std::memcpy(tri_mat, g.tmat.data(), g.tmat.size() * sizeof(uint32_t));
    if (spheres && !g.sph.empty()) std::memcpy(spheres, g.sph.data(), g.sph.size() * sizeof(float));
    if (sph_mat && !g.smat.empty()) std::memcpy(sph_mat, g.smat.data(), g.smat.size() * sizeof(uint32_t));
    if (mats) std::memcpy(mats, g.mats.data(), g.mats.size() * sizeof(pt_material));
    if (sky) { sky[0] = sk[0]; sky[1] = sk[1]; sky[2] = sk[2]; }
    if (cam) {
        std::memset(cam, 0, sizeof *cam);
        const float th = 0.40f; // tan(fov_y/2); image row 0 is the top of the picture
        cam->origin[2] = 3.6f;
        cam->forward[2] = -1.0f;
        cam->right[0] = th;
        cam->up[1] = -th;
        cam->scale = 2.0f / (float)height;
        cam->cx = (float)width / (float)height;
        cam->cy = 1.0f;
        cam->jitter = 1u;
    }
    return PT_OK;
}
