// pt_device.h — gfx950 device functions of the path tracer: RNG, camera, intersection, BSDFs.
// Implements docs/SPEC.md §0-§5 op for op (explicit __builtin_fmaf, -ffp-contract=off, IEEE div/sqrt).
// The only reference-derived piece is ref_sphere_pixel(): RayTracing/Assets/Shaders/Source/Ray/Test.hlsl:1-40
// in the op order of the committed Test.spirv (a1-a4 of SURVEY.md §8a).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define PT_DEV __device__ __forceinline__
#define PT_MISS 0xFFFFFFFFu
#define PT_BVH_EMPTY 0x7fffffff

namespace ptd {

struct V3 { float x, y, z; };

PT_DEV float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
PT_DEV float fmin_(float a, float b) { return __builtin_fminf(a, b); }
PT_DEV float fmax_(float a, float b) { return __builtin_fmaxf(a, b); }
PT_DEV V3 v3(float x, float y, float z) { return V3{ x, y, z }; }
PT_DEV V3 operator-(V3 a, V3 b) { return V3{ a.x - b.x, a.y - b.y, a.z - b.z }; }
PT_DEV V3 neg(V3 a) { return V3{ -a.x, -a.y, -a.z }; }
PT_DEV float dot(V3 a, V3 b) { return fma_(a.z, b.z, fma_(a.y, b.y, a.x * b.x)); }
PT_DEV V3 cross(V3 a, V3 b)
{
    return V3{ fma_(a.y, b.z, -(a.z * b.y)), fma_(a.z, b.x, -(a.x * b.z)), fma_(a.x, b.y, -(a.y * b.x)) };
}
PT_DEV V3 normalize(V3 v)
{
    float s = 1.0f / __builtin_sqrtf(dot(v, v));
    return V3{ v.x * s, v.y * s, v.z * s };
}
PT_DEV V3 madd(float t, V3 d, V3 o) { return V3{ fma_(t, d.x, o.x), fma_(t, d.y, o.y), fma_(t, d.z, o.z) }; }
PT_DEV V3 xyz(float4 q) { return V3{ q.x, q.y, q.z }; }

// ---------------------------------------------------------------- SPEC §2 RNG
PT_DEV uint32_t pcg(uint32_t x)
{
    uint32_t s = x * 747796405u + 2891336453u;
    uint32_t w = ((s >> ((s >> 28) + 4u)) ^ s) * 277803737u;
    return (w >> 22) ^ w;
}
PT_DEV uint32_t path_key(uint32_t seed_hashed, uint32_t pixel, uint32_t sample) { return pcg(pcg(seed_hashed + pixel) + sample); }
PT_DEV float u01(uint32_t key, uint32_t dim) { return (float)(pcg(key + dim * 0x9E3779B9u) >> 8) * 5.9604644775390625e-8f; }

PT_DEV void sincos2pi(float u, float &sn, float &cs)
{
    float x = u * 4.0f;
    int q = (int)x;
    float f = x - (float)q;
    float a = (f - 0.5f) * 1.57079637f;
    float a2 = a * a;
    float s = a * fma_(a2, fma_(a2, fma_(a2, fma_(a2, 2.75573192e-6f, -1.98412698e-4f), 8.33333377e-3f), -1.66666672e-1f), 1.0f);
    float c = fma_(a2, fma_(a2, fma_(a2, fma_(a2, 2.48015876e-5f, -1.38888892e-3f), 4.16666679e-2f), -0.5f), 1.0f);
    float S = (s + c) * 0.707106769f, C = (c - s) * 0.707106769f;
    // quadrant rotation, branch-free: q&1 swaps, sign flips by q
    bool odd = q & 1;
    float ss = odd ? C : S, cc = odd ? S : C;
    sn = (q & 2) ? -ss : ss;
    cs = (((q + 1) & 2)) ? -cc : cc;
}

// ---------------------------------------------------------------- SPEC §3 camera
struct Camera { float origin[3], forward[3], right[3], up[3]; float scale, cx, cy; uint32_t jitter; };

PT_DEV void camera_ray(const Camera &c, uint32_t x, uint32_t y, uint32_t key, V3 &o, V3 &d)
{
    float jx = 0.5f, jy = 0.5f;
    if (c.jitter) { jx = u01(key, 0); jy = u01(key, 1); }
    float sx = ((float)x + jx) * c.scale - c.cx;
    float sy = ((float)y + jy) * c.scale - c.cy;
    V3 v = V3{ fma_(sy, c.up[0], fma_(sx, c.right[0], c.forward[0])),
               fma_(sy, c.up[1], fma_(sx, c.right[1], c.forward[1])),
               fma_(sy, c.up[2], fma_(sx, c.right[2], c.forward[2])) };
    d = normalize(v);
    o = V3{ c.origin[0], c.origin[1], c.origin[2] };
}

// ---------------------------------------------------------------- SPEC §4 intersection
struct Hit { float t; uint32_t id; uint32_t ref; }; // id = original primitive id (tie-break); ref = blob tri index or n_tris + sphere index

PT_DEV void tri_test(float4 r0, float4 r1, float4 r2, uint32_t blob_index, V3 o, V3 d, Hit &h)
{
    V3 v0 = xyz(r0), e1 = xyz(r1), e2 = xyz(r2);
    uint32_t id = __float_as_uint(r0.w);
    V3 p = cross(d, e2);
    float det = dot(e1, p);
    float inv_det = 1.0f / det;
    V3 tv = o - v0;
    float u = dot(tv, p) * inv_det;
    V3 q = cross(tv, e1);
    float v = dot(d, q) * inv_det;
    float t = dot(e2, q) * inv_det;
    bool ok = (det != 0.0f) && (u >= 0.0f) && (u <= 1.0f) && (v >= 0.0f) && (u + v <= 1.0f) && (t > 0.0f)
              && (t < h.t || (t == h.t && id < h.id));
    if (ok) { h.t = t; h.id = id; h.ref = blob_index; }
}

PT_DEV void sphere_test(float4 s, uint32_t id, V3 o, V3 d, Hit &h)
{
    V3 oc = o - xyz(s);
    float b = dot(oc, d);
    float cc = dot(oc, oc) - s.w * s.w;
    float disc = fma_(b, b, -cc);
    if (!(disc > 0.0f)) return;
    float sq = __builtin_sqrtf(disc);
    float t0 = -b - sq, t1 = -b + sq;
    float t = (t0 > 0.0f) ? t0 : t1;
    if (!(t > 0.0f)) return;
    if (t < h.t || (t == h.t && id < h.id)) { h.t = t; h.id = id; h.ref = id; }
}

struct RaySetup { V3 inv, noi; };
PT_DEV float safe_inv(float dk)
{
    float c = (__builtin_fabsf(dk) < 1e-20f) ? __builtin_copysignf(1e-20f, dk) : dk;
    return 1.0f / c;
}
PT_DEV RaySetup ray_setup(V3 o, V3 d)
{
    RaySetup r;
    r.inv = V3{ safe_inv(d.x), safe_inv(d.y), safe_inv(d.z) };
    r.noi = V3{ -(o.x * r.inv.x), -(o.y * r.inv.y), -(o.z * r.inv.z) };
    return r;
}
// slab test of one child slot; returns hit flag and tn
PT_DEV bool box_test(float4 lo, float4 hi, const RaySetup &rs, float t_best, float &tn)
{
    float tax = fma_(lo.x, rs.inv.x, rs.noi.x), tbx = fma_(hi.x, rs.inv.x, rs.noi.x);
    float tay = fma_(lo.y, rs.inv.y, rs.noi.y), tby = fma_(hi.y, rs.inv.y, rs.noi.y);
    float taz = fma_(lo.z, rs.inv.z, rs.noi.z), tbz = fma_(hi.z, rs.inv.z, rs.noi.z);
    tn = fmax_(fmax_(fmin_(tax, tbx), fmin_(tay, tby)), fmax_(fmin_(taz, tbz), 0.0f));
    float tf = fmin_(fmin_(fmax_(tax, tbx), fmax_(tay, tby)), fmin_(fmax_(taz, tbz), t_best)) * 1.0000004f;
    return tn <= tf;
}

// ---------------------------------------------------------------- SPEC §5 BSDFs
struct Material { uint32_t kind; float albedo[3]; float emission[3]; float roughness; float ior; };

PT_DEV void basis(V3 n, V3 &tx, V3 &ty)
{
    float sg = __builtin_copysignf(1.0f, n.z);
    float a = -1.0f / (sg + n.z);
    float b = n.x * n.y * a;
    tx = V3{ fma_(sg * n.x, n.x * a, 1.0f), sg * b, -(sg * n.x) };
    ty = V3{ b, fma_(n.y, n.y * a, sg), -n.y };
}
PT_DEV V3 to_world(V3 l, V3 tx, V3 ty, V3 n)
{
    V3 w = V3{ fma_(l.z, n.x, fma_(l.y, ty.x, l.x * tx.x)),
               fma_(l.z, n.y, fma_(l.y, ty.y, l.x * tx.y)),
               fma_(l.z, n.z, fma_(l.y, ty.z, l.x * tx.z)) };
    return normalize(w);
}
PT_DEV V3 schlick(V3 alb, float cosF)
{
    float m = 1.0f - cosF, m2 = m * m, m5 = m2 * m2 * m;
    return V3{ fma_(1.0f - alb.x, m5, alb.x), fma_(1.0f - alb.y, m5, alb.y), fma_(1.0f - alb.z, m5, alb.z) };
}
PT_DEV V3 reflect_about(V3 d, V3 n, float cosi)
{
    float c2 = 2.0f * cosi;
    return normalize(V3{ fma_(c2, n.x, d.x), fma_(c2, n.y, d.y), fma_(c2, n.z, d.z) });
}
PT_DEV float clamp01(float x) { return fmin_(fmax_(x, 0.0f), 1.0f); }

// What a BSDF sampler returns (by value: outputs through references ended up in a scratch array once the three samplers met
// in one branch of shade_one): direction, throughput weight, side of the surface the next ray leaves from, validity.
struct BsdfSample { V3 wi, W; float side; bool ok; };

PT_DEV BsdfSample sample_lambert(V3 alb, V3 n, float u1, float u2)
{
    V3 wi;
    V3 tx, ty;
    basis(n, tx, ty);
    float r = __builtin_sqrtf(u1), sn, cs;
    sincos2pi(u2, sn, cs);
    V3 l = V3{ r * cs, r * sn, __builtin_sqrtf(fmax_(0.0f, 1.0f - u1)) };
    wi = to_world(l, tx, ty, n);
    return BsdfSample{ wi, alb, 1.0f, true };
}

PT_DEV BsdfSample sample_metal(V3 alb, float al, V3 d, V3 n, float u1, float u2)
{
    float cosi = clamp01(-dot(d, n));
    if (al == 0.0f) return BsdfSample{ reflect_about(d, n, cosi), schlick(alb, cosi), 1.0f, true };
    V3 tx, ty;
    basis(n, tx, ty);
    V3 wo = neg(d);
    V3 wl = V3{ dot(wo, tx), dot(wo, ty), dot(wo, n) };
    V3 Vh = normalize(V3{ al * wl.x, al * wl.y, wl.z });
    float lensq = fma_(Vh.y, Vh.y, Vh.x * Vh.x);
    V3 T1 = V3{ 1.0f, 0.0f, 0.0f };
    if (lensq > 0.0f) { float il = 1.0f / __builtin_sqrtf(lensq); T1 = V3{ -Vh.y * il, Vh.x * il, 0.0f }; }
    V3 T2 = cross(Vh, T1);
    float r = __builtin_sqrtf(u1), sn, cs;
    sincos2pi(u2, sn, cs);
    float t1 = r * cs, t2 = r * sn, s5 = 0.5f * (1.0f + Vh.z);
    t2 = fma_(s5, t2, (1.0f - s5) * __builtin_sqrtf(fmax_(0.0f, 1.0f - t1 * t1)));
    float nz = __builtin_sqrtf(fmax_(0.0f, 1.0f - t1 * t1 - t2 * t2));
    V3 Nh = V3{ fma_(nz, Vh.x, fma_(t2, T2.x, t1 * T1.x)),
                fma_(nz, Vh.y, fma_(t2, T2.y, t1 * T1.y)),
                fma_(nz, Vh.z, fma_(t2, T2.z, t1 * T1.z)) };
    V3 hh = normalize(V3{ al * Nh.x, al * Nh.y, fmax_(0.0f, Nh.z) });
    float dh = dot(wl, hh);
    float cosF = clamp01(dh);
    float c2 = 2.0f * dh;
    V3 wil = V3{ fma_(c2, hh.x, -wl.x), fma_(c2, hh.y, -wl.y), fma_(c2, hh.z, -wl.z) };
    if (!(wil.z > 0.0f)) return BsdfSample{ d, alb, 1.0f, false };
    float wz = wil.z;
    float G1 = 2.0f * wz / (wz + __builtin_sqrtf(fma_(al * al, 1.0f - wz * wz, wz * wz)));
    V3 F = schlick(alb, cosF);
    return BsdfSample{ to_world(wil, tx, ty, n), V3{ F.x * G1, F.y * G1, F.z * G1 }, 1.0f, true };
}

PT_DEV BsdfSample sample_dielectric(V3 alb, float ior, V3 d, V3 n, bool front, float u3)
{
    V3 wi;
    float side;
    float cosi = clamp01(-dot(d, n));
    float eta = front ? 1.0f / ior : ior;
    float sin2t = eta * eta * (1.0f - cosi * cosi);
    bool refl = true;
    float cost = 0.0f;
    if (!(sin2t >= 1.0f)) {
        cost = __builtin_sqrtf(1.0f - sin2t);
        float ni = front ? 1.0f : ior, nt = front ? ior : 1.0f;
        float rp = (nt * cosi - ni * cost) / (nt * cosi + ni * cost);
        float rs = (ni * cosi - nt * cost) / (ni * cosi + nt * cost);
        float F = 0.5f * (rp * rp + rs * rs);
        refl = (u3 < F);
    }
    side = 1.0f;
    if (refl) wi = reflect_about(d, n, cosi);
    else {
        float k = fma_(eta, cosi, -cost);
        wi = normalize(V3{ fma_(k, n.x, eta * d.x), fma_(k, n.y, eta * d.y), fma_(k, n.z, eta * d.z) });
        side = -1.0f;
    }
    return BsdfSample{ wi, alb, side, true };
}

// ---------------------------------------------------------------- SPEC §1 the reference's kernel
PT_DEV uint32_t unorm8(float c)
{
    if (!(c > 0.0f)) return 0u;
    if (c >= 1.0f) return 255u;
    return (uint32_t)__builtin_floorf(c * 255.0f + 0.5f);
}
// Test.hlsl:6-37 for one pixel (CSMain body without the store)
PT_DEV float4 ref_sphere_pixel(uint32_t x, uint32_t y)
{
    const float inv1080 = __uint_as_float(0x3a72b9d6u);      // Test.hlsl:6-7, OpConstant %26
    float uvx = ((float)x * inv1080) * 2.0f - 1.0f;          // Test.hlsl:7
    float uvy = ((float)y * inv1080) * 2.0f - 1.0f;
    float vz = -1.0f;                                        // Test.hlsl:10
    float l = __builtin_sqrtf(uvx * uvx + uvy * uvy + vz * vz);
    float dx = uvx / l, dy = uvy / l, dz = vz / l;
    float a = dx * dx + dy * dy + dz * dz;                   // Test.hlsl:17
    float b = 2.0f * dz;                                     // Test.hlsl:16,18
    float disc = fma_(b, b, a * -3.0f);                      // Test.hlsl:19,21
    if (disc > 0.0f) {                                       // Test.hlsl:24
        float t = fma_(dz, -2.0f, -__builtin_sqrtf(disc)) / (2.0f * a); // Test.hlsl:27
        float px = 0.0f + dx * t, py = 0.0f + dy * t, pz = 1.0f + dz * t; // Test.hlsl:28
        float pl = __builtin_sqrtf(px * px + py * py + pz * pz);
        float nx = px / pl, ny = py / pl, nz = pz / pl;      // Test.hlsl:29
        return make_float4(nx * 0.5f + 0.5f, ny * 0.5f + 0.5f, nz * 0.5f + 0.5f, 1.0f); // Test.hlsl:31
    }
    return make_float4(uvx, uvy, 0.0f, 1.0f);                // Test.hlsl:36
}

} // namespace ptd
