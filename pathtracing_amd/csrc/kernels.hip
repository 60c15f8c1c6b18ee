// kernels.hip — gfx950 wavefront kernels of libptrt (docs/SPEC.md; DESIGN.md "Kernels").
//
//   k_reference_sphere : the reference's CSMain (Test.hlsl:1-40; dispatch Renderer.cs:1020), one lane = one pixel
//   k_generate         : camera rays for sample 0 of every owned pixel, fills the extend queues
//   k_extend<N>        : ray -> closest hit. BVH-N traversal, stack in LDS ([level][lane], conflict-free),
//                        spheres by scalar loads, hits bucketed by material kind with one atomic per wave
//   k_shade            : emission, BSDF sample, Russian roulette, accumulate, in-place regeneration of the
//                        next sample of the same pixel; survivors are compacted into the next extend queue
//   k_assemble         : tile-major slots (of 1..R ranks) -> row-major float4 + RGBA8 frame
//
// One slot per owned pixel, at most one live path per slot => framebuffer RMW without atomics and a
// per-pixel summation order identical to the oracle's `for s in 0..spp`.
// Queues are split into kShards static shards (ptrt_internal.h): blockIdx.y = shard, and every lane of a block only
// ever sees slots of its own shard, so a wavefront's push goes to exactly one per-shard counter.
#include "ptrt_internal.h"
#include "pt_device.h"

using namespace ptd;

namespace ptrt {

PT_DEV uint32_t lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

// Append `value` of every lane with `pred` to queue: one ballot, one atomic per wavefront, mbcnt prefix.
// Must be reached by all live lanes of the wave in uniform control flow.
PT_DEV void wave_push(uint32_t *counter, uint32_t *queue, bool pred, uint32_t value)
{
    const uint64_t m = __ballot(pred);
    if (m == 0) return;
    const uint32_t prefix = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
    const int leader = __ffsll((long long)m) - 1;
    uint32_t base = 0;
    if ((int)lane_id() == leader) base = atomicAdd(counter, (uint32_t)__popcll(m));
    base = __shfl(base, leader, 64);
    if (pred) queue[base + prefix] = value;
}

PT_DEV bool slot_pixel(uint32_t slot, const FrameParams &fp, uint32_t &x, uint32_t &y)
{
    const uint32_t tl = slot >> (2 * kTileShift), inner = slot & (kTilePixels - 1);
    const uint32_t tile = fp.rank + fp.nranks * tl;
    if (tile >= fp.n_tiles) return false;
    const uint32_t tx = tile % fp.tiles_x, ty = tile / fp.tiles_x;
    const uint32_t blk = inner >> 6, ln = inner & 63u;
    x = (tx << kTileShift) + ((blk & 7u) << 3) + (ln & 7u);
    y = (ty << kTileShift) + ((blk >> 3) << 3) + (ln >> 3);
    return x < fp.width && y < fp.height;
}

PT_DEV void camera_ray_of(const pt_camera &c, uint32_t x, uint32_t y, uint32_t key, V3 &o, V3 &d)
{
    Camera k;
    k.origin[0] = c.origin[0]; k.origin[1] = c.origin[1]; k.origin[2] = c.origin[2];
    k.forward[0] = c.forward[0]; k.forward[1] = c.forward[1]; k.forward[2] = c.forward[2];
    k.right[0] = c.right[0]; k.right[1] = c.right[1]; k.right[2] = c.right[2];
    k.up[0] = c.up[0]; k.up[1] = c.up[1]; k.up[2] = c.up[2];
    k.scale = c.scale; k.cx = c.cx; k.cy = c.cy; k.jitter = c.jitter;
    camera_ray(k, x, y, key, o, d);
}

// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock) k_reference_sphere(uint32_t w, uint32_t h, float4 *out_f, uint32_t *out8)
{
    const uint32_t x = blockIdx.x * 64u + threadIdx.x, y = blockIdx.y * 4u + threadIdx.y;
    if (x >= w || y >= h) return; // the reference's rows 1080..1087 are out-of-bounds stores: dropped
    const float4 c = ref_sphere_pixel(x, y);
    const size_t i = (size_t)y * w + x; // Test.hlsl:39 TestImage[id.xy] = color
    out_f[i] = c;
    out8[i] = unorm8(c.x) | (unorm8(c.y) << 8) | (unorm8(c.z) << 16) | (unorm8(c.w) << 24);
}

// ------------------------------------------------------------------------------------------------
// grid (shard_cap/256, kShards): entry j of shard s is slot ((j>>8)*kShards + s)*256 + (j&255)
__global__ void __launch_bounds__(kBlock) k_generate(DeviceScene sc, PathState ps, FrameParams fp)
{
    const uint32_t shard = blockIdx.y;
    const uint32_t j = blockIdx.x * kBlock + threadIdx.x;
    const uint32_t slot = (((j >> 8) * kShards + shard) << 8) | (j & 255u);
    uint32_t x = 0, y = 0;
    const bool in_range = j < ps.shard_cap && slot < ps.n_slots;
    const bool valid = in_range && slot_pixel(slot, fp, x, y);
    if (in_range) ps.acc[slot] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (valid) {
        const uint32_t key = path_key(fp.seed_hashed, y * fp.width + x, fp.sample_offset);
        V3 o, d;
        camera_ray_of(sc.cam, x, y, key, o, d);
        ps.ray_o[slot] = make_float4(o.x, o.y, o.z, 0.f);
        ps.ray_d[slot] = make_float4(d.x, d.y, d.z, 0.f);
        ps.thr[slot] = make_float4(1.f, 1.f, 1.f, __uint_as_float(key));
        ps.sd[slot] = 0u;
    }
    wave_push(&ps.counters[cnt_ext_index(0, shard)], ps.q_ext[0] + (size_t)shard * ps.shard_cap, valid, slot);
}

// ------------------------------------------------------------------------------------------------
template <int N, bool COUNT>
__global__ void __launch_bounds__(kBlock) k_extend(DeviceScene sc, PathState ps, uint32_t parity)
{
    __shared__ int32_t s_stack[kStackLds * kBlock];
    const uint32_t shard = blockIdx.y;
    const uint32_t n = ps.counters[cnt_ext_index(parity, shard)];
    const uint32_t tid = threadIdx.x;
    const uint32_t gid = blockIdx.x * kBlock + tid;          // index inside the shard's queue
    if (gid == 0) {
        ps.counters[cnt_ext_index(parity ^ 1u, shard)] = 0u;   // next iteration's queue: filled by k_shade after us
        unsigned long long *rays = reinterpret_cast<unsigned long long *>(ps.counters + cnt_rays_index(shard));
        *rays += n;                                            // only this thread ever touches rays[shard]
    }
    if (blockIdx.x * kBlock >= n) return;
    const bool active = gid < n;
    const size_t qbase = (size_t)shard * ps.shard_cap;
    const uint32_t slot = active ? ps.q_ext[parity][qbase + gid] : 0u;
    const size_t uid = qbase + gid;                            // unique per thread of this launch

    Hit h{ __builtin_inff(), PT_MISS, PT_MISS };
    uint32_t bucket = B_MISS;
    unsigned long long c_nodes = 0, c_tris = 0, c_sph = 0;

    if (active) {
        const float4 O = ps.ray_o[slot], D = ps.ray_d[slot];
        const V3 o = xyz(O), d = xyz(D);

        for (uint32_t j = 0; j < sc.n_spheres; ++j) { // uniform index => scalar loads
            sphere_test(sc.spheres[j], sc.n_tris + j, o, d, h);
            if (COUNT) c_sph++;
        }

        const RaySetup rs = ray_setup(o, d);
        int32_t cur = sc.n_nodes ? 0 : PT_BVH_EMPTY;
        uint32_t sp = 0, steps = 0;
        const size_t ovf_stride = (size_t)kShards * ps.shard_cap;

        auto push = [&](int32_t v) {
            if (sp < kStackLds) s_stack[sp * kBlock + tid] = v;
            else {
                const uint32_t e = sp - kStackLds;
                if (e < ps.stack_ovf_entries) ps.stack_ovf[(size_t)e * ovf_stride + uid] = v;
                else { atomicOr(&ps.counters[kCntError], 1u); return; }
            }
            ++sp;
        };
        auto pop = [&]() -> int32_t {
            if (sp == 0) return PT_BVH_EMPTY;
            --sp;
            return sp < kStackLds ? s_stack[sp * kBlock + tid] : ps.stack_ovf[(size_t)(sp - kStackLds) * ovf_stride + uid];
        };

        while (cur != PT_BVH_EMPTY) {
            if (++steps > (1u << 22)) { atomicOr(&ps.counters[kCntError], 2u); break; }
            if (cur >= 0) {
                const float4 *nd = sc.nodes + (size_t)cur * (2 * N);
                float4 r[2 * N];
#pragma unroll
                for (int i = 0; i < 2 * N; ++i) r[i] = nd[i];
                if (COUNT) c_nodes++;
                uint32_t key[N];
                int32_t ref[N];
#pragma unroll
                for (int c = 0; c < N; ++c) {
                    float tn;
                    ref[c] = __float_as_int(r[2 * c].w);
                    const bool hb = box_test(r[2 * c], r[2 * c + 1], rs, h.t, tn) && ref[c] != PT_BVH_EMPTY;
                    key[c] = hb ? ((__float_as_uint(tn) & ~3u) | (uint32_t)c) : 0xFFFFFFFFu;
                }
                auto cswap = [&](int a, int b) {
                    if (key[a] > key[b]) {
                        const uint32_t tk = key[a]; key[a] = key[b]; key[b] = tk;
                        const int32_t tr = ref[a]; ref[a] = ref[b]; ref[b] = tr;
                    }
                };
                if (N == 2) { cswap(0, 1); }
                else { cswap(0, 1); cswap(2, 3); cswap(0, 2); cswap(1, 3); cswap(1, 2); }
#pragma unroll
                for (int i = N - 1; i >= 1; --i)
                    if (key[i] != 0xFFFFFFFFu) push(ref[i]); // farthest first, nearest stays in `cur`
                cur = (key[0] != 0xFFFFFFFFu) ? ref[0] : pop();
            } else {
                const uint32_t enc = (uint32_t)~cur, first = enc >> 3, cnt = (enc & 7u) + 1u;
                for (uint32_t j = 0; j < cnt; ++j) {
                    const uint32_t idx = first + j;
                    const float4 *tp = sc.tris + (size_t)idx * 3;
                    tri_test(tp[0], tp[1], tp[2], idx, o, d, h);
                    if (COUNT) c_tris++;
                }
                cur = pop();
            }
        }

        ps.hit[slot] = make_float2(h.t, __uint_as_float(h.ref));
        if (h.ref != PT_MISS) {
            const uint32_t mat = h.ref < sc.n_tris ? __float_as_uint(sc.tris[(size_t)h.ref * 3 + 1].w)
                                                   : sc.sph_mat[h.ref - sc.n_tris];
            bucket = 1u + __float_as_uint(sc.mats[(size_t)mat * 3].x);
        }
        if (COUNT) {
            atomicAdd(reinterpret_cast<unsigned long long *>(ps.counters + kCntNodes), c_nodes);
            atomicAdd(reinterpret_cast<unsigned long long *>(ps.counters + kCntTris), c_tris);
            atomicAdd(reinterpret_cast<unsigned long long *>(ps.counters + kCntSph), c_sph);
        }
    }

#pragma unroll
    for (uint32_t b = 0; b < B_COUNT; ++b)
        wave_push(&ps.counters[cnt_bucket_index(parity, b, shard)], ps.q_bucket[b] + qbase, active && bucket == b, slot);
}

// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock) k_shade(DeviceScene sc, PathState ps, FrameParams fp, uint32_t parity)
{
    const uint32_t shard = blockIdx.y;
    const uint32_t c0 = ps.counters[cnt_bucket_index(parity, 0, shard)], c1 = c0 + ps.counters[cnt_bucket_index(parity, 1, shard)],
                   c2 = c1 + ps.counters[cnt_bucket_index(parity, 2, shard)], total = c2 + ps.counters[cnt_bucket_index(parity, 3, shard)];
    const uint32_t gid = blockIdx.x * kBlock + threadIdx.x;
    if (gid < B_COUNT) // the other parity's buckets were consumed by the previous k_shade; k_extend(i+1) fills them next
        ps.counters[cnt_bucket_index(parity ^ 1u, gid, shard)] = 0u;
    if (blockIdx.x * kBlock >= total) return;
    const bool active = gid < total;
    uint32_t b = B_MISS, qi = gid;
    if (gid >= c2) { b = B_DIELECTRIC; qi = gid - c2; }
    else if (gid >= c1) { b = B_METAL; qi = gid - c1; }
    else if (gid >= c0) { b = B_LAMBERT; qi = gid - c0; }
    const size_t qbase = (size_t)shard * ps.shard_cap;
    const uint32_t slot = active ? ps.q_bucket[b][qbase + qi] : 0u;
    bool alive = false;

    if (active) {
        const float2 hr = ps.hit[slot];
        const float4 O = ps.ray_o[slot], D = ps.ray_d[slot], TK = ps.thr[slot];
        const uint32_t sdv = ps.sd[slot];
        V3 o = xyz(O), d = xyz(D), T = xyz(TK);
        uint32_t key = __float_as_uint(TK.w), sample = sdv >> 8, depth = (sdv & 255u) + 1u;
        const float t = hr.x;
        const uint32_t ref = __float_as_uint(hr.y);
        float4 A = make_float4(0.f, 0.f, 0.f, 0.f);
        bool touched = false, term = false;
        auto add = [&](V3 L) {
            if (!touched) { A = ps.acc[slot]; touched = true; }
            A.x = fma_(T.x, L.x, A.x); A.y = fma_(T.y, L.y, A.y); A.z = fma_(T.z, L.z, A.z);
        };

        if (b == B_MISS) {
            add(v3(sc.sky[0], sc.sky[1], sc.sky[2]));
            term = true;
        } else {
            const V3 P = madd(t, d, o);
            V3 ng;
            uint32_t mat;
            if (ref < sc.n_tris) {
                const float4 r1 = sc.tris[(size_t)ref * 3 + 1], r2 = sc.tris[(size_t)ref * 3 + 2];
                ng = normalize(cross(xyz(r1), xyz(r2)));
                mat = __float_as_uint(r1.w);
            } else {
                const uint32_t j = ref - sc.n_tris;
                const float4 s = sc.spheres[j];
                const float ir = 1.0f / s.w;
                ng = v3((P.x - s.x) * ir, (P.y - s.y) * ir, (P.z - s.z) * ir);
                mat = sc.sph_mat[j];
            }
            const bool front = dot(ng, d) < 0.0f;
            const V3 n = front ? ng : neg(ng);
            const float4 m0 = sc.mats[(size_t)mat * 3], m1 = sc.mats[(size_t)mat * 3 + 1], m2 = sc.mats[(size_t)mat * 3 + 2];
            const V3 alb = v3(m0.y, m0.z, m0.w), emi = xyz(m1);
            if (emi.x != 0.0f || emi.y != 0.0f || emi.z != 0.0f) add(emi);
            if (depth >= fp.max_depth) term = true;
            else {
                const uint32_t bb = depth - 1u;
                V3 wi = d, W = v3(1.f, 1.f, 1.f);
                float side = 1.0f;
                bool ok = true;
                if (b == B_LAMBERT) sample_lambert(alb, n, u01(key, 4u + 4u * bb), u01(key, 5u + 4u * bb), wi, W);
                else if (b == B_METAL) ok = sample_metal(alb, m1.w, d, n, u01(key, 4u + 4u * bb), u01(key, 5u + 4u * bb), wi, W);
                else sample_dielectric(alb, m2.x, d, n, front, u01(key, 6u + 4u * bb), wi, W, side);
                if (!ok) term = true;
                else {
                    T = v3(T.x * W.x, T.y * W.y, T.z * W.z);
                    if (!(fmax_(T.x, fmax_(T.y, T.z)) > 0.0f)) term = true;
                    else if (depth >= fp.rr_start) {
                        const float qrr = fmin_(fmax_(T.x, fmax_(T.y, T.z)), 0.95f);
                        if (!(u01(key, 7u + 4u * bb) < qrr)) term = true;
                        else { const float iq = 1.0f / qrr; T = v3(T.x * iq, T.y * iq, T.z * iq); }
                    }
                    if (!term) { o = madd(side * fp.ray_eps, n, P); d = wi; }
                }
            }
        }

        if (term) {
            if (!touched) { A = ps.acc[slot]; touched = true; }
            A.w += 1.0f;
            ++sample;
            if (sample < fp.spp) { // regenerate the next sample of this pixel in place
                uint32_t x = 0, y = 0;
                slot_pixel(slot, fp, x, y);
                key = path_key(fp.seed_hashed, y * fp.width + x, fp.sample_offset + sample);
                camera_ray_of(sc.cam, x, y, key, o, d);
                T = v3(1.f, 1.f, 1.f);
                depth = 0;
                alive = true;
            }
        } else alive = true;

        if (touched) ps.acc[slot] = A;
        if (alive) {
            ps.ray_o[slot] = make_float4(o.x, o.y, o.z, 0.f);
            ps.ray_d[slot] = make_float4(d.x, d.y, d.z, 0.f);
            ps.thr[slot] = make_float4(T.x, T.y, T.z, __uint_as_float(key));
            ps.sd[slot] = (sample << 8) | depth;
        }
    }
    wave_push(&ps.counters[cnt_ext_index(parity ^ 1u, shard)], ps.q_ext[parity ^ 1u] + qbase, alive, slot);
}

// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock) k_assemble(const float4 *__restrict__ gathered, uint32_t nranks, uint32_t slots_per_rank,
                                                     uint32_t width, uint32_t height, uint32_t tiles_x, uint32_t n_tiles,
                                                     float inv_spp, float4 *__restrict__ fb, uint32_t *__restrict__ fb8)
{
    const uint32_t x = blockIdx.x * 64u + threadIdx.x, y = blockIdx.y * 4u + threadIdx.y;
    if (x >= width || y >= height) return;
    const uint32_t tile = (y >> kTileShift) * tiles_x + (x >> kTileShift);
    const uint32_t rank = tile % nranks, tl = tile / nranks;
    const uint32_t lx = x & (kTile - 1), ly = y & (kTile - 1);
    const uint32_t inner = ((((ly >> 3) << 3) + (lx >> 3)) << 6) | ((ly & 7u) << 3) | (lx & 7u);
    const float4 a = gathered[(size_t)rank * slots_per_rank + ((size_t)tl << (2 * kTileShift)) + inner];
    const float4 c = make_float4(a.x * inv_spp, a.y * inv_spp, a.z * inv_spp, a.w * inv_spp);
    const size_t i = (size_t)y * width + x;
    fb[i] = c;
    fb8[i] = unorm8(c.x) | (unorm8(c.y) << 8) | (unorm8(c.z) << 16) | (unorm8(c.w) << 24);
    (void)n_tiles;
}

// ================================================================================================ launchers
static inline uint32_t blocks_for(uint32_t n) { return n ? (n + kBlock - 1) / kBlock : 1u; }

hipError_t launch_reference_sphere(hipStream_t s, uint32_t w, uint32_t h, float4 *out_f, uint32_t *out8)
{
    // the reference dispatches ceil(1920/32) x ceil(1080/32) groups of 32x32 (Renderer.cs:1020, Test.hlsl:3);
    // here 64x4 lanes per group so one wavefront stores one contiguous 1 KiB row segment
    dim3 block(64, 4, 1), grid((w + 63) / 64, (h + 3) / 4, 1);
    hipLaunchKernelGGL(k_reference_sphere, grid, block, 0, s, w, h, out_f, out8);
    return hipGetLastError();
}

hipError_t launch_generate(hipStream_t s, const DeviceScene &sc, const PathState &ps, const FrameParams &fp)
{
    hipLaunchKernelGGL(k_generate, dim3(blocks_for(ps.shard_cap), kShards), dim3(kBlock), 0, s, sc, ps, fp);
    return hipGetLastError();
}

hipError_t launch_extend(hipStream_t s, const DeviceScene &sc, const PathState &ps, uint32_t parity, uint32_t shard_bound, bool count)
{
    const dim3 grid(blocks_for(shard_bound), kShards), block(kBlock);
    if (sc.bvh_width == 4) {
        if (count) hipLaunchKernelGGL((k_extend<4, true>), grid, block, 0, s, sc, ps, parity);
        else hipLaunchKernelGGL((k_extend<4, false>), grid, block, 0, s, sc, ps, parity);
    } else {
        if (count) hipLaunchKernelGGL((k_extend<2, true>), grid, block, 0, s, sc, ps, parity);
        else hipLaunchKernelGGL((k_extend<2, false>), grid, block, 0, s, sc, ps, parity);
    }
    return hipGetLastError();
}

hipError_t launch_shade(hipStream_t s, const DeviceScene &sc, const PathState &ps, const FrameParams &fp, uint32_t parity, uint32_t shard_bound)
{
    hipLaunchKernelGGL(k_shade, dim3(blocks_for(shard_bound), kShards), dim3(kBlock), 0, s, sc, ps, fp, parity);
    return hipGetLastError();
}

hipError_t launch_assemble(hipStream_t s, const float4 *gathered, uint32_t nranks, uint32_t slots_per_rank,
                           uint32_t width, uint32_t height, uint32_t tiles_x, uint32_t n_tiles, float inv_spp,
                           float4 *fb, uint32_t *fb8)
{
    dim3 block(64, 4, 1), grid((width + 63) / 64, (height + 3) / 4, 1);
    hipLaunchKernelGGL(k_assemble, grid, block, 0, s, gathered, nranks, slots_per_rank, width, height, tiles_x, n_tiles, inv_spp, fb, fb8);
    return hipGetLastError();
}

} // namespace ptrt
