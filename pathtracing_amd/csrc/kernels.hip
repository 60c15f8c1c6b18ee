// kernels.hip — gfx950 wavefront kernels of libptrt (docs/SPEC.md; DESIGN.md "Kernels").
//
//   k_reference_sphere : the reference's CSMain (Test.hlsl:1-40; dispatch Renderer.cs:1020), one lane = one pixel
//   k_generate         : camera ray of the first sample of every (owned pixel, sample stream), fills the extend queues
//   k_extend<L,C,FUSE> : ray -> closest hit. BVH traversal for node layout L, stack in LDS ([level][lane],
//                        conflict-free), spheres by scalar loads, one 4-row fetch per step for node or triangle.
//                        FUSE (the default pipeline): the lane also shades its hit (shade_one: emission, BSDF sample,
//                        Russian roulette, accumulate, in-place regeneration of the stream's next sample) and goes on
//                        for up to `bounces` path vertices with the path state in registers, then queues its slot
//   k_extend_packed<..>: the same, with ballot/mbcnt refill of idle lanes from a per-wave chunk of the queue
//   k_shade<MODE>      : the split pipeline's second kernel (PT_FLAG_SPLIT_KERNELS / PT_FLAG_BUCKET_SPECULAR): walks
//                        the queue k_extend<.., SHADE_NONE> just walked; QUEUE/INLINE shade in queue order, BUCKETS
//                        shades the metal / dielectric hits that QUEUE deferred to per-kind bucket queues
//   k_reduce_streams   : fixed-order sum of a pixel's stream partials
//   k_assemble         : tile-major slots (of 1..R ranks) -> row-major float4 + RGBA8 frame
//
// One slot per (owned pixel, stream), at most one live path per slot => accumulator RMW without atomics and a
// per-stream summation order identical to the oracle's.
// Queues are split into kShards static shards (ptrt_internal.h): a workgroup belongs to one shard (block_pos), and every lane of a block only
// ever sees slots of its own shard, so a wavefront's push goes to exactly one per-shard counter. Queues are carried
// from one iteration to the next in place (entry j stays entry j, dead paths leave holes) and re-packed only when a
// shard's alive/length ratio says so (want_compact); DESIGN.md §3.
#include "ptrt_internal.h"
#include "pt_device.h"
#include <algorithm>
#include <cstdlib>


using namespace ptd;

namespace ptrt {

PT_DEV uint32_t lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

// Scene data that no kernel writes, read at a wave-uniform address: through the constant address space the load is a scalar
// load (s_load_dwordx4 into SGPRs, scalar cache) instead of a vector load whose 64 lanes fetch the same 16 bytes — for the sphere
// list that is 4 vector-memory round trips less per ray, each of which the wave used to wait for.
typedef const float4 __attribute__((address_space(4))) *uniform_f4;
PT_DEV uniform_f4 as_uniform(const float4 *p) { return (uniform_f4)(uintptr_t)p; }
PT_DEV float4 uniform_load(uniform_f4 p, uint32_t i) { return make_float4(p[i].x, p[i].y, p[i].z, p[i].w); }
// The sphere list against one ray, in list order (docs/SPEC.md: ties go to the lower primitive id, and a later sphere never replaces
// an equal hit): four spheres per 64-byte scalar load (api.cpp pads the array to a multiple of four entries), tests of the ones
// that exist. Returns the number of tests.
PT_DEV uint32_t spheres_test(const float4 *spheres, uint32_t n_spheres, uint32_t first_id, V3 o, V3 d, Hit &h)
{
    const uniform_f4 sp = as_uniform(spheres);
    for (uint32_t j = 0; j < n_spheres; j += 4u) {
        const float4 s0 = uniform_load(sp, j), s1 = uniform_load(sp, j + 1u), s2 = uniform_load(sp, j + 2u), s3 = uniform_load(sp, j + 3u);
        sphere_test(s0, first_id + j, o, d, h);
        if (j + 1u < n_spheres) sphere_test(s1, first_id + j + 1u, o, d, h);
        if (j + 2u < n_spheres) sphere_test(s2, first_id + j + 2u, o, d, h);
        if (j + 3u < n_spheres) sphere_test(s3, first_id + j + 3u, o, d, h);
    }
    return n_spheres;
}

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

// Kernel arguments of the extend kernels: ONE by-value block (the kernarg segment). Hot fields (node / triangle / stack
// pointers, counts) are read from the parameter as usual; everything only the shading / regeneration / queueing code wants
// (camera, sky, frame parameters, path-state pointers) is read through cold(): the kernel's own kernarg segment behind an
// opaque copy of its address, so that those scalar loads are issued where they are used instead of being hoisted to the
// kernel entry and kept live in SGPRs across the traversal loop (which is what spilled 57 SGPRs to VGPR lanes and 20 B/lane
// of scratch in round 1).
struct ExtArgs {
    DeviceScene sc; PathState ps; FrameParams fp;
    uint32_t it, compact, bounces, chunk;
};
// Element `i` of a slot-indexed array by a 32-bit byte offset (api.cpp keeps n_slots * 16 below 2^32): the address is an SGPR
// base plus one VGPR offset, instead of a 64-bit address in a VGPR pair per array that stays live across the traversal loop.
template <typename T>
PT_DEV T &at(T *base, uint32_t i) { return *reinterpret_cast<T *>(reinterpret_cast<char *>(base) + (size_t)(i * (uint32_t)sizeof(T))); }

PT_DEV const ExtArgs &cold()
{
    auto p = __builtin_amdgcn_kernarg_segment_ptr(); // constant address space: scalar loads
    asm volatile("" : "+s"(p));
    return *(const ExtArgs *)p;
}

// Slot order. A slot is one (pixel slot, stream) pair; 64 consecutive slots are one 8x8 pixel block of one stream (a
// wavefront). PT_STREAM_INNER: the K streams of a block are K consecutive 64-slot groups, so that wavefronts that run
// at the same time work on the same few tiles of the image (their primary rays want the same corner of the scene,
// which then fits the XCDs' 4 MB L2s). Otherwise stream-major: all blocks of stream 0, then stream 1, ...
#ifndef PT_STREAM_INNER
#define PT_STREAM_INNER 1
#endif
PT_DEV uint32_t slot_stream(uint32_t slot, const FrameParams &fp)
{
    if (!PT_STREAM_INNER) return slot / fp.slots_per_stream;
    const uint32_t g = slot >> 6;
    return g - div_by(g, fp.streams_magic, fp.streams_shift) * fp.streams;
}
PT_DEV uint32_t slot_pixel_slot(uint32_t slot, const FrameParams &fp)
{
    return PT_STREAM_INNER ? ((div_by(slot >> 6, fp.streams_magic, fp.streams_shift) << 6) | (slot & 63u)) : slot % fp.slots_per_stream;
}
PT_DEV size_t slot_of(uint32_t pixel_slot, uint32_t stream, uint32_t streams, uint32_t slots_per_stream)
{
    return PT_STREAM_INNER ? (((size_t)(pixel_slot >> 6) * streams + stream) << 6) | (pixel_slot & 63u)
                           : (size_t)stream * slots_per_stream + pixel_slot;
}

// Which shard and which block of that shard's queue a workgroup is. PT_SHARD_FASTEST: a 1-D grid with the shard as the
// fastest index, so that workgroups are dispatched in slot order (shard s holds every 64th group of 2^kShardGroupShift slots) and, with the
// dispatcher dealing workgroups round-robin to the 8 XCDs, a shard's workgroups always land on the same XCD.
#ifndef PT_SHARD_FASTEST
#define PT_SHARD_FASTEST 1
#endif
PT_DEV void block_pos(const PathState &ps, uint32_t &shard, uint32_t &bx, uint32_t &nbx)
{
    // shard_count is a power of two (kShards / loops): mask and shift, not a division per wave
    const uint32_t sh = 31u - (uint32_t)__builtin_clz(ps.shard_count);
    if (PT_SHARD_FASTEST) { shard = (blockIdx.x & (ps.shard_count - 1u)) + ps.shard_base; bx = blockIdx.x >> sh; nbx = gridDim.x >> sh; }
    else { shard = blockIdx.y + ps.shard_base; bx = blockIdx.x; nbx = gridDim.x; }
}
static inline dim3 shard_grid(uint32_t blocks, uint32_t shard_count)
{
    return PT_SHARD_FASTEST ? dim3(blocks * shard_count) : dim3(blocks, shard_count);
}

PT_DEV bool slot_pixel(uint32_t slot_all, const FrameParams &fp, uint32_t &x, uint32_t &y)
{
    const uint32_t slot = slot_pixel_slot(slot_all, fp); // the K streams of a pixel share its pixel slot
    const uint32_t tl = slot >> (2 * kTileShift), inner = slot & (kTilePixels - 1);
    const uint32_t tile = fp.rank + fp.nranks * tl;
    if (tile >= fp.n_tiles) return false;
    const uint32_t ty = div_by(tile, fp.tiles_x_magic, fp.tiles_x_shift), tx = tile - ty * fp.tiles_x;
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
// One inner-node visit for node layout L (PT_BVH_WIDTH_*): fetch the node, slab-test its children against the ray,
// return (key, ref) sorted by ascending key = (bits(tn) & ~3) | slot; children that are missed/empty get key ~0.
// L = 2 / 4 : N slots of {lo.xyz|ref, hi.xyz|0}  -> 2N 16-byte loads per lane
// L = 4Q    : 64-byte node {origin|exps, refs, qlo x/y/z + qhi x, qhi y/z}, boxes decoded as fma(q, 2^e, origin)
//             -> 4 loads per lane: half the bytes and half the (fully divergent) memory instructions per visit.
// The first four 16-byte rows of the node arrive in r0..r3: the caller loads them (from the node OR, for a lane that
// sits on a leaf, from its triangle) before it branches, so a wave-iteration has one memory round trip, not two.
// L = 8Q    : one 128-byte line per node {origin|exps, 8 refs, qlo x|y, qlo z|qhi x, qhi y|z, pad} -> 6 loads per lane.
//             Beyond the caches a node visit costs the memory system one 128-byte line whether 64 or 96 bytes of it are
//             used (tools/ubench/gather_rows.hip), and an 8-wide tree needs fewer visits. Keys keep 3 bits for the slot.
template <int L>
PT_DEV constexpr int node_rows() { return (L == PT_BVH_WIDTH_4 || L == PT_BVH_WIDTH_8Q || L == PT_BVH_WIDTH_8O) ? 8 : 4; }
template <int L>
PT_DEV constexpr int fanout() { return L == PT_BVH_WIDTH_2 ? 2 : (L == PT_BVH_WIDTH_8Q || L == PT_BVH_WIDTH_8O) ? 8 : 4; }

template <int L>
PT_DEV void visit_node_keys(const float4 *__restrict__ nd, float4 r0, float4 r1, float4 r2, float4 r3, const RaySetup &rs, float t_best,
                            uint32_t (&key)[fanout<L>()], int32_t (&ref)[fanout<L>()])
{
    constexpr int N = fanout<L>();
    if constexpr (L == PT_BVH_WIDTH_8Q) {
        const float4 r4 = nd[4], r5 = nd[5];
        const uint32_t eb = __float_as_uint(r0.w);
        const float sx = __uint_as_float((eb & 0xffu) << 23), sy = __uint_as_float(((eb >> 8) & 0xffu) << 23),
                    sz = __uint_as_float(((eb >> 16) & 0xffu) << 23);
        // 8 bytes per coordinate: children 0-3 in the first dword, 4-7 in the second
        const uint32_t q[6][2] = { { __float_as_uint(r3.x), __float_as_uint(r3.y) }, { __float_as_uint(r3.z), __float_as_uint(r3.w) },
                                   { __float_as_uint(r4.x), __float_as_uint(r4.y) }, { __float_as_uint(r4.z), __float_as_uint(r4.w) },
                                   { __float_as_uint(r5.x), __float_as_uint(r5.y) }, { __float_as_uint(r5.z), __float_as_uint(r5.w) } };
        ref[0] = __float_as_int(r1.x); ref[1] = __float_as_int(r1.y); ref[2] = __float_as_int(r1.z); ref[3] = __float_as_int(r1.w);
        ref[4] = __float_as_int(r2.x); ref[5] = __float_as_int(r2.y); ref[6] = __float_as_int(r2.z); ref[7] = __float_as_int(r2.w);
        // near / far planes picked per axis by the sign of the ray direction, on the packed bytes, once per node: exactly
        // the values min(ta,tb) / max(ta,tb) of the spec's slab test would pick (fma is monotone in q), for 6 selects per node
        // instead of 6 min/max per child — the kernel is VALU-issue bound as much as memory bound
        const bool ng[3] = { rs.inv.x < 0.f, rs.inv.y < 0.f, rs.inv.z < 0.f };
        // docs/SPEC.md §4.1 slab test of a quantised child: t = fma(q, A, B) with A = 2^e * inv (exact: a power of two times inv)
        // and B = fma(origin, inv, noi) per axis and node — one fma per plane instead of decode + slab
        const float qa[3] = { sx * rs.inv.x, sy * rs.inv.y, sz * rs.inv.z };
        const float qb[3] = { fma_(r0.x, rs.inv.x, rs.noi.x), fma_(r0.y, rs.inv.y, rs.noi.y), fma_(r0.z, rs.inv.z, rs.noi.z) };
        uint32_t qn[3][2], qf[3][2];
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int w = 0; w < 2; ++w) { qn[k][w] = ng[k] ? q[3 + k][w] : q[k][w]; qf[k][w] = ng[k] ? q[k][w] : q[3 + k][w]; }
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const int w = c >> 2, sh = 8 * (c & 3);
            float tnk[3], tfk[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                tnk[k] = fma_((float)((qn[k][w] >> sh) & 0xffu), qa[k], qb[k]);
                tfk[k] = fma_((float)((qf[k][w] >> sh) & 0xffu), qa[k], qb[k]);
            }
            const float tn = fmax_(fmax_(tnk[0], tnk[1]), fmax_(tnk[2], 0.0f));
            const float tf = fmin_(fmin_(tfk[0], tfk[1]), fmin_(tfk[2], t_best)) * 1.0000004f;
            const bool hb = tn <= tf && ref[c] != PT_BVH_EMPTY;
            key[c] = hb ? ((__float_as_uint(tn) & ~7u) | (uint32_t)c) : 0xFFFFFFFFu;
        }
    } else if constexpr (L == PT_BVH_WIDTH_4Q) {
        const uint32_t eb = __float_as_uint(r0.w);
        const float sx = __uint_as_float((eb & 0xffu) << 23), sy = __uint_as_float(((eb >> 8) & 0xffu) << 23),
                    sz = __uint_as_float(((eb >> 16) & 0xffu) << 23);
        const uint32_t qlx = __float_as_uint(r2.x), qly = __float_as_uint(r2.y), qlz = __float_as_uint(r2.z),
                       qhx = __float_as_uint(r2.w), qhy = __float_as_uint(r3.x), qhz = __float_as_uint(r3.y);
        ref[0] = __float_as_int(r1.x); ref[1] = __float_as_int(r1.y); ref[2] = __float_as_int(r1.z); ref[3] = __float_as_int(r1.w);
        // (measured and rejected: the (lo, hi) pairs of this decode + slab test as 24 v_pk_fma_f32 instead of 48 v_fma_f32 —
        //  bit-identical, but 6 % slower: packed fp32 issues at half rate here and wants aligned register pairs)
        // near / far planes by the sign of the ray direction, selected on the packed bytes once per node (see BVH8Q above)
        const bool ng[3] = { rs.inv.x < 0.f, rs.inv.y < 0.f, rs.inv.z < 0.f };
        // docs/SPEC.md §4.1 slab test of a quantised child: t = fma(q, A, B) with A = 2^e * inv (exact: a power of two times inv)
        // and B = fma(origin, inv, noi) per axis and node — one fma per plane instead of decode + slab
        const float qa[3] = { sx * rs.inv.x, sy * rs.inv.y, sz * rs.inv.z };
        const float qb[3] = { fma_(r0.x, rs.inv.x, rs.noi.x), fma_(r0.y, rs.inv.y, rs.noi.y), fma_(r0.z, rs.inv.z, rs.noi.z) };
        const uint32_t ql[3] = { qlx, qly, qlz }, qh[3] = { qhx, qhy, qhz };
        uint32_t qn[3], qf[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) { qn[k] = ng[k] ? qh[k] : ql[k]; qf[k] = ng[k] ? ql[k] : qh[k]; }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float tnk[3], tfk[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                tnk[k] = fma_((float)((qn[k] >> (8 * c)) & 0xffu), qa[k], qb[k]);
                tfk[k] = fma_((float)((qf[k] >> (8 * c)) & 0xffu), qa[k], qb[k]);
            }
            const float tn = fmax_(fmax_(tnk[0], tnk[1]), fmax_(tnk[2], 0.0f));
            const float tf = fmin_(fmin_(tfk[0], tfk[1]), fmin_(tfk[2], t_best)) * 1.0000004f;
            const bool hb = tn <= tf && ref[c] != PT_BVH_EMPTY;
            key[c] = hb ? ((__float_as_uint(tn) & ~3u) | (uint32_t)c) : 0xFFFFFFFFu;
        }
    } else {
        float4 r[2 * N];
        r[0] = r0; r[1] = r1; r[2] = r2; r[3] = r3;
#pragma unroll
        for (int i = 4; i < 2 * N; ++i) r[i] = nd[i]; // BVH4/128 B: second half of the node
#pragma unroll
        for (int c = 0; c < N; ++c) {
            float tn;
            ref[c] = __float_as_int(r[2 * c].w);
            const bool hb = box_test(r[2 * c], r[2 * c + 1], rs, t_best, tn) && ref[c] != PT_BVH_EMPTY;
            key[c] = hb ? ((__float_as_uint(tn) & ~3u) | (uint32_t)c) : 0xFFFFFFFFu;
        }
    }
}

// L = 8O: the node bytes of 8Q, but the builder has placed every child in the slot that names its corner of the node (bit k of the
// slot = towards +axis k), so `slot ^ octant` (octant bit k = the ray runs towards -axis k) is a front-to-back order known without
// looking at the distances: no keys, no sorting network. Returns the children in VISIT order: bit p of `hits` / ref[p] = the child in
// slot p ^ octant. The xor permutation is three rounds of conditional pair swaps on the refs and three bit-swizzles on the hit mask.
PT_DEV void visit_node_oct8(const float4 *__restrict__ nd, float4 r0, float4 r1, float4 r2, float4 r3, const RaySetup &rs, float t_best,
                            uint32_t &hits, int32_t (&ref)[8])
{
    const float4 r4 = nd[4], r5 = nd[5];
    const uint32_t eb = __float_as_uint(r0.w);
    const float sx = __uint_as_float((eb & 0xffu) << 23), sy = __uint_as_float(((eb >> 8) & 0xffu) << 23), sz = __uint_as_float(((eb >> 16) & 0xffu) << 23);
    const uint32_t q[6][2] = { { __float_as_uint(r3.x), __float_as_uint(r3.y) }, { __float_as_uint(r3.z), __float_as_uint(r3.w) },
                               { __float_as_uint(r4.x), __float_as_uint(r4.y) }, { __float_as_uint(r4.z), __float_as_uint(r4.w) },
                               { __float_as_uint(r5.x), __float_as_uint(r5.y) }, { __float_as_uint(r5.z), __float_as_uint(r5.w) } };
    ref[0] = __float_as_int(r1.x); ref[1] = __float_as_int(r1.y); ref[2] = __float_as_int(r1.z); ref[3] = __float_as_int(r1.w);
    ref[4] = __float_as_int(r2.x); ref[5] = __float_as_int(r2.y); ref[6] = __float_as_int(r2.z); ref[7] = __float_as_int(r2.w);
    const bool ng[3] = { rs.inv.x < 0.f, rs.inv.y < 0.f, rs.inv.z < 0.f };
    const float qa[3] = { sx * rs.inv.x, sy * rs.inv.y, sz * rs.inv.z };
    const float qb[3] = { fma_(r0.x, rs.inv.x, rs.noi.x), fma_(r0.y, rs.inv.y, rs.noi.y), fma_(r0.z, rs.inv.z, rs.noi.z) };
    uint32_t qn[3][2], qf[3][2];
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int w = 0; w < 2; ++w) { qn[k][w] = ng[k] ? q[3 + k][w] : q[k][w]; qf[k][w] = ng[k] ? q[k][w] : q[3 + k][w]; }
    uint32_t m = 0u; // bit c: slot c is hit
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const int w = c >> 2, sh = 8 * (c & 3);
        float tnk[3], tfk[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            tnk[k] = fma_((float)((qn[k][w] >> sh) & 0xffu), qa[k], qb[k]);
            tfk[k] = fma_((float)((qf[k][w] >> sh) & 0xffu), qa[k], qb[k]);
        }
        const float tn = fmax_(fmax_(tnk[0], tnk[1]), fmax_(tnk[2], 0.0f));
        const float tf = fmin_(fmin_(tfk[0], tfk[1]), fmin_(tfk[2], t_best)) * 1.0000004f;
        m |= (tn <= tf && ref[c] != PT_BVH_EMPTY) ? (1u << c) : 0u;
    }
    // position p holds slot p ^ octant
#pragma unroll
    for (int b = 0; b < 3; ++b) {
        const int d = 1 << b;
        const uint32_t lo = b == 0 ? 0x55u : b == 1 ? 0x33u : 0x0Fu;
        m = ng[b] ? (((m & lo) << d) | ((m >> d) & lo)) : m;
#pragma unroll
        for (int i = 0; i < 8; ++i)
            if (!(i & d)) { const int32_t a = ref[i], bb = ref[i | d]; ref[i] = ng[b] ? bb : a; ref[i | d] = ng[b] ? a : bb; }
    }
    hits = m;
}

// Sorting network over (key, ref), ascending key, in two halves: after sort_head the nearest hit child is in slot 0 (so its
// fetch can be issued), sort_tail finishes the order of the others (which only the pushes need).
template <int N>
PT_DEV void cswap(uint32_t (&key)[N], int32_t (&ref)[N], int a, int b)
{
    if (key[a] > key[b]) {
        const uint32_t tk = key[a]; key[a] = key[b]; key[b] = tk;
        const int32_t tr = ref[a]; ref[a] = ref[b]; ref[b] = tr;
    }
}
// Two independent compare-exchanges with both comparisons issued before the first select: a v_cmp that writes an SGPR pair must be two
// instructions ahead of the v_cndmask that reads it (gfx950 hazard: the compiler otherwise pads every comparator with `s_nop 1`).
template <int N>
PT_DEV void cswap2(uint32_t (&key)[N], int32_t (&ref)[N], int a, int b, int c, int d)
{
    const bool s0 = key[a] > key[b], s1 = key[c] > key[d];
    const uint32_t ka = key[a], kb = key[b], kc = key[c], kd = key[d];
    const int32_t ra = ref[a], rb = ref[b], rc = ref[c], rd = ref[d];
    key[a] = s0 ? kb : ka; key[b] = s0 ? ka : kb; ref[a] = s0 ? rb : ra; ref[b] = s0 ? ra : rb;
    key[c] = s1 ? kd : kc; key[d] = s1 ? kc : kd; ref[c] = s1 ? rd : rc; ref[d] = s1 ? rc : rd;
}
template <int N>
PT_DEV void sort_head(uint32_t (&key)[N], int32_t (&ref)[N])
{
    if constexpr (N == 2) { cswap(key, ref, 0, 1); }
    else if constexpr (N == 4) { cswap2(key, ref, 0, 1, 2, 3); cswap2(key, ref, 0, 2, 1, 3); }
    else { // 19-comparator network for 8 keys, whole
        cswap(key, ref, 0, 1); cswap(key, ref, 2, 3); cswap(key, ref, 4, 5); cswap(key, ref, 6, 7); cswap(key, ref, 0, 2); cswap(key, ref, 1, 3);
        cswap(key, ref, 4, 6); cswap(key, ref, 5, 7); cswap(key, ref, 1, 2); cswap(key, ref, 5, 6); cswap(key, ref, 0, 4); cswap(key, ref, 3, 7);
        cswap(key, ref, 1, 5); cswap(key, ref, 2, 6); cswap(key, ref, 1, 4); cswap(key, ref, 3, 6); cswap(key, ref, 2, 4); cswap(key, ref, 3, 5);
        cswap(key, ref, 3, 4);
    }
}
template <int N>
PT_DEV void sort_tail(uint32_t (&key)[N], int32_t (&ref)[N])
{
    if constexpr (N == 4) { cswap(key, ref, 1, 2); } // (0,1)(2,3) | (0,2)(1,3) | (1,2): the five comparators, in three rounds
}
template <int L>
PT_DEV void visit_node(const float4 *__restrict__ nd, float4 r0, float4 r1, float4 r2, float4 r3, const RaySetup &rs, float t_best,
                       uint32_t (&key)[fanout<L>()], int32_t (&ref)[fanout<L>()])
{
    visit_node_keys<L>(nd, r0, r1, r2, r3, rs, t_best, key, ref);
    sort_head(key, ref); sort_tail(key, ref);
}

// ------------------------------------------------------------------------------------------------
// One path vertex (docs/SPEC.md §5): emission / sky, BSDF sample, throughput, Russian roulette, and in-place
// regeneration of the stream's next camera sample when the path ends. (o, d, t, ref) = the ray and its closest hit.
// Works on the registers in `r`; returns true when `r` holds a ray for the next bounce.
//   SHADE_QUEUE   : Lambert and misses; a specular hit sets `defer` to its bucket and leaves the state untouched
//   SHADE_BUCKETS : the hit's kind is `b` (wave-uniform)
//   SHADE_INLINE  : every kind, by a divergent branch
enum ShadeMode { SHADE_NONE = -1, SHADE_QUEUE = 0, SHADE_BUCKETS = 1, SHADE_INLINE = 2 };
// The path state of a slot, held in registers between path_load and path_store (one bounce in k_shade, several in k_extend).
struct PathRegs {
    V3 o, d, T;                 // ray, throughput
    uint32_t key, sample, depth; // RNG key of the path, sample index, vertices so far
};
PT_DEV void path_load(const PathState &ps, uint32_t slot, PathRegs &r)
{
    const float4 O = at(ps.ray_o, slot), D = at(ps.ray_d, slot), TK = at(ps.thr, slot);
    const uint32_t sdv = at(ps.sd, slot);
    r.o = xyz(O); r.d = xyz(D); r.T = xyz(TK);
    r.key = __float_as_uint(TK.w); r.sample = sdv >> 8; r.depth = sdv & 255u;
}
PT_DEV void path_store(const PathState &ps, uint32_t slot, const PathRegs &r)
{
    at(ps.ray_o, slot) = make_float4(r.o.x, r.o.y, r.o.z, 0.f);
    at(ps.ray_d, slot) = make_float4(r.d.x, r.d.y, r.d.z, 0.f);
    at(ps.thr, slot) = make_float4(r.T.x, r.T.y, r.T.z, __uint_as_float(r.key));
    at(ps.sd, slot) = (r.sample << 8) | r.depth;
}

// The first sample of a slot's stream: stream k takes the samples s with (sample_offset + s) % K == k, in increasing s (docs/SPEC.md §5).
PT_DEV uint32_t first_sample(uint32_t slot, const FrameParams &fp)
{
    const uint32_t v = slot_stream(slot, fp) + fp.streams - fp.offset_mod; // in [1, 2 streams)
    return v >= fp.streams ? v - fp.streams : v;
}
// The state a slot starts the frame with: the camera ray of its stream's first sample (slot_pixel(slot) must be on the image).
PT_DEV void path_init(const DeviceScene &sc, const FrameParams &fp, uint32_t slot, PathRegs &r)
{
    uint32_t x = 0, y = 0;
    slot_pixel(slot, fp, x, y);
    r.sample = first_sample(slot, fp);
    r.key = path_key(fp.seed_hashed, y * fp.width + x, fp.sample_offset + r.sample);
    camera_ray_of(sc.cam, x, y, r.key, r.o, r.d);
    r.T = v3(1.f, 1.f, 1.f);
    r.depth = 0u;
}

// ------------------------------------------------------------------------------------------------
// grid (ceil(shard_cap/256), kShards). Which slots a shard owns is decided here and nowhere else (queues only ever hand a slot on to
// the same shard): entry group t = j >> G of shard s is slot group t * kShards + (s - t) mod kShards, G = kShardGroupShift. Every run of
// kShards consecutive slot groups — with 8 streams: the 8 streams of 8 neighbouring pixel blocks — is dealt over all shards, and the deal
// ROTATES by one shard from run to run: the groups of one sample stream would otherwise all land on the shards s with s % streams ==
// stream, i.e. (a shard's workgroups staying on one XCD) on one XCD — a frame with fewer samples than streams (one progressive sample
// per call) then ran on an eighth of the chip (1080p, 1 spp, 8 streams: 1.45 -> 0.5 ms).
__global__ void __launch_bounds__(kBlock) k_generate(DeviceScene sc, PathState ps, FrameParams fp, uint32_t full)
{
    uint32_t shard, bx, nbx;
    block_pos(ps, shard, bx, nbx);
    const uint32_t j = bx * kBlock + threadIdx.x;
#ifndef PT_DEAL_ROT
#define PT_DEAL_ROT 1
#endif
    const uint32_t t = j >> kShardGroupShift, rot = (shard + kShards - (t * PT_DEAL_ROT) % kShards) % kShards;
    const uint32_t slot = ((t * kShards + rot) << kShardGroupShift) | (j & ((1u << kShardGroupShift) - 1u));
    uint32_t x = 0, y = 0;
    const bool in_range = j < ps.shard_cap && slot < ps.n_slots;
    // stream k takes the samples s with (sample_offset + s) % K == k, in increasing s (docs/SPEC.md §5)
    const uint32_t first = first_sample(slot, fp);
    const bool valid = in_range && first < fp.spp && slot_pixel(slot, fp, x, y);
    if (in_range && !fp.accumulate) ps.acc[slot] = make_float4(0.f, 0.f, 0.f, 0.f);
    // `full` == 0: the first launch of a fused extend kernel builds the state of its slots in registers (path_init) instead of
    // reading it back: 52 bytes per slot not written here and not read there
    if (valid && full == 1u) {
        PathRegs r;
        path_init(sc, fp, slot, r);
        path_store(ps, slot, r);
    }
    // the first queue is the shard's slots in slot order, holes (off-image pixels, streams with no sample) included — or, full == 2
    // (frames in which whole streams have no sample: spp < streams, e.g. one progressive sample per call), only the slots that
    // hold a path, appended per wavefront: the first launch then covers spp/streams of the slots instead of all of them
    if (full == 2u) wave_push(&ps.counters[cnt_ext_index(0, shard)], ps.q_ext[0] + (size_t)shard * ps.shard_cap, valid, slot);
    else if (j < ps.shard_cap) ps.q_ext[0][(size_t)shard * ps.shard_cap + j] = valid ? slot : kInvalidSlot;
    const uint32_t n_alive = (uint32_t)__syncthreads_count(valid);
    if (threadIdx.x == 0) {
        if (n_alive) atomicAdd(&ps.counters[cnt_alive_index(0, shard)], n_alive);
        if (bx == 0 && full != 2u) ps.counters[cnt_ext_index(0, shard)] = ps.shard_cap;
    }
}


template <int MODE>
PT_DEV bool shade_one(const DeviceScene &sc, const PathState &ps, const FrameParams &fp, uint32_t slot, PathRegs &r, float t, uint32_t ref,
                      uint32_t b, uint32_t &defer)
{
    V3 &o = r.o, &d = r.d, &T = r.T;
    uint32_t &key = r.key, &sample = r.sample;
    uint32_t depth = r.depth + 1u;
    // The slot's radiance sum | path count comes from HBM: it is requested up front, whether or not this vertex will touch it, so that
    // its latency runs under the hit-record and material loads below instead of behind them (Cornell 7.04 -> 6.83 ms, Cornell + glass
    // + metal 31.9 -> 31.7, soup 68.2 -> 67.6, 1M-triangle Cornell +-0); written back only if touched.
#ifdef PT_EXP_NOACC // timing probe only (wrong pictures): what the radiance-sum load costs where it stands
    float4 A = make_float4(0.f, 0.f, 0.f, 0.f);
#else
    float4 A = at(ps.acc, slot);
#endif
    bool touched = false, term = false, alive = false;
    auto add = [&](V3 L) {
        touched = true;
        A.x = fma_(T.x, L.x, A.x); A.y = fma_(T.y, L.y, A.y); A.z = fma_(T.z, L.z, A.z);
    };

    if (ref == PT_MISS) {
        add(v3(sc.sky[0], sc.sky[1], sc.sky[2]));
        term = true;
    } else {
        const V3 P = madd(t, d, o);
        V3 ng;
        uint32_t mat;
        if (ref < sc.n_tris) {
            const float4 ts = sc.tris[(size_t)ref * 4 + 3]; // normalize(cross(e1,e2)) precomputed at commit, bit-identical; the
                                                            // 4th row of the line k_extend fetched when it tested this triangle
            ng = xyz(ts);
            mat = __float_as_uint(ts.w);
        } else {
            const uint32_t j = ref - sc.n_tris;
            const float4 s = sc.spheres[j];
            const uint2 mi = sc.sph_mat[j];
            const float ir = __uint_as_float(mi.y); // 1.0f / s.w, correctly rounded on the host exactly as the division here would be
            ng = v3((P.x - s.x) * ir, (P.y - s.y) * ir, (P.z - s.z) * ir);
            mat = mi.x;
        }
        const bool front = dot(ng, d) < 0.0f;
        const V3 n = front ? ng : neg(ng);
        float4 m0 = sc.mats[(size_t)mat * 3], m1 = sc.mats[(size_t)mat * 3 + 1];
        const float4 m2 = sc.mats[(size_t)mat * 3 + 2];
#ifndef PT_EXP_NO_MATPAIR
        // Both rows are requested before either is looked at: left alone, the compiler sinks the second load behind the test of the
        // first row's kind — one more dependent round trip in a step that already chains hit record -> material
        asm volatile("" : "+v"(m0.x), "+v"(m1.x));
#endif
        const V3 alb = v3(m0.y, m0.z, m0.w), emi = xyz(m1);
        const uint32_t kind = __float_as_uint(m0.x);
        if (MODE == SHADE_QUEUE && kind != (uint32_t)PT_LAMBERT) { defer = 1u + kind; return false; } // shaded by k_shade<SHADE_BUCKETS>
        if (emi.x != 0.0f || emi.y != 0.0f || emi.z != 0.0f) add(emi);
        if (depth >= fp.max_depth) term = true;
        else {
            const uint32_t bb = depth - 1u;
            const uint32_t bk = MODE == SHADE_QUEUE ? (uint32_t)B_LAMBERT : MODE == SHADE_BUCKETS ? b : 1u + kind;
            BsdfSample bs;
            if (bk == B_LAMBERT) bs = sample_lambert(alb, n, u01(key, 4u + 4u * bb), u01(key, 5u + 4u * bb));
            else if (bk == B_METAL) bs = sample_metal(alb, m1.w, d, n, u01(key, 4u + 4u * bb), u01(key, 5u + 4u * bb));
            else bs = sample_dielectric(alb, m2.x, d, n, front, u01(key, 6u + 4u * bb));
            const V3 wi = bs.wi, W = bs.W;
            const float side = bs.side;
            if (!bs.ok) term = true;
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
        touched = true;
        A.w += 1.0f;
        sample += fp.streams;
        if (sample < fp.spp) { // regenerate this stream's next sample of the pixel in place
            uint32_t x = 0, y = 0;
            slot_pixel(slot, fp, x, y);
            key = path_key(fp.seed_hashed, y * fp.width + x, fp.sample_offset + sample);
            camera_ray_of(sc.cam, x, y, key, o, d);
            T = v3(1.f, 1.f, 1.f);
            depth = 0;
            alive = true;
        }
    } else alive = true;
    if (touched) at(ps.acc, slot) = A;
    r.depth = depth;
    return alive;
}

// Hand a lane's surviving slot to the next iteration's extend queue (reached by all lanes of the workgroup).
//   compact : append (ballot + one returning atomic per wavefront) -> dense queue, order scrambled by wavefront
//   else    : write position `gid` of the next queue (slot or hole); the shard's length carries over
// Ray accounting: the iteration that fills queue `c` counts the rays it traces in that queue's line (u64, words 2-3); the
// next iteration's first thread folds the count into rays[shard] and clears the line that is two iterations ahead.
PT_DEV unsigned long long *traced_counter(const PathState &ps, uint32_t c, uint32_t shard)
{
    return reinterpret_cast<unsigned long long *>(ps.counters + cnt_traced_index(c, shard));
}
// Run by the first thread of a shard in launch `it`; (n, n_alive) = the shard's line as the previous launch left it. That line also
// goes to the host (PathState::host_ring): a plain 16-byte store to mapped pinned memory instead of a copy dispatch per launch.
PT_DEV void fold_traced(const PathState &ps, uint32_t shard, uint32_t it, uint32_t n, uint32_t n_alive)
{
    *traced_counter(ps, (it + 2u) % 3u, shard) = 0ull;
    const unsigned long long traced = *traced_counter(ps, it % 3u, shard);
    *reinterpret_cast<unsigned long long *>(ps.counters + cnt_rays_index(shard)) += traced;
    if (ps.host_ring && it) ps.host_ring[((it - 1u) % ps.ring_slots) * kShards + shard] = make_uint4(n, n_alive, (uint32_t)traced, (uint32_t)(traced >> 32));
}

// Queue policy, from the shard's own counters so that every workgroup of a launch decides the same and without host lag.
PT_DEV bool want_compact(const PathState &ps, uint32_t len, uint32_t n_alive, uint32_t prev_alive, bool sticky, bool forced)
{
    // A re-pack costs nothing in the launch that does it (tools/exp_compact.py: the returning atomic alone is unmeasurable) but
    // it permutes the shard's wavefronts by arrival order, and every further one permutes them again: consecutive wavefronts
    // stop being the streams of neighbouring pixel blocks and their node fetches stop sharing cache lines (+3..5 % ns per ray
    // after ~100 re-packs with NO path lost). A hole costs a lane for every vertex of the next launch. So:
    //   predicted : re-pack when the queue this launch leaves behind would be emptier than compact_below, counting the holes
    //               it starts with plus as many deaths as the previous launch had (prev_alive - n_alive);
    //   sticky    : (frames of few samples per stream, where all streams run dry within a launch or two and no history
    //               predicts it) once a shard has re-packed - its queue is shorter than k_generate's - every launch does.
    const uint32_t deaths = prev_alive > n_alive ? prev_alive - n_alive : 0u;
    const uint32_t predicted = n_alive > deaths ? n_alive - deaths : 0u;
    return forced || (sticky && len < ps.shard_cap) || (float)predicted < ps.compact_below * (float)len;
}

// EXPERIMENT (-DPT_REPACK_SORT=1; DESIGN.md §4 "coherence-ordered re-packing"): at a re-pack a wave appends its survivors ordered by a
// 6-bit key — octant of the ray direction, octant of the origin about the scene's centre — instead of by lane. Rank of a lane among
// the wave's survivors by (key, lane), from ballots alone: walking the key bits from the top, `eq` keeps the lanes that agree with
// mine so far and `lt` collects those that have a 0 where I have a 1.
#ifndef PT_REPACK_SORT
#define PT_REPACK_SORT 0
#endif
PT_DEV uint32_t wave_rank_by_key(bool pred, uint32_t key)
{
    uint64_t eq = __ballot(pred), lt = 0;
#pragma unroll
    for (int b = 5; b >= 0; --b) {
        const bool one = (key >> b) & 1u;
        const uint64_t bal = __ballot(pred && one);
        if (one) lt |= eq & ~bal;
        eq &= one ? bal : ~bal;
    }
    const uint64_t below = (1ull << lane_id()) - 1ull;
    return (uint32_t)__popcll(lt) + (uint32_t)__popcll(eq & below);
}
PT_DEV void wave_push_sorted(uint32_t *counter, uint32_t *queue, bool pred, uint32_t value, uint32_t key)
{
    const uint64_t m = __ballot(pred);
    if (m == 0) return;
    const uint32_t rank = wave_rank_by_key(pred, key);
    const int leader = __ffsll((long long)m) - 1;
    uint32_t base = 0;
    if ((int)lane_id() == leader) base = atomicAdd(counter, (uint32_t)__popcll(m));
    base = __shfl(base, leader, 64);
    if (pred) queue[base + rank] = value;
}

PT_DEV void queue_next(const PathState &ps, uint32_t shard, uint32_t cnext, uint32_t *q_next, uint32_t gid, uint32_t total, bool alive,
                       uint32_t slot, bool compact, uint32_t sort_key = 0u)
{
    if (compact && PT_REPACK_SORT) wave_push_sorted(&ps.counters[cnt_ext_index(cnext, shard)], q_next, alive, slot, sort_key);
    else if (compact) wave_push(&ps.counters[cnt_ext_index(cnext, shard)], q_next, alive, slot);
    else {
        if (gid < total) at(q_next, gid) = alive ? slot : kInvalidSlot;
        if (gid == 0) ps.counters[cnt_ext_index(cnext, shard)] = total;
    }
    const uint64_t m = __ballot(alive);
    if (m && lane_id() == 0u) atomicAdd(&ps.counters[cnt_alive_index(cnext, shard)], (uint32_t)__popcll(m));
}

// ------------------------------------------------------------------------------------------------
// k_extend<L, COUNT, FUSE>: closest hit of every ray of the iteration's extend queue (one ray per lane, LDS traversal stack).
//   FUSE == SHADE_NONE : writes the hit record; k_shade walks the same queue afterwards
//   FUSE == SHADE_QUEUE (Lambert-only scenes) / SHADE_INLINE : the lane shades its own hit straight from registers and
//       queues its slot for the next iteration. With queues carried in place this needs nothing from any other lane, and
//       it takes the hit record (8 B written + read), the second read of the ray (32 B), one queue pass and one launch per
//       iteration out of the frame; waves that finish traversal early stream their shading traffic to HBM while the
//       others are still gathering nodes.
// Counters are triple-buffered by iteration (cur = it % 3 is read, next is filled, the third is zeroed for the
// iteration after), because a fused kernel fills `next` while other workgroups of the same launch are still starting.
// Waves per SIMD the kernels are compiled for (amdgpu_waves_per_eu pins the register budget: 8 waves = 64 VGPRs, 7 = 72, 6 = 80;
// unconstrained the fused variants take 84-89 VGPRs = 5 waves). Occupancy is what hides the latency of the dependent node gathers;
// measured on MI355X, one-ray-per-lane kernel, Grays/s:
//   1M-triangle Cornell (BVH4Q, Lambert): 6 waves 12.5, 7 waves 13.5, 8 waves 14.0 (64 VGPRs, nothing spilled);
//   Cornell (BVH2, Lambert; the default layout of small scenes): 7 waves 32.8, 8 waves 34.2 (64 VGPRs, nothing spilled);
//   BVH8Q / BVH4: the wider node's visit does not fit 64 registers (Cornell on BVH8Q: 7 waves 29.9, 8 waves 25.4);
//   all-kinds shading (SHADE_INLINE) fits 72 registers: 7 waves; at 8 it spills 4-5 VGPRs (Cornell + glass + metal: +-0, on BVH4Q +1.7 %).
// None of the non-counting k_extend instantiations spills a VGPR or uses scratch (python tools/resources.py); k_extend<BVH2, Lambert>
// keeps 2 SGPRs in VGPR lanes since the root's rows pass through SGPRs.
#ifndef PT_EXT_WAVES
#define PT_EXT_WAVES(L, FUSE) (((L) == PT_BVH_WIDTH_4Q || (L) == PT_BVH_WIDTH_2) && (FUSE) != SHADE_INLINE ? 8 : 7)
#endif
// A lane's traversal stack: kStackLds entries in its LDS column ([level][lane]: conflict-free), the rest in a global
// overflow column sized by the builder's exact worst case (pt_bvh_info.stack_need).
struct StackCtx { int32_t *lds; uint32_t stride, tid, col; }; // col: the lane's column of the overflow area, unique per thread of a launch
PT_DEV void push_slow(const StackCtx &k, uint32_t &sp, int32_t v)
{
    if (sp < kStackLds) k.lds[sp * k.stride + k.tid] = v;
    else {
        const PathState &ps = cold().ps;
        const uint32_t e = sp - kStackLds;
        if (e < ps.stack_ovf_entries) ps.stack_ovf[(size_t)e * ((size_t)kShards * ps.shard_cap) + k.col] = v;
        else { atomicOr(&ps.counters[kCntError], 1u); return; }
    }
    ++sp;
}
PT_DEV int32_t pop_slow(const StackCtx &k, uint32_t &sp)
{
    if (sp == 0) return PT_BVH_EMPTY;
    --sp;
    if (sp < kStackLds) return k.lds[sp * k.stride + k.tid];
    const PathState &ps = cold().ps;
    return ps.stack_ovf[(size_t)(sp - kStackLds) * ((size_t)kShards * ps.shard_cap) + k.col];
}

// One inner-node visit of the lanes that call it: fetch, slab tests, children pushed farthest first, nearest (or the popped
// stack top) becomes `cur`. Stack fast path: while every calling lane still has room for a whole node's pushes in its LDS column
// (the common case), a push is an unconditional LDS store plus a predicated increment and a pop is a plain LDS load — no
// LDS-or-spill branch per push.
template <int L>
PT_DEV void node_visit_rows(const float4 *__restrict__ base, float4 r0, float4 r1, float4 r2, float4 r3, const StackCtx &k, const RaySetup &rs,
                            float t_best, int32_t &cur, uint32_t &sp);
template <int L>
PT_DEV void node_step(const float4 *__restrict__ nodes, const StackCtx &k, const RaySetup &rs, float t_best, int32_t &cur, uint32_t &sp)
{
    const float4 *base = nodes + (size_t)cur * node_rows<L>();
    node_visit_rows<L>(base, base[0], base[1], base[2], base[3], k, rs, t_best, cur, sp);
}
// The visit of the node whose first four rows are r0..r3 (`base`: where layouts with more rows find the rest).
template <int L>
PT_DEV void node_visit_rows(const float4 *__restrict__ base, float4 r0, float4 r1, float4 r2, float4 r3, const StackCtx &k, const RaySetup &rs,
                            float t_best, int32_t &cur, uint32_t &sp)
{
    constexpr int N = fanout<L>();
    const bool deep = __any((int)(sp + (uint32_t)(N - 1) > kStackLds)) != 0;
    if constexpr (L == PT_BVH_WIDTH_8O) { // children arrive in visit order, hits and misses mixed: walk them far to near, the nearest hit stays in `cur`
        uint32_t hits;
        int32_t ref[N];
        visit_node_oct8(base, r0, r1, r2, r3, rs, t_best, hits, ref);
        int32_t prev = PT_BVH_EMPTY;
        if (!deep) {
#pragma unroll
            for (int p = N - 1; p >= 0; --p) {
                const bool h = (hits >> p) & 1u;
                k.lds[sp * k.stride + k.tid] = prev;
                sp += (h && prev != PT_BVH_EMPTY) ? 1u : 0u;
                prev = h ? ref[p] : prev;
            }
            if (prev != PT_BVH_EMPTY) cur = prev;
            else if (sp) { --sp; cur = k.lds[sp * k.stride + k.tid]; }
            else cur = PT_BVH_EMPTY;
        } else {
#pragma unroll
            for (int p = N - 1; p >= 0; --p)
                if ((hits >> p) & 1u) { if (prev != PT_BVH_EMPTY) push_slow(k, sp, prev); prev = ref[p]; }
            cur = prev != PT_BVH_EMPTY ? prev : pop_slow(k, sp);
        }
        return;
    }
    uint32_t key[N];
    int32_t ref[N];
    visit_node<L>(base, r0, r1, r2, r3, rs, t_best, key, ref); // keys sorted ascending, misses = 0xFFFFFFFF at the end
    if (!deep) {
#pragma unroll
        for (int i = N - 1; i >= 1; --i) { // farthest first, nearest stays in `cur`
            k.lds[sp * k.stride + k.tid] = ref[i];
            sp += key[i] != 0xFFFFFFFFu ? 1u : 0u;
        }
        if (key[0] != 0xFFFFFFFFu) cur = ref[0];
        else if (sp) { --sp; cur = k.lds[sp * k.stride + k.tid]; }
        else cur = PT_BVH_EMPTY;
    } else {
#pragma unroll
        for (int i = N - 1; i >= 1; --i)
            if (key[i] != 0xFFFFFFFFu) push_slow(k, sp, ref[i]);
        cur = (key[0] != 0xFFFFFFFFu) ? ref[0] : pop_slow(k, sp);
    }
}

// One leaf of the lanes that call it: its triangles in array order, then the next stack entry. Returns the triangle tests made.
PT_DEV uint32_t leaf_step(const float4 *__restrict__ tris, const StackCtx &k, V3 o, V3 d, Hit &h, int32_t &cur, uint32_t &sp)
{
    const uint32_t enc = (uint32_t)~cur;
    uint32_t first = enc >> 3, more = enc & 7u, n = 0;
    for (;;) {
        const float4 *base = tris + (size_t)first * 4;
        const float4 r0 = base[0], r1 = base[1], r2 = base[2];
        tri_test(r0, r1, r2, first, o, d, h);
        ++n;
        if (more == 0u) break;
        ++first; --more;
    }
    const bool deep = __any((int)(sp > kStackLds)) != 0;
    if (!deep) { if (sp) { --sp; cur = k.lds[sp * k.stride + k.tid]; } else cur = PT_BVH_EMPTY; }
    else cur = pop_slow(k, sp);
    return n;
}

// Finish mode (n_alive <= finish_below): a launch keeps going until its paths end, but never for more than this many
// vertices per lane, so that a launch stays bounded whatever spp and max_depth are; the host loop simply goes on.
constexpr uint32_t kFinishVertices = 256;

template <int L, bool COUNT, int FUSE>
__global__ void __launch_bounds__(kExtBlock) __attribute__((amdgpu_waves_per_eu(PT_EXT_WAVES(L, FUSE), PT_EXT_WAVES(L, FUSE)))) k_extend(ExtArgs a)
{
    // hot arguments (stay in SGPRs across the traversal loop); everything else goes through cold()
    const float4 *__restrict__ nodes = a.sc.nodes, *__restrict__ tris = a.sc.tris, *__restrict__ spheres = a.sc.spheres;
    const uint32_t n_spheres = a.sc.n_spheres, n_tris = a.sc.n_tris, n_nodes = a.sc.n_nodes;
    const uint32_t it = a.it;
    const uint32_t parity = it & 1u, ccur = it % 3u, cnext = (it + 1u) % 3u, czero = (it + 2u) % 3u;
    __shared__ int32_t s_stack[kStackLds * kExtBlock];
    __shared__ uint32_t s_stash[(FUSE != SHADE_NONE ? 5 : 1) * kExtBlock];
    uint32_t *stash = s_stash;
    uint32_t shard, bx, nbx;
    block_pos(a.ps, shard, bx, nbx);
    const uint32_t tid = threadIdx.x;
    const uint32_t gid = bx * kExtBlock + tid;                  // index inside the shard's queue
    uint32_t n, n_alive, slot = kInvalidSlot, n_bounces;
    bool do_compact;
    uint32_t qbase; // first entry of this shard's region of the queues (entries: 32-bit offsets like the slot arrays, see at())
    {
        const PathState &ps = cold().ps;
        n = ps.counters[cnt_ext_index(ccur, shard)]; n_alive = ps.counters[cnt_alive_index(ccur, shard)];
        do_compact = FUSE != SHADE_NONE && want_compact(ps, n, n_alive, ps.counters[cnt_prev_alive_index(ccur, shard)], ps.repack_sticky != 0u, a.compact != 0u);
        if (gid == 0) {
            ps.counters[cnt_prev_alive_index(cnext, shard)] = n_alive;
            ps.counters[cnt_ext_index(czero, shard)] = 0u;         // the queue after next
            ps.counters[cnt_alive_index(czero, shard)] = 0u;
            fold_traced(ps, shard, it, n, n_alive);
            if (FUSE == SHADE_NONE) *traced_counter(ps, cnext, shard) = n_alive; // one ray per alive entry; the fused kernel
                                                                                 // counts what it traces, per wavefront
            if (do_compact && n_alive) atomicAdd(&ps.counters[kCntCompactions], 1u);
        }
        if (bx * kExtBlock >= n || n_alive == 0u) return;
        qbase = shard * ps.shard_cap;
        if (gid < n) slot = at(ps.q_ext[parity], qbase + gid);
        // FUSE: up to `bounces` path vertices per launch with the path state in registers (a terminated path continues with
        // its stream's next camera ray, so most lanes stay busy); the state goes back to memory once, at the end.
        // Once few paths are left in the shard (the frame's tail) the launch runs them to their end instead (bounded by
        // kFinishVertices): launches that small cost more in launch latency and host round trips than the lanes idling
        // behind a wave's longest path. A launch that STARTS sparse (many paths ended during the previous one) runs every
        // wavefront at the cost of its few live lanes; it advances one vertex only, re-packs, and the dense launch after
        // it does the real work.
        const bool sparse = do_compact && (float)n_alive < ps.sparse_below * (float)n;
        n_bounces = FUSE == SHADE_NONE ? 1u : (n_alive <= ps.finish_below ? kFinishVertices : sparse ? 1u : a.bounces);
    }
    const bool active = slot != kInvalidSlot;                  // holes: paths that ended since the queue was last compacted
    const StackCtx stk{ s_stack, kExtBlock, tid, qbase + gid };

    unsigned long long c_nodes = 0, c_tris = 0, c_sph = 0, c_wave_iters = 0, c_wave_iters_late = 0, c_idle_leaf = 0, c_idle_done = 0;
    __shared__ int32_t s_state[COUNT ? 64 : 1]; // COUNT diagnostics: every lane's `cur` (1 = no ray), readable by the lane that counts
    PathRegs r;
    r.o = v3(0.f, 0.f, 0.f); r.d = v3(0.f, 0.f, 1.f); r.T = v3(0.f, 0.f, 0.f); r.key = r.sample = r.depth = 0u;
    if (active) {
        const PathState &ps = cold().ps;
        if (FUSE == SHADE_NONE) { r.o = xyz(at(ps.ray_o, slot)); r.d = xyz(at(ps.ray_d, slot)); }
        else if (it == 0u) { // the frame's first launch makes the state of its slots itself (api.cpp: q_init) and starts their radiance sums
            path_init(cold().sc, cold().fp, slot, r);
            if (!cold().fp.accumulate) at(ps.acc, slot) = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        else path_load(ps, slot, r);
    }
    bool alive = active;
    uint32_t wave_rays = 0;
    for (uint32_t bounce = 0; bounce < n_bounces; ++bounce) {
    const uint64_t alive_mask = __ballot(alive);
    if (alive_mask == 0) break;
    wave_rays += (uint32_t)__popcll(alive_mask);
    if (COUNT && kExtBlock == 64u) { s_state[tid] = alive ? 0 : 1; __syncthreads(); } // 1: neither a node, a leaf nor PT_BVH_EMPTY
    if (alive) {
        const V3 o = r.o, d = r.d;
        Hit h{ __builtin_inff(), PT_MISS, PT_MISS };
        if (FUSE != SHADE_NONE) { // park what traversal does not need in LDS: 5 VGPRs less while the wave gathers nodes
            stash[0 * kExtBlock + tid] = __float_as_uint(r.T.x); stash[1 * kExtBlock + tid] = __float_as_uint(r.T.y);
            stash[2 * kExtBlock + tid] = __float_as_uint(r.T.z); stash[3 * kExtBlock + tid] = r.key;
            stash[4 * kExtBlock + tid] = (r.sample << 8) | r.depth;
            asm volatile("" ::: "memory"); // the values must really leave the registers: no store-to-load forwarding across the traversal
        }

        { const uint32_t ns = spheres_test(spheres, n_spheres, n_tris, o, d, h); if (COUNT) c_sph += ns; }

        const RaySetup rs = ray_setup(o, d);
        int32_t cur = n_nodes ? 0 : PT_BVH_EMPTY;
        uint32_t sp = 0, steps = 0;
        // Every ray starts at node 0, whose rows are the same for the whole wave: this first visit reads them with one scalar load
        // (no vector gather to wait for) and runs at full width before the divergent node loop (headline 17.03 -> 16.88 ms, Cornell
        // 7.31 -> 7.02, Cornell+glass+metal 32.6 -> 31.9).
        if (n_nodes) {
            ++steps;
            if (COUNT) { c_nodes++; if (lane_id() == (uint32_t)(__ffsll((long long)__ballot(1)) - 1)) c_wave_iters++; }
            const uniform_f4 root = as_uniform(nodes);
            node_visit_rows<L>(nodes, uniform_load(root, 0), uniform_load(root, 1), uniform_load(root, 2), uniform_load(root, 3), stk, rs, h.t, cur, sp);
            if (COUNT) s_state[tid & 63u] = cur;
        }

        // while-while: a lane that reaches a leaf waits at the reconvergence point of the node loop until every lane of
        // the wave is at a leaf or done; then the leaves are tested together. Each lane still makes exactly the visits,
        // in exactly the order, of docs/SPEC.md §4.1 (only the interleaving across lanes changes), so hits and visit
        // counters are the oracle's. Why: with one loop for both kinds of step the wave issued the ~160-instruction node
        // code AND the ~125-instruction triangle code in practically every iteration, the latter for ~1 lane in 8; the
        // kernel is VALU-issue bound (rocprofv3: 34 % of wave time waits for an issue slot, 43 % of VALU lanes active).
        bool late_cycle = false; // (COUNT diagnostics) set once this wave has run a leaf phase for the current rays
        for (;;) {
            while ((uint32_t)cur < (uint32_t)PT_BVH_EMPTY) { // ---- node phase: inner-node refs are 0 .. 0x7ffffffe (EMPTY = 0x7fffffff)
                if (++steps > (1u << 22)) { atomicOr(&cold().ps.counters[kCntError], 2u); cur = PT_BVH_EMPTY; break; }
                if (COUNT) { // one lane per wave-iteration counts it; [1]: iterations after the wave's first leaf phase of this ray
                    c_nodes++;
                    if (lane_id() == (uint32_t)(__ffsll((long long)__ballot(1)) - 1)) {
                        c_wave_iters++; if (late_cycle) c_wave_iters_late++;
                        // why the other lanes of this iteration are idle (their state sits in LDS: they are masked off here)
                        uint32_t n_leaf = 0, n_done = 0;
                        for (uint32_t l = 0; l < 64u; ++l) { const int32_t c2 = s_state[l]; n_leaf += c2 < 0 ? 1u : 0u; n_done += c2 == PT_BVH_EMPTY ? 1u : 0u; }
                        c_idle_leaf += n_leaf; c_idle_done += n_done;
                    }
                }
                node_step<L>(nodes, stk, rs, h.t, cur, sp);
                if (COUNT) s_state[tid & 63u] = cur;
            }
            if (cur == PT_BVH_EMPTY) break;
            if (COUNT) late_cycle = true;
            const uint32_t nt = leaf_step(tris, stk, o, d, h, cur, sp); // ---- leaf phase
            if (COUNT) { c_tris += nt; s_state[tid & 63u] = cur; }
        }

        if (FUSE == SHADE_NONE) at(cold().ps.hit, slot) = make_float2(h.t, __uint_as_float(h.ref)); // k_shade walks the same queue in the same order
        else {
            r.T = v3(__uint_as_float(stash[0 * kExtBlock + tid]), __uint_as_float(stash[1 * kExtBlock + tid]), __uint_as_float(stash[2 * kExtBlock + tid]));
            r.key = stash[3 * kExtBlock + tid];
            const uint32_t sdv = stash[4 * kExtBlock + tid];
            r.sample = sdv >> 8; r.depth = sdv & 255u;
            uint32_t defer = 0u;
            const ExtArgs &c = cold();
            alive = shade_one<FUSE>(c.sc, c.ps, c.fp, slot, r, h.t, h.ref, B_LAMBERT, defer);
        }
    }
    }
    const PathState &ps = cold().ps;
    if (COUNT && active) {
        if (c_wave_iters) atomicAdd(reinterpret_cast<unsigned long long *>(ps.counters + kCntWaveNodeIters), c_wave_iters);
        if (c_wave_iters_late) atomicAdd(reinterpret_cast<unsigned long long *>(ps.counters + kCntWaveNodeIters + 2), c_wave_iters_late);
        if (c_idle_leaf) atomicAdd(reinterpret_cast<unsigned long long *>(ps.counters + kCntIdleLeaf), c_idle_leaf);
        if (c_idle_done) atomicAdd(reinterpret_cast<unsigned long long *>(ps.counters + kCntIdleDone), c_idle_done);
        atomicAdd(reinterpret_cast<unsigned long long *>(ps.counters + kCntNodes), c_nodes);
        atomicAdd(reinterpret_cast<unsigned long long *>(ps.counters + kCntTris), c_tris);
        atomicAdd(reinterpret_cast<unsigned long long *>(ps.counters + kCntSph), c_sph);
    }
    if (FUSE != SHADE_NONE) {
        if (alive) path_store(ps, slot, r);
        uint32_t sort_key = 0u;
        if (PT_REPACK_SORT) sort_key = (r.d.x < 0.f ? 32u : 0u) | (r.d.y < 0.f ? 16u : 0u) | (r.d.z < 0.f ? 8u : 0u) |
                                       (r.o.x < 0.f ? 4u : 0u) | (r.o.y < 0.f ? 2u : 0u) | (r.o.z < 0.f ? 1u : 0u);
        queue_next(ps, shard, cnext, &at(ps.q_ext[parity ^ 1u], qbase), gid, n, alive, slot, do_compact, sort_key);
        if (wave_rays && lane_id() == 0u) atomicAdd(traced_counter(ps, cnext, shard), (unsigned long long)wave_rays);
    }
}


// ------------------------------------------------------------------------------------------------
// k_extend_packed: same per-ray work and the same per-ray traversal order as k_extend (so hits AND visit counters
// are identical), but a wavefront owns `chunk` consecutive queue entries instead of 64 and keeps its lanes packed:
// whenever >= kRefillIdle lanes have finished their ray (ballot + popcount), the finished lanes retire (hit record,
// bucket push) and idle lanes pull the next entries of the chunk by mbcnt prefix. With one ray per lane a wave runs
// as long as its slowest ray (measured 19 % lane utilisation on the 1M-triangle soup); with refill the wave's time
// tends to the chunk's mean. No atomics and no cross-wave traffic are added: the chunk is private to the wave.
#ifndef PT_REFILL_IDLE
#define PT_REFILL_IDLE 16
#endif
constexpr uint32_t kRefillIdle = PT_REFILL_IDLE;

//
// FUSE (as in k_extend): a lane that finishes a ray shades it on the spot and, while the path lives and its budget of
// `bounces` vertices lasts, starts the path's next ray itself; only then does it hand the slot to the next iteration
// and pull a new queue entry. Path state stays in registers / LDS across those bounces, and no bounce waits for the wave.
#ifndef PT_PACKED_WAVES
#define PT_PACKED_WAVES(FUSE) ((FUSE) == SHADE_INLINE ? 6 : 7)
#endif
template <int L, bool COUNT, int FUSE>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(PT_PACKED_WAVES(FUSE), PT_PACKED_WAVES(FUSE))))
k_extend_packed(ExtArgs a)
{
    // hot arguments stay in SGPRs; everything else is read through cold() where it is used (see ExtArgs)
    const float4 *__restrict__ nodes = a.sc.nodes, *__restrict__ tris = a.sc.tris, *__restrict__ spheres = a.sc.spheres;
    const uint32_t n_spheres = a.sc.n_spheres, n_tris = a.sc.n_tris, n_nodes = a.sc.n_nodes;
    const uint32_t it = a.it, chunk = a.chunk, compact = a.compact, bounces = a.bounces;
    constexpr int N = fanout<L>();
    __shared__ int32_t s_stack[kStackLds * 64];
    __shared__ uint32_t s_stash[(FUSE != SHADE_NONE ? 5 : 1) * 64];
    uint32_t *stash = s_stash;
    const uint32_t parity = it & 1u, ccur = it % 3u, cnext = (it + 1u) % 3u, czero = (it + 2u) % 3u;
    uint32_t shard, bx, nbx;
    block_pos(a.ps, shard, bx, nbx);
    const uint32_t n = cold().ps.counters[cnt_ext_index(ccur, shard)], n_alive = cold().ps.counters[cnt_alive_index(ccur, shard)];
    const uint32_t lane = threadIdx.x;
    const bool do_compact = FUSE != SHADE_NONE && want_compact(cold().ps, n, n_alive, 0u, false, compact != 0u); // holes cost this kernel one skipped pull: plain ratio
    if (bx == 0 && lane == 0) {
        cold().ps.counters[cnt_prev_alive_index(cnext, shard)] = n_alive; // for a one-ray-per-lane launch that may follow (probe frames)
        cold().ps.counters[cnt_ext_index(czero, shard)] = 0u;
        cold().ps.counters[cnt_alive_index(czero, shard)] = 0u;
        fold_traced(cold().ps, shard, it, n, n_alive);
        if (FUSE == SHADE_NONE) *traced_counter(cold().ps, cnext, shard) = n_alive; // one ray per alive entry
        else if (!do_compact) cold().ps.counters[cnt_ext_index(cnext, shard)] = n;  // carried in place: the length stays
        if (do_compact && n_alive) atomicAdd(&cold().ps.counters[kCntCompactions], 1u);
    }
    uint32_t next = bx * chunk;                               // wave-uniform cursor into the shard's queue
    if (next >= n || n_alive == 0u) return;
    const uint32_t end = min(n, next + chunk);
    const size_t qbase = (size_t)shard * cold().ps.shard_cap;
    const uint32_t *queue = cold().ps.q_ext[parity] + qbase;
    uint32_t *q_next = cold().ps.q_ext[parity ^ 1u] + qbase;
    const size_t uid = qbase + (size_t)bx * 64u + lane;        // unique per thread of this launch (chunk >= 64)
    const size_t ovf_stride = (size_t)kShards * cold().ps.shard_cap;
    const uint32_t budget0 = FUSE == SHADE_NONE ? 1u : (n_alive <= cold().ps.finish_below ? kFinishVertices : bounces);

    bool has = false;                                          // lane holds a ray
    uint32_t slot = 0, pos = 0, budget = 0, sp = 0, steps = 0;
    int32_t cur = PT_BVH_EMPTY;
    PathRegs r;
    r.o = v3(0.f, 0.f, 0.f); r.d = v3(0.f, 0.f, 1.f); r.T = v3(0.f, 0.f, 0.f); r.key = r.sample = r.depth = 0u;
    V3 &o = r.o, &d = r.d;
    RaySetup rs = ray_setup(o, d);
    Hit h{ __builtin_inff(), PT_MISS, PT_MISS };
    unsigned long long c_nodes = 0, c_tris = 0, c_sph = 0;
    uint32_t wave_rays = 0, wave_alive = 0;

    auto push = [&](int32_t v) {
        if (sp < kStackLds) s_stack[sp * 64u + lane] = v;
        else {
            const uint32_t e = sp - kStackLds;
            if (e < cold().ps.stack_ovf_entries) cold().ps.stack_ovf[(size_t)e * ovf_stride + uid] = v;
            else { atomicOr(&cold().ps.counters[kCntError], 1u); return; }
        }
        ++sp;
    };
    auto pop = [&]() -> int32_t {
        if (sp == 0) return PT_BVH_EMPTY;
        --sp;
        return sp < kStackLds ? s_stack[sp * 64u + lane] : cold().ps.stack_ovf[(size_t)(sp - kStackLds) * ovf_stride + uid];
    };

    auto start_ray = [&]() { // the lane's ray is in r.o, r.d
        if (FUSE != SHADE_NONE) { // park what traversal does not need (as in k_extend)
            stash[0 * 64u + lane] = __float_as_uint(r.T.x); stash[1 * 64u + lane] = __float_as_uint(r.T.y);
            stash[2 * 64u + lane] = __float_as_uint(r.T.z); stash[3 * 64u + lane] = r.key;
            stash[4 * 64u + lane] = (r.sample << 8) | r.depth;
            asm volatile("" ::: "memory"); // no store-to-load forwarding: the values leave the registers
        }
        h = Hit{ __builtin_inff(), PT_MISS, PT_MISS };
        { const uint32_t ns = spheres_test(spheres, n_spheres, n_tris, o, d, h); if (COUNT) c_sph += ns; }
        rs = ray_setup(o, d);
        cur = n_nodes ? 0 : PT_BVH_EMPTY;
        sp = 0; steps = 0;
    };

    for (;;) {
        // ---- finished rays (wave-uniform point): shade, continue the path or retire the slot
        const bool fin = has && cur == PT_BVH_EMPTY;
        bool retired = false, retire_alive = false;
        if (fin) {
            if (FUSE == SHADE_NONE) {
                at(cold().ps.hit, slot) = make_float2(h.t, __uint_as_float(h.ref));
                has = false;
            } else {
                r.T = v3(__uint_as_float(stash[0 * 64u + lane]), __uint_as_float(stash[1 * 64u + lane]), __uint_as_float(stash[2 * 64u + lane]));
                r.key = stash[3 * 64u + lane];
                const uint32_t sdv = stash[4 * 64u + lane];
                r.sample = sdv >> 8; r.depth = sdv & 255u;
                uint32_t defer = 0u;
                const bool alive = shade_one<FUSE>(cold().sc, cold().ps, cold().fp, slot, r, h.t, h.ref, B_LAMBERT, defer);
                --budget;
                if (alive && budget != 0u) start_ray(); // next vertex of the same path, state still in registers
                else {
                    if (alive) path_store(cold().ps, slot, r);
                    retired = true; retire_alive = alive; has = false;
                }
            }
        }
        if (FUSE != SHADE_NONE) {
            wave_rays += (uint32_t)__popcll(__ballot(fin && has));
            wave_alive += (uint32_t)__popcll(__ballot(retire_alive));
            if (do_compact) wave_push(&cold().ps.counters[cnt_ext_index(cnext, shard)], q_next, retire_alive, slot);
            else if (retired) q_next[pos] = retire_alive ? slot : kInvalidSlot; // the entry keeps its queue position
        }
        // ---- refill idle lanes from the wave's chunk
        const uint64_t idle = __ballot(!has);
        const uint32_t avail = end - next;
        if (idle && avail) {
            const uint32_t prefix = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));
            const bool pull = !has && prefix < avail;
            if (pull) { pos = next + prefix; slot = queue[pos]; }
            if (pull && slot == kInvalidSlot) { // a hole leaves the lane idle until the next refill
                if (FUSE != SHADE_NONE && !do_compact) q_next[pos] = kInvalidSlot;
            } else if (pull) {
                if (FUSE == SHADE_NONE) { o = xyz(at(cold().ps.ray_o, slot)); d = xyz(at(cold().ps.ray_d, slot)); }
                else if (it == 0u) { // see k_extend
                    path_init(cold().sc, cold().fp, slot, r);
                    if (!cold().fp.accumulate) at(cold().ps.acc, slot) = make_float4(0.f, 0.f, 0.f, 0.f);
                }
                else path_load(cold().ps, slot, r);
                budget = budget0;
                start_ray();
                has = true;
            }
            if (FUSE != SHADE_NONE) wave_rays += (uint32_t)__popcll(__ballot(pull && has));
            next += min((uint32_t)__popcll(idle), avail);
        }
        if (!__ballot(has)) {
            if (next < end) continue; // everything pulled was a hole
            break;
        }
        // ---- traverse until enough lanes went idle to make a refill worth its latency (or nothing is left to pull)
        const uint32_t want = (next < end) ? kRefillIdle : 64u;
        for (;;) {
            if (cur != PT_BVH_EMPTY) {
                const bool inner = cur >= 0;
                const uint32_t enc = (uint32_t)~cur, first = enc >> 3, more = enc & 7u;
                const float4 *base = inner ? nodes + (size_t)cur * node_rows<L>() : tris + (size_t)first * 4;
                const float4 r0 = base[0], r1 = base[1], r2 = base[2], r3 = base[3];
                const bool deep = __any((int)(sp + (uint32_t)(N - 1) > kStackLds)) != 0; // stack fast path as in k_extend
                auto pop_fast = [&]() -> int32_t {
                    if (sp == 0) return PT_BVH_EMPTY;
                    --sp;
                    return s_stack[sp * 64u + lane];
                };
                if (++steps > (1u << 22)) { atomicOr(&cold().ps.counters[kCntError], 2u); cur = PT_BVH_EMPTY; }
                else if (inner && L == PT_BVH_WIDTH_8O) { // octant order: see node_visit_rows
                    uint32_t hits;
                    int32_t ref[N];
                    if constexpr (L == PT_BVH_WIDTH_8O) visit_node_oct8(base, r0, r1, r2, r3, rs, h.t, hits, ref);
                    if (COUNT) c_nodes++;
                    int32_t prev = PT_BVH_EMPTY;
                    if (!deep) {
#pragma unroll
                        for (int p = N - 1; p >= 0; --p) {
                            const bool hp = (hits >> p) & 1u;
                            s_stack[sp * 64u + lane] = prev;
                            sp += (hp && prev != PT_BVH_EMPTY) ? 1u : 0u;
                            prev = hp ? ref[p] : prev;
                        }
                        cur = prev != PT_BVH_EMPTY ? prev : pop_fast();
                    } else {
#pragma unroll
                        for (int p = N - 1; p >= 0; --p)
                            if ((hits >> p) & 1u) { if (prev != PT_BVH_EMPTY) push(prev); prev = ref[p]; }
                        cur = prev != PT_BVH_EMPTY ? prev : pop();
                    }
                } else if (inner) {
                    uint32_t key[N];
                    int32_t ref[N];
                    visit_node<L>(base, r0, r1, r2, r3, rs, h.t, key, ref);
                    if (COUNT) c_nodes++;
                    if (!deep) {
#pragma unroll
                        for (int i = N - 1; i >= 1; --i) {
                            s_stack[sp * 64u + lane] = ref[i];
                            sp += key[i] != 0xFFFFFFFFu ? 1u : 0u;
                        }
                        cur = (key[0] != 0xFFFFFFFFu) ? ref[0] : pop_fast();
                    } else {
#pragma unroll
                        for (int i = N - 1; i >= 1; --i)
                            if (key[i] != 0xFFFFFFFFu) push(ref[i]);
                        cur = (key[0] != 0xFFFFFFFFu) ? ref[0] : pop();
                    }
                } else {
                    tri_test(r0, r1, r2, first, o, d, h);
                    if (COUNT) c_tris++;
                    cur = more ? (int32_t)~(((first + 1u) << 3) | (more - 1u)) : (deep ? pop() : pop_fast());
                }
            }
            const uint32_t busy = (uint32_t)__popcll(__ballot(has && cur != PT_BVH_EMPTY));
            if (FUSE == SHADE_NONE) { if (busy == 0u || 64u - busy >= want) break; }
            else { // lanes that can do something at the uniform point: finished rays to shade, empty lanes that can pull
                const uint32_t waiting = (uint32_t)__popcll(__ballot(has && cur == PT_BVH_EMPTY));
                const uint32_t empty = (next < end) ? (uint32_t)__popcll(__ballot(!has)) : 0u;
                if (busy == 0u || waiting + empty >= kRefillIdle) break;
            }
        }
    }
    if (COUNT) {
        atomicAdd(reinterpret_cast<unsigned long long *>(cold().ps.counters + kCntNodes), c_nodes);
        atomicAdd(reinterpret_cast<unsigned long long *>(cold().ps.counters + kCntTris), c_tris);
        atomicAdd(reinterpret_cast<unsigned long long *>(cold().ps.counters + kCntSph), c_sph);
    }
    if (FUSE != SHADE_NONE && lane == 0u) {
        if (wave_rays) atomicAdd(traced_counter(cold().ps, cnext, shard), (unsigned long long)wave_rays);
        if (wave_alive) atomicAdd(&cold().ps.counters[cnt_alive_index(cnext, shard)], wave_alive);
    }
}

// ------------------------------------------------------------------------------------------------
// k_extend_pool<L, COUNT, FUSE>: the fused pipeline with a per-wavefront ray pool. A wavefront owns kPool (= 64 P) consecutive
// queue entries — with 8 sample streams, P streams of one 8x8 pixel block — and advances ALL of them by `bounces` path vertices:
//   traverse : lanes pull rays from the pool (ballot + mbcnt prefix over a live list in LDS) and re-fill whenever
//              >= kPoolRefill lanes have finished theirs, so a wave's traversal time tends to the pool's MEAN ray length instead
//              of its longest ray (one ray per lane: 48 % node-loop lane utilisation on the 1M-triangle Cornell box, 12 % on the
//              soup). While-while with a threshold: lanes at a leaf wait until kPoolLeafWait of them are there (or no lane has a
//              node left), then their leaves are tested together. Every ray still makes exactly the visits of docs/SPEC.md §4.1 in
//              exactly that order, so hits and visit counters are the oracle's.
//   shade    : then the whole pool is shaded P x 64 lanes wide, coalesced in slot order (k_extend_packed shades each ray when
//              it finishes, i.e. at the width of the refill threshold — that is what made it lose on shallow scenes). Hit
//              records wait in LDS (8 B per entry); the path state takes one round trip through the slot-indexed arrays per
//              vertex, which stays in L2 (a pool's state is 6.6 KB), and the new ray's sphere tests are done here, at full width.
// Same queue contract as k_extend: entry j of the next queue = entry j of this one (slot or hole), or a dense re-pack.
#ifndef PT_POOL_WAVES
#define PT_POOL_WAVES(FUSE) ((FUSE) == SHADE_INLINE ? 6 : 7)
#endif
#ifndef PT_POOL_REFILL
#define PT_POOL_REFILL 16
#endif
#ifndef PT_POOL_LEAFWAIT
#define PT_POOL_LEAFWAIT 16
#endif
constexpr uint32_t kPoolRefill = PT_POOL_REFILL, kPoolLeafWait = PT_POOL_LEAFWAIT;

template <int L, bool COUNT, int FUSE>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(PT_POOL_WAVES(FUSE), PT_POOL_WAVES(FUSE)))) k_extend_pool(ExtArgs a)
{
    static_assert(FUSE != SHADE_NONE, "the pool kernel always shades");
    constexpr uint32_t E = kPool, P = E / 64u;
    static_assert(E % 64u == 0 && E >= 64u && E <= 256u, "kPool: 64, 128, 192 or 256 (live list holds byte indices)");
    const float4 *__restrict__ nodes = a.sc.nodes, *__restrict__ tris = a.sc.tris, *__restrict__ spheres = a.sc.spheres;
    const uint32_t n_spheres = a.sc.n_spheres, n_tris = a.sc.n_tris, n_nodes = a.sc.n_nodes;
    const uint32_t it = a.it;
    const uint32_t parity = it & 1u, ccur = it % 3u, cnext = (it + 1u) % 3u, czero = (it + 2u) % 3u;
    __shared__ int32_t s_stack[kStackLds * 64];
    __shared__ float s_ht[E];       // per pool entry: closest hit so far (before traversal: of the sphere list), t
    __shared__ uint32_t s_href[E];  //                 ... and its primitive ref
    __shared__ uint32_t s_slot[E];  // slot of the entry, kInvalidSlot once its path and stream have ended
    __shared__ uint8_t s_live[E];   // entries that hold a ray for the coming traversal, densely
    uint32_t shard, bx, nbx;
    block_pos(a.ps, shard, bx, nbx);
    const uint32_t lane = threadIdx.x;
    uint32_t n, n_alive, n_bounces;
    bool do_compact;
    size_t qbase;
    {
        const PathState &ps = cold().ps;
        n = ps.counters[cnt_ext_index(ccur, shard)]; n_alive = ps.counters[cnt_alive_index(ccur, shard)];
        do_compact = want_compact(ps, n, n_alive, ps.counters[cnt_prev_alive_index(ccur, shard)], ps.repack_sticky != 0u, a.compact != 0u);
        if (bx == 0 && lane == 0) {
            ps.counters[cnt_prev_alive_index(cnext, shard)] = n_alive;
            ps.counters[cnt_ext_index(czero, shard)] = 0u;
            ps.counters[cnt_alive_index(czero, shard)] = 0u;
            fold_traced(ps, shard, it, n, n_alive);
            if (!do_compact) ps.counters[cnt_ext_index(cnext, shard)] = n; // carried in place: the length stays
            if (do_compact && n_alive) atomicAdd(&ps.counters[kCntCompactions], 1u);
        }
        if (bx * E >= n || n_alive == 0u) return;
        qbase = (size_t)shard * ps.shard_cap;
        n_bounces = n_alive <= ps.finish_below ? kFinishVertices : a.bounces;
    }
    const StackCtx stk{ s_stack, 64u, lane, (uint32_t)qbase + bx * 64u + lane };
    unsigned long long c_nodes = 0, c_tris = 0, c_sph = 0, c_wave_iters = 0;

    auto spheres_of = [&](V3 o, V3 d, uint32_t e) { // the sphere list is tested when a ray is made (full width); traversal starts from its result
        Hit h{ __builtin_inff(), PT_MISS, PT_MISS };
        spheres_test(spheres, n_spheres, n_tris, o, d, h);
        s_ht[e] = h.t; s_href[e] = h.ref;
    };
    auto live_append = [&](bool pred, uint32_t e, uint32_t &count) {
        const uint64_t m = __ballot(pred);
        if (pred) s_live[count + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] = (uint8_t)e;
        count += (uint32_t)__popcll(m);
    };

    // ---- the pool: queue entries -> slots, sphere tests of the rays they hold, live list
    uint32_t n_live = 0;
    {
        const PathState &ps = cold().ps;
#pragma unroll
        for (uint32_t hh = 0; hh < P; ++hh) {
            const uint32_t e = hh * 64u + lane, g = bx * E + e;
            const uint32_t slot = g < n ? ps.q_ext[parity][qbase + g] : kInvalidSlot;
            s_slot[e] = slot;
            const bool valid = slot != kInvalidSlot;
            if (valid) spheres_of(xyz(at(ps.ray_o, slot)), xyz(at(ps.ray_d, slot)), e);
            live_append(valid, e, n_live);
        }
    }
    __syncthreads();

    uint32_t wave_rays = 0;
    for (uint32_t bounce = 0; bounce < n_bounces && n_live; ++bounce) {
        wave_rays += n_live;
        // ---- traverse the pool
        uint32_t next = 0, e = 0, sp = 0, steps = 0;
        bool has = false;
        int32_t cur = PT_BVH_EMPTY;
        V3 o = v3(0.f, 0.f, 0.f), d = v3(0.f, 0.f, 1.f);
        RaySetup rs = ray_setup(o, d);
        Hit h{ __builtin_inff(), PT_MISS, PT_MISS };
        for (;;) {
            if (has && cur == PT_BVH_EMPTY) { s_ht[e] = h.t; s_href[e] = h.ref; has = false; } // finished: hit record to the pool
            const uint64_t idle = __ballot(!has);
            const uint32_t avail = n_live - next;
            if (idle && avail) {
                const uint32_t prefix = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));
                if (!has && prefix < avail) {
                    const PathState &ps = cold().ps;
                    e = s_live[next + prefix];
                    const uint32_t slot = s_slot[e];
                    o = xyz(at(ps.ray_o, slot)); d = xyz(at(ps.ray_d, slot));
                    const uint32_t ref0 = s_href[e];
                    h = Hit{ s_ht[e], ref0, ref0 }; // spheres: id == ref (docs/SPEC.md §3); miss: both PT_MISS
                    cur = n_nodes ? 0 : PT_BVH_EMPTY;
                    rs = ray_setup(o, d);
                    sp = 0; steps = 0; has = true;
                    if (COUNT) c_sph += n_spheres;
                }
                next += min((uint32_t)__popcll(idle), avail);
            }
            if (!__ballot(has)) break;
            for (;;) {
                const bool is_n = has && (uint32_t)cur < (uint32_t)PT_BVH_EMPTY, is_l = has && cur < 0;
                const uint64_t m_n = __ballot(is_n), m_l = __ballot(is_l);
                if (!(m_n | m_l)) break;
                const uint32_t n_l = (uint32_t)__popcll(m_l), busy = (uint32_t)__popcll(m_n) + n_l;
                if (next < n_live && 64u - busy >= kPoolRefill) break; // enough lanes can take a new ray
                if (m_l && (!m_n || n_l >= kPoolLeafWait)) {
                    if (is_l) { const uint32_t nt = leaf_step(tris, stk, o, d, h, cur, sp); if (COUNT) c_tris += nt; }
                } else if (is_n) {
                    if (++steps > (1u << 22)) { atomicOr(&cold().ps.counters[kCntError], 2u); cur = PT_BVH_EMPTY; }
                    else {
                        if (COUNT) { c_nodes++; if (lane == (uint32_t)(__ffsll((long long)m_n) - 1)) c_wave_iters++; }
                        node_step<L>(nodes, stk, rs, h.t, cur, sp);
                    }
                }
            }
        }
        __syncthreads(); // every hit record is in LDS

        // ---- shade the pool, P x 64 lanes wide, in slot order
        const bool last = bounce + 1u == n_bounces;
        uint32_t n_next = 0;
        const ExtArgs &c = cold();
#pragma unroll
        for (uint32_t hh = 0; hh < P; ++hh) {
            const uint32_t e2 = hh * 64u + lane;
            const uint32_t slot = s_slot[e2];
            bool alive = false;
            if (slot != kInvalidSlot) {
                PathRegs r;
                path_load(c.ps, slot, r);
                uint32_t defer = 0u;
                alive = shade_one<FUSE>(c.sc, c.ps, c.fp, slot, r, s_ht[e2], s_href[e2], B_LAMBERT, defer);
                if (alive) {
                    path_store(c.ps, slot, r);
                    if (!last) spheres_of(r.o, r.d, e2);
                } else s_slot[e2] = kInvalidSlot;
            }
            live_append(alive, e2, n_next);
        }
        n_live = n_next;
        __syncthreads(); // the new rays (global) and the live list (LDS) are visible to whichever lane pulls them
    }

    // ---- hand the survivors to the next iteration
    const PathState &ps = cold().ps;
    uint32_t *q_next = ps.q_ext[parity ^ 1u] + qbase;
    uint32_t n_left = 0;
#pragma unroll
    for (uint32_t hh = 0; hh < P; ++hh) {
        const uint32_t e2 = hh * 64u + lane, g = bx * E + e2;
        const uint32_t slot = s_slot[e2];
        const bool alive = slot != kInvalidSlot;
        if (do_compact) wave_push(&ps.counters[cnt_ext_index(cnext, shard)], q_next, alive, slot);
        else if (g < n) q_next[g] = slot;
        n_left += (uint32_t)__popcll(__ballot(alive));
    }
    if (COUNT) {
        if (c_wave_iters) atomicAdd(reinterpret_cast<unsigned long long *>(ps.counters + kCntWaveNodeIters), c_wave_iters);
        atomicAdd(reinterpret_cast<unsigned long long *>(ps.counters + kCntNodes), c_nodes);
        atomicAdd(reinterpret_cast<unsigned long long *>(ps.counters + kCntTris), c_tris);
        atomicAdd(reinterpret_cast<unsigned long long *>(ps.counters + kCntSph), c_sph);
    }
    if (lane == 0u) {
        if (wave_rays) atomicAdd(traced_counter(ps, cnext, shard), (unsigned long long)wave_rays);
        if (n_left) atomicAdd(&ps.counters[cnt_alive_index(cnext, shard)], n_left);
    }
}

// ------------------------------------------------------------------------------------------------
// k_shade<SHADE_QUEUE>   : walks the extend queue of this iteration in the SAME order k_extend did, so ray/throughput/hit
//                          reads are the coalesced, L2-warm lines k_extend just touched. Misses and Lambert hits are shaded
//                          in place; specular kinds (metal, dielectric) are deferred to per-kind bucket queues.
// k_shade<SHADE_BUCKETS> : walks the two specular buckets, concatenated, so the kind switch is wave-uniform.
// k_shade<SHADE_INLINE>  : like SHADE_QUEUE but shades the specular kinds in place too (a divergent branch). Default for scenes
//                          with specular materials: deferring re-appends those slots at the end of the next queue, which
//                          scrambles the queue's slot order a little more every iteration; measured on Cornell + glass + metal,
//                          the same 16.6 M-ray launch went from 0.78 ms to 2.15 ms (shade) and 0.32 to 0.74 ms (extend) within
//                          30 iterations as slot-indexed state lost its coalescing. Order beats divergence here.
// These run after k_extend<.., SHADE_NONE> or k_extend_packed; the default pipeline shades inside k_extend (FUSE).
template <int MODE>
__global__ void __launch_bounds__(kBlock) k_shade(DeviceScene sc, PathState ps, FrameParams fp, uint32_t it, uint32_t compact)
{
    constexpr bool SPEC = MODE == SHADE_BUCKETS;
    const uint32_t parity = it & 1u, ccur = it % 3u, cnext = (it + 1u) % 3u;
    uint32_t shard, bx, nbx;
    block_pos(ps, shard, bx, nbx);
    const size_t qbase = (size_t)shard * ps.shard_cap;
    uint32_t total, c0 = 0u;
    if (SPEC) {
        c0 = ps.counters[cnt_bucket_index(parity, B_METAL, shard)];
        total = c0 + ps.counters[cnt_bucket_index(parity, B_DIELECTRIC, shard)];
        if (bx == 0 && threadIdx.x < 2u) // the other parity's buckets were consumed by the previous k_shade<true>
            ps.counters[cnt_bucket_index(parity ^ 1u, B_METAL + threadIdx.x, shard)] = 0u;
    } else total = ps.counters[cnt_ext_index(ccur, shard)];
    const uint32_t n_alive = ps.counters[cnt_alive_index(ccur, shard)];
    const bool do_compact = SPEC || want_compact(ps, total, n_alive, 0u, false, compact != 0u);
    if (!SPEC && do_compact && n_alive && bx == 0 && threadIdx.x == 0) atomicAdd(&ps.counters[kCntCompactions], 1u);
    // SPEC: a small fixed grid strides over the (usually short, unknown-length) specular buckets, so an empty bucket
    // costs a few hundred trivial blocks instead of one per 256 queue slots. !SPEC: exactly one pass, grid sized by the host.
    for (uint32_t base = bx * kBlock; base < total; base += nbx * kBlock) {
    const uint32_t gid = base + threadIdx.x;
    uint32_t b = B_LAMBERT, slot = kInvalidSlot;
    if (SPEC) {
        b = gid >= c0 ? B_DIELECTRIC : B_METAL;
        if (gid < total) slot = ps.q_bucket[b][qbase + (gid >= c0 ? gid - c0 : gid)];
    } else if (gid < total) slot = ps.q_ext[parity][qbase + gid];
    const bool active = slot != kInvalidSlot;
    bool alive = false;
    uint32_t defer = 0u; // SPEC == false: bucket this lane's hit must be shaded in (0 = handled here)

    if (active) {
        const float2 hr = at(ps.hit, slot);
        PathRegs r;
        path_load(ps, slot, r);
        alive = shade_one<MODE>(sc, ps, fp, slot, r, hr.x, __float_as_uint(hr.y), b, defer);
        if (alive) path_store(ps, slot, r); // a deferred hit stores nothing: its state stays as k_extend left it
    }
    queue_next(ps, shard, cnext, ps.q_ext[parity ^ 1u] + qbase, gid, total, alive, slot, do_compact);
    if (MODE == SHADE_QUEUE) {
        wave_push(&ps.counters[cnt_bucket_index(parity, B_METAL, shard)], ps.q_bucket[B_METAL] + qbase, defer == B_METAL, slot);
        wave_push(&ps.counters[cnt_bucket_index(parity, B_DIELECTRIC, shard)], ps.q_bucket[B_DIELECTRIC] + qbase, defer == B_DIELECTRIC, slot);
    }
    if (!SPEC) break;
    }
}

// ------------------------------------------------------------------------------------------------
// Sum of a pixel's K stream partials in the fixed order ((s0 + s1) + s2) + ... (docs/SPEC.md §5) into the tile-major
// buffer that is assembled / gathered. The partials themselves stay intact so that a later frame can keep accumulating.
__global__ void __launch_bounds__(kBlock) k_reduce_streams(const float4 *__restrict__ acc, float4 *__restrict__ tiles,
                                                           uint32_t slots_per_stream, uint32_t streams)
{
    const uint32_t slot = blockIdx.x * kBlock + threadIdx.x; // pixel slot
    if (slot >= slots_per_stream) return;
    float4 t = acc[slot_of(slot, 0u, streams, slots_per_stream)];
    for (uint32_t k = 1; k < streams; ++k) {
        const float4 a = acc[slot_of(slot, k, streams, slots_per_stream)];
        t.x = t.x + a.x; t.y = t.y + a.y; t.z = t.z + a.z; t.w = t.w + a.w;
    }
    tiles[slot] = t;
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

hipError_t launch_generate(hipStream_t s, const DeviceScene &sc, const PathState &ps, const FrameParams &fp, uint32_t mode)
{
    hipLaunchKernelGGL(k_generate, shard_grid(blocks_for(ps.shard_cap), ps.shard_count), dim3(kBlock), 0, s, sc, ps, fp, mode);
    return hipGetLastError();
}

template <int L, bool C>
static void extend_lc(hipStream_t s, dim3 grid, const ExtArgs &a, int kernel, int fuse)
{
    if (kernel == EXT_POOL && fuse == SHADE_INLINE) hipLaunchKernelGGL((k_extend_pool<L, C, SHADE_INLINE>), grid, dim3(64), 0, s, a);
    else if (kernel == EXT_POOL) hipLaunchKernelGGL((k_extend_pool<L, C, SHADE_QUEUE>), grid, dim3(64), 0, s, a);
    else if (kernel == EXT_PACKED && fuse == SHADE_QUEUE) hipLaunchKernelGGL((k_extend_packed<L, C, SHADE_QUEUE>), grid, dim3(64), 0, s, a);
    else if (kernel == EXT_PACKED && fuse == SHADE_INLINE) hipLaunchKernelGGL((k_extend_packed<L, C, SHADE_INLINE>), grid, dim3(64), 0, s, a);
    else if (kernel == EXT_PACKED) hipLaunchKernelGGL((k_extend_packed<L, C, SHADE_NONE>), grid, dim3(64), 0, s, a);
    else if (fuse == SHADE_QUEUE) hipLaunchKernelGGL((k_extend<L, C, SHADE_QUEUE>), grid, dim3(kExtBlock), 0, s, a);
    else if (fuse == SHADE_INLINE) hipLaunchKernelGGL((k_extend<L, C, SHADE_INLINE>), grid, dim3(kExtBlock), 0, s, a);
    else hipLaunchKernelGGL((k_extend<L, C, SHADE_NONE>), grid, dim3(kExtBlock), 0, s, a);
}

hipError_t launch_extend(hipStream_t s, const DeviceScene &sc, const PathState &ps, const FrameParams &fp, uint32_t it, uint32_t shard_bound, bool count,
                         int kernel, uint32_t packed_chunk, int fuse, bool compact, uint32_t bounces)
{
    // kernel: EXT_SIMPLE one ray per lane; EXT_PACKED a wavefront owns `packed_chunk` (>= 64) queue entries and refills idle lanes,
    // shading each finished ray on the spot; EXT_POOL a wavefront owns kPool entries, refills idle lanes during traversal and shades
    // the whole pool at full width (fused only: without shading it falls back to EXT_PACKED).
    if (kernel == EXT_POOL && fuse == SHADE_NONE) kernel = EXT_PACKED;
    const uint32_t chunk = kernel == EXT_PACKED ? (packed_chunk >= 64u ? packed_chunk : 128u) : 0u;
    const uint32_t per_block = kernel == EXT_PACKED ? chunk : kernel == EXT_POOL ? kPool : kExtBlock;
    const dim3 grid = shard_grid(shard_bound ? (shard_bound + per_block - 1) / per_block : 1u, ps.shard_count);
    ExtArgs a;
    a.sc = sc; a.ps = ps; a.fp = fp; a.it = it; a.compact = compact ? 1u : 0u; a.bounces = bounces ? bounces : 1u; a.chunk = chunk;
    switch (sc.bvh_width) {
    case PT_BVH_WIDTH_2:  count ? extend_lc<PT_BVH_WIDTH_2, true>(s, grid, a, kernel, fuse) : extend_lc<PT_BVH_WIDTH_2, false>(s, grid, a, kernel, fuse); break;
    case PT_BVH_WIDTH_4:  count ? extend_lc<PT_BVH_WIDTH_4, true>(s, grid, a, kernel, fuse) : extend_lc<PT_BVH_WIDTH_4, false>(s, grid, a, kernel, fuse); break;
    case PT_BVH_WIDTH_4Q: count ? extend_lc<PT_BVH_WIDTH_4Q, true>(s, grid, a, kernel, fuse) : extend_lc<PT_BVH_WIDTH_4Q, false>(s, grid, a, kernel, fuse); break;
    case PT_BVH_WIDTH_8Q: count ? extend_lc<PT_BVH_WIDTH_8Q, true>(s, grid, a, kernel, fuse) : extend_lc<PT_BVH_WIDTH_8Q, false>(s, grid, a, kernel, fuse); break;
    case PT_BVH_WIDTH_8O: count ? extend_lc<PT_BVH_WIDTH_8O, true>(s, grid, a, kernel, fuse) : extend_lc<PT_BVH_WIDTH_8O, false>(s, grid, a, kernel, fuse); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_shade(hipStream_t s, const DeviceScene &sc, const PathState &ps, const FrameParams &fp, uint32_t it, uint32_t shard_bound, int mode, bool compact)
{
    const uint32_t parity = it; // k_shade derives queue parity and counter rotation from the iteration index
    const dim3 grid = shard_grid(blocks_for(shard_bound), ps.shard_count), block(kBlock);
    const dim3 sgrid = shard_grid(std::min(blocks_for(shard_bound), 16u), ps.shard_count); // grid-stride over the specular buckets
    const uint32_t cm = compact ? 1u : 0u;
    if (mode == SHADE_BUCKETS) hipLaunchKernelGGL(k_shade<SHADE_BUCKETS>, sgrid, block, 0, s, sc, ps, fp, parity, 1u);
    else if (mode == SHADE_INLINE) hipLaunchKernelGGL(k_shade<SHADE_INLINE>, grid, block, 0, s, sc, ps, fp, parity, cm);
    else hipLaunchKernelGGL(k_shade<SHADE_QUEUE>, grid, block, 0, s, sc, ps, fp, parity, cm);
    return hipGetLastError();
}

hipError_t launch_reduce_streams(hipStream_t s, const float4 *acc, float4 *tiles, uint32_t slots_per_stream, uint32_t streams)
{
    hipLaunchKernelGGL(k_reduce_streams, dim3(blocks_for(slots_per_stream)), dim3(kBlock), 0, s, acc, tiles, slots_per_stream, streams);
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
