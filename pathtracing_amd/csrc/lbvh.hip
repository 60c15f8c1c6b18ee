// lbvh.hip — GPU construction of a binary LBVH (the reference has no acceleration structure).
//   k_tri_boxes   : per triangle, the padded box of docs/SPEC.md §4.1 and its centroid; scene centroid bounds by atomics
//   k_morton      : 30-bit Morton code of the centroid
//   (rocPRIM)     : radix sort of (code, triangle) pairs
//   k_hierarchy   : Karras 2012 — one thread per internal node finds its key range and split by binary search on the
//                   common-prefix length (ties between equal codes are broken by the index, so duplicates form a balanced subtree)
//   k_refit       : leaves walk up; the second arrival at a node unites the children's boxes (agent-scope fences between)
// Then either build_lbvh_blob4q_device packs the default BVH4Q blob right here on the device (second half of this file), or — other
// node layouts — the binary tree is copied back and packed by build_bvh_from_binary() on the host. The closest hit does not depend
// on the tree (SPEC §4), so a scene committed with this builder renders the same picture bit for bit.
#include <cstring> // rocprim's texture_cache_iterator.hpp needs ::memset
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>
#include <algorithm>
#include <chrono>
#include <vector>
#include "bvh_build.h"

namespace ptrt {
namespace {

__device__ __forceinline__ uint32_t f_ord(float f) { const uint32_t b = __float_as_uint(f); return (b & 0x80000000u) ? ~b : (b | 0x80000000u); }
inline float f_unord(uint32_t u) { const uint32_t b = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u; float f; std::memcpy(&f, &b, 4); return f; }
__device__ __forceinline__ float pad_of(float c) { return 1e-6f * fmaxf(1.0f, fabsf(c)); }

__global__ void __launch_bounds__(256) k_tri_boxes(const float *__restrict__ verts, uint32_t n, float *__restrict__ leaf_box,
                                                   float *__restrict__ cent, uint32_t *__restrict__ bounds /* 3 min, 3 max (ordered uints) */)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float *p = verts + (size_t)i * 9;
    for (int k = 0; k < 3; ++k) {
        const float lo = fminf(p[k], fminf(p[3 + k], p[6 + k])), hi = fmaxf(p[k], fmaxf(p[3 + k], p[6 + k]));
        leaf_box[(size_t)i * 6 + k] = lo - pad_of(lo);
        leaf_box[(size_t)i * 6 + 3 + k] = hi + pad_of(hi);
        const float c = 0.5f * (lo + hi);
        cent[(size_t)i * 3 + k] = c;
        atomicMin(&bounds[k], f_ord(c));
        atomicMax(&bounds[3 + k], f_ord(c));
    }
}

__device__ __forceinline__ uint32_t spread10(uint32_t v)
{
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}

__global__ void __launch_bounds__(256) k_morton(const float *__restrict__ cent, uint32_t n, float3 cmin, float3 inv_ext,
                                                uint32_t *__restrict__ keys, uint32_t *__restrict__ vals)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float x = (cent[(size_t)i * 3 + 0] - cmin.x) * inv_ext.x, y = (cent[(size_t)i * 3 + 1] - cmin.y) * inv_ext.y,
                z = (cent[(size_t)i * 3 + 2] - cmin.z) * inv_ext.z;
    const uint32_t xi = (uint32_t)fminf(fmaxf(x * 1024.0f, 0.0f), 1023.0f), yi = (uint32_t)fminf(fmaxf(y * 1024.0f, 0.0f), 1023.0f),
                   zi = (uint32_t)fminf(fmaxf(z * 1024.0f, 0.0f), 1023.0f);
    keys[i] = (spread10(xi) << 2) | (spread10(yi) << 1) | spread10(zi);
    vals[i] = i;
}

__device__ __forceinline__ int delta(const uint32_t *__restrict__ keys, int n, int i, int j)
{
    if (j < 0 || j >= n) return -1;
    const uint32_t a = keys[i], b = keys[j];
    return a == b ? 32 + __clz((uint32_t)(i ^ j)) : __clz(a ^ b);
}

// children: >= 0 internal node, < 0 leaf ~j (j = position in the sorted order)
__global__ void __launch_bounds__(256) k_hierarchy(const uint32_t *__restrict__ keys, int n, int32_t *__restrict__ left, int32_t *__restrict__ right,
                                                   uint32_t *__restrict__ first, uint32_t *__restrict__ last,
                                                   int32_t *__restrict__ parent_node, int32_t *__restrict__ parent_leaf)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n - 1) return;
    const int d = (delta(keys, n, i, i + 1) - delta(keys, n, i, i - 1)) >= 0 ? 1 : -1;
    const int dmin = delta(keys, n, i, i - d);
    int lmax = 2;
    while (delta(keys, n, i, i + lmax * d) > dmin) lmax <<= 1;
    int l = 0;
    for (int t = lmax >> 1; t >= 1; t >>= 1)
        if (delta(keys, n, i, i + (l + t) * d) > dmin) l += t;
    const int j = i + l * d;
    const int dnode = delta(keys, n, i, j);
    int s = 0, t = l;
    do {
        t = (t + 1) >> 1;
        if (delta(keys, n, i, i + (s + t) * d) > dnode) s += t;
    } while (t > 1);
    const int gamma = i + s * d + (d < 0 ? d : 0);
    const int lo = i < j ? i : j, hi = i < j ? j : i;
    const int32_t lc = (lo == gamma) ? ~gamma : gamma, rc = (hi == gamma + 1) ? ~(gamma + 1) : gamma + 1;
    left[i] = lc; right[i] = rc; first[i] = (uint32_t)lo; last[i] = (uint32_t)hi;
    if (lc >= 0) parent_node[lc] = i; else parent_leaf[~lc] = i;
    if (rc >= 0) parent_node[rc] = i; else parent_leaf[~rc] = i;
    if (i == 0) parent_node[0] = -1;
}

__global__ void __launch_bounds__(256) k_refit(const uint32_t *__restrict__ order, const float *__restrict__ leaf_box, int n,
                                               const int32_t *__restrict__ left, const int32_t *__restrict__ right,
                                               const int32_t *__restrict__ parent_node, const int32_t *__restrict__ parent_leaf,
                                               float *__restrict__ node_box, uint32_t *__restrict__ arrived)
{
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    int cur = parent_leaf[j];
    while (cur >= 0) {
        __threadfence();                                 // release what this thread (or its child visit) wrote
        if (atomicAdd(&arrived[cur], 1u) == 0u) return;  // first of the two children to arrive: the sibling finishes the node
        __threadfence();                                 // acquire the sibling subtree's boxes
        float b[6];
        const int32_t c[2] = { left[cur], right[cur] };
        for (int k = 0; k < 3; ++k) { b[k] = __builtin_inff(); b[3 + k] = -__builtin_inff(); }
        for (int s = 0; s < 2; ++s) {
            const float *src = c[s] >= 0 ? node_box + (size_t)c[s] * 6 : leaf_box + (size_t)order[~c[s]] * 6;
            for (int k = 0; k < 3; ++k) {
                b[k] = fminf(b[k], __hip_atomic_load(src + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                b[3 + k] = fmaxf(b[3 + k], __hip_atomic_load(src + 3 + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            }
        }
        for (int k = 0; k < 6; ++k) __hip_atomic_store(node_box + (size_t)cur * 6 + k, b[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        cur = parent_node[cur];
    }
}

template <typename T> struct Dev {
    T *p = nullptr;
    hipError_t alloc(size_t n) { return hipMalloc((void **)&p, (n ? n : 1) * sizeof(T)); }
    T *release() { T *q = p; p = nullptr; return q; }
    ~Dev() { if (p) (void)hipFree(p); }
};

#define LB_TRY(x) do { hipError_t _e = (x); if (_e != hipSuccess) return _e; } while (0)

// The binary LBVH on the device: construction shared by the two consumers below.
struct Lbvh {
    uint32_t n = 0;
    Dev<float> verts, leaf_box, cent, node_box;
    Dev<uint32_t> bounds, keys, vals, keys2, order, first, last, arrived;
    Dev<int32_t> left, right, pn, pl;
    Dev<unsigned char> tmp;
    hipError_t construct(hipStream_t stream, const float *verts9, uint32_t n_)
    {
        n = n_;
        LB_TRY(verts.alloc((size_t)n * 9)); LB_TRY(leaf_box.alloc((size_t)n * 6)); LB_TRY(cent.alloc((size_t)n * 3));
        LB_TRY(node_box.alloc((size_t)(n - 1) * 6)); LB_TRY(bounds.alloc(6));
        LB_TRY(keys.alloc(n)); LB_TRY(vals.alloc(n)); LB_TRY(keys2.alloc(n)); LB_TRY(order.alloc(n));
        LB_TRY(first.alloc(n - 1)); LB_TRY(last.alloc(n - 1)); LB_TRY(arrived.alloc(n - 1));
        LB_TRY(left.alloc(n - 1)); LB_TRY(right.alloc(n - 1)); LB_TRY(pn.alloc(n - 1)); LB_TRY(pl.alloc(n));
        LB_TRY(hipMemcpyAsync(verts.p, verts9, (size_t)n * 36, hipMemcpyHostToDevice, stream));
        const uint32_t init_bounds[6] = { 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u, 0u };
        LB_TRY(hipMemcpyAsync(bounds.p, init_bounds, sizeof init_bounds, hipMemcpyHostToDevice, stream));
        LB_TRY(hipMemsetAsync(arrived.p, 0, (size_t)(n - 1) * 4, stream));
        const dim3 grid((n + 255) / 256), block(256);
        hipLaunchKernelGGL(k_tri_boxes, grid, block, 0, stream, verts.p, n, leaf_box.p, cent.p, bounds.p);
        uint32_t hb[6];
        LB_TRY(hipMemcpyAsync(hb, bounds.p, sizeof hb, hipMemcpyDeviceToHost, stream));
        LB_TRY(hipStreamSynchronize(stream));
        const float lo[3] = { f_unord(hb[0]), f_unord(hb[1]), f_unord(hb[2]) }, hi[3] = { f_unord(hb[3]), f_unord(hb[4]), f_unord(hb[5]) };
        const float3 cmin = make_float3(lo[0], lo[1], lo[2]);
        const float3 inv = make_float3(hi[0] > lo[0] ? 1.0f / (hi[0] - lo[0]) : 0.f, hi[1] > lo[1] ? 1.0f / (hi[1] - lo[1]) : 0.f, hi[2] > lo[2] ? 1.0f / (hi[2] - lo[2]) : 0.f);
        hipLaunchKernelGGL(k_morton, grid, block, 0, stream, cent.p, n, cmin, inv, keys.p, vals.p);
        size_t tmp_bytes = 0;
        LB_TRY(rocprim::radix_sort_pairs(nullptr, tmp_bytes, keys.p, keys2.p, vals.p, order.p, (size_t)n, 0, 30, stream));
        LB_TRY(tmp.alloc(tmp_bytes));
        LB_TRY(rocprim::radix_sort_pairs(tmp.p, tmp_bytes, keys.p, keys2.p, vals.p, order.p, (size_t)n, 0, 30, stream));
        hipLaunchKernelGGL(k_hierarchy, dim3((n - 1 + 255) / 256), block, 0, stream, keys2.p, (int)n, left.p, right.p, first.p, last.p, pn.p, pl.p);
        hipLaunchKernelGGL(k_refit, grid, block, 0, stream, order.p, leaf_box.p, (int)n, left.p, right.p, pn.p, pl.p, node_box.p, arrived.p);
        return hipGetLastError();
    }
};

// =================================================================================================
// Packing on the device: binary LBVH -> BVH4Q blob (64-byte quantised nodes, breadth-first) + 64-byte triangle records, without
// the tree ever visiting the host. Only the top storey does: the LBVH is cut into clusters of <= kClusterTris triangles, their
// boxes (a few thousand) go to the host's binned-SAH builder and the small binary tree over them comes back (bvh_build.cpp has the
// same two-storey scheme for the other node layouts). Then, level by level: expand every node of the level to <= 4 children by
// opening the child of largest area (as emit_blob does), scan the inner children to number the next level breadth-first,
// quantise and write the node. Triangles are emitted in Morton order, so every leaf's range [first, last] is contiguous as it is.
constexpr int32_t kTopBase = 0x40000000;  // binary refs: >= kTopBase top-storey node, 0 .. n-2 LBVH node, < 0 LBVH leaf ~j
constexpr int32_t kNoKid = 0x7fffffff;
constexpr uint32_t kLeafTris = 4;         // kMaxLeaf of bvh_build.cpp

struct TreeView {
    const int32_t *left, *right; const uint32_t *first, *last, *order; const float *node_box, *leaf_box; const uint8_t *leaf_flag;
    const int32_t *top_left, *top_right; const float *top_box;
};
struct BoxF { float lo[3], hi[3]; };
__device__ __forceinline__ BoxF box_of(const TreeView &t, int32_t r)
{
    const float *p = r >= kTopBase ? t.top_box + (size_t)(r - kTopBase) * 6 : r >= 0 ? t.node_box + (size_t)r * 6 : t.leaf_box + (size_t)t.order[~r] * 6;
    BoxF b;
    for (int k = 0; k < 3; ++k) { b.lo[k] = p[k]; b.hi[k] = p[3 + k]; }
    return b;
}
__device__ __forceinline__ float area_of(const BoxF &b)
{
    const float dx = b.hi[0] - b.lo[0], dy = b.hi[1] - b.lo[1], dz = b.hi[2] - b.lo[2];
    return dx < 0.f ? 0.f : 2.f * (dx * dy + dy * dz + dz * dx);
}
__device__ __forceinline__ uint32_t count_of(const TreeView &t, int32_t r) { return r < 0 ? 1u : t.last[r] - t.first[r] + 1u; } // LBVH refs only
__device__ __forceinline__ bool blob_leaf(const TreeView &t, int32_t r) { return r < 0 || (r < kTopBase && t.leaf_flag[r]); }

// per LBVH node: does it become a leaf of the blob (<= kLeafTris triangles and splitting does not lower the SAH cost)? and is it a cluster root?
__global__ void __launch_bounds__(256) k_mark(TreeView t, uint32_t n, uint32_t cluster_tris, const int32_t *__restrict__ parent_node,
                                              const int32_t *__restrict__ parent_leaf, uint8_t *__restrict__ leaf_flag, int32_t *__restrict__ clusters,
                                              uint32_t *__restrict__ n_clusters)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < n - 1) {
        const uint32_t cnt = t.last[i] - t.first[i] + 1u;
        bool leaf = false;
        if (cnt <= kLeafTris) {
            float split = 0.f;
            const int32_t c[2] = { t.left[i], t.right[i] };
            for (int s = 0; s < 2; ++s) split += area_of(box_of(t, c[s])) * (float)count_of(t, c[s]);
            leaf = !(split < area_of(box_of(t, (int32_t)i)) * (float)cnt);
        }
        leaf_flag[i] = leaf ? 1 : 0;
        const int32_t p = parent_node[i];
        if (cnt <= cluster_tris && (p < 0 || t.last[p] - t.first[p] + 1u > cluster_tris)) clusters[atomicAdd(n_clusters, 1u)] = (int32_t)i;
    }
    if (i < n) { // single triangles hanging off a node that is too big to be a cluster
        const int32_t p = parent_leaf[i];
        if (t.last[p] - t.first[p] + 1u > cluster_tris) clusters[atomicAdd(n_clusters, 1u)] = ~(int32_t)i;
    }
}

__global__ void __launch_bounds__(256) k_cluster_info(TreeView t, const int32_t *__restrict__ clusters, uint32_t nc, float *__restrict__ boxes, uint32_t *__restrict__ firsts)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= nc) return;
    const int32_t r = clusters[i];
    const BoxF b = box_of(t, r);
    for (int k = 0; k < 3; ++k) { boxes[(size_t)i * 6 + k] = b.lo[k]; boxes[(size_t)i * 6 + 3 + k] = b.hi[k]; }
    firsts[i] = r < 0 ? (uint32_t)~r : t.first[r];
}

// one level, first half: the (up to) 4 children of every node of the level, and how many of them are inner nodes
__global__ void __launch_bounds__(256) k_expand(TreeView t, const int32_t *__restrict__ queue, uint32_t n_cur, uint32_t base, int32_t *__restrict__ kids,
                                                uint32_t *__restrict__ inner_count)
{
    const uint32_t q = blockIdx.x * 256 + threadIdx.x;
    if (q >= n_cur) return;
    const int32_t r = queue[q];
    int32_t kid[4] = { kNoKid, kNoKid, kNoKid, kNoKid };
    int nk = 0;
    if (blob_leaf(t, r)) kid[nk++] = r; // (the whole scene is one leaf) single child
    else {
        kid[0] = r >= kTopBase ? t.top_left[r - kTopBase] : t.left[r];
        kid[1] = r >= kTopBase ? t.top_right[r - kTopBase] : t.right[r];
        nk = 2;
        while (nk < 4) {
            int best = -1; float ba = -1.f;
            for (int i = 0; i < nk; ++i)
                if (!blob_leaf(t, kid[i])) { const float a = area_of(box_of(t, kid[i])); if (a > ba) { ba = a; best = i; } }
            if (best < 0) break;
            const int32_t c = kid[best];
            for (int i = nk; i > best + 1; --i) kid[i] = kid[i - 1];
            kid[best] = c >= kTopBase ? t.top_left[c - kTopBase] : t.left[c];
            kid[best + 1] = c >= kTopBase ? t.top_right[c - kTopBase] : t.right[c];
            nk++;
        }
    }
    uint32_t ni = 0;
    for (int i = 0; i < 4; ++i) {
        kids[(size_t)(base + q) * 4 + i] = kid[i];
        if (kid[i] != kNoKid && !blob_leaf(t, kid[i])) ++ni;
    }
    inner_count[q] = ni;
}

__device__ __forceinline__ float scale_of(uint32_t e) { return __uint_as_float(e << 23); }

// one level, second half: number the inner children breadth-first (they are the next level's queue), quantise, write the node
__global__ void __launch_bounds__(256) k_finalize(TreeView t, uint32_t n_cur, uint32_t base, const int32_t *__restrict__ kids, const uint32_t *__restrict__ offs,
                                                  int32_t *__restrict__ queue_next, uint8_t *__restrict__ nodes, float *__restrict__ cost, float inv_root_area)
{
    const uint32_t q = blockIdx.x * 256 + threadIdx.x;
    if (q >= n_cur) return;
    const uint32_t node = base + q;
    int32_t ref[4];
    BoxF box[4];
    uint32_t rank = 0;
    float my_cost = 0.f;
    for (int c = 0; c < 4; ++c) {
        const int32_t k = kids[(size_t)node * 4 + c];
        if (k == kNoKid) { ref[c] = kNoKid; continue; }
        box[c] = box_of(t, k);
        if (blob_leaf(t, k)) {
            const uint32_t first = k < 0 ? (uint32_t)~k : t.first[k], cnt = count_of(t, k);
            ref[c] = (int32_t)~((first << 3) | (cnt - 1u));
            my_cost += area_of(box[c]) * inv_root_area * (float)cnt;
        } else {
            const uint32_t pos = offs[q] + rank++;
            queue_next[pos] = k;
            ref[c] = (int32_t)(base + n_cur + pos);
            my_cost += area_of(box[c]) * inv_root_area;
        }
    }
    cost[node] = my_cost;
    // ---- docs/SPEC.md §4.1 BVH4Q: per axis a power-of-two grid from the children's union; every decoded box must enclose its float box,
    // checked with the traversal's own expression fma((float)q, scale, origin) (bvh_build.cpp quantize_nodes is the host twin)
    uint8_t *nd = nodes + (size_t)node * 64;
    float org[3]; uint32_t ex[3];
    uint32_t qlo[3] = { 0, 0, 0 }, qhi[3] = { 0, 0, 0 }; // 4 bytes each, child c in byte c
    for (int a = 0; a < 3; ++a) {
        float lo = __builtin_inff(), hi = -__builtin_inff();
        for (int c = 0; c < 4; ++c) if (ref[c] != kNoKid) { lo = fminf(lo, box[c].lo[a]); hi = fmaxf(hi, box[c].hi[a]); }
        if (!(lo <= hi)) lo = hi = 0.f;
        org[a] = lo;
        int e = 1;
        {
            const float ext = hi - lo;
            int ee; const float m = frexpf(ext / 255.0f, &ee);
            e = (ext > 0.f) ? ee + 127 - (m == 0.5f ? 1 : 0) : 1;
            e = min(max(e, 1), 254);
        }
        for (;;) {
            const float sc = scale_of((uint32_t)e);
            bool ok = true;
            uint32_t pl = 0, ph = 0;
            for (int c = 0; c < 4 && ok; ++c) {
                if (ref[c] == kNoKid) continue;
                int ql = (int)floorf((box[c].lo[a] - lo) / sc), qh = (int)ceilf((box[c].hi[a] - lo) / sc);
                ql = min(max(ql, 0), 255); qh = min(max(qh, 0), 255);
                while (ql > 0 && !(__builtin_fmaf((float)ql, sc, lo) <= box[c].lo[a])) --ql;
                while (qh < 255 && !(__builtin_fmaf((float)qh, sc, lo) >= box[c].hi[a])) ++qh;
                if (!(__builtin_fmaf((float)ql, sc, lo) <= box[c].lo[a]) || !(__builtin_fmaf((float)qh, sc, lo) >= box[c].hi[a])) { ok = false; break; }
                pl |= (uint32_t)ql << (8 * c); ph |= (uint32_t)qh << (8 * c);
            }
            if (ok || e >= 254) { qlo[a] = pl; qhi[a] = ph; break; }
            ++e;
        }
        ex[a] = (uint32_t)e;
    }
    uint32_t *w = reinterpret_cast<uint32_t *>(nd);
    w[0] = __float_as_uint(org[0]); w[1] = __float_as_uint(org[1]); w[2] = __float_as_uint(org[2]);
    w[3] = ex[0] | (ex[1] << 8) | (ex[2] << 16);
    for (int c = 0; c < 4; ++c) w[4 + c] = (uint32_t)ref[c];
    w[8] = qlo[0]; w[9] = qlo[1]; w[10] = qlo[2]; w[11] = qhi[0]; w[12] = qhi[1]; w[13] = qhi[2]; w[14] = 0u; w[15] = 0u;
}

// depth and worst-case traversal-stack need, one level at a time from the bottom (children live in the next level)
__global__ void __launch_bounds__(256) k_depth(const uint8_t *__restrict__ nodes, uint32_t base, uint32_t n_cur, uint32_t *__restrict__ depth, uint32_t *__restrict__ need)
{
    const uint32_t q = blockIdx.x * 256 + threadIdx.x;
    if (q >= n_cur) return;
    const int32_t *ref = reinterpret_cast<const int32_t *>(nodes + (size_t)(base + q) * 64 + 16);
    uint32_t k = 0, dmax = 1, nmax = 0;
    for (int c = 0; c < 4; ++c) {
        const int32_t r = ref[c];
        if (r == kNoKid) continue;
        ++k;
        if (r >= 0) { dmax = max(dmax, depth[r]); nmax = max(nmax, need[r]); }
    }
    depth[base + q] = dmax + 1;
    need[base + q] = (k ? k - 1 : 0) + nmax;
}

// the device triangle record (one 64-byte line, Morton order): the blob's three rows + the shading row normalize(cross(e1, e2)) | material
// in the op order of docs/SPEC.md §0 (api.cpp builds the same record on the host for the host-built trees)
__global__ void __launch_bounds__(256) k_tri_records(const float *__restrict__ verts, const uint32_t *__restrict__ mats, const uint32_t *__restrict__ order, uint32_t n,
                                                     float4 *__restrict__ rec)
{
    const uint32_t j = blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    const uint32_t id = order[j];
    const float *v = verts + (size_t)id * 9;
    const float a[3] = { v[3] - v[0], v[4] - v[1], v[5] - v[2] }, b[3] = { v[6] - v[0], v[7] - v[1], v[8] - v[2] };
    const uint32_t m = mats ? mats[id] : 0u;
    const float cx = __builtin_fmaf(a[1], b[2], -(a[2] * b[1])), cy = __builtin_fmaf(a[2], b[0], -(a[0] * b[2])), cz = __builtin_fmaf(a[0], b[1], -(a[1] * b[0]));
    const float inv = 1.0f / __builtin_sqrtf(__builtin_fmaf(cz, cz, __builtin_fmaf(cy, cy, cx * cx)));
    rec[(size_t)j * 4 + 0] = make_float4(v[0], v[1], v[2], __uint_as_float(id));
    rec[(size_t)j * 4 + 1] = make_float4(a[0], a[1], a[2], __uint_as_float(m));
    rec[(size_t)j * 4 + 2] = make_float4(b[0], b[1], b[2], 0.f);
    rec[(size_t)j * 4 + 3] = make_float4(cx * inv, cy * inv, cz * inv, __uint_as_float(m));
}

} // namespace

#ifndef PT_LBVH_CLUSTER
#define PT_LBVH_CLUSTER 32
#endif

hipError_t build_lbvh_device(hipStream_t stream, const float *verts9, uint32_t n, BinaryBvh &out)
{
    const auto t0 = std::chrono::steady_clock::now();
    out = BinaryBvh{};
    if (n < 2) return hipErrorInvalidValue;
    Lbvh t;
    LB_TRY(t.construct(stream, verts9, n));
    out.order.resize(n); out.left.resize(n - 1); out.right.resize(n - 1); out.first.resize(n - 1); out.last.resize(n - 1);
    out.box.resize((size_t)(n - 1) * 6);
    LB_TRY(hipMemcpyAsync(out.order.data(), t.order.p, (size_t)n * 4, hipMemcpyDeviceToHost, stream));
    LB_TRY(hipMemcpyAsync(out.left.data(), t.left.p, (size_t)(n - 1) * 4, hipMemcpyDeviceToHost, stream));
    LB_TRY(hipMemcpyAsync(out.right.data(), t.right.p, (size_t)(n - 1) * 4, hipMemcpyDeviceToHost, stream));
    LB_TRY(hipMemcpyAsync(out.first.data(), t.first.p, (size_t)(n - 1) * 4, hipMemcpyDeviceToHost, stream));
    LB_TRY(hipMemcpyAsync(out.last.data(), t.last.p, (size_t)(n - 1) * 4, hipMemcpyDeviceToHost, stream));
    LB_TRY(hipMemcpyAsync(out.box.data(), t.node_box.p, (size_t)(n - 1) * 24, hipMemcpyDeviceToHost, stream));
    LB_TRY(hipStreamSynchronize(stream));
    out.device_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return hipSuccess;
}

hipError_t build_lbvh_blob4q_device(hipStream_t stream, const float *verts9, const uint32_t *mats, uint32_t n, DeviceBlob4Q &out)
{
    const auto t0 = std::chrono::steady_clock::now();
    out = DeviceBlob4Q{};
    if (n < 2) return hipErrorInvalidValue;
    Lbvh t;
    LB_TRY(t.construct(stream, verts9, n));
    const dim3 block(256);
    auto blocks = [](uint32_t k) { return dim3((k + 255) / 256); };

    // ---- leaf flags + cluster roots
    Dev<uint8_t> leaf_flag; Dev<int32_t> clusters; Dev<uint32_t> n_clusters, d_mats;
    LB_TRY(leaf_flag.alloc(n)); LB_TRY(clusters.alloc(n)); LB_TRY(n_clusters.alloc(1));
    LB_TRY(hipMemsetAsync(n_clusters.p, 0, 4, stream));
    if (mats) { LB_TRY(d_mats.alloc(n)); LB_TRY(hipMemcpyAsync(d_mats.p, mats, (size_t)n * 4, hipMemcpyHostToDevice, stream)); }
    TreeView tv{ t.left.p, t.right.p, t.first.p, t.last.p, t.order.p, t.node_box.p, t.leaf_box.p, leaf_flag.p, nullptr, nullptr, nullptr };
    hipLaunchKernelGGL(k_mark, blocks(n), block, 0, stream, tv, n, (uint32_t)PT_LBVH_CLUSTER, t.pn.p, t.pl.p, leaf_flag.p, clusters.p, n_clusters.p);
    uint32_t nc = 0;
    LB_TRY(hipMemcpyAsync(&nc, n_clusters.p, 4, hipMemcpyDeviceToHost, stream));
    LB_TRY(hipStreamSynchronize(stream));
    if (nc == 0 || nc > n) return hipErrorUnknown;
    Dev<float> d_cbox; Dev<uint32_t> d_cfirst;
    LB_TRY(d_cbox.alloc((size_t)nc * 6)); LB_TRY(d_cfirst.alloc(nc));
    hipLaunchKernelGGL(k_cluster_info, blocks(nc), block, 0, stream, tv, clusters.p, nc, d_cbox.p, d_cfirst.p);
    std::vector<int32_t> h_clusters(nc); std::vector<float> h_cbox((size_t)nc * 6); std::vector<uint32_t> h_cfirst(nc);
    LB_TRY(hipMemcpyAsync(h_clusters.data(), clusters.p, (size_t)nc * 4, hipMemcpyDeviceToHost, stream));
    LB_TRY(hipMemcpyAsync(h_cbox.data(), d_cbox.p, (size_t)nc * 24, hipMemcpyDeviceToHost, stream));
    LB_TRY(hipMemcpyAsync(h_cfirst.data(), d_cfirst.p, (size_t)nc * 4, hipMemcpyDeviceToHost, stream));
    // the triangle records need nothing of the above: they run while the host builds the top storey
    Dev<float4> tris;
    LB_TRY(tris.alloc((size_t)n * 4));
    hipLaunchKernelGGL(k_tri_records, blocks(n), block, 0, stream, t.verts.p, mats ? d_mats.p : nullptr, t.order.p, n, tris.p);
    LB_TRY(hipStreamSynchronize(stream));

    // ---- top storey on the host: binned SAH over the cluster boxes, in Morton order of the clusters (the append order above is not deterministic)
    std::vector<uint32_t> perm(nc);
    for (uint32_t i = 0; i < nc; ++i) perm[i] = i;
    std::sort(perm.begin(), perm.end(), [&](uint32_t a, uint32_t b) { return h_cfirst[a] < h_cfirst[b]; });
    std::vector<float> sorted_box((size_t)nc * 6);
    for (uint32_t i = 0; i < nc; ++i) std::memcpy(&sorted_box[(size_t)i * 6], &h_cbox[(size_t)perm[i] * 6], 24);
    std::vector<int32_t> top_left, top_right; std::vector<float> top_box; int32_t top_root = 0;
    build_sah_over_boxes(sorted_box.data(), nc, top_left, top_right, top_box, top_root);
    auto to_ref = [&](int32_t c) { return c < 0 ? h_clusters[perm[(uint32_t)~c]] : kTopBase + c; }; // SAH leaf ~i = cluster i
    for (auto &c : top_left) c = to_ref(c);
    for (auto &c : top_right) c = to_ref(c);
    const int32_t root_ref = to_ref(top_root);
    const float *rb = top_root < 0 ? &sorted_box[(size_t)(uint32_t)~top_root * 6] : &top_box[(size_t)top_root * 6];
    const float rdx = rb[3] - rb[0], rdy = rb[4] - rb[1], rdz = rb[5] - rb[2];
    const float root_area = std::max(2.f * (rdx * rdy + rdy * rdz + rdz * rdx), 1e-30f);
    Dev<int32_t> d_tl, d_tr; Dev<float> d_tb;
    const size_t nt_top = top_left.size();
    LB_TRY(d_tl.alloc(nt_top)); LB_TRY(d_tr.alloc(nt_top)); LB_TRY(d_tb.alloc(nt_top * 6));
    if (nt_top) {
        LB_TRY(hipMemcpyAsync(d_tl.p, top_left.data(), nt_top * 4, hipMemcpyHostToDevice, stream));
        LB_TRY(hipMemcpyAsync(d_tr.p, top_right.data(), nt_top * 4, hipMemcpyHostToDevice, stream));
        LB_TRY(hipMemcpyAsync(d_tb.p, top_box.data(), nt_top * 24, hipMemcpyHostToDevice, stream));
    }
    tv.top_left = d_tl.p; tv.top_right = d_tr.p; tv.top_box = d_tb.p;

    // ---- level by level: expand, scan, finalize. A 4-wide inner node has >= 2 children, so there are < n nodes in all.
    const uint32_t cap = n;
    Dev<uint8_t> nodes; Dev<int32_t> kids, queue_a, queue_b; Dev<uint32_t> inner_count, offs, depth, need; Dev<float> cost; Dev<unsigned char> scan_tmp;
    LB_TRY(nodes.alloc((size_t)cap * 64)); LB_TRY(kids.alloc((size_t)cap * 4)); LB_TRY(queue_a.alloc(cap)); LB_TRY(queue_b.alloc(cap));
    LB_TRY(inner_count.alloc(cap + 1)); LB_TRY(offs.alloc(cap + 1)); LB_TRY(depth.alloc(cap)); LB_TRY(need.alloc(cap)); LB_TRY(cost.alloc(cap));
    size_t scan_bytes = 0;
    LB_TRY(rocprim::exclusive_scan(nullptr, scan_bytes, inner_count.p, offs.p, 0u, (size_t)cap + 1, rocprim::plus<uint32_t>(), stream));
    LB_TRY(scan_tmp.alloc(scan_bytes));
    LB_TRY(hipMemcpyAsync(queue_a.p, &root_ref, 4, hipMemcpyHostToDevice, stream));
    std::vector<uint32_t> level_base;
    uint32_t base = 0, n_cur = 1;
    int32_t *qc = queue_a.p, *qn = queue_b.p;
    while (n_cur) {
        if (base + n_cur > cap || level_base.size() > 200) return hipErrorUnknown;
        level_base.push_back(base);
        hipLaunchKernelGGL(k_expand, blocks(n_cur), block, 0, stream, tv, qc, n_cur, base, kids.p, inner_count.p);
        LB_TRY(hipMemsetAsync(inner_count.p + n_cur, 0, 4, stream)); // the scan's extra element: offs[n_cur] = the next level's size
        LB_TRY(rocprim::exclusive_scan(scan_tmp.p, scan_bytes, inner_count.p, offs.p, 0u, (size_t)n_cur + 1, rocprim::plus<uint32_t>(), stream));
        hipLaunchKernelGGL(k_finalize, blocks(n_cur), block, 0, stream, tv, n_cur, base, kids.p, offs.p, qn, nodes.p, cost.p, 1.0f / root_area);
        uint32_t n_next = 0;
        LB_TRY(hipMemcpyAsync(&n_next, offs.p + n_cur, 4, hipMemcpyDeviceToHost, stream));
        LB_TRY(hipStreamSynchronize(stream));
        base += n_cur; n_cur = n_next;
        std::swap(qc, qn);
    }
    const uint32_t n_nodes = base;
    level_base.push_back(n_nodes);
    for (size_t l = level_base.size() - 1; l-- > 0;)
        hipLaunchKernelGGL(k_depth, blocks(level_base[l + 1] - level_base[l]), block, 0, stream, nodes.p, level_base[l], level_base[l + 1] - level_base[l], depth.p, need.p);
    size_t red_bytes = 0;
    Dev<float> d_sum; Dev<unsigned char> red_tmp;
    LB_TRY(d_sum.alloc(1));
    LB_TRY(rocprim::reduce(nullptr, red_bytes, cost.p, d_sum.p, 0.f, (size_t)n_nodes, rocprim::plus<float>(), stream));
    LB_TRY(red_tmp.alloc(red_bytes));
    LB_TRY(rocprim::reduce(red_tmp.p, red_bytes, cost.p, d_sum.p, 0.f, (size_t)n_nodes, rocprim::plus<float>(), stream));
    uint32_t h_depth = 0, h_need = 0; float h_cost = 0.f;
    LB_TRY(hipMemcpyAsync(&h_depth, depth.p, 4, hipMemcpyDeviceToHost, stream));
    LB_TRY(hipMemcpyAsync(&h_need, need.p, 4, hipMemcpyDeviceToHost, stream));
    LB_TRY(hipMemcpyAsync(&h_cost, d_sum.p, 4, hipMemcpyDeviceToHost, stream));
    // exact-size node array for the scene (the work array is sized for the worst case)
    Dev<float4> final_nodes;
    LB_TRY(final_nodes.alloc((size_t)n_nodes * 4));
    LB_TRY(hipMemcpyAsync(final_nodes.p, nodes.p, (size_t)n_nodes * 64, hipMemcpyDeviceToDevice, stream));
    LB_TRY(hipStreamSynchronize(stream));
    LB_TRY(hipGetLastError());
    out.nodes = final_nodes.release(); out.tris = tris.release();
    out.n_nodes = n_nodes; out.max_depth = h_depth; out.stack_need = h_need; out.sah_cost = h_cost;
    out.device_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return hipSuccess;
}

} // namespace ptrt
