// lbvh.hip — GPU construction of a binary LBVH (the reference has no acceleration structure).
//   k_tri_boxes   : per triangle, the padded box of docs/SPEC.md §4.1 and its centroid; scene centroid bounds by atomics
//   k_morton      : 30-bit Morton code of the centroid
//   (rocPRIM)     : radix sort of (code, triangle) pairs
//   k_hierarchy   : Karras 2012 — one thread per internal node finds its key range and split by binary search on the
//                   common-prefix length (ties between equal codes are broken by the index, so duplicates form a balanced subtree)
//   k_refit       : leaves walk up; the second arrival at a node unites the children's boxes (agent-scope fences between)
// The binary tree is copied back and packed into the blob layouts by build_bvh_from_binary() on the host: the closest hit
// does not depend on the tree (SPEC §4), so a scene committed with this builder renders the same picture bit for bit.
#include <cstring> // rocprim's texture_cache_iterator.hpp needs ::memset
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>
#include <chrono>
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
    ~Dev() { if (p) (void)hipFree(p); }
};

} // namespace

#define LB_TRY(x) do { hipError_t _e = (x); if (_e != hipSuccess) return _e; } while (0)

hipError_t build_lbvh_device(hipStream_t stream, const float *verts9, uint32_t n, BinaryBvh &out)
{
    const auto t0 = std::chrono::steady_clock::now();
    out = BinaryBvh{};
    if (n < 2) return hipErrorInvalidValue;
    Dev<float> d_verts, d_leaf_box, d_cent, d_node_box;
    Dev<uint32_t> d_bounds, d_keys, d_vals, d_keys2, d_vals2, d_first, d_last, d_arrived;
    Dev<int32_t> d_left, d_right, d_pn, d_pl;
    Dev<unsigned char> d_tmp;
    LB_TRY(d_verts.alloc((size_t)n * 9)); LB_TRY(d_leaf_box.alloc((size_t)n * 6)); LB_TRY(d_cent.alloc((size_t)n * 3));
    LB_TRY(d_node_box.alloc((size_t)(n - 1) * 6)); LB_TRY(d_bounds.alloc(6));
    LB_TRY(d_keys.alloc(n)); LB_TRY(d_vals.alloc(n)); LB_TRY(d_keys2.alloc(n)); LB_TRY(d_vals2.alloc(n));
    LB_TRY(d_first.alloc(n - 1)); LB_TRY(d_last.alloc(n - 1)); LB_TRY(d_arrived.alloc(n - 1));
    LB_TRY(d_left.alloc(n - 1)); LB_TRY(d_right.alloc(n - 1)); LB_TRY(d_pn.alloc(n - 1)); LB_TRY(d_pl.alloc(n));

    LB_TRY(hipMemcpyAsync(d_verts.p, verts9, (size_t)n * 36, hipMemcpyHostToDevice, stream));
    const uint32_t init_bounds[6] = { 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u, 0u };
    LB_TRY(hipMemcpyAsync(d_bounds.p, init_bounds, sizeof init_bounds, hipMemcpyHostToDevice, stream));
    LB_TRY(hipMemsetAsync(d_arrived.p, 0, (size_t)(n - 1) * 4, stream));
    const dim3 grid((n + 255) / 256), block(256);
    hipLaunchKernelGGL(k_tri_boxes, grid, block, 0, stream, d_verts.p, n, d_leaf_box.p, d_cent.p, d_bounds.p);
    uint32_t hb[6];
    LB_TRY(hipMemcpyAsync(hb, d_bounds.p, sizeof hb, hipMemcpyDeviceToHost, stream));
    LB_TRY(hipStreamSynchronize(stream));
    float3 cmin, inv;
    {
        const float lo[3] = { f_unord(hb[0]), f_unord(hb[1]), f_unord(hb[2]) }, hi[3] = { f_unord(hb[3]), f_unord(hb[4]), f_unord(hb[5]) };
        cmin = make_float3(lo[0], lo[1], lo[2]);
        inv = make_float3(hi[0] > lo[0] ? 1.0f / (hi[0] - lo[0]) : 0.f, hi[1] > lo[1] ? 1.0f / (hi[1] - lo[1]) : 0.f, hi[2] > lo[2] ? 1.0f / (hi[2] - lo[2]) : 0.f);
    }
    hipLaunchKernelGGL(k_morton, grid, block, 0, stream, d_cent.p, n, cmin, inv, d_keys.p, d_vals.p);
    size_t tmp_bytes = 0;
    LB_TRY(rocprim::radix_sort_pairs(nullptr, tmp_bytes, d_keys.p, d_keys2.p, d_vals.p, d_vals2.p, (size_t)n, 0, 30, stream));
    LB_TRY(d_tmp.alloc(tmp_bytes));
    LB_TRY(rocprim::radix_sort_pairs(d_tmp.p, tmp_bytes, d_keys.p, d_keys2.p, d_vals.p, d_vals2.p, (size_t)n, 0, 30, stream));
    hipLaunchKernelGGL(k_hierarchy, dim3((n - 1 + 255) / 256), block, 0, stream, d_keys2.p, (int)n, d_left.p, d_right.p, d_first.p, d_last.p, d_pn.p, d_pl.p);
    hipLaunchKernelGGL(k_refit, grid, block, 0, stream, d_vals2.p, d_leaf_box.p, (int)n, d_left.p, d_right.p, d_pn.p, d_pl.p, d_node_box.p, d_arrived.p);
    LB_TRY(hipGetLastError());

    out.order.resize(n); out.left.resize(n - 1); out.right.resize(n - 1); out.first.resize(n - 1); out.last.resize(n - 1);
    out.box.resize((size_t)(n - 1) * 6);
    LB_TRY(hipMemcpyAsync(out.order.data(), d_vals2.p, (size_t)n * 4, hipMemcpyDeviceToHost, stream));
    LB_TRY(hipMemcpyAsync(out.left.data(), d_left.p, (size_t)(n - 1) * 4, hipMemcpyDeviceToHost, stream));
    LB_TRY(hipMemcpyAsync(out.right.data(), d_right.p, (size_t)(n - 1) * 4, hipMemcpyDeviceToHost, stream));
    LB_TRY(hipMemcpyAsync(out.first.data(), d_first.p, (size_t)(n - 1) * 4, hipMemcpyDeviceToHost, stream));
    LB_TRY(hipMemcpyAsync(out.last.data(), d_last.p, (size_t)(n - 1) * 4, hipMemcpyDeviceToHost, stream));
    LB_TRY(hipMemcpyAsync(out.box.data(), d_node_box.p, (size_t)(n - 1) * 24, hipMemcpyDeviceToHost, stream));
    LB_TRY(hipStreamSynchronize(stream));
    out.device_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return hipSuccess;
}

} // namespace ptrt
