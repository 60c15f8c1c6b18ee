// Microbenchmark (developer aid): random gather of 64-byte records, one record per lane per step, dependent chain.
//  A: every lane loads its own record with 4 x 16-B loads (what k_extend does per node visit)
//  B: quad-cooperative: 4 instructions, in instruction j each lane loads row (lane&3) of the record of quad-lane j,
//     so the 4 lanes of a quad read one contiguous 64-B line per instruction; rows exchanged by DPP quad_perm
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ uint32_t pcg(uint32_t x) { uint32_t s = x * 747796405u + 2891336453u; uint32_t w = ((s >> ((s >> 28) + 4u)) ^ s) * 277803737u; return (w >> 22) ^ w; }

template <int QP> __device__ __forceinline__ uint32_t quad_bcast(uint32_t v) { return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, QP, 0xf, 0xf, true); }
// quad_perm selectors broadcasting quad lane j: (j | j<<2 | j<<4 | j<<6)
__device__ __forceinline__ uint32_t qb(uint32_t v, int j)
{
    switch (j) { case 0: return quad_bcast<0x00>(v); case 1: return quad_bcast<0x55>(v); case 2: return quad_bcast<0xAA>(v); default: return quad_bcast<0xFF>(v); }
}

template <int MODE>
__global__ void __launch_bounds__(64) k_gather(const float4 *__restrict__ table, uint32_t n_rec, uint32_t steps, float *out)
{
    const uint32_t gid = blockIdx.x * 64 + threadIdx.x, lane = threadIdx.x;
    uint32_t idx = pcg(gid) % n_rec;
    float acc = 0.f;
    for (uint32_t s = 0; s < steps; ++s) {
        float4 r0, r1, r2, r3;
        if (MODE == 0) {
            const float4 *p = table + (size_t)idx * 4;
            r0 = p[0]; r1 = p[1]; r2 = p[2]; r3 = p[3];
        } else {
            float4 v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t ij = qb(idx, j);
                v[j] = table[(size_t)ij * 4 + (lane & 3u)];
            }
            // lane q needs row k of ITS record = v[q] of quad-lane k. Exchange: for each k, candidate c_j = v[j] from lane k; pick j = q.
            const uint32_t q = lane & 3u;
            float4 rr[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float4 pick = make_float4(0, 0, 0, 0);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float4 c;
                    c.x = __uint_as_float(qb(__float_as_uint(v[j].x), k)); c.y = __uint_as_float(qb(__float_as_uint(v[j].y), k));
                    c.z = __uint_as_float(qb(__float_as_uint(v[j].z), k)); c.w = __uint_as_float(qb(__float_as_uint(v[j].w), k));
                    if (q == (uint32_t)j) pick = c;
                }
                rr[k] = pick;
            }
            r0 = rr[0]; r1 = rr[1]; r2 = rr[2]; r3 = rr[3];
        }
        acc += r0.x + r1.y + r2.z + r3.w;
        idx = pcg(idx ^ __float_as_uint(r3.w) ^ s) % n_rec; // dependent chain like a traversal
    }
    out[gid] = acc;
}

int main(int argc, char **argv)
{
    const uint32_t n_rec = argc > 1 ? atoi(argv[1]) : 480000, steps = 16, waves = 256 * 28 * 4;
    std::vector<float> h((size_t)n_rec * 16);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)(i % 97) * 0.01f;
    float4 *d; float *o;
    CK(hipMalloc(&d, h.size() * 4)); CK(hipMalloc(&o, (size_t)waves * 64 * 4));
    CK(hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int mode = 0; mode < 2; ++mode)
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(a));
            if (mode == 0) hipLaunchKernelGGL(k_gather<0>, dim3(waves), dim3(64), 0, 0, d, n_rec, steps, o);
            else hipLaunchKernelGGL(k_gather<1>, dim3(waves), dim3(64), 0, 0, d, n_rec, steps, o);
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            const double recs = (double)waves * 64 * steps;
            printf("mode %d (%s) table %.1f MB: %.3f ms, %.2f G records/s, %.2f TB/s\n", mode, mode ? "quad-cooperative" : "per-lane 4x16B",
                   n_rec * 64 / 1e6, ms, recs / ms / 1e6, recs * 64 / ms / 1e9);
        }
    return 0;
}
