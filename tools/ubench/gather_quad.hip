// Microbenchmark (developer aid): a lane needs the 64-byte record of ITS OWN random node, as in a BVH4Q visit. Two ways to fetch it:
//   mode 0  each lane issues 4 x global_load_dwordx4 on its own line: 4 L1 accesses per lane and record (what k_extend does);
//   mode 1  the 4 lanes of a quad fetch one another's records together: in step k every lane of the quad loads row (lane & 3) of
//           the record lane k wants (one coalesced 64-byte access per record), then a 4x4 transpose across the quad (DPP quad_perm +
//           v_cndmask, 64 instructions) hands every lane its own 4 rows;
//   mode 2  as mode 1 with the transpose written as v_cndmask_b32_dpp (32 instructions).
// Per step a lane is active with probability `act`/256 and spends `alu` dependent FMAs on the record, like a traversal step.
//   usage: gather_quad <mode> <levels> <node_mb> <alu> <act>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
#define DEV __device__ __forceinline__

DEV uint32_t pcg(uint32_t x) { uint32_t s = x * 747796405u + 2891336453u; uint32_t w = ((s >> ((s >> 28) + 4u)) ^ s) * 277803737u; return (w >> 22) ^ w; }
template <int CTRL> DEV uint32_t dpp(uint32_t v) { return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, CTRL, 0xf, 0xf, true); }
template <int CTRL> DEV void xchg(bool b, uint32_t &x, uint32_t &y)
{
    const uint32_t px = dpp<CTRL>(x), py = dpp<CTRL>(y);
    const uint32_t nx = b ? py : x, ny = b ? y : px;
    x = nx; y = ny;
}
DEV void transpose4(uint32_t lane, uint32_t (&a)[4])
{
    const bool b0 = lane & 1u, b1 = lane & 2u;
    xchg<0xB1>(b0, a[0], a[1]); xchg<0xB1>(b0, a[2], a[3]);
    xchg<0x4E>(b1, a[0], a[2]); xchg<0x4E>(b1, a[1], a[3]);
}
// the same with v_cndmask_b32_dpp: D = vcc ? src1 : dpp(src0). One stage over 4 registers (pairs (0,1),(2,3) or (0,2),(1,3)).
#define STAGE_ASM(CTRLSTR, BITMASK, X0, Y0, X1, Y1)                                                                              \
    {                                                                                                                            \
        uint32_t nx0, ny0, nx1, ny1;                                                                                             \
        asm volatile("v_and_b32 %4, %9, %8\n"                                                                                     \
                     "v_cmp_ne_u32 vcc, 0, %4\n"                                                                                  \
                     "s_nop 1\n"                                                                                                  \
                     "v_cndmask_b32_dpp %1, %5, %6, vcc " CTRLSTR " row_mask:0xf bank_mask:0xf bound_ctrl:1\n" /* y' = b ? y : dpp(x) */ \
                     "v_cndmask_b32_dpp %3, %7, %10, vcc " CTRLSTR " row_mask:0xf bank_mask:0xf bound_ctrl:1\n"                   \
                     "s_not_b64 vcc, vcc\n"                                                                                       \
                     "v_cndmask_b32_dpp %0, %6, %5, vcc " CTRLSTR " row_mask:0xf bank_mask:0xf bound_ctrl:1\n" /* x' = !b ? x : dpp(y) */ \
                     "v_cndmask_b32_dpp %2, %10, %7, vcc " CTRLSTR " row_mask:0xf bank_mask:0xf bound_ctrl:1\n"                   \
                     : "=&v"(nx0), "=&v"(ny0), "=&v"(nx1), "=&v"(ny1), "=&v"(tmp)                                                 \
                     : "v"(X0), "v"(Y0), "v"(X1), "v"(lane), "v"(BITMASK), "v"(Y1)                                                \
                     : "vcc");                                                                                                    \
        X0 = nx0; Y0 = ny0; X1 = nx1; Y1 = ny1;                                                                                   \
    }
DEV void transpose4_asm(uint32_t lane, uint32_t (&a)[4])
{
    uint32_t tmp;
    const uint32_t one = 1u, two = 2u;
    STAGE_ASM("quad_perm:[1,0,3,2]", one, a[0], a[1], a[2], a[3])
    STAGE_ASM("quad_perm:[2,3,0,1]", two, a[0], a[2], a[1], a[3])
    (void)tmp;
}

template <int MODE>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(8, 8)))
k_quad(const float4 *__restrict__ nodes, uint32_t n_nodes, uint32_t levels, uint32_t alu, uint32_t act256, uint32_t rays_per_lane, float *out)
{
    const uint32_t lane = threadIdx.x & 63u, gid = blockIdx.x * 64 + threadIdx.x, j = lane & 3u;
    uint32_t rnd = pcg(gid);
    float acc = 0.f;
    for (uint32_t ray = 0; ray < rays_per_lane; ++ray) {
        uint32_t first = 0, width = 1;
        for (uint32_t k = 0; k < levels; ++k) {
            const uint32_t w = width < n_nodes - first ? width : n_nodes - first;
            const bool act = (pcg(rnd ^ 0x9e37u) & 255u) < act256;
            const int idx = act ? (int)(first + rnd % w) : -1;
            float4 r0 = make_float4(0, 0, 0, 0), r1 = r0, r2 = r0, r3 = r0;
            if (MODE == 0) {
                if (act) { const float4 *p = nodes + (size_t)idx * 4; r0 = p[0]; r1 = p[1]; r2 = p[2]; r3 = p[3]; }
            } else {
                float4 R0 = r0, R1 = r0, R2 = r0, R3 = r0;
                const int c0 = __builtin_amdgcn_mov_dpp(idx, 0x00, 0xf, 0xf, true), c1 = __builtin_amdgcn_mov_dpp(idx, 0x55, 0xf, 0xf, true),
                          c2 = __builtin_amdgcn_mov_dpp(idx, 0xAA, 0xf, 0xf, true), c3 = __builtin_amdgcn_mov_dpp(idx, 0xFF, 0xf, 0xf, true);
                if (c0 >= 0) R0 = nodes[(size_t)c0 * 4 + j];
                if (c1 >= 0) R1 = nodes[(size_t)c1 * 4 + j];
                if (c2 >= 0) R2 = nodes[(size_t)c2 * 4 + j];
                if (c3 >= 0) R3 = nodes[(size_t)c3 * 4 + j];
                uint32_t cx[4] = { __float_as_uint(R0.x), __float_as_uint(R1.x), __float_as_uint(R2.x), __float_as_uint(R3.x) };
                uint32_t cy[4] = { __float_as_uint(R0.y), __float_as_uint(R1.y), __float_as_uint(R2.y), __float_as_uint(R3.y) };
                uint32_t cz[4] = { __float_as_uint(R0.z), __float_as_uint(R1.z), __float_as_uint(R2.z), __float_as_uint(R3.z) };
                uint32_t cw[4] = { __float_as_uint(R0.w), __float_as_uint(R1.w), __float_as_uint(R2.w), __float_as_uint(R3.w) };
                if (MODE == 1) { transpose4(lane, cx); transpose4(lane, cy); transpose4(lane, cz); transpose4(lane, cw); }
                else { transpose4_asm(lane, cx); transpose4_asm(lane, cy); transpose4_asm(lane, cz); transpose4_asm(lane, cw); }
                r0 = make_float4(__uint_as_float(cx[0]), __uint_as_float(cy[0]), __uint_as_float(cz[0]), __uint_as_float(cw[0]));
                r1 = make_float4(__uint_as_float(cx[1]), __uint_as_float(cy[1]), __uint_as_float(cz[1]), __uint_as_float(cw[1]));
                r2 = make_float4(__uint_as_float(cx[2]), __uint_as_float(cy[2]), __uint_as_float(cz[2]), __uint_as_float(cw[2]));
                r3 = make_float4(__uint_as_float(cx[3]), __uint_as_float(cy[3]), __uint_as_float(cz[3]), __uint_as_float(cw[3]));
            }
            if (act) {
                float t = r0.x + r0.y + r0.z + r0.w + r1.x + r1.y + r1.z + r1.w + r2.x + r2.y + r2.z + r2.w + r3.x + r3.y + r3.z + r3.w;
                for (uint32_t a = 0; a < alu; ++a) t = __builtin_fmaf(t, 1.0001f, 0.25f);
                acc += t;
                rnd = pcg(rnd ^ __float_as_uint(t) ^ k);
                first += width; width <<= 2;
                if (first >= n_nodes) { first = n_nodes - w; width = w; }
            } else rnd = pcg(rnd + k);
        }
    }
    out[gid] = acc;
}

int main(int argc, char **argv)
{
    const int mode = argc > 1 ? atoi(argv[1]) : 0;
    const uint32_t levels = argc > 2 ? atoi(argv[2]) : 16;
    const double node_mb = argc > 3 ? atof(argv[3]) : 17.2;
    const uint32_t alu = argc > 4 ? atoi(argv[4]) : 130, act = argc > 5 ? atoi(argv[5]) : 128;
    const uint32_t n_nodes = (uint32_t)(node_mb * 1e6 / 64), rays = 4, waves = 256 * 32 * 4;
    std::vector<float> h((size_t)n_nodes * 16);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)(i % 97) * 0.01f;
    float4 *dn; float *o;
    CK(hipMalloc(&dn, (size_t)n_nodes * 64)); CK(hipMalloc(&o, (size_t)waves * 64 * 4));
    CK(hipMemcpy(dn, h.data(), (size_t)n_nodes * 64, hipMemcpyHostToDevice));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    double best = 1e30; std::vector<float> res(64);
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(a));
        if (mode == 0) hipLaunchKernelGGL(k_quad<0>, dim3(waves), dim3(64), 0, 0, dn, n_nodes, levels, alu, act, rays, o);
        else if (mode == 1) hipLaunchKernelGGL(k_quad<1>, dim3(waves), dim3(64), 0, 0, dn, n_nodes, levels, alu, act, rays, o);
        else hipLaunchKernelGGL(k_quad<2>, dim3(waves), dim3(64), 0, 0, dn, n_nodes, levels, alu, act, rays, o);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        if (rep && ms < best) best = ms;
    }
    CK(hipMemcpy(res.data(), o, 64 * 4, hipMemcpyDeviceToHost));
    double sum = 0; for (float v : res) sum += v;
    const double steps = (double)waves * 64 * rays * levels; // lane-steps; records fetched = steps * act / 256
    printf("mode %d levels %u node_mb %.1f alu %u act %u/256: %.3f ms, %.1f G lane-steps/s, %.1f G records/s (checksum %.6g)\n", mode, levels, node_mb, alu, act,
           best, steps / best / 1e6, steps * act / 256.0 / best / 1e6, sum);
    return 0;
}
