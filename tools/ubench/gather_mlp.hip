// Microbenchmark (developer aid): what raises the rate of DEPENDENT random gathers on MI355X? One record per lane and step from a
// uniformly random node of a breadth-first 4-ary tree (17.2 MB by default), next index from the data (a traversal's memory pattern).
//   ROWS   16-byte loads per record (4 = the 64-byte BVH4Q node, 3 = a 48-byte node, 2 = 32 bytes)
//   CHAINS independent chains per lane (memory-level parallelism inside a lane)
//   WAVES  waves per SIMD (occupancy)
// Half of the lanes are active per step, as in the tracer's node loop. usage: gather_mlp [levels] [node_mb]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
#define DEV __device__ __forceinline__
DEV uint32_t pcg(uint32_t x) { uint32_t s = x * 747796405u + 2891336453u; uint32_t w = ((s >> ((s >> 28) + 4u)) ^ s) * 277803737u; return (w >> 22) ^ w; }

template <int ROWS, int CHAINS, int WAVES>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES)))
k(const float4 *__restrict__ nodes, uint32_t n_nodes, uint32_t levels, uint32_t rays_per_lane, float *out)
{
    const uint32_t gid = blockIdx.x * 64 + threadIdx.x;
    uint32_t rnd[CHAINS];
    for (int c = 0; c < CHAINS; ++c) rnd[c] = pcg(gid * 4u + c);
    float acc = 0.f;
    for (uint32_t ray = 0; ray < rays_per_lane; ++ray) {
        uint32_t first = 0, width = 1;
        for (uint32_t k2 = 0; k2 < levels; ++k2) {
            const uint32_t w = width < n_nodes - first ? width : n_nodes - first;
            float4 r[CHAINS][ROWS];
            bool act[CHAINS];
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) {
                act[c] = (pcg(rnd[c] ^ 0x9e37u) & 1u) != 0u;
                if (act[c]) {
                    const float4 *p = nodes + (size_t)(first + rnd[c] % w) * 4;
#pragma unroll
                    for (int q = 0; q < ROWS; ++q) r[c][q] = p[q];
                }
            }
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) {
                if (act[c]) {
                    float t = 0.f;
#pragma unroll
                    for (int q = 0; q < ROWS; ++q) t += r[c][q].x + r[c][q].y + r[c][q].z + r[c][q].w;
                    acc += t;
                    rnd[c] = pcg(rnd[c] ^ __float_as_uint(t) ^ k2);
                } else rnd[c] = pcg(rnd[c] + k2);
            }
            first += width; width <<= 2;
            if (first >= n_nodes) { first = n_nodes - w; width = w; }
        }
    }
    out[gid] = acc;
}

template <int ROWS, int CHAINS, int WAVES>
static void run(const float4 *dn, uint32_t n_nodes, uint32_t levels, float *o)
{
    const uint32_t rays = 4, waves = 256 * 4 * WAVES * 4;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    double best = 1e30;
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL((k<ROWS, CHAINS, WAVES>), dim3(waves), dim3(64), 0, 0, dn, n_nodes, levels, rays, o);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        if (rep && ms < best) best = ms;
    }
    const double recs = (double)waves * 64 * rays * levels * CHAINS * 0.5;
    printf("rows %d chains %d waves/SIMD %2d: %7.3f ms  %6.1f G records/s  %6.2f TB/s of rows\n", ROWS, CHAINS, WAVES, best, recs / best / 1e6, recs * ROWS * 16 / best / 1e9);
}

int main(int argc, char **argv)
{
    const uint32_t levels = argc > 1 ? atoi(argv[1]) : 16;
    const double node_mb = argc > 2 ? atof(argv[2]) : 17.2;
    const uint32_t n_nodes = (uint32_t)(node_mb * 1e6 / 64);
    std::vector<float> h((size_t)n_nodes * 16);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)(i % 97) * 0.01f;
    float4 *dn; float *o;
    CK(hipMalloc(&dn, (size_t)n_nodes * 64)); CK(hipMalloc(&o, (size_t)256 * 4 * 16 * 4 * 64 * 4));
    CK(hipMemcpy(dn, h.data(), (size_t)n_nodes * 64, hipMemcpyHostToDevice));
    run<4, 1, 4>(dn, n_nodes, levels, o); run<4, 1, 6>(dn, n_nodes, levels, o); run<4, 1, 8>(dn, n_nodes, levels, o);
    run<4, 1, 10>(dn, n_nodes, levels, o); run<4, 1, 12>(dn, n_nodes, levels, o); run<4, 1, 16>(dn, n_nodes, levels, o);
    run<3, 1, 8>(dn, n_nodes, levels, o); run<2, 1, 8>(dn, n_nodes, levels, o); run<1, 1, 8>(dn, n_nodes, levels, o);
    run<4, 2, 8>(dn, n_nodes, levels, o); run<4, 2, 4>(dn, n_nodes, levels, o); run<2, 2, 8>(dn, n_nodes, levels, o);
    return 0;
}
