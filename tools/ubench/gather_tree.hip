// Microbenchmark (developer aid, and the "gather ceiling" bench.py quotes): what MI355X sustains for the MEMORY pattern of a
// BVH4Q traversal with no arithmetic at all. One ray per lane; a "ray" is a dependent chain of `levels` 64-byte node fetches,
// the k-th from a uniformly random node of level k of a complete 4-ary tree laid out breadth-first (level k = 4^k nodes; the top
// stays in L1/L2, the bottom is as wide as the scene's node array), followed by `tris` 64-byte triangle-record fetches from a
// table of `tri_mb` MB. Fetch = 4 x 16 B per lane, like k_extend. Reported: records/s at full occupancy (7 waves/SIMD).
//   usage: gather_tree <levels> <node_mb> <tris_per_ray> <tri_mb>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ uint32_t pcg(uint32_t x) { uint32_t s = x * 747796405u + 2891336453u; uint32_t w = ((s >> ((s >> 28) + 4u)) ^ s) * 277803737u; return (w >> 22) ^ w; }

__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(7, 7)))
k_tree(const float4 *__restrict__ nodes, uint32_t n_nodes, uint32_t levels, const float4 *__restrict__ tris, uint32_t n_tris, uint32_t tri_steps,
       uint32_t rays_per_lane, float *out)
{
    const uint32_t gid = blockIdx.x * 64 + threadIdx.x;
    uint32_t rnd = pcg(gid);
    float acc = 0.f;
    for (uint32_t ray = 0; ray < rays_per_lane; ++ray) {
        uint32_t first = 0, width = 1; // level k occupies nodes [first, first + width)
        for (uint32_t k = 0; k < levels; ++k) {
            uint32_t w = width < n_nodes - first ? width : n_nodes - first; // the last level is clipped to the array
            const uint32_t idx = first + rnd % w;
            const float4 *p = nodes + (size_t)idx * 4;
            const float4 r0 = p[0], r1 = p[1], r2 = p[2], r3 = p[3];
            acc += r0.x + r1.y + r2.z + r3.w;
            rnd = pcg(rnd ^ __float_as_uint(r3.w) ^ k); // dependent chain like a traversal
            first += width; width <<= 2;
            if (first >= n_nodes) { first = n_nodes - w; width = w; }
        }
        for (uint32_t k = 0; k < tri_steps; ++k) {
            const float4 *p = tris + (size_t)(rnd % n_tris) * 4;
            const float4 r0 = p[0], r1 = p[1], r2 = p[2];
            acc += r0.x + r1.y + r2.z;
            rnd = pcg(rnd ^ __float_as_uint(r2.z) ^ k);
        }
    }
    out[gid] = acc;
}

int main(int argc, char **argv)
{
    const uint32_t levels = argc > 1 ? atoi(argv[1]) : 10;
    const double node_mb = argc > 2 ? atof(argv[2]) : 17.2, tri_mb = argc > 4 ? atof(argv[4]) : 66.8;
    const uint32_t tri_steps = argc > 3 ? atoi(argv[3]) : 1;
    const uint32_t n_nodes = (uint32_t)(node_mb * 1e6 / 64), n_tris = (uint32_t)(tri_mb * 1e6 / 64), rays = 4, waves = 256 * 28 * 4;
    std::vector<float> h((size_t)(n_nodes > n_tris ? n_nodes : n_tris) * 16);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)(i % 97) * 0.01f;
    float4 *dn, *dt; float *o;
    CK(hipMalloc(&dn, (size_t)n_nodes * 64)); CK(hipMalloc(&dt, (size_t)n_tris * 64)); CK(hipMalloc(&o, (size_t)waves * 64 * 4));
    CK(hipMemcpy(dn, h.data(), (size_t)n_nodes * 64, hipMemcpyHostToDevice));
    CK(hipMemcpy(dt, h.data(), (size_t)n_tris * 64, hipMemcpyHostToDevice));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    double best = 0;
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL(k_tree, dim3(waves), dim3(64), 0, 0, dn, n_nodes, levels, dt, n_tris, tri_steps, rays, o);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        const double recs = (double)waves * 64 * rays * (levels + tri_steps);
        if (rep && recs / ms / 1e6 > best) best = recs / ms / 1e6;
    }
    printf("{\"levels\": %u, \"node_mb\": %.1f, \"tris_per_ray\": %u, \"tri_mb\": %.1f, \"g_records_per_s\": %.2f}\n", levels, node_mb, tri_steps, tri_mb, best);
    return 0;
}
