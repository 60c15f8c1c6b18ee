// Microbenchmark (developer aid): dependent random gather of records of R 16-byte rows placed at a stride of S rows,
// one record per lane per step (what a BVH node visit costs). Question: does a 96-byte node in a 128-byte line (BVH8)
// cost more per visit than a 64-byte node (BVH4Q)?   ./gather_rows <table MB>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
__device__ __forceinline__ uint32_t pcg(uint32_t x) { uint32_t s = x * 747796405u + 2891336453u; uint32_t w = ((s >> ((s >> 28) + 4u)) ^ s) * 277803737u; return (w >> 22) ^ w; }

template <int R, int S>
__global__ void __launch_bounds__(64) k_gather(const float4 *__restrict__ table, uint32_t n_rec, uint32_t steps, float *out)
{
    const uint32_t gid = blockIdx.x * 64 + threadIdx.x;
    uint32_t idx = pcg(gid) % n_rec;
    float acc = 0.f;
    for (uint32_t s = 0; s < steps; ++s) {
        const float4 *p = table + (size_t)idx * S;
        float4 r[R];
#pragma unroll
        for (int i = 0; i < R; ++i) r[i] = p[i];
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < R; ++i) t += r[i].x + r[i].w;
        acc += t;
        idx = pcg(idx ^ __float_as_uint(t) ^ s) % n_rec;
    }
    out[gid] = acc;
}

template <int R, int S>
void run(const float4 *d, size_t table_bytes, float *o, const char *what)
{
    const uint32_t n_rec = (uint32_t)(table_bytes / (S * 16)), steps = 16, waves = 256 * 28 * 4;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL((k_gather<R, S>), dim3(waves), dim3(64), 0, 0, d, n_rec, steps, o);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        if (rep) best = ms < best ? ms : best;
    }
    const double recs = (double)waves * 64 * steps;
    printf("%-34s table %6.1f MB (%8u records): %7.3f ms  %6.2f G records/s  %5.2f TB/s of rows read\n", what, table_bytes / 1e6, n_rec, best,
           recs / best / 1e6, recs * R * 16 / best / 1e9);
}

int main(int argc, char **argv)
{
    const size_t mb = argc > 1 ? atoi(argv[1]) : 32, bytes = mb << 20;
    std::vector<float> h(bytes / 4);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)(i % 97) * 0.01f;
    float4 *d; float *o;
    CK(hipMalloc(&d, bytes)); CK(hipMalloc(&o, (size_t)256 * 28 * 4 * 64 * 4));
    CK(hipMemcpy(d, h.data(), bytes, hipMemcpyHostToDevice));
    run<4, 4>(d, bytes, o, "64-B record, 64-B stride");
    run<4, 8>(d, bytes, o, "64-B record, 128-B stride");
    run<6, 8>(d, bytes, o, "96-B record, 128-B stride");
    run<8, 8>(d, bytes, o, "128-B record, 128-B stride");
    run<5, 5>(d, bytes, o, "80-B record, 80-B stride");
    return 0;
}
