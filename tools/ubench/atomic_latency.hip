// How long does a wavefront wait for a returning atomicAdd on a shared counter (the append of a queue re-pack)?
// 64 counters on separate 64-byte lines, one wave per workgroup, counter = workgroup % 64 (the tracer's shard mapping).
// Reports the mean wait per wave for agent scope (what the tracer uses) and, for comparison only, workgroup scope
// (performed in the XCD's own L2: not coherent across XCDs, so not usable for queues shared by a whole launch).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int SCOPE>
__global__ void __launch_bounds__(64) k(uint32_t *cnt, uint32_t *out, uint32_t *wait, int spin)
{
    const uint32_t c = (blockIdx.x & 63u) * 16u;
    // some ALU work first so that waves do not all arrive at once
    float x = (float)threadIdx.x;
    for (int i = 0; i < spin; ++i) x = __builtin_fmaf(x, 1.0001f, 0.5f);
    const unsigned long long t0 = wall_clock64();
    uint32_t base = 0;
    if (threadIdx.x == 0) base = __hip_atomic_fetch_add(&cnt[c], 64u, __ATOMIC_RELAXED, SCOPE);
    base = __shfl(base, 0, 64);
    out[(base + threadIdx.x) & 0xfffffu] = (uint32_t)x;
    const unsigned long long t1 = wall_clock64();
    if (threadIdx.x == 0) wait[blockIdx.x] = (uint32_t)(t1 - t0); // per wave: a shared sum would itself be a one-address atomic (12 ns each)
}
int main()
{
    uint32_t *cnt, *out, *wait;
    hipMalloc(&cnt, 64 * 64); hipMalloc(&out, 4u << 20); const uint32_t waves = 1u << 18; hipMalloc(&wait, 4 * waves); uint32_t *h = new uint32_t[waves];
    int rate = 0; hipDeviceGetAttribute(&rate, hipDeviceAttributeWallClockRate, 0); // kHz
    for (int spin : { 0, 2000, 20000 })
        for (int scope = 0; scope < 2; ++scope) {
            float best = 1e30f; unsigned long long ticks = 0;
            for (int rep = 0; rep < 3; ++rep) {
                hipMemset(cnt, 0, 64 * 64); 
                hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
                hipEventRecord(e0);
                if (scope == 0) hipLaunchKernelGGL(k<__HIP_MEMORY_SCOPE_AGENT>, dim3(waves), dim3(64), 0, 0, cnt, out, wait, spin);
                else hipLaunchKernelGGL(k<__HIP_MEMORY_SCOPE_WORKGROUP>, dim3(waves), dim3(64), 0, 0, cnt, out, wait, spin);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) { best = ms; (void)hipMemcpy(h, wait, 4 * waves, hipMemcpyDeviceToHost); ticks = 0; for (uint32_t i = 0; i < waves; ++i) ticks += h[i]; }
            }
            printf("spin %5d scope %-9s: %.3f ms for %u waves, mean wait per wave %.0f ns (launch / atomics per counter = %.1f ns)\n", spin,
                   scope == 0 ? "agent" : "workgroup", best, waves, (double)ticks / waves * 1e6 / rate, best * 1e6 / (waves / 64));
        }
    return 0;
}
