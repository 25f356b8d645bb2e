// Developer microbenchmark (not product code): throughput of 64-bit atomicMax on random slots of a
// table, by memory scope and working-set size.  Informs the alive-key pass design.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>

__device__ __forceinline__ uint64_t mix64(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

template <int SCOPE, bool RET>
__global__ __launch_bounds__(256) void k_atomic(unsigned long long *table, uint64_t mask, uint64_t n, uint64_t seed,
                                                unsigned long long *sink)
{
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    unsigned long long acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        const uint64_t slot = mix64(seed + i) & mask;
        const unsigned long long v = (i << 1) | 1ull;
        if (RET) acc += __hip_atomic_fetch_max(&table[slot], v, __ATOMIC_RELAXED, SCOPE);
        else (void)__hip_atomic_fetch_max(&table[slot], v, __ATOMIC_RELAXED, SCOPE);
    }
    if (RET && acc == 0x1234567) *sink = acc;
}

// plain (non-atomic) read-modify-write for comparison
__global__ __launch_bounds__(256) void k_plain(unsigned long long *table, uint64_t mask, uint64_t n, uint64_t seed)
{
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        const uint64_t slot = mix64(seed + i) & mask;
        const unsigned long long v = (i << 1) | 1ull;
        if (table[slot] < v) table[slot] = v;
    }
}

int main()
{
    const uint64_t n = 1ull << 26;
    unsigned long long *table, *sink;
    const uint64_t max_slots = 1ull << 32;
    if (hipMalloc(&table, max_slots * 8) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMalloc(&sink, 8);
    hipMemset(table, 0, max_slots * 8);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    const int grid = 256 * 8;
    for (int log2slots : {14, 17, 20, 23, 26, 29, 32}) {
        const uint64_t mask = (1ull << log2slots) - 1;
        for (int mode = 0; mode < 5; mode++) {
            float best = 1e9;
            for (int rep = 0; rep < 3; rep++) {
                hipEventRecord(a);
                switch (mode) {
                case 0: hipLaunchKernelGGL((k_atomic<__HIP_MEMORY_SCOPE_AGENT, false>), dim3(grid), dim3(256), 0, 0, table, mask, n, 7 + rep, sink); break;
                case 1: hipLaunchKernelGGL((k_atomic<__HIP_MEMORY_SCOPE_AGENT, true>), dim3(grid), dim3(256), 0, 0, table, mask, n, 7 + rep, sink); break;
                case 2: hipLaunchKernelGGL((k_atomic<__HIP_MEMORY_SCOPE_WORKGROUP, false>), dim3(grid), dim3(256), 0, 0, table, mask, n, 7 + rep, sink); break;
                case 3: hipLaunchKernelGGL((k_atomic<__HIP_MEMORY_SCOPE_WORKGROUP, true>), dim3(grid), dim3(256), 0, 0, table, mask, n, 7 + rep, sink); break;
                default: hipLaunchKernelGGL(k_plain, dim3(grid), dim3(256), 0, 0, table, mask, n, 7 + rep); break;
                }
                hipEventRecord(b);
                hipEventSynchronize(b);
                float ms; hipEventElapsedTime(&ms, a, b);
                if (ms < best) best = ms;
            }
            const char *names[5] = {"agent-scope atomicMax (no return)", "agent-scope atomicMax (returning)",
                                    "workgroup-scope atomicMax (no return; L2-local, NOT cross-XCD coherent)",
                                    "workgroup-scope atomicMax (returning)", "plain load/compare/store (racy)"};
            printf("slots=2^%-2d (%8.1f MB)  %-72s %7.3f ms  %6.2f G/s\n", log2slots, (double)(8ull << log2slots) / 1e6,
                   names[mode], best, n / best / 1e6);
        }
    }
    return 0;
}
