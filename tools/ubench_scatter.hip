// Developer microbenchmark (not product code): what the partitioned alive-key pass is made of.
//   (1) scattered writes of 2^26 8-byte pairs into a 1 GiB workspace, by write granularity: 8 B per lane at
//       random places, aligned 32 B / 64 B / 128 B blocks written by one lane or by a lane group;
//   (2) LDS atomic throughput: 32- and 64-bit add / max / CAS on random slots of a 64 KiB table.
// hipcc --offload-arch=gfx950 -O3 tools/ubench_scatter.hip -o tools/ubench_scatter
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

__device__ __forceinline__ uint64_t mix64(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// GROUP lanes write one aligned block of GROUP * PER * 8 bytes; PER = u64 words per lane (1, 2 = 16 B store)
template <int GROUP, int PER>
__global__ __launch_bounds__(256) void k_scatter(unsigned long long *ws, uint64_t ws_words, uint64_t n_words, uint64_t seed)
{
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    constexpr uint64_t kBlock = (uint64_t)GROUP * PER;          // words per block
    const uint64_t nblocks_ws = ws_words / kBlock;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i * PER < n_words; i += stride) {
        const uint64_t blk = mix64(seed + i / GROUP) % nblocks_ws;
        unsigned long long *p = ws + blk * kBlock + (i % GROUP) * PER;
        if (PER == 1) {
            p[0] = i;
        } else if (PER == 2) {
            *reinterpret_cast<ulonglong2 *>(p) = make_ulonglong2(i, i + 1);
        } else {
#pragma unroll
            for (int k = 0; k < PER; k += 2) *reinterpret_cast<ulonglong2 *>(p + k) = make_ulonglong2(i, i + k);
        }
    }
}

// MODE 0: u32 atomicAdd returning, 1: u64 atomicMax (no return), 2: u64 read + CAS, 3: u32 atomicMax, 4: u64 plain read + write
template <int MODE>
__global__ __launch_bounds__(512) void k_lds(uint64_t iters, uint64_t seed, unsigned long long *sink)
{
    __shared__ unsigned long long tbl[8192];
    for (int e = threadIdx.x; e < 8192; e += 512) tbl[e] = 0;
    __syncthreads();
    uint32_t *t32 = reinterpret_cast<uint32_t *>(tbl);
    unsigned long long acc = 0;
    uint64_t r = mix64(seed + blockIdx.x * 512 + threadIdx.x);
    for (uint64_t i = 0; i < iters; i++) {
        r = r * 6364136223846793005ull + 1442695040888963407ull;
        const uint32_t pos = (uint32_t)(r >> 40);
        if (MODE == 0) acc += atomicAdd(&t32[pos & 16383u], 1u);
        else if (MODE == 1) atomicMax(&tbl[pos & 8191u], r);
        else if (MODE == 2) {
            const unsigned long long cur = tbl[pos & 8191u];
            acc += atomicCAS(&tbl[pos & 8191u], cur, r);
        } else if (MODE == 3) atomicMax(&t32[pos & 16383u], (uint32_t)r);
        else {
            const unsigned long long cur = tbl[pos & 8191u];
            if (cur < r) tbl[pos & 8191u] = r;
            acc += cur;
        }
    }
    if (acc == 0x123456789ull) *sink = acc;
}

template <typename F>
float time_ms(F f, int reps)
{
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    float best = 1e9f;
    for (int r = 0; r < reps; r++) {
        hipEventRecord(a);
        f(r);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
    }
    return best;
}

int main()
{
    const uint64_t ws_words = 1ull << 27;    // 1 GiB workspace
    const uint64_t n_words = 1ull << 26;     // 512 MiB of pairs
    unsigned long long *ws, *sink;
    if (hipMalloc(&ws, ws_words * 8) != hipSuccess) return 1;
    hipMalloc(&sink, 8);
    hipMemset(ws, 0, ws_words * 8);
    const int grid = 256 * 8;
#define RUN(G, P, label)                                                                                         \
    {                                                                                                            \
        float ms = time_ms([&](int r) { hipLaunchKernelGGL((k_scatter<G, P>), dim3(grid), dim3(256), 0, 0, ws,    \
                                                           ws_words, n_words, 11 + r); }, 3);                    \
        printf("scatter %-44s %7.3f ms  %6.1f GB/s\n", label, ms, n_words * 8 / ms / 1e6);                       \
    }
    RUN(1, 1, "8 B per lane, random places");
    RUN(2, 1, "16 B blocks, 2 lanes x 8 B");
    RUN(4, 1, "32 B blocks, 4 lanes x 8 B");
    RUN(2, 2, "32 B blocks, 2 lanes x 16 B");
    RUN(1, 4, "32 B blocks, 1 lane x 2 x 16 B");
    RUN(8, 1, "64 B blocks, 8 lanes x 8 B");
    RUN(4, 2, "64 B blocks, 4 lanes x 16 B");
    RUN(1, 8, "64 B blocks, 1 lane x 4 x 16 B");
    RUN(16, 1, "128 B blocks, 16 lanes x 8 B");
    RUN(8, 2, "128 B blocks, 8 lanes x 16 B");
    RUN(64, 2, "1 KiB blocks, 64 lanes x 16 B (coalesced)");
    const uint64_t iters = 4096;
    const int lgrid = 512;      // 2 workgroups of 512 per CU
#define LDS(M, label)                                                                                            \
    {                                                                                                            \
        float ms = time_ms([&](int r) { hipLaunchKernelGGL((k_lds<M>), dim3(lgrid), dim3(512), 0, 0, iters,       \
                                                           (uint64_t)(5 + r), sink); }, 3);                      \
        printf("lds %-48s %7.3f ms  %7.1f G ops/s chip (%5.2f lanes/clk/CU @2.4GHz)\n", label, ms,              \
               (double)lgrid * 512 * iters / ms / 1e6, (double)lgrid * 512 * iters / ms / 1e6 / 256 / 2.4);      \
    }
    LDS(0, "u32 atomicAdd returning");
    LDS(3, "u32 atomicMax no return");
    LDS(1, "u64 atomicMax no return");
    LDS(2, "u64 read + CAS returning");
    LDS(4, "u64 read + conditional write");
    return 0;
}
