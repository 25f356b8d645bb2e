// Developer microbenchmark (not product code): issue rate of the integer operations the FNV chain is made of.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_valu.hip -o tools/ubench_valu
// Each wave runs ITER rounds of 4 independent dependency chains of the operation; 4 waves per SIMD (1024-thread
// workgroups, one per CU).  Prints cycles per wave instruction and SIMD (2.4 GHz assumed).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

template <int MODE>
__global__ __launch_bounds__(1024) void k_ops(uint32_t iters, uint32_t seed, uint32_t *sink)
{
    uint32_t a = seed + threadIdx.x, b = a * 3u + 1u, c = a * 5u + 2u, d = a * 7u + 3u;
    const uint32_t m = 0x811c9dc5u;
    for (uint32_t i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if (MODE == 0) { a *= m; b *= m; c *= m; d *= m; }                                   // v_mul_lo_u32
            else if (MODE == 1) { a = (a & 0xFFFFFFu) * 0x9dc5u; b = (b & 0xFFFFFFu) * 0x9dc5u; c = (c & 0xFFFFFFu) * 0x9dc5u; d = (d & 0xFFFFFFu) * 0x9dc5u; }
            else if (MODE == 2) { a ^= (a >> 7); b ^= (b >> 7); c ^= (c >> 7); d ^= (d >> 7); }   // 2 full-rate ops (or one)
            else if (MODE == 3) { a = (a ^ (b & 0xFFu)) * m; b = (b ^ (c & 0xFFu)) * m; c = (c ^ (d & 0xFFu)) * m; d = (d ^ (a >> 24)) * m; }   // an FNV byte step
            else if (MODE == 4) { unsigned long long t = (unsigned long long)a * m; a = (uint32_t)t ^ (uint32_t)(t >> 32);
                                  t = (unsigned long long)b * m; b = (uint32_t)t ^ (uint32_t)(t >> 32);
                                  t = (unsigned long long)c * m; c = (uint32_t)t ^ (uint32_t)(t >> 32);
                                  t = (unsigned long long)d * m; d = (uint32_t)t ^ (uint32_t)(t >> 32); }  // mul_lo + mul_hi
        }
    }
    if ((a ^ b ^ c ^ d) == 0x12345u) *sink = a;
}

template <int MODE>
void run(const char *label, int ops_per_round)
{
    uint32_t *sink;
    (void)hipMalloc(&sink, 4);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    const uint32_t iters = 20000;
    float best = 1e9f;
    for (int r = 0; r < 3; r++) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k_ops<MODE>), dim3(256), dim3(1024), 0, 0, iters, 7u + r, sink);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    // per SIMD: 4 waves x iters x 8 x ops_per_round wave instructions
    const double insts = 4.0 * iters * 8 * ops_per_round;
    printf("%-40s %8.3f ms  %6.2f cycles per wave instruction and SIMD (2.4 GHz)\n", label, best, best * 1e-3 * 2.4e9 / insts);
}

int main()
{
    run<0>("v_mul_lo_u32 (4 chains)", 4);
    run<1>("v_mul_u32_u24 (4 chains)", 4);
    run<2>("xor + shift (4 chains, 2 ops each)", 8);
    run<3>("FNV byte step: and, xor, mul_lo", 12);
    run<4>("mul_lo + mul_hi + xor", 12);
    return 0;
}
