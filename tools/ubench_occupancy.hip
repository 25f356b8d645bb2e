// Developer microbenchmark (not product code): how many workgroups of a given size / LDS footprint run on a CU
// at once.  Every workgroup spins ~100 us; the launch time over (grid / 256 CUs) rounds tells the residency.
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int THREADS>
__global__ __launch_bounds__(THREADS) void k_spin(unsigned long long ticks, unsigned *sink)
{
    extern __shared__ unsigned s[];
    if (threadIdx.x == 0) s[0] = 1;
    __syncthreads();
    const unsigned long long t0 = wall_clock64();
    unsigned acc = 0;
    while (wall_clock64() - t0 < ticks) acc += s[threadIdx.x & 31];
    if (acc == 0xdeadbeef) *sink = acc;
}
int main()
{
    unsigned *sink; hipMalloc(&sink, 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int lds[] = {1024, 40 * 1024, 64 * 1024, 72 * 1024, 80 * 1024};
    for (int threads : {1024, 512, 256}) for (int l : lds) {
        const int grid = 256 * 8;
        auto launch = [&]() {
            if (threads == 1024) { hipFuncSetAttribute((const void *)k_spin<1024>, hipFuncAttributeMaxDynamicSharedMemorySize, l); hipLaunchKernelGGL(k_spin<1024>, dim3(grid), dim3(1024), l, 0, 10000ull, sink); }
            else if (threads == 512) { hipFuncSetAttribute((const void *)k_spin<512>, hipFuncAttributeMaxDynamicSharedMemorySize, l); hipLaunchKernelGGL(k_spin<512>, dim3(grid), dim3(512), l, 0, 10000ull, sink); }
            else { hipFuncSetAttribute((const void *)k_spin<256>, hipFuncAttributeMaxDynamicSharedMemorySize, l); hipLaunchKernelGGL(k_spin<256>, dim3(grid), dim3(256), l, 0, 10000ull, sink); }
        };
        launch(); hipDeviceSynchronize();
        hipEventRecord(a); launch(); hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("threads=%4d lds=%6d: %.3f ms for 8 workgroups per CU of 100 us each -> %.1f resident per CU\n", threads, l, ms, 0.8 / ms);
    }
    return 0;
}
