// Developer harness (not product code): the partitioned alive-key pass of kta_alive.hip on synthetic
// 16-byte keys, with the per-phase tick counters compiled in.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I kafka_topic_analyzer_amd/csrc tools/ubench_alive.hip -o /tmp/uba
//   /tmp/uba [log2 n = 26] [distinct keys = 10000000] [state: 0 bit set, 1 table] [segment workgroups = 0] [seq column: 0 / 1]
//            [both handlers in the pass (FUSE, 256 partitions): 0 / 1]
#ifndef KTA_NO_PHASES
#define KTA_ALIVE_PHASES 1
#endif
#define KTA_UBENCH_EVENTS 1   // bit set state: events between the kernels of a pair (partition / apply / fallback)
#include "../kafka_topic_analyzer_amd/csrc/kta_alive.hip"

#include <stdio.h>
#include <stdlib.h>

__device__ __forceinline__ uint64_t mix64(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__global__ void k_fill2(int32_t *pt, int64_t *ts, uint64_t *sq, uint64_t n)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        if (pt) pt[i] = (int32_t)(mix64(i ^ 0x77) % 256);
        if (ts) ts[i] = 1600000000000ll + (int64_t)(mix64(i ^ 0x99) % 7200000);
        if (sq) sq[i] = 8 * i + mix64(i ^ 0x33) % 8;      // ascending, with gaps: what one of eight ranks sees
    }
}

__global__ void k_fill(int32_t *kl, int32_t *vl, uint32_t *ko, uint64_t *kb, uint64_t n, uint64_t distinct)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint64_t id = mix64(i) % distinct;
        kl[i] = 16;
        vl[i] = mix64(i ^ 0x5555) % 10 == 0 ? -1 : 100;
        ko[i] = (uint32_t)(16 * i);
        kb[2 * i] = mix64(id * 2 + 77);
        kb[2 * i + 1] = mix64(id * 2 + 78);
    }
}

__global__ void k_bump(uint64_t *sq, uint64_t n, uint64_t by)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) sq[i] += by;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char **argv)
{
    const int log2n = argc > 1 ? atoi(argv[1]) : 26;
    const uint64_t distinct = argc > 2 ? strtoull(argv[2], 0, 10) : 10000000ull;
    const int table_state = argc > 3 ? atoi(argv[3]) : 0;     // 0: bit set state, 1: table state
    const int wgs = argc > 4 ? atoi(argv[4]) : 0;
    const int with_seq = argc > 5 ? atoi(argv[5]) : 0;
    const int with_fuse = argc > 6 ? atoi(argv[6]) : 0;
    const uint64_t n = 1ull << log2n;
    int32_t *kl, *vl;
    uint32_t *ko, *bitmap = nullptr, *counts, *fail_from;
    uint64_t *kb, *table = nullptr, *pairs, *pool;
    void *ctl;
    int64_t *running;
    CK(hipMalloc(&kl, n * 4)); CK(hipMalloc(&vl, n * 4)); CK(hipMalloc(&ko, n * 4)); CK(hipMalloc(&kb, n * 16 + 64));
    if (table_state) { CK(hipMalloc(&table, 8ull << 32)); CK(hipMemset(table, 0, 8ull << 32)); }
    else { CK(hipMalloc(&bitmap, 1ull << 29)); CK(hipMemset(bitmap, 0, 1ull << 29)); }
    CK(hipMalloc(&running, 8)); CK(hipMemset(running, 0, 8));
    hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, 0, kl, vl, ko, kb, n, distinct);
    int32_t *pt = nullptr;
    int64_t *ts = nullptr;
    uint64_t *sq = nullptr, *partials = nullptr;
    if (with_seq) CK(hipMalloc(&sq, n * 8 + 64));
    if (with_fuse) { CK(hipMalloc(&pt, n * 4)); CK(hipMalloc(&ts, n * 8)); CK(hipMalloc(&partials, (size_t)1024 * kta::scan_row_len(256, false) * 8)); }
    hipLaunchKernelGGL(k_fill2, dim3(2048), dim3(256), 0, 0, pt, ts, sq, n);
    kta::AlivePartitionPlan pl = kta::plan_alive_partition(n, wgs, 256, !table_state);
    CK(hipMalloc(&pairs, pl.pair_words * 8)); CK(hipMalloc(&counts, pl.count_words * 4)); CK(hipMalloc(&pool, (pl.pool_words + 8) * 8));
    CK(hipMalloc(&ctl, pl.ctl_bytes)); CK(hipMalloc(&fail_from, 12u << pl.bucket_log2));
    printf("n=2^%d distinct=%llu %s state buckets=2^%u segment_wgs=%u tiles/wg=%u cap=%u workspace=%.0f MB\n", log2n,
           (unsigned long long)distinct, table_state ? "table" : "bit set", pl.bucket_log2, pl.segment_wgs, pl.tiles_per_wg, pl.cap,
           pl.pair_words * 8 / 1e6);
    kta::AliveColumns c{kl, vl, ko, reinterpret_cast<const uint8_t *>(kb), sq};
    const kta::AliveFuse fz{pt, ts, 256, partials, kta::scan_row_len(256, false)};
    printf("seq column: %s   both handlers in the pass: %s\n", with_seq ? "yes" : "no", with_fuse ? "yes" : "no");
    kta::AliveState st{table, bitmap, running};
    kta::AliveWorkspace ws{pairs, counts, pool, ctl, fail_from, nullptr};
    uint64_t *d_stats;
    CK(hipMalloc(&d_stats, 32));
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 4; i++) hipEventCreate(&kta::g_ub_ev[i]);
    // (a GPU that has idled starts at low clocks, and 4 passes do not wake it: without this pre-roll the same binary measured
    // 1.80 and 2.12 ms for the same kernel in two processes of one gpurun call)
    const int reps = getenv("UB_REPS") ? atoi(getenv("UB_REPS")) : 4;
    for (int warm = 0; warm < (getenv("UB_WARM") ? atoi(getenv("UB_WARM")) : 60); warm++) {
        CK(hipMemset(d_stats, 0, 32));
        CK(kta::launch_alive_partitioned(c, n, 0, st, pl, ws, d_stats, 0, with_fuse ? &fz : nullptr));
    }
    if (table_state) { CK(hipDeviceSynchronize()); CK(hipMemset(table, 0, 8ull << 32)); }
    else CK(hipMemset(bitmap, 0, 1ull << 29));
    CK(hipMemset(running, 0, 8));
    CK(hipDeviceSynchronize());
    for (int rep = 0; rep < reps; rep++) {
        unsigned long long zero[16] = {0};
#ifdef KTA_ALIVE_PHASES
        CK(hipMemcpyToSymbol(HIP_SYMBOL(g_kta_phase), zero, sizeof zero));
#endif
        CK(hipDeviceSynchronize());
        hipEventRecord(a);
        CK(hipMemset(d_stats, 0, 32));
        if (sq && rep) { hipLaunchKernelGGL(k_bump, dim3(2048), dim3(256), 0, 0, sq, n, 8 * n); CK(hipDeviceSynchronize()); hipEventRecord(a); }   // later sequence numbers every pass
        CK(kta::launch_alive_partitioned(c, n, (uint64_t)rep * n, st, pl, ws, d_stats, 0, with_fuse ? &fz : nullptr));
        hipEventRecord(b);
        CK(hipEventSynchronize(b));
        float ms;
        hipEventElapsedTime(&ms, a, b);
        unsigned long long ph[16] = {0}, pc[2] = {0};
#ifdef KTA_ALIVE_PHASES
        CK(hipMemcpyFromSymbol(ph, HIP_SYMBOL(g_kta_phase), sizeof ph));
#endif
        long long alive;
        CK(hipMemcpy(&alive, running, 8, hipMemcpyDeviceToHost));
        CK(hipMemcpy(pc, ctl, 16, hipMemcpyDeviceToHost));
        unsigned long long stt[4];
        CK(hipMemcpy(stt, d_stats, 32, hipMemcpyDeviceToHost));
        printf("   pairs %llu claims %llu instalments %llu list entries %llu\n", stt[0], stt[1], stt[2], stt[3]);
        const double w1 = pl.segment_wgs, w2 = (double)(1u << pl.bucket_log2);
        printf("rep %d: %.3f ms = %.1f G records/s  alive=%lld  pool pairs=%llu  buckets given to the fallback=%llu\n", rep, ms,
               n / ms / 1e6, alive, pc[0], pc[1]);
        {
            float t1 = 0, t2 = 0, t3 = 0;
            hipEventElapsedTime(&t1, kta::g_ub_ev[0], kta::g_ub_ev[1]);
            hipEventElapsedTime(&t2, kta::g_ub_ev[1], kta::g_ub_ev[2]);
            hipEventElapsedTime(&t3, kta::g_ub_ev[2], kta::g_ub_ev[3]);
            printf("kernels %d: partition %.3f  apply %.3f  %s %.3f ms\n", rep, t1, t2, table_state ? "pool" : "fallback", t3);
        }
        printf("   partition per workgroup (us, thread 0's wave): wait+hash %.1f  positions %.1f  inserts %.1f  queueing %.1f  write-out %.1f  ring wait %.1f  tail %.1f\n",
               ph[0] / w1 / 100, ph[1] / w1 / 100, ph[2] / w1 / 100, ph[3] / w1 / 100, ph[5] / w1 / 100, ph[4] / w1 / 100, ph[7] / w1 / 100);
        printf("   apply per workgroup (us):     init %.1f  loads+merge %.1f  checkpoints %.1f  rest %.1f  end %.1f   (x %.0f workgroups / 256 CUs)\n",
               ph[8] / w2 / 100, ph[11] / w2 / 100, ph[12] / w2 / 100, ph[9] / w2 / 100, ph[10] / w2 / 100, w2);
    }
    return 0;
}
