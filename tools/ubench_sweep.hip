// Developer harness (not product code): what an in-place read-modify-write sweep of the 512 MiB bit set costs by shape —
// the fixed part of pass 2 of the alive-key pass (kta_alive_apply streams every bucket's 512 KiB region through LDS).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench_sweep.hip -o tools/ubench_sweep && tools/ubench_sweep
// Prints ms and TB/s (bytes read + bytes written) per variant.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

typedef uint32_t u4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// apply's shape: one workgroup of 1024 threads per 512 KiB region, wave v owns 32 KiB, walked in pieces of PIECE x 16 bytes
// per lane with DEPTH pieces requested ahead; NT: non-temporal loads / stores
template <int PIECE, int DEPTH, bool NTL, bool NTS, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void sweep_apply_shape(u4 *bits, uint32_t regions_per_wg)
{
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    constexpr uint32_t kWaveBytes = (512u << 10) / WAVES, kPieceBytes = PIECE * 16u * 64u, kPieces = kWaveBytes / kPieceBytes;
    for (uint32_t rr = 0; rr < regions_per_wg; rr++) {
        const uint32_t b = blockIdx.x * regions_per_wg + rr;
        u4 *mine = bits + ((size_t)b * (512u << 10) + (size_t)wave * kWaveBytes) / 16u;
        const uint32_t start = (wave + b * 5u) % kPieces;
        u4 q[DEPTH][PIECE];
        auto request = [&](uint32_t i, u4 (&r)[PIECE]) __attribute__((always_inline)) {
            const uint32_t m = ((i < kPieces ? i : 0u) + start) % kPieces;
#pragma unroll
            for (int u = 0; u < PIECE; u++) {
                const u4 *p = mine + (size_t)m * (kPieceBytes / 16u) + lane + 64u * u;
                r[u] = NTL ? __builtin_nontemporal_load(p) : *p;
            }
        };
#pragma unroll
        for (int d = 0; d < DEPTH; d++) request(d, q[d]);
        for (uint32_t i = 0; i < kPieces; i += DEPTH) {
#pragma unroll
            for (int d = 0; d < DEPTH; d++) {
                u4 t[PIECE];
#pragma unroll
                for (int u = 0; u < PIECE; u++) t[u] = q[d][u];
                request(i + d + DEPTH, q[d]);
                const uint32_t m = ((i + d) + start) % kPieces;
                if (i + d < kPieces) {
#pragma unroll
                    for (int u = 0; u < PIECE; u++) {
                        u4 v = t[u];
                        v.x ^= 1u;
                        u4 *p = mine + (size_t)m * (kPieceBytes / 16u) + lane + 64u * u;
                        if (NTS) __builtin_nontemporal_store(v, p); else *p = v;
                    }
                }
            }
        }
    }
}

// a plain grid-stride in-place stream: 256 threads, UNR x 16 bytes per lane in flight
template <int UNR, bool NTL, bool NTS>
__global__ __launch_bounds__(256) void sweep_stream(u4 *bits, uint64_t n16)
{
    const uint64_t stride = (uint64_t)gridDim.x * 256u * UNR;
    for (uint64_t i = (uint64_t)blockIdx.x * 256u * UNR + threadIdx.x; i < n16; i += stride) {
        u4 v[UNR];
#pragma unroll
        for (int u = 0; u < UNR; u++) v[u] = NTL ? __builtin_nontemporal_load(bits + i + 256u * u) : bits[i + 256u * u];
#pragma unroll
        for (int u = 0; u < UNR; u++) {
            v[u].x ^= 1u;
            if (NTS) __builtin_nontemporal_store(v[u], bits + i + 256u * u); else bits[i + 256u * u] = v[u];
        }
    }
}

// read only (what a recount costs), for reference
__global__ __launch_bounds__(256) void sweep_read(const u4 *bits, uint64_t n16, unsigned long long *out)
{
    unsigned long long c = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * 256u) {
        const u4 v = __builtin_nontemporal_load(bits + i);
        c += __popc(v.x) + __popc(v.y) + __popc(v.z) + __popc(v.w);
    }
    if (c == 0x123456789ull) *out = c;
}

template <typename F>
static float time_it(F launch)
{
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    float best = 1e9f;
    for (int rep = 0; rep < 6; rep++) {
        (void)hipEventRecord(a);
        launch();
        (void)hipEventRecord(b);
        (void)hipEventSynchronize(b);
        float ms;
        (void)hipEventElapsedTime(&ms, a, b);
        if (rep && ms < best) best = ms;
    }
    return best;
}

int main()
{
    const size_t bytes = 512ull << 20;
    u4 *bits;
    unsigned long long *out;
    CK(hipMalloc(&bits, bytes)); CK(hipMemset(bits, 0, bytes)); CK(hipMalloc(&out, 8));
    const uint64_t n16 = bytes / 16;
#define ROW(name, rw, ...) do { const float ms = time_it([&] { __VA_ARGS__; }); \
        printf("%-64s %7.3f ms  %6.2f TB/s\n", name, ms, (rw) * bytes / (ms * 1e-3) / 1e12); } while (0)
    ROW("read only, 2048 x 256, NT", 1.0, hipLaunchKernelGGL(sweep_read, dim3(2048), dim3(256), 0, 0, bits, n16, out));
    ROW("stream in place 2048 x 256, 4 x 16 B, plain", 2.0, hipLaunchKernelGGL((sweep_stream<4, false, false>), dim3(2048), dim3(256), 0, 0, bits, n16));
    ROW("stream in place 2048 x 256, 4 x 16 B, NT loads", 2.0, hipLaunchKernelGGL((sweep_stream<4, true, false>), dim3(2048), dim3(256), 0, 0, bits, n16));
    ROW("stream in place 2048 x 256, 4 x 16 B, NT loads + stores", 2.0, hipLaunchKernelGGL((sweep_stream<4, true, true>), dim3(2048), dim3(256), 0, 0, bits, n16));
    ROW("stream in place 1024 x 256, 8 x 16 B, plain", 2.0, hipLaunchKernelGGL((sweep_stream<8, false, false>), dim3(1024), dim3(256), 0, 0, bits, n16));
    ROW("stream in place 768 x 256, 4 x 16 B, plain", 2.0, hipLaunchKernelGGL((sweep_stream<4, false, false>), dim3(768), dim3(256), 0, 0, bits, n16));
    ROW("apply shape: 1024 wg x 16 waves, 2 KiB pieces, 2 ahead (today)", 2.0, hipLaunchKernelGGL((sweep_apply_shape<2, 2, false, false, 16>), dim3(1024), dim3(1024), 0, 0, bits, 1u));
    ROW("apply shape: 1024 wg x 16 waves, 2 KiB pieces, 4 ahead", 2.0, hipLaunchKernelGGL((sweep_apply_shape<2, 4, false, false, 16>), dim3(1024), dim3(1024), 0, 0, bits, 1u));
    ROW("apply shape: 1024 wg x 16 waves, 4 KiB pieces, 2 ahead", 2.0, hipLaunchKernelGGL((sweep_apply_shape<4, 2, false, false, 16>), dim3(1024), dim3(1024), 0, 0, bits, 1u));
    ROW("apply shape: 1024 wg x 16 waves, 2 KiB pieces, 2 ahead, NT loads", 2.0, hipLaunchKernelGGL((sweep_apply_shape<2, 2, true, false, 16>), dim3(1024), dim3(1024), 0, 0, bits, 1u));
    ROW("apply shape: 1024 wg x 16 waves, 2 KiB pieces, 2 ahead, NT l + s", 2.0, hipLaunchKernelGGL((sweep_apply_shape<2, 2, true, true, 16>), dim3(1024), dim3(1024), 0, 0, bits, 1u));
    ROW("apply shape: 256 wg x 16 waves x 4 regions, 2 KiB pieces, 2 ahead", 2.0, hipLaunchKernelGGL((sweep_apply_shape<2, 2, false, false, 16>), dim3(256), dim3(1024), 0, 0, bits, 4u));
    ROW("apply shape: 1024 wg x 4 waves (4 wg / CU), 2 KiB pieces, 2 ahead", 2.0, hipLaunchKernelGGL((sweep_apply_shape<2, 2, false, false, 4>), dim3(1024), dim3(256), 0, 0, bits, 1u));
    ROW("apply shape: 1024 wg x 4 waves, 4 KiB pieces, 4 ahead", 2.0, hipLaunchKernelGGL((sweep_apply_shape<4, 4, false, false, 4>), dim3(1024), dim3(256), 0, 0, bits, 1u));
    return 0;
}
