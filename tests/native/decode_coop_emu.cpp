// decode_coop_emu.cpp — test infrastructure: kafka_decode_coop<G, W, R> (csrc/kta_decode_coop.h, the kernel's own
// source) compiled for the host over tests/native/wave_emu.h and run workgroup by workgroup, 64 lanes as fibers.
// tests/test_decode_emu.py builds this into a shared object and compares its columns with the oracle, the encoder's
// expectations and kta_kafka_decode_rounds_host.  The blob is handed to the kernel as the device sees it: 16-byte
// aligned, with 64 readable bytes behind blob_len — poisoned here, and differently on request, because nothing read
// there may decide anything.
#include "wave_emu.h"

#include "../../include/kta_kafka.h"
#include "kta_records.h"

namespace {
#include "kta_decode_coop.h"
}

#include <new>

namespace {

template <int G, uint32_t W, uint32_t R>
const char *run(uint32_t grid, int order, uint32_t seed, const uint4 *blocks, const kta_kafka_batch_desc *descs,
                uint64_t n_batches, int want_keys, int32_t *part, int32_t *klen, int32_t *vlen, int64_t *ts,
                uint32_t *koff, uint64_t *seq, uint64_t seq_base, unsigned long long *n_bad, unsigned long long *n_keyb)
{
    return wave_emu::launch(grid, order, seed, [&] {
        kafka_decode_coop<G, W, R>(blocks, descs, n_batches, want_keys, part, klen, vlen, ts, koff, (uint64_t)0, seq, seq_base,
                                   n_bad, n_keyb);
    });
}

char g_error[320];

// The emulator's own checks: a hand-over through __shared__ with and without the barrier it needs, and a barrier
// under divergent control flow.
void handover_kernel(int with_barrier, uint32_t *out)
{
    __shared__ uint32_t s_word;
    const uint32_t lane = threadIdx.x;
    if (lane == 0) s_word = 0;
    __syncthreads();
    if (lane == 17) s_word = 42u + blockIdx.x;
    if (with_barrier) __syncthreads();
    out[blockIdx.x * 64 + lane] = s_word;
    unsigned long long v = lane;
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    if (!__any(lane == 63)) v = 0;
    out[blockIdx.x * 64 + lane] += (uint32_t)v << 16;          // 2016 << 16 on every lane
}

void divergent_kernel()
{
    if (threadIdx.x < 32) {
        __syncthreads();
    } else {
        __syncthreads();
    }
}

} // namespace

extern "C" {

const char *kta_emu_last_error(void) { return g_error; }

int kta_emu_selftest_handover(int order, uint32_t seed, int with_barrier, uint32_t grid, uint32_t *out)
{
    const char *err = wave_emu::launch(grid, order, seed, [&] { handover_kernel(with_barrier, out); });
    snprintf(g_error, sizeof g_error, "%s", err ? err : "");
    return err ? -2 : 0;
}

int kta_emu_selftest_divergent(void)
{
    const char *err = wave_emu::launch(1, 0, 0, [] { divergent_kernel(); });
    snprintf(g_error, sizeof g_error, "%s", err ? err : "");
    return err ? -2 : 0;
}

// lanes / window / per_round: the geometry as kta_kafka_decode_rounds_host names it (G = 64 / lanes); prefetch: must be 0 (the prefetching form of the
// kernel was deleted in round 5).  order, seed: wave_emu::launch.  poison: the byte behind the blob's end.  Returns 0, -1 for a geometry that is not instantiated
// below, -2 when the emulator reports divergent meeting points (kta_emu_last_error).
int kta_emu_decode_coop(uint32_t lanes, uint32_t window, uint32_t per_round, int prefetch, int order, uint32_t seed, uint8_t poison,
                        const uint8_t *blob, uint64_t blob_len, const kta_kafka_batch_desc *descs, uint64_t n_batches,
                        int32_t *partition, int32_t *key_len, int32_t *val_len, int64_t *ts_ms, uint32_t *key_off,
                        uint64_t *seq, uint64_t seq_base, uint64_t *n_key_bytes, uint64_t *n_bad_batches)
{
    g_error[0] = 0;
    const uint64_t padded = ((blob_len + 15) & ~15ull) + 64;
    uint8_t *buf = static_cast<uint8_t *>(aligned_alloc(16, padded));
    if (!buf) return -3;
    memset(buf, poison, padded);
    memcpy(buf, blob, blob_len);
    const uint4 *blocks = reinterpret_cast<const uint4 *>(buf);
    unsigned long long bad = 0, keyb = 0;
    const int wk = key_off ? 1 : 0;
    const uint32_t G = lanes ? 64 / lanes : 0;
    const uint32_t grid = G ? (uint32_t)((n_batches + G - 1) / G) : 0;
    const char *err = nullptr;
    int rc = 0;
#define GEOMETRY(g, w, r)                                                                                             \
    else if (G == (g) && window == (w) && per_round == (r) && !prefetch)                                              \
        err = run<g, w, r>(grid, order, seed, blocks, descs, n_batches, wk, partition, key_len, val_len, ts_ms,       \
                           key_off, seq, seq_base, &bad, n_key_bytes ? &keyb : nullptr)
    if (!G || 64 % lanes) rc = -1;
    GEOMETRY(1, 8192u, 256u);   // the dispatcher's geometries (kta_kafka.hip: KTA_DECODE_COOP)
    GEOMETRY(4, 3072u, 16u);
    GEOMETRY(2, 8192u, 32u);
    GEOMETRY(8, 1024u, 16u);    // eight leaders
    GEOMETRY(8, 256u, 8u);      // small ones: a window edge in almost every record
    GEOMETRY(16, 64u, 4u);
    else rc = -1;
#undef GEOMETRY
    free(buf);
    if (err) {
        snprintf(g_error, sizeof g_error, "%s", err);
        return -2;
    }
    if (n_key_bytes) *n_key_bytes = keyb;
    if (n_bad_batches) *n_bad_batches = bad;
    return rc;
}

} // extern "C"
