// kta_synth.hip — the synthetic topic source (include/kta_synth.h): host generator, device
// generator (HBM-resident batches for benchmarks) and the BASELINE.json config presets.
// Input generation only: nothing here is on the measured hot path.
#include "../../include/kta_synth.h"

#include <hip/hip_runtime.h>
#include <string.h>

#include <string>

hipStream_t kta_internal_stream(kta_ctx *ctx);
int kta_internal_device(kta_ctx *ctx);
void kta_internal_set_error(kta_ctx *ctx, const char *msg);

namespace {

constexpr int kWG = 256;
constexpr int kScanItems = 16;                 // records per thread in the offset scan
constexpr int kScanChunk = kWG * kScanItems;   // records per workgroup

__global__ __launch_bounds__(kWG) void synth_fill_cols(kta_synth_spec sp, uint64_t first, uint64_t n,
                                                       int32_t *part, int32_t *klen, int32_t *vlen,
                                                       int64_t *ts, uint64_t *seq)
{
    const uint64_t stride = (uint64_t)gridDim.x * kWG;
    for (uint64_t i = (uint64_t)blockIdx.x * kWG + threadIdx.x; i < n; i += stride) {
        int32_t p, kl, vl;
        int64_t t;
        kta_synth_record(&sp, first + i, &p, &kl, &vl, &t);
        part[i] = p;
        klen[i] = kl;
        vlen[i] = vl;
        ts[i] = t;
        if (seq) seq[i] = first + i;
    }
}

__device__ __forceinline__ uint32_t klen_bytes(const int32_t *klen, uint64_t i, uint64_t n)
{
    if (i >= n) return 0u;
    const int32_t k = klen[i];
    return k > 0 ? (uint32_t)k : 0u;
}

// phase A: per-chunk byte totals
__global__ __launch_bounds__(kWG) void synth_chunk_sums(const int32_t *klen, uint64_t n, uint64_t *chunk_sum)
{
    __shared__ uint64_t s[kWG];
    const uint64_t base = (uint64_t)blockIdx.x * kScanChunk + (uint64_t)threadIdx.x * kScanItems;
    uint64_t t = 0;
    for (int j = 0; j < kScanItems; j++) t += klen_bytes(klen, base + j, n);
    s[threadIdx.x] = t;
    __syncthreads();
    for (int off = kWG / 2; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) s[threadIdx.x] += s[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) chunk_sum[blockIdx.x] = s[0];
}

// phase B: exclusive scan of the chunk totals (one workgroup; each thread owns a contiguous span)
__global__ __launch_bounds__(1024) void synth_scan_chunks(uint64_t *chunk_sum, uint64_t n_chunks, uint64_t *total)
{
    __shared__ uint64_t s[1024];
    const uint64_t per = (n_chunks + 1023) / 1024;
    const uint64_t b = (uint64_t)threadIdx.x * per;
    const uint64_t e = b + per < n_chunks ? b + per : n_chunks;
    uint64_t t = 0;
    for (uint64_t i = b; i < e; i++) t += chunk_sum[i];
    s[threadIdx.x] = t;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint64_t run = 0;
        for (int i = 0; i < 1024; i++) {
            const uint64_t v = s[i];
            s[i] = run;
            run += v;
        }
        *total = run;
    }
    __syncthreads();
    uint64_t run = s[threadIdx.x];
    for (uint64_t i = b; i < e; i++) {
        const uint64_t v = chunk_sum[i];
        chunk_sum[i] = run;
        run += v;
    }
}

// phase C: batch-local key offsets (packed in record order)
__global__ __launch_bounds__(kWG) void synth_key_offsets(const int32_t *klen, uint64_t n, const uint64_t *chunk_off,
                                                         uint32_t *key_off)
{
    __shared__ uint64_t s[kWG];
    const uint64_t base = (uint64_t)blockIdx.x * kScanChunk + (uint64_t)threadIdx.x * kScanItems;
    uint32_t len[kScanItems];
    uint64_t t = 0;
    for (int j = 0; j < kScanItems; j++) {
        len[j] = klen_bytes(klen, base + j, n);
        t += len[j];
    }
    s[threadIdx.x] = t;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint64_t run = chunk_off[blockIdx.x];
        for (int i = 0; i < kWG; i++) {
            const uint64_t v = s[i];
            s[i] = run;
            run += v;
        }
    }
    __syncthreads();
    uint64_t run = s[threadIdx.x];
    for (int j = 0; j < kScanItems; j++) {
        if (base + j < n) key_off[base + j] = (uint32_t)run;
        run += len[j];
    }
}

__global__ __launch_bounds__(kWG) void synth_fill_keys(kta_synth_spec sp, uint64_t first, uint64_t n,
                                                       const int32_t *klen, const uint32_t *key_off,
                                                       uint8_t *key_bytes)
{
    const uint64_t stride = (uint64_t)gridDim.x * kWG;
    for (uint64_t i = (uint64_t)blockIdx.x * kWG + threadIdx.x; i < n; i += stride) {
        const int32_t kl = klen[i];
        if (kl <= 0) continue;
        const uint64_t kid = (uint64_t)kta_synth_key_id(&sp, first + i);
        uint8_t *dst = key_bytes + key_off[i];
        for (uint32_t w = 0; w * 8u < (uint32_t)kl; w++) {
            uint64_t word = kta_synth_key_word(&sp, kid, w);
            const uint32_t m = (uint32_t)kl - w * 8u < 8u ? (uint32_t)kl - w * 8u : 8u;
            for (uint32_t j = 0; j < m; j++) {
                dst[w * 8u + j] = (uint8_t)word;
                word >>= 8;
            }
        }
    }
}

int check_spec(const kta_synth_spec *sp)
{
    if (!sp) return KTA_ERR_INVALID;
    if (sp->n_partitions == 0 || sp->shard_count == 0 || sp->shard_index >= sp->shard_count) return KTA_ERR_INVALID;
    if (kta_synth_local_partitions(sp) == 0) return KTA_ERR_INVALID;
    if (sp->n_key_lens == 0 || sp->n_key_lens > KTA_SYNTH_MAX_KEY_LENS) return KTA_ERR_INVALID;
    if (sp->key_null_permille > 1000 || sp->tombstone_permille + sp->val_empty_permille > 1000) return KTA_ERR_INVALID;
    return KTA_OK;
}

} // namespace

extern "C" {

int kta_synth_fill_host(const kta_synth_spec *spec, uint64_t first, uint64_t n, const kta_batch *b,
                        uint64_t *n_key_bytes)
{
    int rc = check_spec(spec);
    if (rc != KTA_OK || !b) return KTA_ERR_INVALID;
    if (n > b->capacity) return KTA_ERR_CAPACITY;
    uint64_t kb = 0;
    for (uint64_t i = 0; i < n; i++) {
        int32_t p, kl, vl;
        int64_t t;
        kta_synth_record(spec, first + i, &p, &kl, &vl, &t);
        b->partition[i] = p;
        b->key_len[i] = kl;
        b->val_len[i] = vl;
        b->ts_ms[i] = t;
        if (b->seq) b->seq[i] = first + i;
        if (b->key_off) {
            if (kb >= (1ull << 32)) return KTA_ERR_CAPACITY;
            b->key_off[i] = (uint32_t)kb;
        }
        if (kl > 0) {
            if (b->key_bytes) {
                if (kb + (uint64_t)kl > b->key_bytes_capacity) return KTA_ERR_CAPACITY;
                const uint64_t kid = (uint64_t)kta_synth_key_id(spec, first + i);
                for (uint32_t j = 0; j < (uint32_t)kl; j++) b->key_bytes[kb + j] = kta_synth_key_byte(spec, kid, j);
            }
            kb += (uint64_t)kl;
        }
    }
    if (n_key_bytes) *n_key_bytes = kb;
    return KTA_OK;
}

int kta_synth_fill_device(kta_ctx *ctx, const kta_synth_spec *spec, uint64_t first, uint64_t n,
                          const kta_batch *b, uint64_t *n_key_bytes)
{
    if (!ctx || !b) return KTA_ERR_INVALID;
    if (check_spec(spec) != KTA_OK) {
        kta_internal_set_error(ctx, "invalid synthetic spec");
        return KTA_ERR_INVALID;
    }
    if (n > b->capacity) {
        kta_internal_set_error(ctx, "synthetic batch larger than the device batch capacity");
        return KTA_ERR_CAPACITY;
    }
    if (n_key_bytes) *n_key_bytes = 0;
    if (n == 0) return KTA_OK;
    hipStream_t s = kta_internal_stream(ctx);
    hipError_t e = hipSetDevice(kta_internal_device(ctx));
    uint64_t *d_chunks = nullptr;
    std::string what;
#define SY(call)                                   \
    do {                                           \
        if (e == hipSuccess) {                     \
            e = (call);                            \
            if (e != hipSuccess) what = #call;     \
        }                                          \
    } while (0)
    const uint32_t grid = (uint32_t)((n + kWG - 1) / kWG < 8192 ? (n + kWG - 1) / kWG : 8192);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(synth_fill_cols, dim3(grid), dim3(kWG), 0, s, *spec, first, n, b->partition, b->key_len,
                           b->val_len, b->ts_ms, b->seq);
        SY(hipGetLastError());
    }
    uint64_t total = 0;
    if (b->key_off && b->key_bytes) {
        const uint64_t n_chunks = (n + kScanChunk - 1) / kScanChunk;
        SY(hipMalloc((void **)&d_chunks, (n_chunks + 1) * sizeof(uint64_t)));
        if (e == hipSuccess) {
            hipLaunchKernelGGL(synth_chunk_sums, dim3((uint32_t)n_chunks), dim3(kWG), 0, s, b->key_len, n, d_chunks);
            hipLaunchKernelGGL(synth_scan_chunks, dim3(1), dim3(1024), 0, s, d_chunks, n_chunks, d_chunks + n_chunks);
            SY(hipGetLastError());
        }
        SY(hipMemcpyAsync(&total, d_chunks + n_chunks, sizeof(uint64_t), hipMemcpyDeviceToHost, s));
        SY(hipStreamSynchronize(s));
        if (e == hipSuccess && (total > b->key_bytes_capacity || total >= (1ull << 32))) {
            (void)hipFree(d_chunks);
            kta_internal_set_error(ctx, "synthetic key bytes exceed the device batch key capacity");
            if (n_key_bytes) *n_key_bytes = total;
            return KTA_ERR_CAPACITY;
        }
        if (e == hipSuccess) {
            hipLaunchKernelGGL(synth_key_offsets, dim3((uint32_t)n_chunks), dim3(kWG), 0, s, b->key_len, n, d_chunks,
                               b->key_off);
            hipLaunchKernelGGL(synth_fill_keys, dim3(grid), dim3(kWG), 0, s, *spec, first, n, b->key_len, b->key_off,
                               b->key_bytes);
            SY(hipGetLastError());
        }
        SY(hipStreamSynchronize(s));
        if (d_chunks) (void)hipFree(d_chunks);
    }
#undef SY
    if (e != hipSuccess) {
        std::string m = what + ": " + hipGetErrorString(e);
        kta_internal_set_error(ctx, m.c_str());
        return KTA_ERR_HIP;
    }
    if (n_key_bytes) *n_key_bytes = total;
    return KTA_OK;
}

// The five BASELINE.json configs made concrete.  Mean record =
// key (~61 B incl. nulls) + value (~195 B incl. tombstones) ~ 256 B for the mixed law.
int kta_synth_preset(const char *name, kta_synth_spec *sp, uint64_t *n_records)
{
    if (!name || !sp) return KTA_ERR_INVALID;
    memset(sp, 0, sizeof(*sp));
    sp->shard_index = 0;
    sp->shard_count = 1;
    sp->ts_base_ms = 1600000000000ll;
    uint64_t n = 0;
    auto mixed = [&]() {
        sp->key_null_permille = 50;
        sp->key_empty_permille = 10;
        sp->n_key_lens = 5;
        const uint32_t lens[5] = {8, 16, 36, 64, 200};
        for (int i = 0; i < 5; i++) sp->key_lens[i] = lens[i];
        sp->tombstone_permille = 50;
        sp->val_empty_permille = 10;
        sp->val_mode = KTA_VAL_EXP;
        sp->val_mean = 208;
        sp->val_cap = 65536;
        sp->ts_missing_permille = 1;
        sp->ts_step_us = 10;
        sp->ts_jitter_ms = 3600000;
        sp->part_mode = KTA_PART_RANDOM;
    };
    auto keyed16 = [&](uint64_t distinct, uint32_t tomb_permille) {
        sp->key_null_permille = 0;
        sp->key_empty_permille = 0;
        sp->n_key_lens = 1;
        sp->key_lens[0] = 16;
        sp->n_distinct_keys = distinct;
        sp->tombstone_permille = tomb_permille;
        sp->val_mode = KTA_VAL_FIXED;
        sp->val_mean = 240;
        sp->ts_step_us = 10;
        sp->ts_jitter_ms = 1000;
        sp->part_mode = KTA_PART_KEY_AFFINE;
    };
    if (!strcmp(name, "c1")) { // 1 partition, 1M records, 64 B keys, 256 B values, no nulls
        sp->seed = 1; sp->n_partitions = 1; n = 1000000ull;
        sp->n_key_lens = 1; sp->key_lens[0] = 64; sp->n_distinct_keys = 1000000ull;
        sp->val_mode = KTA_VAL_FIXED; sp->val_mean = 256; sp->ts_step_us = 1000;
        sp->part_mode = KTA_PART_RANDOM;
    } else if (!strcmp(name, "c2")) { // 8 partitions, 100M records, mixed sizes
        sp->seed = 2; sp->n_partitions = 8; n = 100000000ull; mixed();
        sp->n_distinct_keys = 50000000ull;
    } else if (!strcmp(name, "c3")) { // 64 partitions, 1B records, -c, 10M distinct 16 B keys
        sp->seed = 3; sp->n_partitions = 64; n = 1000000000ull; keyed16(10000000ull, 100);
    } else if (!strcmp(name, "c4")) { // 256 partitions, 1B records, mixed sizes, 8-way sharded
        sp->seed = 4; sp->n_partitions = 256; n = 1000000000ull; mixed();
        sp->n_distinct_keys = 100000000ull;
    } else if (!strcmp(name, "c5")) { // 256 partitions, 10B records, -c, 100M keys, 50% tombstones
        sp->seed = 5; sp->n_partitions = 256; n = 10000000000ull; keyed16(100000000ull, 500);
    } else {
        return KTA_ERR_INVALID;
    }
    if (n_records) *n_records = n;
    return KTA_OK;
}

} // extern "C"
