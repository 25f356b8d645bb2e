// kta_kernels.h — internal launch interface between the C-ABI layer (kta_api.hip) and the
// gfx950 kernels (kta_kernels.hip).  Not part of the public ABI.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace kta {

constexpr int kWG = 256;                    // threads per workgroup (4 wave64)
constexpr uint32_t kScanCols = 5;           // per-partition partial columns written by the scan:
                                            //   count, tombstones, key_null, key_size_sum, value_size_sum
constexpr uint32_t kScanGlobals = 8;        // per-workgroup partial globals (see ScanGlobal)
constexpr uint64_t kAliveSlots = 1ull << 32; // the reference hashes into a usize from a u32 (metric.rs:259)

enum ScanGlobal : uint32_t {
    SG_TMIN = 0, SG_TMAX = 1, SG_SMIN = 2, SG_SMAX = 3, SG_BAD = 4, SG_NREC = 5
};

struct ScanColumns {
    const int32_t *partition;
    const int32_t *key_len;
    const int32_t *val_len;
    const int64_t *ts_ms;
};

struct AliveColumns {
    const int32_t *key_len;
    const int32_t *val_len;
    const uint32_t *key_off;
    const uint8_t *key_bytes;
    const uint64_t *seq; // may be null
};

// Table state: the slots this context ever wrote, in the order they were first written — what a rank exports in the
// exchange instead of sweeping its 32 GiB table.  Every kernel that writes the table appends a slot when it finds the
// entry 0 (never written).  n counts past cap when the list overflows (the exchange then sweeps the table).
struct WrittenList {
    uint32_t *slots;            // null: not tracked
    unsigned long long *n;      // device counter
    uint64_t cap;
};

#ifdef __HIPCC__
// wave-aggregated append (works under divergence: the ballot covers the active lanes)
__device__ __forceinline__ void note_new_slot(const WrittenList &wl, bool is_new, uint32_t slot)
{
    if (!wl.slots) return;
    const unsigned long long m = __ballot(is_new);
    if (m == 0ull) return;
    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
    unsigned long long base = 0;
    if (is_new && rank == 0u) base = atomicAdd(wl.n, (unsigned long long)__popcll(m));
    base = __shfl(base, __ffsll((long long)m) - 1);
    if (is_new && base + rank < wl.cap) wl.slots[base + rank] = slot;
}
#endif

struct ScanPlan {
    uint32_t workgroups;   // grid size
    uint32_t rep_log2;     // LDS replication of each partition's slots
    uint32_t lds_bytes;    // dynamic LDS per workgroup
    uint32_t variant;      // 0 = accumulate (three 64-bit LDS atomics per record or quad), 9 = loads only (diagnostic)
    bool nontemporal;      // stream the columns with non-temporal loads
    bool analytics;        // additive outputs: size histograms + per-partition extrema
    uint32_t row_len;      // u64 words per workgroup row of the partial workspace
};

constexpr uint32_t kAnalyticsHist = 2 * 34; // key-size and value-size log2 histograms
// analytics vector: u64[2*34 + 4*P] = histograms, then per partition [~min ts, max ts, ~smallest, largest]
inline uint32_t analytics_len(uint32_t P) { return kAnalyticsHist + 4 * P; }

// u64 words one workgroup writes into the partial workspace
inline uint32_t scan_row_len(uint32_t P, bool analytics)
{
    return P * kScanCols + kScanGlobals + (analytics ? 4 * P + 2 * 34 : 0);
}

ScanPlan plan_scan(uint32_t P, uint64_t n, int cu_count, int req_workgroups, int req_variant, bool analytics);

// K1: per-record metric accumulation (metric.rs:207-252) over one struct-of-arrays batch.
hipError_t launch_metrics_scan(const ScanPlan &plan, const ScanColumns &c, uint64_t n, uint32_t P,
                               uint64_t *partials, hipStream_t s);
// K5: fold the per-workgroup partial rows into the persistent counter vector.
hipError_t launch_fold_partials(const uint64_t *partials, uint32_t rows, uint32_t P, uint64_t *vec,
                                uint32_t row_len, uint64_t *analytics_vec, hipStream_t s);
// reset the counter vector to the MessageMetrics::new state (metric.rs:30-46)
hipError_t launch_init_vector(uint64_t *vec, uint32_t P, uint64_t *analytics_vec, hipStream_t s);

// K2+K3: FNV (fnv32.rs:92-101) + last-writer-wins table update (metric.rs:289-304)
// variant 0 = fused; 1 = fused + running alive count (returning atomics, default); 8 / 9 = ablation halves
// (hash -> scratch, scratch -> table)
hipError_t launch_alive_update(const AliveColumns &c, uint64_t n, uint64_t base_seq, uint64_t *table,
                               int workgroups, int variant, uint32_t *scratch, int64_t *running, hipStream_t s,
                               const uint32_t *only_if /* variant 2: device word, run only when non-zero; may be null */,
                               const WrittenList &written);
// K2'+K3' (kta_alive.hip): the same update as two kernels — hash + partition the batch's (hash, index, alive)
// pairs by the hash's top bits into workgroup-private segments, then one workgroup per bucket merges its
// pairs in LDS and applies the survivors to the region of the persistent state it alone writes.  The state is
// either the reference's bit set (batches applied in submission order) or the u64 last-writer table (global
// sequence numbers).  A seq column must ascend inside a batch (checked on the device: alive_order_flag).
constexpr uint64_t kAlivePartitionMin = 1ull << 21;   // table state: below this the single-kernel update is used
constexpr uint64_t kAlivePartitionMax = 1ull << 28;   // records per launch pair: larger batches are sliced
struct AlivePartitionPlan {
    bool pair32;            // bit set state: 4-byte pairs with implicit order (kta_alive.hip); table state: 8-byte pairs
    uint32_t bucket_log2;   // buckets = state regions = apply workgroups
    uint32_t segment_wgs;   // partition workgroups = segments per bucket
    uint32_t tiles_per_wg;  // 256-record tiles each partition workgroup takes (a contiguous range)
    uint32_t cap;           // pairs per segment
    uint64_t max_records;   // records one launch pair takes (larger batches are sliced)
    uint64_t pair_words;    // u64 words of the pair workspace
    uint64_t count_words;   // u32 words of the segment-count workspace
    uint64_t pool_pairs;    // pairs the pool takes (whose segment was full); bit set state: in blocks of 16, their tags behind them
    uint64_t pool_words;    // u64 words of the pool's allocation
    uint64_t ctl_bytes;     // pool control words + pool histogram + order flag
};
struct AliveState {
    uint64_t *table;        // u64[2^32], or null
    uint32_t *bitmap;       // u32[2^27] = 2^32 bits, or null (exactly one of the two)
    int64_t *running;       // running alive count
    WrittenList written;    // table state: the slots ever written
};
struct AliveWorkspace {
    uint64_t *pairs;
    uint32_t *counts;
    uint64_t *pool;
    void *pool_ctl;         // ctl_bytes
    uint32_t *fail_from;    // u32[3 x buckets] (bit set state): per bucket the first segment pass 2 left to kta_alive_fallback
                            // (or none), then the list of the buckets it gave up, then the list of the buckets whose slots
                            // did not fit pass 2's table in one piece (applied in slot-range passes)
    uint64_t *failed_total; // bit set state: += the buckets a launch pair handed to kta_alive_fallback (device word; may be null)
};
AlivePartitionPlan plan_alive_partition(uint64_t n, int req_wgs, int cu_count, bool pair32);
// Both handlers in one pass (bit set state): pass 1 also reads partition and ts_ms and writes one row of the scan's partial
// workspace per partition workgroup (plan.segment_wgs rows of row_len words), to be folded by launch_fold_partials —
// MessageMetrics::handle_message (metric.rs:207-252) without a second reading of key_len and val_len.
struct AliveFuse {
    const int32_t *partition;
    const int64_t *ts_ms;
    uint32_t P;
    uint64_t *partials;
    uint32_t row_len;
};
bool alive_fuse_possible(const AlivePartitionPlan &plan, uint32_t P);   // P <= 256, and less than 2^21 records per workgroup
hipError_t launch_alive_partitioned(const AliveColumns &c, uint64_t n, uint64_t base_seq, const AliveState &st,
                                    const AlivePartitionPlan &plan, const AliveWorkspace &ws,
                                    uint64_t *stats /* [pairs, claims] += ; may be null */, hipStream_t s,
                                    const AliveFuse *fuse = nullptr /* null: the alive-key pass only */);
// device word that the launch pair sets when the batch's seq column does not ascend (the pair then did nothing)
const uint32_t *alive_order_flag(const AliveWorkspace &ws, int bucket_log2);
// popcount of the bit set -> *out += (u64)
hipError_t launch_bitmap_count(const uint32_t *bitmap, uint64_t *out, hipStream_t s);
// K4: sum_all_alive (metric.rs:282-284): count table entries whose low bit is set -> *out (u64)
hipError_t launch_alive_count(const uint64_t *table, uint64_t n_slots, uint64_t *out, hipStream_t s);
// compact (slot, value) export / import of the entries ever written: what sharded GPUs exchange
hipError_t launch_alive_count_span(const uint64_t *table, uint64_t lo, uint64_t hi, uint64_t *out, hipStream_t s);
hipError_t launch_alive_count_written(const uint64_t *table, uint64_t n_slots, uint64_t *out, hipStream_t s);
hipError_t launch_alive_export(const uint64_t *table, uint64_t n_slots, uint32_t *out_slots, uint64_t *out_vals,
                               uint64_t *counter, uint64_t cap, hipStream_t s);
hipError_t launch_alive_export_span(const uint64_t *table, uint64_t lo, uint64_t hi, uint32_t *out_slots,
                                    uint64_t *out_vals, uint64_t *counter, uint64_t cap, hipStream_t s);
hipError_t launch_alive_count_written_span(const uint64_t *table, uint64_t lo, uint64_t hi, uint64_t *out, hipStream_t s);
hipError_t launch_alive_import(const uint32_t *slots, const uint64_t *vals, uint64_t n, uint64_t *table,
                               int64_t *running, const WrittenList &written, hipStream_t s);
// the exchange over the written list: entries per owner rank (owner(slot) = (slot * R) >> 32), their export as
// one contiguous (slot, value) list per owner, and the alive count of one owner's range
// (the list's length is read on the device; *overflow = 1 when it exceeds the list's capacity)
hipError_t launch_written_count(const WrittenList &wl, int nranks, uint64_t *counts /* [nranks] += */, uint64_t *overflow, hipStream_t s);
hipError_t launch_written_export(const WrittenList &wl, const uint64_t *table, int nranks, int skip_rank,
                                 const uint64_t *owner_at /* device [nranks] */, uint64_t *cursors /* device [nranks], zero */,
                                 uint32_t *out_slots, uint64_t *out_vals, hipStream_t s);
hipError_t launch_written_alive_count(const WrittenList &wl, const uint64_t *table, uint64_t lo, uint64_t hi,
                                      uint64_t *out /* = */, hipStream_t s);
// table -> 2^32-bit bitmap (u32 words)
hipError_t launch_alive_bitmap(const uint64_t *table, uint64_t n_slots, uint32_t *bitmap, hipStream_t s);
// hash only (tests)
hipError_t launch_fnv32(const uint8_t *key_bytes, const uint32_t *key_off, const int32_t *key_len,
                        uint64_t n, uint32_t *out, hipStream_t s);

} // namespace kta
