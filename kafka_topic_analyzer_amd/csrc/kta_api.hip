// kta_api.hip — the extern "C" boundary of libkta_hip.so (include/kta_hip.h): context,
// pinned struct-of-arrays staging ring, H2D/compute streams, result decode.
// Host code only; the kernels are in kta_kernels.hip.  There is no CPU fallback: every
// entry point that computes anything needs a gfx950 device.
#include "../../include/kta_hip.h"
#include "kta_kernels.h"

#include <limits.h>
#include <stdio.h>
#include <string.h>

#include <new>
#include <cstdlib>
#include <string>
#include <time.h>
#include <vector>

namespace {

thread_local std::string g_create_error = "";

// One staging batch = ONE pinned slab and ONE device slab with the same layout: the columns at 16-byte
// aligned offsets in the order [partition | key_len | val_len | ts_ms | key_off | seq | key_bytes], sized for
// the batch capacity, the variable part (key bytes) last.  A submit is then a single H2D copy of the slab's
// used prefix (a copy per column — up to seven DMA launches per batch — was 19 % of the GPU time of a
// host-fed run, profiles/r01).
struct Stage {
    kta_batch host{};  // column pointers into host_slab (pinned)
    kta_batch dev{};   // column pointers into dev_slab
    uint8_t *host_slab = nullptr, *dev_slab = nullptr;
    size_t metric_bytes = 0;      // [partition .. ts_ms]
    size_t key_bytes_off = 0;     // where key_bytes starts (0 without -c)
    size_t slab_bytes = 0;
    hipEvent_t done = nullptr;
    bool busy = false;
};

} // namespace

struct kta_ctx {
    int device = 0;
    uint32_t P = 0;
    bool alive = false;
    int cu_count = 256;
    hipStream_t s_compute = nullptr, s_copy = nullptr;
    hipStream_t s_own = nullptr;    // the context's own compute stream (s_compute may be caller-owned)
    hipEvent_t ev_copied = nullptr;
    bool analytics = false;
    bool stage_seq = false;         // KTA_FLAG_SEQ_COLUMN
    uint64_t *d_avec = nullptr;     // analytics vector u64[2*34 + 4*P] (KTA_FLAG_ANALYTICS)
    uint64_t *d_vec = nullptr;      // u64[P*7 + KTA_NGLOBALS]: the live accumulator
    uint64_t *d_vec_out = nullptr;  // its snapshot (kta_finish_device): what kta_result_vector hands out and the
                                    // exchange reduces in place — the accumulator itself is never reduced
    uint64_t *d_partials = nullptr; // scan workspace: max_rows x row_len
    uint32_t max_rows = 0;
    // -c: the persistent state is ONE of
    //   d_bitmap  the reference's bit set, u32[2^27] (default: batches are applied in submission order), or
    //   d_table   u64[2^32] last-writer table (KTA_FLAG_ALIVE_TABLE / KTA_FLAG_SEQ_COLUMN: global sequence numbers)
    bool alive_table = false;
    uint32_t *d_bitmap = nullptr;
    uint64_t *d_table = nullptr;
    int64_t *d_alive_running = nullptr; // running alive count
    bool running_valid = true;          // false once an update ran without counting
    // table state: the slots ever written (what the exchange exports); invalid once somebody else wrote the table
    uint32_t *d_written = nullptr;
    unsigned long long *d_written_n = nullptr;
    uint64_t written_cap = 0;
    bool written_valid = true;
    uint32_t *d_exp_slots = nullptr;    // kta_alive_export_entries buffers
    uint64_t *d_exp_vals = nullptr, *d_exp_count = nullptr;
    uint64_t exp_cap = 0;
    uint32_t *d_hash_scratch = nullptr; // ablation variants only
    uint64_t hash_scratch_cap = 0;
    uint64_t *d_pairs = nullptr;        // partitioned alive pass: (hash, local seq, alive) pairs by [workgroup][bucket]
    uint32_t *d_pair_counts = nullptr;  //   and the fill of every segment, [bucket][workgroup]
    uint64_t *d_pool = nullptr;         //   pairs whose segment was full
    void *d_pool_ctl = nullptr;         //   pool cursor, histogram, order flag
    uint32_t *d_fail_from = nullptr;    //   buckets handed to the fallback kernel (bit set state)
    uint64_t pairs_cap = 0, pair_counts_cap = 0, pool_cap = 0;
    // Feedback for the automatic choice (alive_variant 3 / 4): the partitioned pass pays when records die in
    // LDS (a compacted topic repeats its keys inside a batch); a batch of mostly unique keys is cheaper in
    // the single-kernel update.  Every partitioned batch reports [pairs, entries claimed]; a batch that
    // claimed more than kAliveUniqueNum / kAliveUniqueDen of its pairs sends the next kAliveBackoff batches
    // down the single-kernel path before the partitioned pass is tried again.
    uint64_t *d_alive_stats = nullptr, *h_alive_stats = nullptr;
    hipEvent_t ev_alive_stats = nullptr;
    bool alive_stats_pending = false;
    bool fuse_handlers = true;      // both handlers of a batch in one pass where that is possible (KTA_NO_FUSE=1: never)
    uint64_t *d_failed_total = nullptr; // buckets handed to kta_alive_fallback since create / reset (kta_alive_pass_info)
    int alive_backoff = 0;
    // what the partitioned pass did since kta_create / kta_reset (kta_alive_pass_info): launch pairs, of them with both
    // handlers in the one pass, of them with the metrics handler through the scan although the batch began fused, and the
    // buckets that the sampled launches (the last one of every batch of 2^24 records and more, bit set state) handed to
    // kta_alive_fallback; the word comes back with the stream and is never waited for
    uint64_t info_slices = 0, info_fused = 0, info_scanned = 0, info_failed_buckets = 0;
    std::vector<Stage> stages;
    uint64_t batch_capacity = 0, key_bytes_capacity = 0;
    int cur = 0;
    bool acquired = false;
    uint64_t fill_n = 0, fill_kb = 0; // kta_handle_message fill state
    uint64_t msg_count = 0, msg_flushes = 0, msg_flush_ns = 0, msg_wait_ns = 0;   // kta_handle_message_stats
    uint64_t next_seq = 0;
    // tuning / profiling
    int scan_wgs = 0, scan_variant = 16, alive_wgs = 0, alive_variant = 3; // 16: non-temporal loads; 3: partitioned pass for large batches
    bool timing = false;
    // HIP-event pairs recorded around each kernel on the compute stream (no host sync while
    // recording); drained by kta_kernel_time_stats.  kind: 0 scan, 1 fold, 2 alive update.
    std::vector<hipEvent_t> ev_pool[3];
    size_t ev_used[3] = {0, 0, 0};
    double ms_sum[3] = {0, 0, 0};
    uint64_t ms_cnt[3] = {0, 0, 0};
    // extension state owned by another translation unit of the library (kta_kafka.hip)
    void *ext_state = nullptr;
    void (*ext_free)(void *) = nullptr;
    void *comm_state = nullptr;     // kta_comm.hip
    void (*comm_free)(void *) = nullptr;
    std::string err;
};

namespace {

uint64_t now_ns()
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (uint64_t)ts.tv_sec * 1000000000ull + (uint64_t)ts.tv_nsec;
}

int fail(kta_ctx *ctx, int code, const std::string &msg)
{
    if (ctx) ctx->err = msg; else g_create_error = msg;
    return code;
}

int hip_fail(kta_ctx *ctx, hipError_t e, const char *what)
{
    std::string m = std::string(what) + ": " + hipGetErrorString(e);
    return fail(ctx, e == hipErrorOutOfMemory ? KTA_ERR_NOMEM : KTA_ERR_HIP, m);
}

#define KTA_HIP(ctx, call)                                         \
    do {                                                           \
        hipError_t e__ = (call);                                   \
        if (e__ != hipSuccess) return hip_fail(ctx, e__, #call);   \
    } while (0)

size_t pad16(size_t b) { return (b + 15) & ~(size_t)15; }

int alloc_device_batch(kta_ctx *ctx, uint64_t cap, uint64_t kcap, bool keys, bool seq, kta_batch *b)
{
    memset(b, 0, sizeof(*b));
    b->capacity = cap;
    b->key_bytes_capacity = keys ? kcap : 0;
    const size_t c4 = pad16(cap * 4 + 16), c8 = pad16(cap * 8 + 16);
    KTA_HIP(ctx, hipMalloc((void **)&b->partition, c4));
    KTA_HIP(ctx, hipMalloc((void **)&b->key_len, c4));
    KTA_HIP(ctx, hipMalloc((void **)&b->val_len, c4));
    KTA_HIP(ctx, hipMalloc((void **)&b->ts_ms, c8));
    if (keys) {
        KTA_HIP(ctx, hipMalloc((void **)&b->key_off, c4));
        KTA_HIP(ctx, hipMalloc((void **)&b->key_bytes, pad16(kcap + 16)));
    }
    if (seq) KTA_HIP(ctx, hipMalloc((void **)&b->seq, c8));
    return KTA_OK;
}

void free_device_batch(kta_batch *b)
{
    if (b->partition) (void)hipFree(b->partition);
    if (b->key_len) (void)hipFree(b->key_len);
    if (b->val_len) (void)hipFree(b->val_len);
    if (b->ts_ms) (void)hipFree(b->ts_ms);
    if (b->key_off) (void)hipFree(b->key_off);
    if (b->key_bytes) (void)hipFree(b->key_bytes);
    if (b->seq) (void)hipFree(b->seq);
    memset(b, 0, sizeof(*b));
}

bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

constexpr size_t kMaxTimedPairs = 2048;
constexpr uint64_t kAliveUniqueNum = 2, kAliveUniqueDen = 5;   // > 40 % of a batch's pairs claimed an entry: mostly unique keys
constexpr int kAliveBackoff = 7;

int drain_timers(kta_ctx *ctx)
{
    KTA_HIP(ctx, hipStreamSynchronize(ctx->s_compute));
    for (int k = 0; k < 3; k++) {
        for (size_t i = 0; i + 1 < ctx->ev_used[k]; i += 2) {
            float ms = 0.f;
            KTA_HIP(ctx, hipEventElapsedTime(&ms, ctx->ev_pool[k][i], ctx->ev_pool[k][i + 1]));
            ctx->ms_sum[k] += ms;
            ctx->ms_cnt[k] += 1;
        }
        ctx->ev_used[k] = 0;
    }
    return KTA_OK;
}

// next (start, stop) event pair of kernel kind k
int timer_pair(kta_ctx *ctx, int k, hipEvent_t *a, hipEvent_t *b)
{
    if (ctx->ev_used[k] + 2 > 2 * kMaxTimedPairs) {
        int rc = drain_timers(ctx);
        if (rc != KTA_OK) return rc;
    }
    while (ctx->ev_pool[k].size() < ctx->ev_used[k] + 2) {
        hipEvent_t e;
        KTA_HIP(ctx, hipEventCreate(&e));
        ctx->ev_pool[k].push_back(e);
    }
    *a = ctx->ev_pool[k][ctx->ev_used[k]];
    *b = ctx->ev_pool[k][ctx->ev_used[k] + 1];
    ctx->ev_used[k] += 2;
    return KTA_OK;
}

kta::WrittenList written_list(kta_ctx *ctx) { return kta::WrittenList{ctx->d_written, ctx->d_written_n, ctx->written_cap}; }

// Launch the handlers over device-resident columns on the compute stream.
int run_device_batch(kta_ctx *ctx, const kta_batch *c, uint64_t n, uint64_t base_seq, int which)
{
    if (n == 0) return KTA_OK;
    hipEvent_t a = nullptr, b = nullptr;
    // every refusal comes before the first launch: a batch is counted by both handlers or by neither
    if ((which & 2) && ctx->alive && (!c->key_len || !c->val_len || !c->key_off || !c->key_bytes))
        return fail(ctx, KTA_ERR_INVALID, "key columns missing (count_alive_keys)");
    // Table state: which kernels take the batch is decided before anything is launched (the fused pass below depends on it).
    // 3 = the partitioned pass for batches of >= 2^21 records (13: for batches of any size — tests), with the automatic
    // fall-back to the single-kernel filtered update (2) for batches of mostly unique keys; 1 / 2 / 8 / 9 = the
    // single-kernel variants.  Bit set state: every batch takes the partitioned pass.
    const bool part_kind = ctx->alive_variant == 3 || ctx->alive_variant == 13 || ctx->alive_variant == 4 || ctx->alive_variant == 14;
    const bool partitioned = !ctx->alive_table || (part_kind && (n >= kta::kAlivePartitionMin || ctx->alive_variant >= 13));
    bool use_partitioned = partitioned;
    if ((which & 2) && ctx->alive && partitioned && ctx->alive_table && ctx->alive_variant < 13) {   // automatic choice only (13 forces it)
        if (ctx->alive_stats_pending && hipEventQuery(ctx->ev_alive_stats) == hipSuccess) {
            ctx->alive_stats_pending = false;
            const uint64_t pairs = ctx->h_alive_stats[0], claims = ctx->h_alive_stats[1];
            if (pairs && claims * kAliveUniqueDen > pairs * kAliveUniqueNum) ctx->alive_backoff = kAliveBackoff;
        }
        if (ctx->alive_backoff > 0) {
            ctx->alive_backoff--;
            use_partitioned = false;
        }
    }
    // Both handlers over one batch (what kafka.rs:107-109 does with every message): with at most 256 partitions, pass 1 of
    // the partitioned alive-key pass does the metrics handler's work as well (kta_alive.hip, FuseArgs) — the batch is read
    // once: 40 B + key per record instead of 20 + 28 in the bit set state, 48 B + key instead of 20 + 36 for a sharded
    // rank's batch with its seq column (table state).
    bool fuse = false;
    if (which == 3 && ctx->alive && use_partitioned && !ctx->analytics && ctx->fuse_handlers) {
        const uint64_t first = n > kta::kAlivePartitionMax ? kta::kAlivePartitionMax : n;
        const kta::AlivePartitionPlan pl0 = kta::plan_alive_partition(first, ctx->alive_wgs, ctx->cu_count, !ctx->alive_table);
        fuse = kta::alive_fuse_possible(pl0, ctx->P) && pl0.segment_wgs <= ctx->max_rows;
    }
    if (which & 1) {
        if (!c->partition || !c->key_len || !c->val_len || !c->ts_ms)
            return fail(ctx, KTA_ERR_INVALID, "metric columns missing");
        if (!aligned16(c->partition) || !aligned16(c->key_len) || !aligned16(c->val_len) ||
            !aligned16(c->ts_ms))
            return fail(ctx, KTA_ERR_INVALID, "device columns must be 16-byte aligned");
    }
    if ((which & 1) && !fuse) {
        kta::ScanColumns sc{c->partition, c->key_len, c->val_len, c->ts_ms};
        kta::ScanPlan pl = kta::plan_scan(ctx->P, n, ctx->cu_count, ctx->scan_wgs, ctx->scan_variant,
                                          ctx->analytics);
        if (pl.workgroups > ctx->max_rows) pl.workgroups = ctx->max_rows;
        if (ctx->timing) {
            int rc = timer_pair(ctx, 0, &a, &b);
            if (rc != KTA_OK) return rc;
            KTA_HIP(ctx, hipEventRecord(a, ctx->s_compute));
        }
        KTA_HIP(ctx, kta::launch_metrics_scan(pl, sc, n, ctx->P, ctx->d_partials, ctx->s_compute));
        if (ctx->timing) {
            KTA_HIP(ctx, hipEventRecord(b, ctx->s_compute));
            int rc = timer_pair(ctx, 1, &a, &b);
            if (rc != KTA_OK) return rc;
            KTA_HIP(ctx, hipEventRecord(a, ctx->s_compute));
        }
        KTA_HIP(ctx, kta::launch_fold_partials(ctx->d_partials, pl.workgroups, ctx->P, ctx->d_vec, pl.row_len,
                                               ctx->d_avec, ctx->s_compute));
        if (ctx->timing) KTA_HIP(ctx, hipEventRecord(b, ctx->s_compute));
    }
    if ((which & 2) && ctx->alive) {
        kta::AliveColumns ac{c->key_len, c->val_len, c->key_off, c->key_bytes, c->seq};
        if (ctx->timing) {
            int rc = timer_pair(ctx, 2, &a, &b);
            if (rc != KTA_OK) return rc;
            KTA_HIP(ctx, hipEventRecord(a, ctx->s_compute));
        }
        if (ctx->alive_table && (ctx->alive_variant == 8 || ctx->alive_variant == 9) && ctx->hash_scratch_cap < n) {
            KTA_HIP(ctx, hipStreamSynchronize(ctx->s_compute));
            if (ctx->d_hash_scratch) (void)hipFree(ctx->d_hash_scratch);
            ctx->d_hash_scratch = nullptr;
            KTA_HIP(ctx, hipMalloc((void **)&ctx->d_hash_scratch, n * sizeof(uint32_t)));
            ctx->hash_scratch_cap = n;
        }
        // Bit set state: every batch takes the partitioned pass (hash + partition, per-bucket merge in LDS, the
        // bucket's bitmap region streamed through LDS); batches are applied in submission order, base_seq and a
        // seq column are not looked at.
        // Table state: the choice made above.  A seq column has to ascend inside a batch for the partitioned pass (the
        // merge orders a batch's records by their index): checked on the device, and a batch that fails the check runs
        // the filtered update instead, conditionally, without a host round trip.  (Both partition kernels read their
        // columns 4 bytes per lane: any alignment will do.)
        if (ctx->alive_table && ctx->alive_variant != 1 && ctx->alive_variant != 2 && !part_kind) ctx->running_valid = false;   // non-counting kernels
        if (use_partitioned) {
            if (!ctx->d_alive_stats) {   // (four words: builds with KTA_ALIVE_PHASES add instalments and side-table entries)
                KTA_HIP(ctx, hipMalloc((void **)&ctx->d_alive_stats, 4 * sizeof(uint64_t)));
                KTA_HIP(ctx, hipHostMalloc((void **)&ctx->h_alive_stats, 4 * sizeof(uint64_t), hipHostMallocDefault));
                KTA_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_alive_stats, hipEventDisableTiming));
            }
            const bool report = ctx->alive_table && !ctx->alive_stats_pending;       // one report in flight at a time
            // Bit set state: a bucket with more distinct slots than pass 2's LDS table takes is applied in instalments (careful
            // mode, groups of segments sized to fit the table: kta_alive.hip); what defeats that too — hot keys that overflow
            // their segments into the pool, one group's segments all in one set — goes to kta_alive_fallback, exact and slow.
            // The number of such buckets is sampled for kta_alive_pass_info.  (Round 4 applied the batches that followed one
            // with such buckets in slices of 2^26 records; with the groups sized from the fills a whole batch of config 5's
            // law takes 8.2 ms where four slices took 10.1, and the slicing went.)
            if (!ctx->d_failed_total) {
                KTA_HIP(ctx, hipMalloc((void **)&ctx->d_failed_total, sizeof(uint64_t)));
                KTA_HIP(ctx, hipMemsetAsync(ctx->d_failed_total, 0, sizeof(uint64_t), ctx->s_compute));
            }
            if (report) KTA_HIP(ctx, hipMemsetAsync(ctx->d_alive_stats, 0, 4 * sizeof(uint64_t), ctx->s_compute));
            for (uint64_t at = 0; at < n;) {
                const uint64_t left = n - at;
                kta::AlivePartitionPlan pl = kta::plan_alive_partition(left, ctx->alive_wgs, ctx->cu_count, !ctx->alive_table);
                const uint64_t take = left < pl.max_records ? left : pl.max_records;
                if (ctx->pairs_cap < pl.pair_words || ctx->pair_counts_cap < pl.count_words || ctx->pool_cap < pl.pool_words) {
                    KTA_HIP(ctx, hipStreamSynchronize(ctx->s_compute));
                    if (ctx->d_pairs) (void)hipFree(ctx->d_pairs);
                    if (ctx->d_pair_counts) (void)hipFree(ctx->d_pair_counts);
                    if (ctx->d_pool) (void)hipFree(ctx->d_pool);
                    ctx->d_pairs = nullptr;
                    ctx->d_pair_counts = nullptr;
                    ctx->d_pool = nullptr;
                    ctx->pairs_cap = ctx->pair_counts_cap = ctx->pool_cap = 0;
                    KTA_HIP(ctx, hipMalloc((void **)&ctx->d_pairs, pl.pair_words * sizeof(uint64_t)));
                    KTA_HIP(ctx, hipMalloc((void **)&ctx->d_pair_counts, pl.count_words * sizeof(uint32_t)));
                    KTA_HIP(ctx, hipMalloc((void **)&ctx->d_pool, (pl.pool_words + 8) * sizeof(uint64_t)));
                    ctx->pairs_cap = pl.pair_words;
                    ctx->pair_counts_cap = pl.count_words;
                    ctx->pool_cap = pl.pool_words;
                }
                if (!ctx->d_pool_ctl) KTA_HIP(ctx, hipMalloc(&ctx->d_pool_ctl, pl.ctl_bytes));
                if (!ctx->d_fail_from) KTA_HIP(ctx, hipMalloc((void **)&ctx->d_fail_from, 3 * sizeof(uint32_t) << pl.bucket_log2));   // (per bucket + the list of given-up buckets + the list for the slot-range passes)
                kta::AliveColumns sl{c->key_len + at, c->val_len + at, c->key_off + at, c->key_bytes,
                                     ctx->alive_table && c->seq ? c->seq + at : nullptr};
                kta::AliveState st{ctx->alive_table ? ctx->d_table : nullptr, ctx->alive_table ? nullptr : ctx->d_bitmap,
                                   ctx->d_alive_running, written_list(ctx)};
                kta::AliveWorkspace ws{ctx->d_pairs, ctx->d_pair_counts, ctx->d_pool, ctx->d_pool_ctl, ctx->d_fail_from, ctx->d_failed_total};
                ctx->info_slices++;
                if (fuse && kta::alive_fuse_possible(pl, ctx->P) && pl.segment_wgs <= ctx->max_rows) {
                    ctx->info_fused++;
                    const uint32_t row_len = kta::scan_row_len(ctx->P, false);
                    const kta::AliveFuse fz{c->partition + at, c->ts_ms + at, ctx->P, ctx->d_partials, row_len};
                    KTA_HIP(ctx, kta::launch_alive_partitioned(sl, take, base_seq + at, st, pl, ws, report ? ctx->d_alive_stats : nullptr,
                                                               ctx->s_compute, &fz));
                    KTA_HIP(ctx, kta::launch_fold_partials(ctx->d_partials, pl.segment_wgs, ctx->P, ctx->d_vec, row_len, ctx->d_avec,
                                                           ctx->s_compute));
                } else {
                    if (fuse) {      // (a slice the fused pass does not take: its records go through the scan)
                        ctx->info_scanned++;
                        kta::ScanColumns sc{c->partition + at, c->key_len + at, c->val_len + at, c->ts_ms + at};
                        kta::ScanPlan spl = kta::plan_scan(ctx->P, take, ctx->cu_count, ctx->scan_wgs, ctx->scan_variant, ctx->analytics);
                        if (spl.workgroups > ctx->max_rows) spl.workgroups = ctx->max_rows;
                        KTA_HIP(ctx, kta::launch_metrics_scan(spl, sc, take, ctx->P, ctx->d_partials, ctx->s_compute));
                        KTA_HIP(ctx, kta::launch_fold_partials(ctx->d_partials, spl.workgroups, ctx->P, ctx->d_vec, spl.row_len, ctx->d_avec,
                                                               ctx->s_compute));
                    }
                    KTA_HIP(ctx, kta::launch_alive_partitioned(sl, take, base_seq + at, st, pl, ws,
                                                               report ? ctx->d_alive_stats : nullptr, ctx->s_compute));
                }
                if (sl.seq)   // the batch's seq column did not ascend: the pair did nothing, this runs instead
                    KTA_HIP(ctx, kta::launch_alive_update(sl, take, base_seq + at, ctx->d_table, 0, 2, nullptr,
                                                          ctx->d_alive_running, ctx->s_compute,
                                                          kta::alive_order_flag(ws, (int)pl.bucket_log2), written_list(ctx)));
                at += take;
            }
            if (report) {
                KTA_HIP(ctx, hipMemcpyAsync(ctx->h_alive_stats, ctx->d_alive_stats, 2 * sizeof(uint64_t),
                                            hipMemcpyDeviceToHost, ctx->s_compute));
                KTA_HIP(ctx, hipEventRecord(ctx->ev_alive_stats, ctx->s_compute));
                ctx->alive_stats_pending = true;
            }
        } else {
            const int v = part_kind ? 2 : ctx->alive_variant;
            KTA_HIP(ctx, kta::launch_alive_update(ac, n, base_seq, ctx->d_table, ctx->alive_wgs > 2048 ? 0 : ctx->alive_wgs,
                                                  v, ctx->d_hash_scratch, ctx->d_alive_running, ctx->s_compute, nullptr,
                                                  written_list(ctx)));
        }
        if (ctx->timing) KTA_HIP(ctx, hipEventRecord(b, ctx->s_compute));
    }
    return KTA_OK;
}

int reset_state(kta_ctx *ctx)
{
    KTA_HIP(ctx, kta::launch_init_vector(ctx->d_vec, ctx->P, ctx->d_avec, ctx->s_compute));
    if (ctx->alive) {
        if (ctx->alive_table) {
            KTA_HIP(ctx, hipMemsetAsync(ctx->d_table, 0, kta::kAliveSlots * sizeof(uint64_t), ctx->s_compute));
            KTA_HIP(ctx, hipMemsetAsync(ctx->d_written_n, 0, sizeof(unsigned long long), ctx->s_compute));
            ctx->written_valid = true;
        } else {
            KTA_HIP(ctx, hipMemsetAsync(ctx->d_bitmap, 0, (size_t)(kta::kAliveSlots / 8), ctx->s_compute));
        }
        KTA_HIP(ctx, hipMemsetAsync(ctx->d_alive_running, 0, sizeof(int64_t), ctx->s_compute));
        ctx->running_valid = true;
        // a new topic: what the old one's batches taught about backing off does not carry over (a word still on its way
        // lands in h_alive_stats before any later copy — same stream — and is never looked at)
        if (ctx->d_failed_total) KTA_HIP(ctx, hipMemsetAsync(ctx->d_failed_total, 0, sizeof(uint64_t), ctx->s_compute));
        ctx->alive_stats_pending = false;
        ctx->alive_backoff = 0;
        ctx->info_slices = ctx->info_fused = ctx->info_scanned = ctx->info_failed_buckets = 0;
    }
    ctx->next_seq = 0;
    ctx->msg_count = ctx->msg_flushes = ctx->msg_flush_ns = ctx->msg_wait_ns = 0;
    return KTA_OK;
}

} // namespace

extern "C" {

int kta_abi_version(void) { return KTA_ABI_VERSION; }

int kta_device_count(int *n)
{
    if (!n) return KTA_ERR_INVALID;
    *n = 0;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return KTA_ERR_NO_DEVICE;
    *n = ndev;
    return KTA_OK;
}

const char *kta_last_error(const kta_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int kta_create(const kta_config *cfg, kta_ctx **out)
{
    if (!cfg || !out) return fail(nullptr, KTA_ERR_INVALID, "kta_create: null argument");
    *out = nullptr;
    if (cfg->n_partitions <= 0 || cfg->n_partitions > 4096)
        return fail(nullptr, KTA_ERR_INVALID, "n_partitions must be in [1, 4096]");
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return fail(nullptr, KTA_ERR_NO_DEVICE, "no HIP device visible (libkta_hip has no CPU fallback)");
    if (cfg->device_id < 0 || cfg->device_id >= ndev)
        return fail(nullptr, KTA_ERR_INVALID, "device_id out of range");
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, cfg->device_id);
    if (e != hipSuccess) return hip_fail(nullptr, e, "hipGetDeviceProperties");
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(nullptr, KTA_ERR_NO_DEVICE,
                    std::string("device is ") + prop.gcnArchName + ", libkta_hip is built for gfx950 only");
    e = hipSetDevice(cfg->device_id);
    if (e != hipSuccess) return hip_fail(nullptr, e, "hipSetDevice");

    kta_ctx *ctx = new (std::nothrow) kta_ctx();
    if (!ctx) return fail(nullptr, KTA_ERR_NOMEM, "out of host memory");
    ctx->device = cfg->device_id;
    ctx->P = (uint32_t)cfg->n_partitions;
    ctx->alive = cfg->count_alive_keys != 0;
    ctx->analytics = (cfg->flags & KTA_FLAG_ANALYTICS) != 0;
    {
        const char *nf = getenv("KTA_NO_FUSE");      // A/B switch of bench.py and the tests: the two handlers as two passes
        ctx->fuse_handlers = !(nf && nf[0] == '1');
    }
    ctx->stage_seq = (cfg->flags & KTA_FLAG_SEQ_COLUMN) != 0;
    ctx->alive_table = ctx->alive && (cfg->flags & (KTA_FLAG_SEQ_COLUMN | KTA_FLAG_ALIVE_TABLE)) != 0;
    ctx->cu_count = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    ctx->batch_capacity = cfg->batch_capacity ? cfg->batch_capacity : (1ull << 22);
    ctx->key_bytes_capacity = cfg->key_bytes_capacity ? cfg->key_bytes_capacity : 64ull * ctx->batch_capacity;
    int rc = KTA_OK;
    auto bail = [&](int code) {
        g_create_error = ctx->err;
        kta_destroy(ctx);
        return code;
    };
    if (ctx->key_bytes_capacity >= (1ull << 32)) {
        ctx->err = "key_bytes_capacity must be < 4 GiB (key_off is a batch-local u32)";
        return bail(KTA_ERR_INVALID);
    }
#define KTA_TRY(call)                                                     \
    do {                                                                  \
        hipError_t e__ = (call);                                          \
        if (e__ != hipSuccess) return bail(hip_fail(ctx, e__, #call));    \
    } while (0)
    KTA_TRY(hipStreamCreateWithFlags(&ctx->s_compute, hipStreamNonBlocking));
    ctx->s_own = ctx->s_compute;
    KTA_TRY(hipStreamCreateWithFlags(&ctx->s_copy, hipStreamNonBlocking));
    KTA_TRY(hipEventCreateWithFlags(&ctx->ev_copied, hipEventDisableTiming));
    const size_t vec_words = (size_t)ctx->P * KTA_NCOUNTERS + KTA_NGLOBALS;
    KTA_TRY(hipMalloc((void **)&ctx->d_vec, vec_words * sizeof(uint64_t)));
    KTA_TRY(hipMalloc((void **)&ctx->d_vec_out, vec_words * sizeof(uint64_t)));
    ctx->max_rows = (uint32_t)ctx->cu_count * 8u;
    KTA_TRY(hipMalloc((void **)&ctx->d_partials,
                      (size_t)ctx->max_rows * kta::scan_row_len(ctx->P, ctx->analytics) * sizeof(uint64_t)));
    if (ctx->analytics)
        KTA_TRY(hipMalloc((void **)&ctx->d_avec, (size_t)kta::analytics_len(ctx->P) * sizeof(uint64_t)));
    if (ctx->alive) {
        if (ctx->alive_table) {
            KTA_TRY(hipMalloc((void **)&ctx->d_table, kta::kAliveSlots * sizeof(uint64_t)));
            ctx->written_cap = 1ull << 28;                 // 1 GiB of slot numbers; beyond that the exchange sweeps the table
            KTA_TRY(hipMalloc((void **)&ctx->d_written, ctx->written_cap * sizeof(uint32_t)));
            KTA_TRY(hipMalloc((void **)&ctx->d_written_n, sizeof(unsigned long long)));
        } else {
            KTA_TRY(hipMalloc((void **)&ctx->d_bitmap, (size_t)(kta::kAliveSlots / 8)));
        }
        KTA_TRY(hipMalloc((void **)&ctx->d_alive_running, sizeof(int64_t)));
    }
#undef KTA_TRY
    rc = reset_state(ctx);
    if (rc != KTA_OK) return bail(rc);
    // staging ring is allocated lazily (first kta_batch_acquire / kta_handle_message)
    ctx->stages.resize(cfg->n_staging > 0 ? (size_t)cfg->n_staging : 2);
    *out = ctx;
    return KTA_OK;
}

void kta_destroy(kta_ctx *ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->s_compute) (void)hipStreamSynchronize(ctx->s_compute);
    if (ctx->s_copy) (void)hipStreamSynchronize(ctx->s_copy);
    if (ctx->ext_state && ctx->ext_free) ctx->ext_free(ctx->ext_state);
    if (ctx->comm_state && ctx->comm_free) ctx->comm_free(ctx->comm_state);
    for (auto &st : ctx->stages) {
        if (st.host_slab) (void)hipHostFree(st.host_slab);
        if (st.dev_slab) (void)hipFree(st.dev_slab);
        if (st.done) (void)hipEventDestroy(st.done);
    }
    if (ctx->d_vec) (void)hipFree(ctx->d_vec);
    if (ctx->d_vec_out) (void)hipFree(ctx->d_vec_out);
    if (ctx->d_partials) (void)hipFree(ctx->d_partials);
    if (ctx->d_avec) (void)hipFree(ctx->d_avec);
    if (ctx->d_table) (void)hipFree(ctx->d_table);
    if (ctx->d_bitmap) (void)hipFree(ctx->d_bitmap);
    if (ctx->d_written) (void)hipFree(ctx->d_written);
    if (ctx->d_written_n) (void)hipFree(ctx->d_written_n);
    if (ctx->d_pool) (void)hipFree(ctx->d_pool);
    if (ctx->d_pool_ctl) (void)hipFree(ctx->d_pool_ctl);
    if (ctx->d_fail_from) (void)hipFree(ctx->d_fail_from);
    if (ctx->d_failed_total) (void)hipFree(ctx->d_failed_total);
    if (ctx->d_hash_scratch) (void)hipFree(ctx->d_hash_scratch);
    if (ctx->d_alive_stats) (void)hipFree(ctx->d_alive_stats);
    if (ctx->h_alive_stats) (void)hipHostFree(ctx->h_alive_stats);
    if (ctx->ev_alive_stats) (void)hipEventDestroy(ctx->ev_alive_stats);
    if (ctx->d_pairs) (void)hipFree(ctx->d_pairs);
    if (ctx->d_pair_counts) (void)hipFree(ctx->d_pair_counts);
    if (ctx->d_alive_running) (void)hipFree(ctx->d_alive_running);
    if (ctx->d_exp_slots) (void)hipFree(ctx->d_exp_slots);
    if (ctx->d_exp_vals) (void)hipFree(ctx->d_exp_vals);
    if (ctx->d_exp_count) (void)hipFree(ctx->d_exp_count);
    if (ctx->ev_copied) (void)hipEventDestroy(ctx->ev_copied);
    for (auto &pool : ctx->ev_pool)
        for (auto ev : pool) (void)hipEventDestroy(ev);
    if (ctx->s_own) (void)hipStreamDestroy(ctx->s_own);
    if (ctx->s_copy) (void)hipStreamDestroy(ctx->s_copy);
    delete ctx;
}

int kta_reset(kta_ctx *ctx)
{
    if (!ctx) return KTA_ERR_INVALID;
    KTA_HIP(ctx, hipSetDevice(ctx->device));
    ctx->fill_n = ctx->fill_kb = 0;
    return reset_state(ctx);
}

static int ensure_stage(kta_ctx *ctx, Stage &st)
{
    if (st.host_slab) return KTA_OK;
    const uint64_t cap = ctx->batch_capacity, kcap = ctx->key_bytes_capacity;
    const bool keys = ctx->alive, seq = ctx->alive && ctx->stage_seq;
    size_t off = 0, o_part, o_klen, o_vlen, o_ts, o_koff = 0, o_seq = 0, o_kb = 0;
    o_part = off; off += pad16(cap * 4);
    o_klen = off; off += pad16(cap * 4);
    o_vlen = off; off += pad16(cap * 4);
    o_ts = off; off += pad16(cap * 8);
    st.metric_bytes = off;
    if (keys) {
        o_koff = off; off += pad16(cap * 4);
        if (seq) { o_seq = off; off += pad16(cap * 8); }
        o_kb = off; off += pad16(kcap + 16);
    }
    st.key_bytes_off = o_kb;
    st.slab_bytes = off;
    KTA_HIP(ctx, hipHostMalloc((void **)&st.host_slab, off, hipHostMallocDefault));
    KTA_HIP(ctx, hipMalloc((void **)&st.dev_slab, off));
    auto point = [&](kta_batch &b, uint8_t *base) {
        memset(&b, 0, sizeof b);
        b.capacity = cap;
        b.key_bytes_capacity = keys ? kcap : 0;
        b.partition = reinterpret_cast<int32_t *>(base + o_part);
        b.key_len = reinterpret_cast<int32_t *>(base + o_klen);
        b.val_len = reinterpret_cast<int32_t *>(base + o_vlen);
        b.ts_ms = reinterpret_cast<int64_t *>(base + o_ts);
        if (keys) {
            b.key_off = reinterpret_cast<uint32_t *>(base + o_koff);
            b.key_bytes = base + o_kb;
            if (seq) b.seq = reinterpret_cast<uint64_t *>(base + o_seq);
        }
    };
    point(st.host, st.host_slab);
    point(st.dev, st.dev_slab);
    KTA_HIP(ctx, hipEventCreateWithFlags(&st.done, hipEventDisableTiming));
    return KTA_OK;
}

int kta_batch_acquire(kta_ctx *ctx, kta_batch *out)
{
    if (!ctx || !out) return KTA_ERR_INVALID;
    KTA_HIP(ctx, hipSetDevice(ctx->device));
    Stage &st = ctx->stages[ctx->cur];
    int rc = ensure_stage(ctx, st);
    if (rc != KTA_OK) return rc;
    if (st.busy) { // the ring wrapped: wait until the kernels that read this stage are done
        KTA_HIP(ctx, hipEventSynchronize(st.done));
        st.busy = false;
    }
    *out = st.host;
    ctx->acquired = true;
    return KTA_OK;
}

int kta_batch_submit(kta_ctx *ctx, uint64_t n, uint64_t n_key_bytes, uint64_t base_seq)
{
    if (!ctx) return KTA_ERR_INVALID;
    if (!ctx->acquired) return fail(ctx, KTA_ERR_INVALID, "kta_batch_submit without kta_batch_acquire");
    if (n > ctx->batch_capacity || (ctx->alive && n_key_bytes > ctx->key_bytes_capacity))
        return fail(ctx, KTA_ERR_CAPACITY, "batch larger than the staging capacity");
    KTA_HIP(ctx, hipSetDevice(ctx->device));
    Stage &st = ctx->stages[ctx->cur];
    ctx->acquired = false;
    if (n == 0) return KTA_OK;
    hipStream_t cs = ctx->s_copy;
    const size_t used = ctx->alive ? st.key_bytes_off + n_key_bytes : st.metric_bytes;
    if (n * 2 >= ctx->batch_capacity) {
        // a batch that is at least half full (every batch of a stream but its last): ONE copy of the slab's
        // used prefix — the unused tails of the columns travel along, the launches do not multiply
        KTA_HIP(ctx, hipMemcpyAsync(st.dev_slab, st.host_slab, used, hipMemcpyHostToDevice, cs));
    } else {
        KTA_HIP(ctx, hipMemcpyAsync(st.dev.partition, st.host.partition, n * 4, hipMemcpyHostToDevice, cs));
        KTA_HIP(ctx, hipMemcpyAsync(st.dev.key_len, st.host.key_len, n * 4, hipMemcpyHostToDevice, cs));
        KTA_HIP(ctx, hipMemcpyAsync(st.dev.val_len, st.host.val_len, n * 4, hipMemcpyHostToDevice, cs));
        KTA_HIP(ctx, hipMemcpyAsync(st.dev.ts_ms, st.host.ts_ms, n * 8, hipMemcpyHostToDevice, cs));
        if (ctx->alive) {
            KTA_HIP(ctx, hipMemcpyAsync(st.dev.key_off, st.host.key_off, n * 4, hipMemcpyHostToDevice, cs));
            if (n_key_bytes)
                KTA_HIP(ctx, hipMemcpyAsync(st.dev.key_bytes, st.host.key_bytes, n_key_bytes, hipMemcpyHostToDevice, cs));
            if (st.host.seq)   // KTA_FLAG_SEQ_COLUMN: the producer wrote every record's global sequence number
                KTA_HIP(ctx, hipMemcpyAsync(st.dev.seq, st.host.seq, n * 8, hipMemcpyHostToDevice, cs));
        }
    }
    KTA_HIP(ctx, hipEventRecord(ctx->ev_copied, cs));
    KTA_HIP(ctx, hipStreamWaitEvent(ctx->s_compute, ctx->ev_copied, 0));
    int rc = run_device_batch(ctx, &st.dev, n, base_seq, 3);
    if (rc != KTA_OK) return rc;
    KTA_HIP(ctx, hipEventRecord(st.done, ctx->s_compute));
    st.busy = true;
    ctx->cur = (ctx->cur + 1) % (int)ctx->stages.size();
    return KTA_OK;
}

int kta_flush(kta_ctx *ctx)
{
    if (!ctx) return KTA_ERR_INVALID;
    if (ctx->fill_n == 0) return KTA_OK;
    const uint64_t n = ctx->fill_n, kb = ctx->fill_kb;
    const uint64_t base = ctx->next_seq;
    ctx->fill_n = ctx->fill_kb = 0;
    ctx->next_seq += n;
    return kta_batch_submit(ctx, n, kb, base);
}

int kta_seek_seq(kta_ctx *ctx, uint64_t next_seq)
{
    if (!ctx) return KTA_ERR_INVALID;
    int rc = kta_flush(ctx);   // records already staged keep the numbers they were given
    if (rc != KTA_OK) return rc;
    ctx->next_seq = next_seq;
    return KTA_OK;
}

int kta_handle_message(kta_ctx *ctx, int32_t partition, int64_t ts_ms, const void *key, int64_t key_len,
                       int64_t val_len)
{
    if (!ctx) return KTA_ERR_INVALID;
    if (key_len > INT32_MAX || val_len > INT32_MAX)
        return fail(ctx, KTA_ERR_INVALID, "key/value length above i32 range");
    if (!key) key_len = -1; // m.key() is None iff librdkafka's key pointer is null
    const uint64_t kb = (ctx->alive && key_len > 0) ? (uint64_t)key_len : 0;
    if (kb > ctx->key_bytes_capacity) return fail(ctx, KTA_ERR_CAPACITY, "key larger than key_bytes_capacity");
    if (ctx->fill_n > 0 && (ctx->fill_n == ctx->batch_capacity || ctx->fill_kb + kb > ctx->key_bytes_capacity)) {
        const uint64_t t0 = now_ns();
        int rc = kta_flush(ctx);
        ctx->msg_flush_ns += now_ns() - t0;
        ctx->msg_flushes++;
        if (rc != KTA_OK) return rc;
    }
    if (ctx->fill_n == 0) {
        kta_batch tmp;
        const uint64_t t0 = now_ns();
        int rc = kta_batch_acquire(ctx, &tmp);
        ctx->msg_wait_ns += now_ns() - t0;
        if (rc != KTA_OK) return rc;
    }
    kta_batch &h = ctx->stages[ctx->cur].host;
    const uint64_t i = ctx->fill_n;
    h.partition[i] = partition;
    h.ts_ms[i] = ts_ms;
    h.key_len[i] = key_len < 0 ? -1 : (int32_t)key_len;
    h.val_len[i] = val_len < 0 ? -1 : (int32_t)val_len;
    if (ctx->alive) {
        if (h.seq) h.seq[i] = ctx->next_seq + i;
        h.key_off[i] = (uint32_t)ctx->fill_kb;
        if (kb) {
            memcpy(h.key_bytes + ctx->fill_kb, key, kb);
            ctx->fill_kb += kb;
        }
    }
    ctx->fill_n = i + 1;
    ctx->msg_count++;
    return KTA_OK;
}

int kta_replay_messages(kta_ctx *ctx, const kta_batch *c, uint64_t n)
{
    if (!ctx || !c) return KTA_ERR_INVALID;
    if (!c->partition || !c->key_len || !c->val_len || !c->ts_ms) return fail(ctx, KTA_ERR_INVALID, "metric columns missing");
    if (ctx->alive && (!c->key_off || !c->key_bytes)) return fail(ctx, KTA_ERR_INVALID, "key columns missing (count_alive_keys)");
    // through a pointer the compiler cannot see through: the loop pays the call a foreign caller pays per message
    static int (*volatile entry)(kta_ctx *, int32_t, int64_t, const void *, int64_t, int64_t) = kta_handle_message;
    static const uint8_t no_bytes[1] = {0};
    for (uint64_t i = 0; i < n; i++) {
        const int32_t kl = c->key_len[i];
        const void *key = kl < 0 ? nullptr : (c->key_bytes && c->key_off ? (const void *)(c->key_bytes + c->key_off[i]) : (const void *)no_bytes);
        int rc = entry(ctx, c->partition[i], c->ts_ms[i], key, kl, c->val_len[i]);
        if (rc != KTA_OK) return rc;
    }
    return KTA_OK;
}

int kta_handle_message_stats(kta_ctx *ctx, uint64_t out[4])
{
    if (!ctx || !out) return KTA_ERR_INVALID;
    out[0] = ctx->msg_count, out[1] = ctx->msg_flushes, out[2] = ctx->msg_flush_ns, out[3] = ctx->msg_wait_ns;
    return KTA_OK;
}

int kta_submit_device_ex(kta_ctx *ctx, const kta_batch *cols, uint64_t n, uint64_t base_seq, int which)
{
    if (!ctx || !cols) return KTA_ERR_INVALID;
    if (which < 1 || which > 3) return fail(ctx, KTA_ERR_INVALID, "which must be 1, 2 or 3");
    KTA_HIP(ctx, hipSetDevice(ctx->device));
    return run_device_batch(ctx, cols, n, base_seq, which);
}

int kta_submit_device(kta_ctx *ctx, const kta_batch *cols, uint64_t n, uint64_t base_seq)
{
    return kta_submit_device_ex(ctx, cols, n, base_seq, 3);
}

int kta_device_batch_alloc(kta_ctx *ctx, uint64_t capacity, uint64_t key_bytes_capacity, int with_seq,
                           kta_batch *out)
{
    if (!ctx || !out) return KTA_ERR_INVALID;
    KTA_HIP(ctx, hipSetDevice(ctx->device));
    if (key_bytes_capacity >= (1ull << 32))
        return fail(ctx, KTA_ERR_INVALID, "key_bytes_capacity must be < 4 GiB per batch");
    int rc = alloc_device_batch(ctx, capacity, key_bytes_capacity, key_bytes_capacity > 0, with_seq != 0, out);
    if (rc != KTA_OK) free_device_batch(out);
    return rc;
}

int kta_device_batch_free(kta_ctx *ctx, kta_batch *cols)
{
    if (!ctx || !cols) return KTA_ERR_INVALID;
    KTA_HIP(ctx, hipSetDevice(ctx->device));
    KTA_HIP(ctx, hipStreamSynchronize(ctx->s_compute));
    free_device_batch(cols);
    return KTA_OK;
}

int kta_copy_to_device(kta_ctx *ctx, void *dst, const void *src, size_t bytes)
{
    if (!ctx) return KTA_ERR_INVALID;
    KTA_HIP(ctx, hipSetDevice(ctx->device));
    KTA_HIP(ctx, hipStreamSynchronize(ctx->s_compute));
    KTA_HIP(ctx, hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
    return KTA_OK;
}

int kta_copy_to_host(kta_ctx *ctx, void *dst, const void *src, size_t bytes)
{
    if (!ctx) return KTA_ERR_INVALID;
    KTA_HIP(ctx, hipSetDevice(ctx->device));
    KTA_HIP(ctx, hipStreamSynchronize(ctx->s_compute));
    KTA_HIP(ctx, hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
    return KTA_OK;
}

int kta_set_compute_stream(kta_ctx *ctx, void *hip_stream)
{
    if (!ctx) return KTA_ERR_INVALID;
    KTA_HIP(ctx, hipSetDevice(ctx->device));
    KTA_HIP(ctx, hipStreamSynchronize(ctx->s_compute));   // nothing of ours may still be in flight on the old stream
    ctx->s_compute = hip_stream ? static_cast<hipStream_t>(hip_stream) : ctx->s_own;
    return KTA_OK;
}

int kta_sync(kta_ctx *ctx)
{
    if (!ctx) return KTA_ERR_INVALID;
    KTA_HIP(ctx, hipSetDevice(ctx->device));
    KTA_HIP(ctx, hipStreamSynchronize(ctx->s_copy));
    KTA_HIP(ctx, hipStreamSynchronize(ctx->s_compute));
    return KTA_OK;
}

int kta_finish_device(kta_ctx *ctx)
{
    if (!ctx) return KTA_ERR_INVALID;
    KTA_HIP(ctx, hipSetDevice(ctx->device));
    int rc = kta_flush(ctx);
    if (rc != KTA_OK) return rc;
    const size_t words = (size_t)ctx->P * KTA_NCOUNTERS + KTA_NGLOBALS;
    KTA_HIP(ctx, hipMemcpyAsync(ctx->d_vec_out, ctx->d_vec, words * sizeof(uint64_t), hipMemcpyDeviceToDevice,
                                ctx->s_compute));
    if (ctx->alive) {
        uint64_t *dst = ctx->d_vec_out + (size_t)ctx->P * KTA_NCOUNTERS + KTA_G_ALIVE_KEYS;
        if (ctx->running_valid)  // exact running count (every update so far ran a counting kernel): no table scan
            KTA_HIP(ctx, hipMemcpyAsync(dst, ctx->d_alive_running, sizeof(uint64_t), hipMemcpyDeviceToDevice,
                                        ctx->s_compute));
        else if (ctx->alive_table)
            KTA_HIP(ctx, kta::launch_alive_count(ctx->d_table, kta::kAliveSlots, dst, ctx->s_compute));
        else {
            KTA_HIP(ctx, hipMemsetAsync(dst, 0, sizeof(uint64_t), ctx->s_compute));
            KTA_HIP(ctx, kta::launch_bitmap_count(ctx->d_bitmap, dst, ctx->s_compute));
        }
    }
    return KTA_OK;
}

int kta_result_vector(kta_ctx *ctx, void **device_ptr, size_t *n_u64)
{
    if (!ctx || !device_ptr || !n_u64) return KTA_ERR_INVALID;
    *device_ptr = ctx->d_vec_out;
    *n_u64 = (size_t)ctx->P * KTA_NCOUNTERS + KTA_NGLOBALS;
    return KTA_OK;
}

int kta_decode_vector(const uint64_t *vec, uint32_t P, int count_alive_keys, kta_result *out,
                      uint64_t *counters_out)
{
    if (!vec || !out || P == 0) return KTA_ERR_INVALID;
    const uint64_t *g = vec + (size_t)P * KTA_NCOUNTERS;
    memset(out, 0, sizeof(*out));
    out->n_partitions = P;
    uint64_t total = 0, live = 0, size = 0;
    for (uint32_t p = 0; p < P; p++) {
        const uint64_t *c = vec + (size_t)p * KTA_NCOUNTERS;
        total += c[KTA_C_TOTAL];
        live += c[KTA_C_ALIVE];
        size += c[KTA_C_KEY_SIZE_SUM] + c[KTA_C_VALUE_SIZE_SUM];
    }
    if (counters_out) memcpy(counters_out, vec, (size_t)P * KTA_NCOUNTERS * sizeof(uint64_t));
    out->any_records = total > 0;
    out->any_live = live > 0;
    out->count_alive_keys = count_alive_keys ? 1u : 0u;
    // metric.rs:210: timestamp / 1000 truncates toward zero, and is monotone, so applying it
    // to the extrema of the millisecond values equals the extrema of the per-record seconds.
    out->min_ts_sec = out->any_records ? (int64_t)~g[KTA_G_NOT_MIN_TS_MS] / 1000 : 0;
    out->max_ts_sec = out->any_records ? (int64_t)g[KTA_G_MAX_TS_MS] / 1000 : 0;
    out->smallest_message = out->any_live ? ~g[KTA_G_NOT_SMALLEST] : UINT64_MAX; // metric.rs:42
    out->largest_message = g[KTA_G_LARGEST];
    out->overall_count = total;
    out->overall_size = size;
    out->alive_keys = count_alive_keys ? g[KTA_G_ALIVE_KEYS] : 0;
    out->bad_partition_records = g[KTA_G_BAD_PARTITION];
    // metric.rs:210 / kafka.rs:104: NaiveDateTime::from_timestamp panics outside chrono's range, on the first
    // such record; one such record among those counted puts an extremum outside the range
    if (out->any_records && (out->min_ts_sec < KTA_CHRONO_MIN_SEC || out->max_ts_sec > KTA_CHRONO_MAX_SEC))
        return KTA_ERR_TIMESTAMP_RANGE;
    return out->bad_partition_records ? KTA_ERR_BAD_PARTITION : KTA_OK;
}

int kta_merge_vectors(uint64_t *acc, const uint64_t *other, uint32_t P)
{
    if (!acc || !other || P == 0) return KTA_ERR_INVALID;
    const size_t nc = (size_t)P * KTA_NCOUNTERS;
    for (size_t i = 0; i < nc + KTA_NSUM_GLOBALS; i++) acc[i] += other[i];
    for (size_t i = nc + KTA_NSUM_GLOBALS; i < nc + KTA_NGLOBALS; i++)
        if ((int64_t)other[i] > (int64_t)acc[i]) acc[i] = other[i];
    return KTA_OK;
}

int kta_finish(kta_ctx *ctx, kta_result *out, uint64_t *counters_out)
{
    if (!ctx || !out) return KTA_ERR_INVALID;
    int rc = kta_finish_device(ctx);
    if (rc != KTA_OK) return rc;
    return kta_exchange_result(ctx, out, counters_out);
}

int kta_exchange_result(kta_ctx *ctx, kta_result *out, uint64_t *counters_out)
{
    if (!ctx || !out) return KTA_ERR_INVALID;
    KTA_HIP(ctx, hipSetDevice(ctx->device));
    const size_t words = (size_t)ctx->P * KTA_NCOUNTERS + KTA_NGLOBALS;
    std::vector<uint64_t> host(words);
    KTA_HIP(ctx, hipMemcpyAsync(host.data(), ctx->d_vec_out, words * sizeof(uint64_t), hipMemcpyDeviceToHost,
                                ctx->s_compute));
    KTA_HIP(ctx, hipStreamSynchronize(ctx->s_compute));
    int rc = kta_decode_vector(host.data(), ctx->P, ctx->alive ? 1 : 0, out, counters_out);
    if (rc == KTA_ERR_BAD_PARTITION) {
        char buf[128];
        snprintf(buf, sizeof buf, "%llu record(s) had a partition id outside [0, %u)",
                 (unsigned long long)out->bad_partition_records, ctx->P);
        ctx->err = buf;
    } else if (rc == KTA_ERR_TIMESTAMP_RANGE) {
        ctx->err = "a record's timestamp / 1000 is outside chrono's NaiveDateTime range: the reference panics on it "
                   "('invalid or out-of-range datetime', metric.rs:210)";
    }
    return rc;
}

int kta_analytics_vector(kta_ctx *ctx, void **device_ptr, size_t *n_u64)
{
    if (!ctx || !device_ptr || !n_u64) return KTA_ERR_INVALID;
    if (!ctx->analytics) return fail(ctx, KTA_ERR_INVALID, "context was created without KTA_FLAG_ANALYTICS");
    *device_ptr = ctx->d_avec;
    *n_u64 = kta::analytics_len(ctx->P);
    return KTA_OK;
}

int kta_get_analytics(kta_ctx *ctx, kta_analytics *out, int64_t *part_min_ts_sec, int64_t *part_max_ts_sec,
                      uint64_t *part_smallest, uint64_t *part_largest)
{
    if (!ctx || !out) return KTA_ERR_INVALID;
    if (!ctx->analytics) return fail(ctx, KTA_ERR_INVALID, "context was created without KTA_FLAG_ANALYTICS");
    KTA_HIP(ctx, hipSetDevice(ctx->device));
    int rc = kta_flush(ctx);
    if (rc != KTA_OK) return rc;
    std::vector<uint64_t> host(kta::analytics_len(ctx->P));
    KTA_HIP(ctx, hipMemcpyAsync(host.data(), ctx->d_avec, host.size() * sizeof(uint64_t), hipMemcpyDeviceToHost,
                                ctx->s_compute));
    KTA_HIP(ctx, hipStreamSynchronize(ctx->s_compute));
    for (int b = 0; b < KTA_HIST_BUCKETS; b++) {
        out->key_size_hist[b] = host[b];
        out->value_size_hist[b] = host[KTA_HIST_BUCKETS + b];
    }
    for (uint32_t p = 0; p < ctx->P; p++) {
        const int64_t *x = reinterpret_cast<const int64_t *>(host.data()) + kta::kAnalyticsHist + 4 * (size_t)p;
        const bool seen = x[1] != INT64_MIN;     // max ts never written => no record in this partition
        const bool live = x[3] != INT64_MIN;     // largest never written => no non-tombstone
        if (part_min_ts_sec) part_min_ts_sec[p] = seen ? (~x[0]) / 1000 : INT64_MAX;  // metric.rs:210 (monotone)
        if (part_max_ts_sec) part_max_ts_sec[p] = seen ? x[1] / 1000 : INT64_MIN;
        if (part_smallest) part_smallest[p] = live ? (uint64_t)~x[2] : UINT64_MAX;
        if (part_largest) part_largest[p] = live ? (uint64_t)x[3] : 0;
    }
    return KTA_OK;
}

static const char *const kNeedsTable =
    "the context keeps the alive set as a bit set: create it with KTA_FLAG_ALIVE_TABLE for sequence-numbered entries";

int kta_export_alive_bitmap(kta_ctx *ctx, void *dst)
{
    if (!ctx || !dst) return KTA_ERR_INVALID;
    if (!ctx->alive) return fail(ctx, KTA_ERR_INVALID, "context was created without count_alive_keys");
    KTA_HIP(ctx, hipSetDevice(ctx->device));
    int rc = kta_flush(ctx);
    if (rc != KTA_OK) return rc;
    uint32_t *d_bm = nullptr;
    const size_t bytes = (size_t)(kta::kAliveSlots / 8);
    if (!ctx->alive_table) {       // the state IS the reference's bit set
        KTA_HIP(ctx, hipMemcpyAsync(dst, ctx->d_bitmap, bytes, hipMemcpyDeviceToHost, ctx->s_compute));
        KTA_HIP(ctx, hipStreamSynchronize(ctx->s_compute));
        return KTA_OK;
    }
    KTA_HIP(ctx, hipMalloc((void **)&d_bm, bytes));
    hipError_t e = kta::launch_alive_bitmap(ctx->d_table, kta::kAliveSlots, d_bm, ctx->s_compute);
    if (e == hipSuccess) e = hipMemcpyAsync(dst, d_bm, bytes, hipMemcpyDeviceToHost, ctx->s_compute);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->s_compute);
    (void)hipFree(d_bm);
    if (e != hipSuccess) return hip_fail(ctx, e, "alive bitmap export");
    return KTA_OK;
}

int kta_alive_table(kta_ctx *ctx, void **device_ptr, size_t *n_u64)
{
    if (!ctx || !device_ptr || !n_u64) return KTA_ERR_INVALID;
    if (!ctx->alive) return fail(ctx, KTA_ERR_INVALID, "context was created without count_alive_keys");
    if (!ctx->alive_table) return fail(ctx, KTA_ERR_INVALID, kNeedsTable);
    *device_ptr = ctx->d_table;
    *n_u64 = (size_t)kta::kAliveSlots;
    return KTA_OK;
}

int kta_alive_export_entries(kta_ctx *ctx, void **d_slots, void **d_vals, uint64_t *n)
{
    if (!ctx || !d_slots || !d_vals || !n) return KTA_ERR_INVALID;
    if (!ctx->alive) return fail(ctx, KTA_ERR_INVALID, "context was created without count_alive_keys");
    if (!ctx->alive_table) return fail(ctx, KTA_ERR_INVALID, kNeedsTable);
    KTA_HIP(ctx, hipSetDevice(ctx->device));
    int rc = kta_flush(ctx);
    if (rc != KTA_OK) return rc;
    hipStream_t s = ctx->s_compute;
    if (!ctx->d_exp_count) KTA_HIP(ctx, hipMalloc((void **)&ctx->d_exp_count, sizeof(uint64_t)));
    uint64_t written = 0;
    KTA_HIP(ctx, kta::launch_alive_count_written(ctx->d_table, kta::kAliveSlots, ctx->d_exp_count, s));
    KTA_HIP(ctx, hipMemcpyAsync(&written, ctx->d_exp_count, sizeof(uint64_t), hipMemcpyDeviceToHost, s));
    KTA_HIP(ctx, hipStreamSynchronize(s));
    if (ctx->exp_cap < written) {
        if (ctx->d_exp_slots) (void)hipFree(ctx->d_exp_slots);
        if (ctx->d_exp_vals) (void)hipFree(ctx->d_exp_vals);
        ctx->d_exp_slots = nullptr;
        ctx->d_exp_vals = nullptr;
        ctx->exp_cap = written + written / 8 + 1024;
        KTA_HIP(ctx, hipMalloc((void **)&ctx->d_exp_slots, ctx->exp_cap * sizeof(uint32_t)));
        KTA_HIP(ctx, hipMalloc((void **)&ctx->d_exp_vals, ctx->exp_cap * sizeof(uint64_t)));
    }
    if (written)
        KTA_HIP(ctx, kta::launch_alive_export(ctx->d_table, kta::kAliveSlots, ctx->d_exp_slots, ctx->d_exp_vals,
                                              ctx->d_exp_count, ctx->exp_cap, s));
    KTA_HIP(ctx, hipStreamSynchronize(s));
    *d_slots = ctx->d_exp_slots;
    *d_vals = ctx->d_exp_vals;
    *n = written;
    return KTA_OK;
}

int kta_alive_import_entries(kta_ctx *ctx, const void *d_slots, const void *d_vals, uint64_t n)
{
    if (!ctx || (n && (!d_slots || !d_vals))) return KTA_ERR_INVALID;
    if (!ctx->alive) return fail(ctx, KTA_ERR_INVALID, "context was created without count_alive_keys");
    if (!ctx->alive_table) return fail(ctx, KTA_ERR_INVALID, kNeedsTable);
    KTA_HIP(ctx, hipSetDevice(ctx->device));
    int rc = kta_flush(ctx);
    if (rc != KTA_OK) return rc;
    KTA_HIP(ctx, kta::launch_alive_import(static_cast<const uint32_t *>(d_slots), static_cast<const uint64_t *>(d_vals),
                                          n, ctx->d_table, ctx->d_alive_running, written_list(ctx), ctx->s_compute));
    return KTA_OK;
}

int kta_alive_count_range(kta_ctx *ctx, uint64_t slot_lo, uint64_t slot_hi, uint64_t *count)
{
    if (!ctx || !count || slot_lo > slot_hi || slot_hi > kta::kAliveSlots) return KTA_ERR_INVALID;
    if (!ctx->alive) return fail(ctx, KTA_ERR_INVALID, "context was created without count_alive_keys");
    if (!ctx->alive_table) return fail(ctx, KTA_ERR_INVALID, kNeedsTable);
    KTA_HIP(ctx, hipSetDevice(ctx->device));
    int rc = kta_flush(ctx);
    if (rc != KTA_OK) return rc;
    hipStream_t s = ctx->s_compute;
    if (!ctx->d_exp_count) KTA_HIP(ctx, hipMalloc((void **)&ctx->d_exp_count, sizeof(uint64_t)));
    KTA_HIP(ctx, kta::launch_alive_count_span(ctx->d_table, slot_lo, slot_hi, ctx->d_exp_count, s));
    KTA_HIP(ctx, hipMemcpyAsync(count, ctx->d_exp_count, sizeof(uint64_t), hipMemcpyDeviceToHost, s));
    KTA_HIP(ctx, hipStreamSynchronize(s));
    return KTA_OK;
}

int kta_alive_table_modified(kta_ctx *ctx)
{
    if (!ctx) return KTA_ERR_INVALID;
    if (!ctx->alive) return fail(ctx, KTA_ERR_INVALID, "context was created without count_alive_keys");
    ctx->running_valid = false;  // the next kta_finish recounts from the table
    ctx->written_valid = false;  // and the exchange sweeps it: somebody else wrote entries
    return KTA_OK;
}

int kta_fnv32_device(kta_ctx *ctx, const uint8_t *key_bytes, const uint32_t *key_off, const int32_t *key_len,
                     uint64_t n, uint64_t n_key_bytes, uint32_t *hash_out)
{
    if (!ctx || !key_off || !key_len || !hash_out) return KTA_ERR_INVALID;
    if (n == 0) return KTA_OK;
    KTA_HIP(ctx, hipSetDevice(ctx->device));
    uint8_t *d_kb = nullptr;
    uint32_t *d_off = nullptr, *d_out = nullptr;
    int32_t *d_len = nullptr;
    hipError_t e = hipMalloc((void **)&d_kb, pad16(n_key_bytes + 16));
    if (e == hipSuccess) e = hipMalloc((void **)&d_off, n * 4);
    if (e == hipSuccess) e = hipMalloc((void **)&d_len, n * 4);
    if (e == hipSuccess) e = hipMalloc((void **)&d_out, n * 4);
    hipStream_t s = ctx->s_compute;
    if (e == hipSuccess && n_key_bytes) e = hipMemcpyAsync(d_kb, key_bytes, n_key_bytes, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(d_off, key_off, n * 4, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(d_len, key_len, n * 4, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = kta::launch_fnv32(d_kb, d_off, d_len, n, d_out, s);
    if (e == hipSuccess) e = hipMemcpyAsync(hash_out, d_out, n * 4, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (d_kb) (void)hipFree(d_kb);
    if (d_off) (void)hipFree(d_off);
    if (d_len) (void)hipFree(d_len);
    if (d_out) (void)hipFree(d_out);
    if (e != hipSuccess) return hip_fail(ctx, e, "kta_fnv32_device");
    return KTA_OK;
}

int kta_set_timing(kta_ctx *ctx, int enable)
{
    if (!ctx) return KTA_ERR_INVALID;
    ctx->timing = enable != 0;
    return KTA_OK;
}

int kta_kernel_time_stats(kta_ctx *ctx, float avg_ms[3], uint64_t launches[3])
{
    if (!ctx || !avg_ms || !launches) return KTA_ERR_INVALID;
    KTA_HIP(ctx, hipSetDevice(ctx->device));
    int rc = drain_timers(ctx);
    if (rc != KTA_OK) return rc;
    for (int k = 0; k < 3; k++) {
        launches[k] = ctx->ms_cnt[k];
        avg_ms[k] = ctx->ms_cnt[k] ? (float)(ctx->ms_sum[k] / (double)ctx->ms_cnt[k]) : -1.f;
        ctx->ms_sum[k] = 0;
        ctx->ms_cnt[k] = 0;
    }
    return KTA_OK;
}

int kta_set_tuning(kta_ctx *ctx, int scan_workgroups, int scan_variant, int alive_workgroups, int alive_variant)
{
    if (!ctx) return KTA_ERR_INVALID;
    if (scan_workgroups < 0 || alive_workgroups < 0) return fail(ctx, KTA_ERR_INVALID, "negative workgroup count");
    ctx->scan_wgs = scan_workgroups;
    ctx->scan_variant = scan_variant;
    ctx->alive_wgs = alive_workgroups;
    ctx->alive_variant = alive_variant;
    return KTA_OK;
}

int kta_set_fuse(kta_ctx *ctx, int enable)
{
    if (!ctx) return KTA_ERR_INVALID;
    ctx->fuse_handlers = enable != 0;
    return KTA_OK;
}

int kta_alive_pass_info(kta_ctx *ctx, uint64_t out[6])
{
    if (!ctx || !out) return KTA_ERR_INVALID;
    // the device's own count (every launch pair's fallback kernel adds what pass 2 handed it): waits for the compute stream
    ctx->info_failed_buckets = 0;
    if (ctx->d_failed_total) {
        KTA_HIP(ctx, hipSetDevice(ctx->device));
        KTA_HIP(ctx, hipMemcpyAsync(&ctx->info_failed_buckets, ctx->d_failed_total, sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->s_compute));
        KTA_HIP(ctx, hipStreamSynchronize(ctx->s_compute));
    }
    out[0] = kta::kAlivePartitionMax;
    out[1] = ctx->info_slices;
    out[2] = ctx->info_fused;
    out[3] = ctx->info_scanned;
    out[4] = ctx->info_failed_buckets;
    out[5] = ctx->fuse_handlers ? 1 : 0;
    return KTA_OK;
}

} // extern "C"

// exported for kta_synth.hip / kta_kafka.hip (same shared object)
void **kta_internal_ext_slot(kta_ctx *ctx, void (*free_fn)(void *))
{
    ctx->ext_free = free_fn;
    return &ctx->ext_state;
}
void **kta_internal_comm_slot(kta_ctx *ctx, void (*free_fn)(void *))
{
    ctx->comm_free = free_fn;
    return &ctx->comm_state;
}
uint64_t *kta_internal_vec_out(kta_ctx *ctx) { return ctx->d_vec_out; }
uint32_t kta_internal_partitions(kta_ctx *ctx) { return ctx->P; }
uint64_t *kta_internal_table(kta_ctx *ctx) { return ctx->d_table; }
bool kta_internal_alive_table(kta_ctx *ctx) { return ctx->alive_table; }
bool kta_internal_written(kta_ctx *ctx, kta::WrittenList *out)
{
    *out = kta::WrittenList{ctx->d_written, ctx->d_written_n, ctx->written_cap};
    return ctx->written_valid && ctx->d_written != nullptr;
}
int64_t *kta_internal_running(kta_ctx *ctx) { return ctx->d_alive_running; }
uint64_t kta_internal_take_seq(kta_ctx *ctx, uint64_t n)
{
    const uint64_t base = ctx->next_seq;
    ctx->next_seq += n;
    return base;
}
bool kta_internal_timing(kta_ctx *ctx) { return ctx->timing; }
hipStream_t kta_internal_copy_stream(kta_ctx *ctx) { return ctx->s_copy; }
bool kta_internal_count_alive(kta_ctx *ctx) { return ctx->alive; }
hipStream_t kta_internal_stream(kta_ctx *ctx) { return ctx->s_compute; }
int kta_internal_device(kta_ctx *ctx) { return ctx->device; }
void kta_internal_set_error(kta_ctx *ctx, const char *msg) { ctx->err = msg; }
