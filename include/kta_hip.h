/*
 * kta_hip.h — C ABI of libkta_hip.so: the MI355X (gfx950) implementation of the
 * per-record metric-accumulation hot path of xenji/kafka-topic-analyzer.
 *
 * This is the drop-in boundary.  Reference interface being replaced (paths under
 * /root/reference):
 *
 *   src/kafka.rs:18-20    trait MetricHandler { fn handle_message(&mut self, m: &BorrowedMessage) }
 *   src/kafka.rs:107-109  dispatch: every handler, every polled message, registration order
 *   src/metric.rs:206-253 impl MetricHandler for MessageMetrics
 *   src/metric.rs:288-305 impl MetricHandler for LogCompactionInMemoryMetrics (-c)
 *   src/metric.rs:104-195 accessors the report reads (main.rs:130-170)
 *   src/metric.rs:282-284 sum_all_alive()
 *
 * A Rust `MetricHandler` shim binds these entry points over FFI (INTEGRATION.md shows
 * the `extern "C"` block): `handle_message` becomes `kta_handle_message` (copies what
 * the borrowed message exposes into pinned struct-of-arrays staging and launches the
 * HIP kernels whenever a batch fills), and the accessors read a `kta_result` obtained
 * from `kta_finish`.  Plain pointers and sizes only; every function returns a status
 * (0 = OK, negative = error) and never throws or aborts across the boundary.
 * There is NO CPU fallback: without a usable gfx950 device `kta_create` fails.
 *
 * Threading: one producer thread per context (the reference's poll loop is single
 * threaded, kafka.rs:92-135).  Internally the context owns two HIP streams (H2D copy,
 * compute) and a ring of pinned staging batches.
 */
#ifndef KTA_HIP_H
#define KTA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KTA_ABI_VERSION 6   /* 6: kta_replay_messages, kta_handle_message_stats, kta_zstd_inflate_host_small (kta_kafka.h); the table state takes the fused pass; kta_kafka_set_variant takes 0, 1, 2, 10, 11 only (the other geometries went in round 5); 5: kta_set_fuse, kta_alive_pass_info; 4: KTA_FLAG_ALIVE_TABLE, the default -c state is the bit set (submission order); 3: kta_comm_* / kta_exchange*, kta_result_vector is a snapshot; 2: kta_kafka_batch_desc.scratch_end */

/* status codes */
#define KTA_OK 0
#define KTA_ERR_INVALID (-1)       /* bad argument / misuse                         */
#define KTA_ERR_HIP (-2)           /* a HIP runtime call failed (see kta_last_error) */
#define KTA_ERR_NOMEM (-3)         /* host or device allocation failed              */
#define KTA_ERR_NO_DEVICE (-4)     /* no usable gfx950 device                        */
#define KTA_ERR_BAD_PARTITION (-5) /* a record's partition id was outside [0, P)     */
#define KTA_ERR_CAPACITY (-6)      /* batch / key-byte capacity exceeded             */
#define KTA_ERR_DIV_BY_ZERO (-7)   /* where the reference panics (metric.rs:135,144,153) */
#define KTA_ERR_COMM (-8)          /* RCCL missing or a collective failed (see kta_last_error) */
#define KTA_ERR_TIMESTAMP_RANGE (-9) /* where the reference panics: a record's ts / 1000 outside chrono's range (metric.rs:210, kafka.rs:104) */

/* chrono 0.4.19 (Cargo.lock:84-85) NaiveDateTime::from_timestamp(secs, 0) — metric.rs:210, kafka.rs:104 —
 * .expect()s a date inside NaiveDate's years [i32::MIN >> 13, i32::MAX >> 13] = [-262144, 262143]:
 * -262144-01-01 00:00:00 .. 262143-12-31 23:59:59 in seconds since the epoch.  A record outside ends the
 * reference ("invalid or out-of-range datetime"); this library counts it like any other and says so at
 * kta_finish / kta_decode_vector (the extrema of the scan tell: one such record moves one of them out). */
#define KTA_CHRONO_MIN_SEC (-8334632851200LL)
#define KTA_CHRONO_MAX_SEC (8210298412799LL)

/* Per-partition counters, in the field order of `struct MessageMetrics`
 * (metric.rs:13-19). */
enum {
    KTA_C_TOTAL = 0,          /* total_messages  metric.rs:13 */
    KTA_C_TOMBSTONES = 1,     /* tombstones      metric.rs:14 */
    KTA_C_ALIVE = 2,          /* alive           metric.rs:15 */
    KTA_C_KEY_NULL = 3,       /* key_null        metric.rs:16 */
    KTA_C_KEY_NON_NULL = 4,   /* key_non_null    metric.rs:17 */
    KTA_C_KEY_SIZE_SUM = 5,   /* key_size_sum    metric.rs:18 */
    KTA_C_VALUE_SIZE_SUM = 6, /* value_size_sum  metric.rs:19 */
    KTA_NCOUNTERS = 7
};

/* The device result vector ("counter vector") is u64[P*7 + KTA_NGLOBALS]:
 * counters[p*7 + c], then KTA_NSUM_GLOBALS SUM-type globals, then four MAX-type
 * globals.  It is the unit that is reduced across GPUs when partitions are sharded:
 * ONE all-reduce SUM over the first P*7 + KTA_NSUM_GLOBALS words (u64 wrap-around ==
 * i64 wrap-around) and ONE all-reduce MAX (signed i64) over the last four.  Minima are
 * stored bit-complemented (~x is an order-reversing bijection on i64 without overflow),
 * so that every extremum is a MAX. */
enum {
    KTA_G_BAD_PARTITION = 0, /* SUM: records whose partition id was out of range (ignored)      */
    KTA_G_ALIVE_KEYS = 1,    /* SUM: alive keys of this context's table (kta_finish); in an exchange, of the
                                hash range this rank owns — disjoint ranges, so the SUM is the job's count */
    KTA_G_RECORDS = 2,       /* SUM: records scanned (== overall_count, metric.rs:25)            */
    KTA_G_RESERVED = 3,      /* SUM: reserved, zero                                              */
    KTA_NSUM_GLOBALS = 4,
    KTA_G_NOT_MIN_TS_MS = 4, /* MAX: ~min(ts_ms) (i64); ~INT64_MAX when no record seen; a raw
                                timestamp of -1 (not available) was mapped to 0 first          */
    KTA_G_MAX_TS_MS = 5,     /* MAX: max(ts_ms), INT64_MIN when no record seen                   */
    KTA_G_NOT_SMALLEST = 6,  /* MAX: ~(size of the smallest non-tombstone); ~INT64_MAX if none   */
    KTA_G_LARGEST = 7,       /* MAX: size of the largest non-tombstone, 0 if none (metric.rs:41) */
    KTA_NGLOBALS = 8
};

typedef struct kta_ctx kta_ctx;

typedef struct kta_config {
    int32_t device_id;           /* HIP device ordinal                                        */
    int32_t n_partitions;        /* P: partition ids are dense in [0, P) (Kafka's are)         */
    int32_t count_alive_keys;    /* 1 == the reference's -c/--count-alive-keys (main.rs:77-80) */
    int32_t n_staging;           /* pinned staging batches in the ring; 0 -> 2                */
    uint64_t batch_capacity;     /* records per staging batch; 0 -> 1<<22                     */
    uint64_t key_bytes_capacity; /* key bytes per staging batch; 0 -> 64 * batch_capacity;
                                    must be < 4 GiB (key_off is u32, batch-local)             */
    uint32_t flags;              /* KTA_FLAG_*                                                */
    uint32_t reserved;           /* 0                                                         */
} kta_config;

/* Additive analytics (NOT in the reference, never printed by the reference report): log2
 * histograms of key and value sizes and per-partition timestamp / message-size extrema,
 * accumulated by the same scan kernel in extra LDS arrays.  Opt-in: costs LDS, not bandwidth. */
#define KTA_FLAG_ANALYTICS 1u
/* The staging batches of kta_batch_acquire carry a `seq` column (with count_alive_keys): the producer writes
 * every record's GLOBAL consumption index, kta_batch_submit's base_seq is ignored.  For a rank of a
 * partition-sharded run, whose records are not consecutive in the topic's consumption order. */
#define KTA_FLAG_SEQ_COLUMN 2u
/* How a -c context keeps the alive set (the reference: one BitSet, metric.rs:262-264):
 *   default               the reference's own bit set, 2^32 bits = 512 MiB.  Batches are applied IN SUBMISSION
 *                         ORDER (what the reference does with the messages it polls); base_seq is not looked at.
 *   KTA_FLAG_ALIVE_TABLE  a last-writer table u64[2^32] = 32 GiB of ((seq + 1) << 1 | alive): batches, shards and
 *                         ranks may arrive in any order, the largest GLOBAL sequence number of a slot wins.  What a
 *                         rank of a sharded run needs (kta_comm_create with nranks > 1, kta_alive_table,
 *                         kta_alive_export_entries / import_entries / count_range).  Implied by KTA_FLAG_SEQ_COLUMN.
 *                         A seq column has to ascend inside each batch to take the fast path (checked on the device;
 *                         other batches are still exact, through the single-kernel update). */
#define KTA_FLAG_ALIVE_TABLE 4u
#define KTA_HIST_BUCKETS 34 /* [0] None, [1] length 0, [2+k] 2^k <= length < 2^(k+1), k = 0..31 */

typedef struct kta_analytics {
    uint64_t key_size_hist[KTA_HIST_BUCKETS];
    uint64_t value_size_hist[KTA_HIST_BUCKETS];
} kta_analytics;

/* One batch of decoded records as struct-of-arrays columns.  What the reference's
 * handlers read from a BorrowedMessage (metric.rs:208-209, 218, 233, 291-293):
 *   partition[i]  m.partition()                                          i32
 *   ts_ms[i]      raw rdkafka timestamp in ms; -1 == not available       i64
 *   key_len[i]    m.key():  -1 == None, >= 0 == Some(k).len()            i32
 *   val_len[i]    m.payload(): -1 == None (tombstone), >= 0 == len       i32
 *   key_off[i]    offset of the key's bytes in key_bytes (if key_len>0)  u32  (-c only)
 *   key_bytes     concatenated key bytes; device buffers must be readable  u8   (-c only)
 *                 for 16 bytes past the last key (kta_device_batch_alloc pads)
 *   seq[i]        optional global consumption index; NULL => base_seq+i  u64  (-c only)
 * Value bytes are never read by the reference path (only their length). */
typedef struct kta_batch {
    int32_t *partition;
    int32_t *key_len;
    int32_t *val_len;
    int64_t *ts_ms;
    uint32_t *key_off;
    uint8_t *key_bytes;
    uint64_t *seq;
    uint64_t capacity;           /* records the columns can hold   */
    uint64_t key_bytes_capacity; /* bytes key_bytes can hold       */
} kta_batch;

/* Decoded results: everything the reference's report reads (main.rs:130-170). */
typedef struct kta_result {
    uint32_t n_partitions;
    uint32_t any_records;      /* 1 if at least one record was scanned                 */
    uint32_t any_live;         /* 1 if at least one non-tombstone was scanned           */
    uint32_t count_alive_keys; /* 1 if alive_keys is valid                              */
    int64_t min_ts_sec;        /* min over records of trunc(ts_ms/1000) (metric.rs:210) */
    int64_t max_ts_sec;
    uint64_t smallest_message; /* raw state: u64::MAX if no non-tombstone (metric.rs:42) */
    uint64_t largest_message;  /* raw state: 0 if none (metric.rs:41)                   */
    uint64_t overall_count;    /* metric.rs:25 */
    uint64_t overall_size;     /* metric.rs:24 */
    uint64_t alive_keys;       /* sum_all_alive(), metric.rs:282-284                    */
    uint64_t bad_partition_records;
} kta_result;

/* ---- lifecycle ---------------------------------------------------------------- */
/* MessageMetrics::new + LogCompactionInMemoryMetrics::new (metric.rs:30-46, 267-271):
 * allocates the device counter vector, the scan workspace and — with
 * count_alive_keys — the 2^32-slot last-writer table (32 GiB of HBM). */
int kta_create(const kta_config *cfg, kta_ctx **out);
void kta_destroy(kta_ctx *ctx);
/* Message of the last failure on this context (ctx may be NULL: last kta_create failure
 * on the calling thread).  Never NULL. */
const char *kta_last_error(const kta_ctx *ctx);
int kta_abi_version(void);
/* HIP devices visible to the process (a sharded run places rank r on device r). */
int kta_device_count(int *n);
/* Zero all accumulated state (counters, extrema, alive table). */
int kta_reset(kta_ctx *ctx);

/* ---- per-message entry: what MetricHandler::handle_message binds to ----------- */
/* kafka.rs:107-109 / metric.rs:207-252 / metric.rs:289-304.  key == NULL or
 * key_len < 0 is key None; val_len < 0 is payload None.  The call copies lengths,
 * timestamp, partition and (with -c) the key bytes into the current pinned staging
 * batch and submits it when full.  Records get consecutive sequence numbers. */
int kta_handle_message(kta_ctx *ctx, int32_t partition, int64_t ts_ms, const void *key,
                       int64_t key_len, int64_t val_len);
/* Submit the partially filled staging batch, if any. */
int kta_flush(kta_ctx *ctx);
/* kta_handle_message for each of the n records of HOST columns, in order — what the reference's consume loop
 * (kafka.rs:92-135) does with the messages it polls, as a native loop: a host in a language with an expensive foreign
 * call (the ctypes mirror, a JVM) replays decoded records through the per-message entry without paying that call per
 * message, and bench.py times the entry itself with it (`boundary_per_message`).  cols: partition, key_len, val_len,
 * ts_ms; with -c also key_off and key_bytes (a key of length >= 0 is passed as a non-null pointer, -1 as key None). */
int kta_replay_messages(kta_ctx *ctx, const kta_batch *host_cols, uint64_t n);
/* Host-side cost of the per-message entry since kta_create / kta_reset: out[0] messages taken, out[1] staging batches
 * submitted by it, out[2] nanoseconds spent submitting them (the copy's and the kernels' launches), out[3] nanoseconds
 * blocked because the staging ring had wrapped onto a batch still in flight (the GPU, or the link, was the slower side). */
int kta_handle_message_stats(kta_ctx *ctx, uint64_t out[4]);
/* The next record of kta_handle_message / kta_kafka_consume gets this global sequence number (a rank of a
 * sharded run positions itself at the first record of each of its partitions' stretches). */
int kta_seek_seq(kta_ctx *ctx, uint64_t next_seq);

/* ---- batch entry: a decoder that already produces columns --------------------- */
/* Borrow the current pinned staging batch (blocks until the ring has a free one). */
int kta_batch_acquire(kta_ctx *ctx, kta_batch *out);
/* Submit the first n_records of the acquired batch: async H2D copy + kernels.
 * base_seq = global consumption index of record 0 (ignored without -c). */
int kta_batch_submit(kta_ctx *ctx, uint64_t n_records, uint64_t n_key_bytes, uint64_t base_seq);

/* ---- device-resident batches (HBM-resident topic shards, benchmarks) ----------- */
/* Column pointers in `cols` are device pointers (16-byte aligned).  Asynchronous on
 * the context's compute stream. */
int kta_submit_device(kta_ctx *ctx, const kta_batch *cols, uint64_t n_records, uint64_t base_seq);
/* Run only one of the two handlers over a device batch (profiling / benchmarks):
 * which = 1 MessageMetrics, 2 LogCompactionInMemoryMetrics, 3 both.  With both, -c, the bit set state and at most 256
 * partitions the batch is read ONCE: the first kernel of the alive-key pass also does the metrics handler's work
 * (src/kafka.rs:107-109 hands every message to every handler; same results bit for bit; kta_set_fuse(ctx, 0)
 * keeps the two passes).  kta_batch_submit and kta_handle_message submit with which = 3. */
int kta_submit_device_ex(kta_ctx *ctx, const kta_batch *cols, uint64_t n_records,
                         uint64_t base_seq, int which);
int kta_device_batch_alloc(kta_ctx *ctx, uint64_t capacity, uint64_t key_bytes_capacity,
                           int with_seq, kta_batch *out);
int kta_device_batch_free(kta_ctx *ctx, kta_batch *cols);
int kta_copy_to_device(kta_ctx *ctx, void *dst_device, const void *src_host, size_t bytes);
int kta_copy_to_host(kta_ctx *ctx, void *dst_host, const void *src_device, size_t bytes);

/* Run the context's kernels on a caller-owned HIP stream (e.g. the stream RCCL collectives are issued
 * on), so that submit -> collective -> next submit needs no host synchronisation.  NULL restores the
 * context's own compute stream — so the null (default) stream, whose handle is 0, cannot be selected:
 * pass a created stream (PyTorch: a torch.cuda.Stream, not the default stream).  The stream must belong to
 * the context's device and outlive its use. */
int kta_set_compute_stream(kta_ctx *ctx, void *hip_stream);

/* ---- results --------------------------------------------------------------------- */
/* Wait for everything submitted so far. */
int kta_sync(kta_ctx *ctx);
/* Fold everything submitted so far, count alive keys (with -c), copy the counter vector
 * to the host and decode it.  counters_out (may be NULL) receives P*7 u64.  Returns
 * KTA_ERR_BAD_PARTITION (results still filled in) if any record was out of range, and
 * KTA_ERR_TIMESTAMP_RANGE (results filled in as well, and taking precedence) if the reference would not
 * have got this far: some record's ts / 1000 lies outside [KTA_CHRONO_MIN_SEC, KTA_CHRONO_MAX_SEC].
 * Non-destructive: more batches may follow and kta_finish may be called again. */
int kta_finish(kta_ctx *ctx, kta_result *out, uint64_t *counters_out);
/* Device pointer and length (in u64) of the SNAPSHOT of the counter vector that kta_finish_device
 * takes: collectives may reduce it in place, the live accumulator is never touched, so the counters of
 * kta_finish and of further batches after an exchange stay correct.  One exception, -c after an exchange with
 * nranks > 1: the exchange merges other ranks' entries of this rank's hash range INTO this rank's table, so the
 * `alive_keys` of a later kta_finish on this context is neither the rank's own count nor the job's — read the
 * job's count from kta_exchange_result (every exchange recomputes it). */
int kta_result_vector(kta_ctx *ctx, void **device_ptr, size_t *n_u64);
/* As kta_finish but leaves the snapshot on the device (no D2H, asynchronous). */
int kta_finish_device(kta_ctx *ctx);

/* ---- multi-GPU exchange: one context = one rank = one GPU, RCCL over xGMI --------------------------- */
/* The reference is one process, one MessageMetrics, one BitSet (main.rs:77-82).  Kafka partitions shard
 * across GPUs with no data-path collective (every per-partition counter depends on its own partition's
 * records only, metric.rs:74-100); what replaces "the report reads the handlers" (main.rs:121-179) is ONE
 * exchange step.  Records of a sharded -c run carry GLOBAL sequence numbers (kta_batch.seq / base_seq).
 *   kta_comm_unique_id  on one rank; hand the 128 bytes to the others out of band
 *   kta_comm_create     every rank, collectively (ncclCommInitRank); nranks == 1 needs no id and no RCCL
 *   kta_exchange        kta_finish_device, then on the compute stream: (-c) every rank sends the table
 *                       entries it ever wrote to the owner of their hash range (rank r owns the slots
 *                       [ceil(r 2^32 / R), ceil((r+1) 2^32 / R)); one grouped ncclSend / ncclRecv), the
 *                       owner merges by last writer and counts its range; then ONE grouped launch of
 *                       all-reduce SUM over vec[0 : P*7+4] and all-reduce MAX over vec[P*7+4 : P*7+8]
 *                       What a rank sends is found through the list of slots the context wrote for the first
 *                       time (4 bytes per distinct key hash), not by sweeping the 32 GiB table; the one host
 *                       synchronisation of the step is for the sizes of the sends.  A rank that fails locally
 *                       aborts its communicator (ncclCommAbort), so that its peers get an error instead of
 *                       waiting in their collectives; the context's communicator is unusable afterwards.
 *   kta_exchange_result the decoded snapshot: after kta_exchange the whole job's result on every rank
 * RCCL is bound at run time (KTA_RCCL_LIBRARY, /opt/rocm/lib/librccl.so.1). */
#define KTA_COMM_ID_BYTES 128
int kta_comm_unique_id(uint8_t id[KTA_COMM_ID_BYTES]);
int kta_comm_create(kta_ctx *ctx, int nranks, int rank, const uint8_t id[KTA_COMM_ID_BYTES]);
int kta_comm_destroy(kta_ctx *ctx);
int kta_exchange(kta_ctx *ctx);
int kta_exchange_result(kta_ctx *ctx, kta_result *out, uint64_t *counters_out);
/* Small host-side vectors of the same job (per-partition start / end offsets of the report,
 * main.rs:155-157): all-reduce SUM (op_max 0) or MAX (1), in place, synchronous. */
int kta_comm_allreduce_i64(kta_ctx *ctx, int64_t *host_values, size_t n, int op_max);
/* Communicator facts and the alive entries the last kta_exchange sent / received (profiling). */
int kta_comm_info(kta_ctx *ctx, int *nranks, int *rank, uint64_t *entries_sent, uint64_t *entries_received);
/* Host-side decode of a (possibly all-reduced) counter vector. */
int kta_decode_vector(const uint64_t *vec, uint32_t n_partitions, int count_alive_keys,
                      kta_result *out, uint64_t *counters_out);
/* Host-side merge of two counter vectors (acc <- acc (+) other) with the per-field
 * reduction operator (SUM over the prefix, signed MAX over the last four words) — exactly
 * the reduction the two collectives implement. */
int kta_merge_vectors(uint64_t *acc, const uint64_t *other, uint32_t n_partitions);

/* Analytics of everything submitted so far (context created with KTA_FLAG_ANALYTICS).  The four
 * per-partition arrays (length P, any may be NULL) use the reference's conventions: seconds =
 * trunc(ms / 1000) with a raw timestamp of -1 counted as 0; a partition without records reports
 * min_ts_sec = INT64_MAX / max_ts_sec = INT64_MIN; without non-tombstones smallest = UINT64_MAX,
 * largest = 0. */
int kta_get_analytics(kta_ctx *ctx, kta_analytics *out, int64_t *part_min_ts_sec, int64_t *part_max_ts_sec,
                      uint64_t *part_smallest, uint64_t *part_largest);
/* Device pointer / length (u64) of the analytics vector: [2 x 34 histogram (SUM)] then per
 * partition [~min ts_ms, max ts_ms, ~smallest, largest] (signed MAX) — reducible across GPUs like
 * the counter vector. */
int kta_analytics_vector(kta_ctx *ctx, void **device_ptr, size_t *n_u64);

/* ---- alive-key table access (tests, multi-GPU merge) ------------------------------ */
/* Export the alive set as a 2^32-bit little-endian bitmap (bit h%32 of u32 word h/32;
 * 512 MiB) into host memory — the same layout as BitSet's storage (metric.rs:263). */
int kta_export_alive_bitmap(kta_ctx *ctx, void *dst_host_512MiB);
/* Device pointer of the last-writer table: u64[2^32], entry = ((seq+1)<<1)|alive, 0 = never
 * written.  Element-wise MAX across GPUs merges shards exactly (then call
 * kta_alive_table_modified).  sum_all_alive is normally kept as a running count by the update
 * kernel (returning atomicMax: +flag(new) - flag(old) whenever an entry is replaced), so
 * kta_finish does not scan the table. */
int kta_alive_table(kta_ctx *ctx, void **device_ptr, size_t *n_u64);
/* Compact exchange of the table between partition-sharded GPUs.  Export: the entries ever written
 * (value != 0) as device arrays slot u32[n] / value u64[n] (owned by the context, valid until the
 * next export); at most one per distinct key hash the shard has seen, i.e. 12 bytes per key instead
 * of the 32 GiB table.  Import: table[slot] = max(table[slot], value) for foreign entries (device
 * pointers), keeping the running alive count exact.  Importing every other shard's export into one
 * context reproduces the global last-writer state. */
int kta_alive_export_entries(kta_ctx *ctx, void **d_slots, void **d_vals, uint64_t *n);
int kta_alive_import_entries(kta_ctx *ctx, const void *d_slots, const void *d_vals, uint64_t n);
/* Alive keys whose hash slot lies in [slot_lo, slot_hi) (0 <= lo <= hi <= 2^32): the share of the owner
 * of a hash range after the hash-range exchange of a multi-GPU run (distributed.py,
 * exchange_alive_by_hash_range; SURVEY section 8(e) option ii).  Synchronous. */
int kta_alive_count_range(kta_ctx *ctx, uint64_t slot_lo, uint64_t slot_hi, uint64_t *count);

/* Tell the context that the table was changed behind its back (e.g. merged with other GPUs' tables
 * by an all-reduce MAX): the running alive count is dropped and the next kta_finish recounts by
 * scanning the table. */
int kta_alive_table_modified(kta_ctx *ctx);
/* Hash `n` keys on the device with the reference's FNV variant (fnv32.rs:92-101). */
int kta_fnv32_device(kta_ctx *ctx, const uint8_t *key_bytes_host, const uint32_t *key_off_host,
                     const int32_t *key_len_host, uint64_t n, uint64_t n_key_bytes,
                     uint32_t *hash_out_host);

/* ---- report -------------------------------------------------------------------------- */
/* Render what the reference prints after the scan (src/main.rs:123-178) from a counter vector:
 * the text block, chrono's `DateTime<Utc>` Display, `{:.4}` of the f32 dirty ratio and the
 * prettytable.  `now_*` stands in for Utc::now() at MessageMetrics::new (metric.rs:39);
 * start/end offsets may be NULL (0 / per-partition record count).  *out_len receives the full
 * length; the text is truncated to out_cap-1 bytes + NUL.  Returns KTA_ERR_DIV_BY_ZERO where
 * the reference panics (a partition with key bytes but no alive record, metric.rs:135) and
 * KTA_ERR_TIMESTAMP_RANGE, without a text, for a vector the reference could not have produced (above). */
int kta_render_report(const char *topic, uint64_t duration_secs, const uint64_t *vec,
                      uint32_t n_partitions, int count_alive_keys, int64_t now_sec, uint32_t now_ns,
                      const int64_t *start_offsets, const int64_t *end_offsets, char *out,
                      size_t out_cap, size_t *out_len);

/* ---- profiling hooks --------------------------------------------------------------- */
/* With kta_set_timing(ctx, 1) every kernel launch is bracketed by a pair of HIP events recorded
 * on the compute stream (no host synchronisation while recording).  kta_kernel_time_stats waits
 * for the stream and returns, per kernel kind ([0] metrics scan, [1] partial fold, [2] alive-key
 * update), the average duration in ms and the number of launches since the previous call
 * (avg -1 when none). */
int kta_set_timing(kta_ctx *ctx, int enable);
int kta_kernel_time_stats(kta_ctx *ctx, float avg_ms[3], uint64_t launches[3]);
/* Launch-geometry knobs (0 = default): scan workgroups, scan kernel flavour (16 = non-temporal loads),
 * alive workgroups, alive kernel: 0 plain atomicMax, 1 returning atomicMax + running alive count, 2 the
 * same walked backwards with a pre-read that skips superseded records, 3 (default) / 4 the partitioned
 * pass (hash + partition by the hash's top 10 / 9 bits, then per-bucket merge in LDS) for batches of 2^21
 * records and more without a seq column — kernel 2 otherwise, and for the batches that follow one whose
 * keys were mostly unique; 13 / 14 the partitioned pass for every batch (tests); 8 / 9 ablation halves. */
int kta_set_tuning(kta_ctx *ctx, int scan_workgroups, int scan_variant, int alive_workgroups,
                   int alive_variant);
/* Both handlers of a batch (kafka.rs:107-109: every handler for every message) as ONE pass over it where that is
 * possible (bit set state, at most 256 partitions, no analytics): on by default; 0 = always two passes (scan, then
 * the alive-key pass).  Results are bit-identical either way.  The environment variable KTA_NO_FUSE=1, read by
 * kta_create, only sets the initial value.  In the fused pass all kernel time is booked on timer kind 2 (the
 * alive-key update) and kinds 0 / 1 see only the fold: kta_kernel_time_stats returns launches[0] == 0 there. */
int kta_set_fuse(kta_ctx *ctx, int enable);
/* What the partitioned alive-key pass did since kta_create / kta_reset:
 * out[0] the most records one launch pair takes (larger batches are applied piece after piece), out[1] launch pairs,
 * out[2] of them with both handlers in the one pass, out[3] of them whose metrics handler ran as a scan although the
 * batch began fused, out[4] buckets handed to the fallback kernel (hot keys that overflow their segments; exact, slow) —
 * counted on the device by every launch pair; this call waits for the context's compute stream to read the word —,
 * out[5] the fuse switch. */
int kta_alive_pass_info(kta_ctx *ctx, uint64_t out[6]);

#ifdef __cplusplus
}
#endif
#endif /* KTA_HIP_H */
