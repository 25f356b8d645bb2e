/*
 * kta_oracle.h — CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 *
 * A plain-C restatement of the per-record metric-accumulation hot path of
 * xenji/kafka-topic-analyzer (reference mounted at /root/reference):
 *     src/metric.rs:29-305   MessageMetrics + LogCompactionInMemoryMetrics
 *     src/fnv32.rs:76-101    FnvHasher (32 bit, multiplier == offset basis)
 *     src/kafka.rs:107-109   handler dispatch order
 *
 * PARITY STATUS: **parity unpinned for handle_message; accessors and report pinned by the one output of the
 * reference that the reference ships.**  The reference has no tests, golden vectors or fixtures for this path
 *  and its Rust toolchain is not available in the build image, so the reference itself cannot be
 * executed.  What pins the oracle:
 *   (a) REFERENCE-PRODUCED: /root/reference/demo_output.png (README.md:28), a real run against a 10-partition
 *       topic, transcribed into tests/golden/reference_demo_output.json (the transcription checks itself against
 *       the screenshot's own sums).  tests/test_reference_demo.py asserts, through this oracle, the Python oracle
 *       and the product: floor division of the three averages by `alive` (metric.rs:132-157), the `0.0000` dirty
 *       ratio (metric.rs:159-167), Msg/s and Topic Size (main.rs:130,137), P-Bytes, the DateTime display and the
 *       table geometry (main.rs:125-178) — rows a8 and f1 of SURVEY §8.  The topic behind the screenshot is gone:
 *       it cannot pin handle_message (rows a2-a7, a9-a13);
 *   (b) known-answer vectors derived by hand from the source (tests/golden/);
 *   (c) agreement with a second, independently written pure-Python restatement (oracle/oracle_py.py) on
 *       randomised inputs.
 * oracle/ref_gen/ compiles the reference's OWN metric.rs / fnv32.rs for a box with cargo; until its output is
 * committed, (b) and (c) are all that stands behind the handlers.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may link,
 * import or call anything in oracle/.  The product (libkta_hip.so and everything
 * under kafka_topic_analyzer_amd/) never does.
 */
#ifndef KTA_ORACLE_H
#define KTA_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- fnv32.rs ---------------------------------------------------------- */
/* fnv32.rs:79-81 (default), :92-101 (write), :87-89 (finish) */
uint32_t kto_fnv32(const uint8_t *bytes, size_t len);
/* metric.rs:256-260  fn fnv1a(bytes) -> usize  (u32 zero-extended) */
uint64_t kto_fnv1a(const uint8_t *bytes, size_t len);

/* ---- metric.rs: MessageMetrics ---------------------------------------- */
typedef struct kto_metrics kto_metrics;

/* metric.rs:30-46.  `now_sec/now_ns` stand in for Utc::now() (metric.rs:39),
 * which is non-deterministic in the reference; the caller supplies it. */
kto_metrics *kto_metrics_new(int64_t now_sec, uint32_t now_ns);
void kto_metrics_free(kto_metrics *m);

/* metric.rs:207-252.  key_len / val_len: -1 == None, >=0 == Some(len).
 * ts_raw_ms: the raw rdkafka timestamp; ts_available==0 or ts_raw_ms==-1 is
 * Timestamp::to_millis() == None -> unwrap_or(0) (metric.rs:209). */
void kto_metrics_handle_message(kto_metrics *m, int32_t partition, int64_t ts_raw_ms,
                                int ts_available, int64_t key_len, int64_t val_len);

/* metric.rs:210 (and kafka.rs:104): NaiveDateTime::from_timestamp(timestamp / 1000, 0) panics ("invalid or
 * out-of-range datetime") outside chrono 0.4.19's NaiveDate years [i32::MIN >> 13, i32::MAX >> 13] =
 * [-262144, 262143] ([3P], restated: the crate's source is not under /root/reference — parity unpinned):
 *   -262144-01-01 00:00:00 = -8 334 632 851 200 s,  262143-12-31 23:59:59 = 8 210 298 412 799 s.
 * A panicked oracle has stopped: the record that did it and everything after it is not counted. */
#define KTO_CHRONO_MIN_SEC (-8334632851200LL)
#define KTO_CHRONO_MAX_SEC (8210298412799LL)
int kto_metrics_panicked(const kto_metrics *m);

/* metric.rs:104-130 */
uint64_t kto_total(const kto_metrics *m, int32_t p);
uint64_t kto_tombstones(const kto_metrics *m, int32_t p);
uint64_t kto_alive(const kto_metrics *m, int32_t p);
uint64_t kto_key_null(const kto_metrics *m, int32_t p);
uint64_t kto_key_non_null(const kto_metrics *m, int32_t p);
uint64_t kto_key_size_sum(const kto_metrics *m, int32_t p);
uint64_t kto_value_size_sum(const kto_metrics *m, int32_t p);
/* metric.rs:132-157.  Return 0 on success and write *out; return -1 where the
 * reference panics with "attempt to divide by zero" (sum > 0 && alive == 0). */
int kto_key_size_avg(const kto_metrics *m, int32_t p, uint64_t *out);
int kto_value_size_avg(const kto_metrics *m, int32_t p, uint64_t *out);
int kto_message_size_avg(const kto_metrics *m, int32_t p, uint64_t *out);
/* metric.rs:159-167 (f32 arithmetic, two roundings) */
float kto_dirty_ratio(const kto_metrics *m, int32_t p);
/* metric.rs:169-175: DateTime<Utc> as (seconds, nanoseconds) */
void kto_latest_message(const kto_metrics *m, int64_t *sec, uint32_t *ns);
void kto_earliest_message(const kto_metrics *m, int64_t *sec, uint32_t *ns);
/* metric.rs:177-195 */
uint64_t kto_smallest_message(const kto_metrics *m);
uint64_t kto_largest_message(const kto_metrics *m);
uint64_t kto_overall_count(const kto_metrics *m);
uint64_t kto_overall_size(const kto_metrics *m);
/* number of distinct partitions seen in total_messages + their ids (sorted) */
size_t kto_partitions(const kto_metrics *m, int32_t *out, size_t cap);

/* ---- metric.rs: LogCompactionInMemoryMetrics -------------------------- */
typedef struct kto_logcompaction kto_logcompaction;

kto_logcompaction *kto_lc_new(void);                 /* metric.rs:267-271 */
void kto_lc_free(kto_logcompaction *lc);
void kto_lc_mark_key_alive(kto_logcompaction *lc, const uint8_t *key, size_t len); /* :273-276 */
void kto_lc_mark_key_dead(kto_logcompaction *lc, const uint8_t *key, size_t len);  /* :278-280 */
uint64_t kto_lc_sum_all_alive(const kto_logcompaction *lc);                        /* :282-284 */
/* metric.rs:289-304 */
void kto_lc_handle_message(kto_logcompaction *lc, const uint8_t *key, int64_t key_len,
                           int64_t val_len);
/* test helpers on the BitSet: membership and number of allocated bits */
int kto_lc_contains(const kto_logcompaction *lc, uint32_t slot);
uint64_t kto_lc_nbits(const kto_logcompaction *lc);
/* copy the set as a little-endian bitmap of u32 words (bit h%32 of word h/32),
 * zero-padded/truncated to n_words words */
void kto_lc_export_words(const kto_logcompaction *lc, uint32_t *dst, uint64_t n_words);

/* ---- kafka.rs:107-109: run both handlers over a struct-of-arrays batch --- */
/* Record i is (part[i], ts_ms[i], key = key_bytes[key_off[i] .. +key_len[i]] or
 * None when key_len[i] < 0, payload length val_len[i] or None when < 0).
 * Records are consumed in index order.  `lc` may be NULL (no -c flag);
 * key_off/key_bytes may be NULL when lc is NULL. */
void kto_run_soa(kto_metrics *m, kto_logcompaction *lc, uint64_t n, const int32_t *part,
                 const int32_t *key_len, const int32_t *val_len, const int64_t *ts_ms,
                 const uint32_t *key_off, const uint8_t *key_bytes);
/* TEST HARNESS (no reference counterpart): the LogCompaction handler over the records whose key hashes into
 * [slot_lo, slot_hi) — K instances over disjoint slot ranges hold between them what one instance holds (kta_oracle.c). */
void kto_lc_run_soa_slot_range(kto_logcompaction *lc, uint64_t n, const int32_t *key_len, const int32_t *val_len,
                               const uint32_t *key_off, const uint8_t *key_bytes, const uint32_t *slots /* or NULL */,
                               uint64_t slot_lo, uint64_t slot_hi);
/* TEST HARNESS: fnv1a of the keys of the records [first, first + n) into out[first ...] (0 for key None) */
void kto_fnv1a_soa(uint64_t first, uint64_t n, const int32_t *key_len, const uint32_t *key_off, const uint8_t *key_bytes,
                   uint32_t *out);

/* dense export for comparisons: out[p*7 + c], c in reference field order
 * (total, tombstones, alive, key_null, key_non_null, key_size_sum, value_size_sum) */
void kto_export_counters(const kto_metrics *m, int32_t n_partitions, uint64_t *out);

#ifdef __cplusplus
}
#endif

/* ---- additive analytics: NO reference counterpart ------------------------------------------
 * The reference keeps only sums and global extrema (metric.rs:18-23).  The product can
 * additionally report log2 size histograms and per-partition extrema (KTA_FLAG_ANALYTICS,
 * DESIGN.md §3.6); this restates THAT definition on the CPU so the kernel can be checked.
 * bucket(None) = 0, bucket(len 0) = 1, bucket(len) = 2 + floor(log2(len)). */
#ifdef __cplusplus
extern "C" {
#endif
typedef struct kto_analytics kto_analytics;
kto_analytics *kto_analytics_new(int32_t n_partitions);
void kto_analytics_free(kto_analytics *a);
void kto_analytics_run_soa(kto_analytics *a, uint64_t n, const int32_t *part, const int32_t *key_len,
                           const int32_t *val_len, const int64_t *ts_ms);
/* out: key_hist[34], val_hist[34]; per partition min/max seconds (INT64_MAX / INT64_MIN if none),
 * smallest/largest non-tombstone (UINT64_MAX / 0 if none) */
void kto_analytics_export(const kto_analytics *a, uint64_t *key_hist, uint64_t *val_hist, int64_t *min_ts_sec,
                          int64_t *max_ts_sec, uint64_t *smallest, uint64_t *largest);
#ifdef __cplusplus
}
#endif
#endif /* KTA_ORACLE_H */
