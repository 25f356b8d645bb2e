/*
 * kta_oracle.c — CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 * See kta_oracle.h for scope and the "parity unpinned" statement.
 *
 * Every function cites the reference lines (under /root/reference) it restates.
 * The structure deliberately stays close to the reference's *data structures*
 * (seven separate hash maps keyed by partition, one growable bit set) so that
 * the 1-thread timing of this file is a fair stand-in for the Rust original
 * when bench.py reports it as cpu_baseline {"kind": "port"}.
 */
#include "kta_oracle.h"

#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------ */
/* fnv32.rs                                                                   */
/* ------------------------------------------------------------------------ */

/* fnv32.rs:79-81: the initial state is 0x811c9dc5.
 * fnv32.rs:95-98: for each byte: hash ^= byte; hash = hash.wrapping_mul(0x811c9dc5).
 * NB the multiplier is the offset basis again, NOT the FNV-32 prime 0x01000193. */
uint32_t kto_fnv32(const uint8_t *bytes, size_t len)
{
    uint32_t hash = 0x811c9dc5u;
    for (size_t i = 0; i < len; i++) {
        hash = hash ^ (uint32_t)bytes[i];
        hash = hash * 0x811c9dc5u; /* wrapping_mul on u32 */
    }
    return hash;
}

/* metric.rs:256-260 */
uint64_t kto_fnv1a(const uint8_t *bytes, size_t len) { return (uint64_t)kto_fnv32(bytes, len); }

/* ------------------------------------------------------------------------ */
/* PartitionedCounterBucket = HashMap<i32, u64>   (metric.rs:8-9)             */
/* std's HashMap hashes with SipHash-1-3; we do the same (fixed zero key) over */
/* an open-addressed table so the per-record cost is of the same kind.        */
/* ------------------------------------------------------------------------ */

#define ROTL64(x, b) (((x) << (b)) | ((x) >> (64 - (b))))
#define SIPROUND(v0, v1, v2, v3)                                                           \
    do {                                                                                   \
        v0 += v1; v1 = ROTL64(v1, 13); v1 ^= v0; v0 = ROTL64(v0, 32);                      \
        v2 += v3; v3 = ROTL64(v3, 16); v3 ^= v2;                                           \
        v0 += v3; v3 = ROTL64(v3, 21); v3 ^= v0;                                           \
        v2 += v1; v1 = ROTL64(v1, 17); v1 ^= v2; v2 = ROTL64(v2, 32);                      \
    } while (0)

static uint64_t siphash13_i32(int32_t key)
{
    /* k0 = k1 = 0; message = 4 little-endian bytes of the key */
    uint64_t v0 = 0x736f6d6570736575ull, v1 = 0x646f72616e646f6dull;
    uint64_t v2 = 0x6c7967656e657261ull, v3 = 0x7465646279746573ull;
    uint64_t b = ((uint64_t)4 << 56) | (uint64_t)(uint32_t)key;
    v3 ^= b;
    SIPROUND(v0, v1, v2, v3); /* c = 1 */
    v0 ^= b;
    v2 ^= 0xff;
    SIPROUND(v0, v1, v2, v3); /* d = 3 */
    SIPROUND(v0, v1, v2, v3);
    SIPROUND(v0, v1, v2, v3);
    return v0 ^ v1 ^ v2 ^ v3;
}

typedef struct {
    int32_t *keys;
    uint64_t *vals;
    uint8_t *used;
    size_t cap; /* power of two */
    size_t len;
} pmap;

static void pmap_init(pmap *m)
{
    m->cap = 16;
    m->len = 0;
    m->keys = (int32_t *)calloc(m->cap, sizeof(int32_t));
    m->vals = (uint64_t *)calloc(m->cap, sizeof(uint64_t));
    m->used = (uint8_t *)calloc(m->cap, 1);
}

static void pmap_free(pmap *m)
{
    free(m->keys);
    free(m->vals);
    free(m->used);
}

static uint64_t *pmap_entry_or_insert0(pmap *m, int32_t key);

static void pmap_grow(pmap *m)
{
    pmap n;
    n.cap = m->cap * 2;
    n.len = 0;
    n.keys = (int32_t *)calloc(n.cap, sizeof(int32_t));
    n.vals = (uint64_t *)calloc(n.cap, sizeof(uint64_t));
    n.used = (uint8_t *)calloc(n.cap, 1);
    for (size_t i = 0; i < m->cap; i++)
        if (m->used[i]) *pmap_entry_or_insert0(&n, m->keys[i]) = m->vals[i];
    pmap_free(m);
    *m = n;
}

/* `*map.entry(p).or_insert(0u64)`  (metric.rs:75,79,83,87,91,95,99) */
static uint64_t *pmap_entry_or_insert0(pmap *m, int32_t key)
{
    if ((m->len + 1) * 8 > m->cap * 7) pmap_grow(m);
    size_t i = (size_t)siphash13_i32(key) & (m->cap - 1);
    while (m->used[i]) {
        if (m->keys[i] == key) return &m->vals[i];
        i = (i + 1) & (m->cap - 1);
    }
    m->used[i] = 1;
    m->keys[i] = key;
    m->vals[i] = 0;
    m->len++;
    return &m->vals[i];
}

/* metric.rs:198-203  fn metric(): Some(v) => *v, None => 0 */
static uint64_t pmap_get(const pmap *m, int32_t key)
{
    size_t i = (size_t)siphash13_i32(key) & (m->cap - 1);
    while (m->used[i]) {
        if (m->keys[i] == key) return m->vals[i];
        i = (i + 1) & (m->cap - 1);
    }
    return 0;
}

/* ------------------------------------------------------------------------ */
/* MessageMetrics   (metric.rs:11-26)                                         */
/* ------------------------------------------------------------------------ */

typedef struct {
    int64_t sec;
    uint32_t ns;
} kto_datetime; /* DateTime<Utc>: ordered by (sec, ns) */

struct kto_metrics {
    pmap total_messages, tombstones, alive, key_null, key_non_null, key_size_sum, value_size_sum;
    kto_datetime earliest_message, latest_message;
    uint64_t smallest_message, largest_message, overall_size, overall_count;
    int panicked; /* a record's timestamp was outside chrono's range: the process is gone (see handle_message) */
};

static int dt_gt(kto_datetime a, kto_datetime b)
{
    return a.sec > b.sec || (a.sec == b.sec && a.ns > b.ns);
}
static int dt_lt(kto_datetime a, kto_datetime b) { return dt_gt(b, a); }

/* metric.rs:30-46 */
kto_metrics *kto_metrics_new(int64_t now_sec, uint32_t now_ns)
{
    kto_metrics *m = (kto_metrics *)calloc(1, sizeof(*m));
    pmap_init(&m->total_messages);
    pmap_init(&m->tombstones);
    pmap_init(&m->alive);
    pmap_init(&m->key_null);
    pmap_init(&m->key_non_null);
    pmap_init(&m->key_size_sum);
    pmap_init(&m->value_size_sum);
    m->earliest_message.sec = now_sec; /* Utc::now()             :39 */
    m->earliest_message.ns = now_ns;
    m->latest_message.sec = 0;         /* from_timestamp(0, 0)   :40 */
    m->latest_message.ns = 0;
    m->largest_message = 0;            /* :41 */
    m->smallest_message = UINT64_MAX;  /* :42 */
    m->overall_size = 0;               /* :43 */
    m->overall_count = 0;              /* :44 */
    return m;
}

void kto_metrics_free(kto_metrics *m)
{
    if (!m) return;
    pmap_free(&m->total_messages);
    pmap_free(&m->tombstones);
    pmap_free(&m->alive);
    pmap_free(&m->key_null);
    pmap_free(&m->key_non_null);
    pmap_free(&m->key_size_sum);
    pmap_free(&m->value_size_sum);
    free(m);
}

/* metric.rs:56-63 */
static void cmp_and_set_message_size(kto_metrics *m, uint64_t size)
{
    if (m->largest_message < size) m->largest_message = size;
    if (m->smallest_message > size) m->smallest_message = size;
}

/* metric.rs:65-72 */
static void cmp_and_set_message_timestamp(kto_metrics *m, kto_datetime cmp)
{
    if (dt_gt(m->earliest_message, cmp)) m->earliest_message = cmp;
    if (dt_lt(m->latest_message, cmp)) m->latest_message = cmp;
}

/* metric.rs:207-252 */
void kto_metrics_handle_message(kto_metrics *m, int32_t partition, int64_t ts_raw_ms,
                                int ts_available, int64_t key_len, int64_t val_len)
{
    /* :209  m.timestamp().to_millis().unwrap_or(0); rdkafka 0.25 Timestamp::to_millis
     * yields None for NotAvailable and for a CreateTime/LogAppendTime of -1. */
    int64_t timestamp = (!ts_available || ts_raw_ms == -1) ? 0 : ts_raw_ms;
    /* :210-211  from_timestamp(timestamp / 1000, 0): i64 division truncates toward zero */
    kto_datetime timestamp_dt;
    timestamp_dt.sec = timestamp / 1000;
    timestamp_dt.ns = 0;
    /* [3P chrono 0.4.19, Cargo.lock:84-85] NaiveDateTime::from_timestamp = from_timestamp_opt(secs, 0)
     * .expect("invalid or out-of-range datetime"): the day number has to fit NaiveDate, whose years run from
     * MIN_YEAR = i32::MIN >> 13 = -262144 to MAX_YEAR = i32::MAX >> 13 = 262143.  Outside
     * [-262144-01-01 00:00:00, 262143-12-31 23:59:59] the reference panics here — before anything is counted
     * (and, in the running program, already at the same call in kafka.rs:104, before any handler): the process
     * ends, no later record is looked at. */
    if (m->panicked) return;
    if (timestamp_dt.sec < KTO_CHRONO_MIN_SEC || timestamp_dt.sec > KTO_CHRONO_MAX_SEC) {
        m->panicked = 1;
        return;
    }
    uint64_t message_size = 0; /* :212 */
    int empty_value = 0;       /* :213 */

    m->overall_count += 1;                                      /* :215 -> :52-54 */
    *pmap_entry_or_insert0(&m->total_messages, partition) += 1; /* :216 -> :74-76 */

    if (key_len >= 0) { /* Some(k)  :219-226 */
        *pmap_entry_or_insert0(&m->key_non_null, partition) += 1;
        uint64_t k_len = (uint64_t)key_len;
        message_size += k_len;
        *pmap_entry_or_insert0(&m->key_size_sum, partition) += k_len;
        m->overall_size += k_len;
    } else { /* None  :227-230 */
        *pmap_entry_or_insert0(&m->key_null, partition) += 1;
    }

    if (val_len >= 0) { /* Some(v)  :234-240 */
        uint64_t v_len = (uint64_t)val_len;
        message_size += v_len;
        *pmap_entry_or_insert0(&m->value_size_sum, partition) += v_len;
        m->overall_size += v_len;
        *pmap_entry_or_insert0(&m->alive, partition) += 1;
    } else { /* None  :241-244 */
        empty_value = 1;
        *pmap_entry_or_insert0(&m->tombstones, partition) += 1;
    }

    cmp_and_set_message_timestamp(m, timestamp_dt); /* :247 */

    if (!empty_value) cmp_and_set_message_size(m, message_size); /* :249-251 */
}

int kto_metrics_panicked(const kto_metrics *m) { return m->panicked; }

uint64_t kto_total(const kto_metrics *m, int32_t p) { return pmap_get(&m->total_messages, p); }
uint64_t kto_tombstones(const kto_metrics *m, int32_t p) { return pmap_get(&m->tombstones, p); }
uint64_t kto_alive(const kto_metrics *m, int32_t p) { return pmap_get(&m->alive, p); }
uint64_t kto_key_null(const kto_metrics *m, int32_t p) { return pmap_get(&m->key_null, p); }
uint64_t kto_key_non_null(const kto_metrics *m, int32_t p) { return pmap_get(&m->key_non_null, p); }
uint64_t kto_key_size_sum(const kto_metrics *m, int32_t p) { return pmap_get(&m->key_size_sum, p); }
uint64_t kto_value_size_sum(const kto_metrics *m, int32_t p)
{
    return pmap_get(&m->value_size_sum, p);
}

/* metric.rs:132-139 */
int kto_key_size_avg(const kto_metrics *m, int32_t p, uint64_t *out)
{
    uint64_t s = kto_key_size_sum(m, p);
    if (s > 0) {
        uint64_t a = kto_alive(m, p);
        if (a == 0) return -1; /* Rust: panic "attempt to divide by zero" */
        *out = s / a;
    } else {
        *out = 0;
    }
    return 0;
}

/* metric.rs:141-148 */
int kto_value_size_avg(const kto_metrics *m, int32_t p, uint64_t *out)
{
    uint64_t s = kto_value_size_sum(m, p);
    if (s > 0) {
        uint64_t a = kto_alive(m, p);
        if (a == 0) return -1;
        *out = s / a;
    } else {
        *out = 0;
    }
    return 0;
}

/* metric.rs:150-157 */
int kto_message_size_avg(const kto_metrics *m, int32_t p, uint64_t *out)
{
    uint64_t s = kto_key_size_sum(m, p) + kto_value_size_sum(m, p);
    if (s > 0) {
        uint64_t a = kto_alive(m, p);
        if (a == 0) return -1;
        *out = s / a;
    } else {
        *out = 0;
    }
    return 0;
}

/* metric.rs:159-167: tombstones as f32 / (total_messages as f32 / 100.0f32).
 * volatile forces each f32 rounding to happen (no double-precision contraction). */
float kto_dirty_ratio(const kto_metrics *m, int32_t p)
{
    uint64_t total_messages = kto_total(m, p);
    uint64_t tombstones = kto_tombstones(m, p);
    if (total_messages > 0 && tombstones > 0) {
        volatile float t = (float)tombstones;
        volatile float tm = (float)total_messages;
        volatile float d = tm / 100.0f;
        volatile float r = t / d;
        return r;
    }
    return 0.0f;
}

void kto_latest_message(const kto_metrics *m, int64_t *sec, uint32_t *ns)
{
    *sec = m->latest_message.sec;
    *ns = m->latest_message.ns;
}

void kto_earliest_message(const kto_metrics *m, int64_t *sec, uint32_t *ns)
{
    *sec = m->earliest_message.sec;
    *ns = m->earliest_message.ns;
}

/* metric.rs:177-183 */
uint64_t kto_smallest_message(const kto_metrics *m)
{
    return m->smallest_message == UINT64_MAX ? 0 : m->smallest_message;
}
uint64_t kto_largest_message(const kto_metrics *m) { return m->largest_message; }
uint64_t kto_overall_count(const kto_metrics *m) { return m->overall_count; }
uint64_t kto_overall_size(const kto_metrics *m) { return m->overall_size; }

static int cmp_i32(const void *a, const void *b)
{
    int32_t x = *(const int32_t *)a, y = *(const int32_t *)b;
    return (x > y) - (x < y);
}

size_t kto_partitions(const kto_metrics *m, int32_t *out, size_t cap)
{
    size_t n = 0;
    for (size_t i = 0; i < m->total_messages.cap; i++)
        if (m->total_messages.used[i]) {
            if (n < cap) out[n] = m->total_messages.keys[i];
            n++;
        }
    qsort(out, n < cap ? n : cap, sizeof(int32_t), cmp_i32);
    return n;
}

void kto_export_counters(const kto_metrics *m, int32_t n_partitions, uint64_t *out)
{
    for (int32_t p = 0; p < n_partitions; p++) {
        out[(size_t)p * 7 + 0] = kto_total(m, p);
        out[(size_t)p * 7 + 1] = kto_tombstones(m, p);
        out[(size_t)p * 7 + 2] = kto_alive(m, p);
        out[(size_t)p * 7 + 3] = kto_key_null(m, p);
        out[(size_t)p * 7 + 4] = kto_key_non_null(m, p);
        out[(size_t)p * 7 + 5] = kto_key_size_sum(m, p);
        out[(size_t)p * 7 + 6] = kto_value_size_sum(m, p);
    }
}

/* ------------------------------------------------------------------------ */
/* LogCompactionInMemoryMetrics  (metric.rs:262-305)                          */
/* store: Box<BitSet>  — bit-set 0.5.2 over bit-vec 0.6.3 (Cargo.lock:45-57); */
/* third-party source is not in the reference tree, semantics restated from   */
/* the published crates:                                                     */
/*   insert(v): if contains(v) return; grow bit_vec to v+1 bits; set bit      */
/*   remove(v): if !contains(v) return; clear bit                             */
/*   len():     sum of count_ones over the u32 blocks                         */
/* ------------------------------------------------------------------------ */

struct kto_logcompaction {
    uint32_t *blocks;
    uint64_t nbits;       /* BitVec::len() */
    uint64_t cap_blocks;  /* Vec capacity */
};

kto_logcompaction *kto_lc_new(void)
{
    return (kto_logcompaction *)calloc(1, sizeof(kto_logcompaction));
}

void kto_lc_free(kto_logcompaction *lc)
{
    if (!lc) return;
    free(lc->blocks);
    free(lc);
}

int kto_lc_contains(const kto_logcompaction *lc, uint32_t slot)
{
    if ((uint64_t)slot >= lc->nbits) return 0;
    return (lc->blocks[slot >> 5] >> (slot & 31)) & 1u;
}

uint64_t kto_lc_nbits(const kto_logcompaction *lc) { return lc->nbits; }

static void bitvec_grow_to(kto_logcompaction *lc, uint64_t nbits)
{
    uint64_t need = (nbits + 31) / 32;
    if (need > lc->cap_blocks) {
        uint64_t cap = lc->cap_blocks ? lc->cap_blocks : 4;
        while (cap < need) cap *= 2;
        lc->blocks = (uint32_t *)realloc(lc->blocks, cap * sizeof(uint32_t));
        memset(lc->blocks + lc->cap_blocks, 0, (cap - lc->cap_blocks) * sizeof(uint32_t));
        lc->cap_blocks = cap;
    }
    lc->nbits = nbits;
}

/* metric.rs:273-276 */
void kto_lc_mark_key_alive(kto_logcompaction *lc, const uint8_t *key, size_t len)
{
    uint64_t k = kto_fnv1a(key, len);
    if (kto_lc_contains(lc, (uint32_t)k)) return;
    if (k >= lc->nbits) bitvec_grow_to(lc, k + 1);
    lc->blocks[k >> 5] |= (1u << (k & 31));
}

/* metric.rs:278-280 */
void kto_lc_mark_key_dead(kto_logcompaction *lc, const uint8_t *key, size_t len)
{
    uint64_t k = kto_fnv1a(key, len);
    if (!kto_lc_contains(lc, (uint32_t)k)) return;
    lc->blocks[k >> 5] &= ~(1u << (k & 31));
}

/* metric.rs:282-284 */
uint64_t kto_lc_sum_all_alive(const kto_logcompaction *lc)
{
    uint64_t n = 0, nb = (lc->nbits + 31) / 32;
    for (uint64_t i = 0; i < nb; i++) n += (uint64_t)__builtin_popcount(lc->blocks[i]);
    return n;
}

/* metric.rs:289-304 */
void kto_lc_handle_message(kto_logcompaction *lc, const uint8_t *key, int64_t key_len,
                           int64_t val_len)
{
    if (key_len >= 0) {                                   /* Some(k) :292 */
        if (val_len >= 0)                                 /* Some(_) :294 */
            kto_lc_mark_key_alive(lc, key, (size_t)key_len);
        else                                              /* None    :297 */
            kto_lc_mark_key_dead(lc, key, (size_t)key_len);
    }                                                     /* None => {} :302 */
}

void kto_lc_export_words(const kto_logcompaction *lc, uint32_t *dst, uint64_t n_words)
{
    uint64_t nb = (lc->nbits + 31) / 32;
    uint64_t c = nb < n_words ? nb : n_words;
    if (c) memcpy(dst, lc->blocks, c * sizeof(uint32_t));
    if (n_words > c) memset(dst + c, 0, (n_words - c) * sizeof(uint32_t));
}

/* ------------------------------------------------------------------------ */
/* kafka.rs:107-109: for mh in metric_handlers { mh.handle_message(&m) }       */
/* registration order main.rs:108-115: MessageMetrics first, then the          */
/* log-compaction handler.                                                    */
/* ------------------------------------------------------------------------ */
void kto_run_soa(kto_metrics *m, kto_logcompaction *lc, uint64_t n, const int32_t *part,
                 const int32_t *key_len, const int32_t *val_len, const int64_t *ts_ms,
                 const uint32_t *key_off, const uint8_t *key_bytes)
{
    for (uint64_t i = 0; i < n; i++) {
        if (m) {
            kto_metrics_handle_message(m, part[i], ts_ms[i], 1, key_len[i], val_len[i]);
            if (m->panicked) return; /* the process died in the first handler: the second never sees the record */
        }
        if (lc) {
            const uint8_t *k = (key_len[i] >= 0 && key_bytes) ? key_bytes + key_off[i] : NULL;
            kto_lc_handle_message(lc, k, key_len[i], val_len[i]);
        }
    }
}

/* TEST HARNESS, not part of the restatement: kto_run_soa's loop for the LogCompaction handler alone, over the records whose
 * key hashes into [slot_lo, slot_hi) — the others are passed by.  A record touches ONE bit, the one at fnv1a(key)
 * (metric.rs:273-280, 294-303), so K instances over disjoint slot ranges, each fed the whole topic in consumption order,
 * hold between them exactly what one instance holds: every slot sees its own records in order.  The tests run the instances
 * on K threads (the 2^30-record oracle of config 3 took 207 s of a 1200 s GPU step on one). */
void kto_lc_run_soa_slot_range(kto_logcompaction *lc, uint64_t n, const int32_t *key_len, const int32_t *val_len,
                               const uint32_t *key_off, const uint8_t *key_bytes, const uint32_t *slots,
                               uint64_t slot_lo, uint64_t slot_hi)
{
    for (uint64_t i = 0; i < n; i++) {
        if (key_len[i] < 0) continue;                    /* key None: ignored by the handler (metric.rs:302) */
        const uint8_t *k = key_bytes + key_off[i];
        const uint64_t slot = slots ? slots[i] : kto_fnv1a(k, (size_t)key_len[i]);   /* (slots: kto_fnv1a_soa's column) */
        if (slot < slot_lo || slot >= slot_hi) continue;
        kto_lc_handle_message(lc, k, key_len[i], val_len[i]);
    }
}

/* TEST HARNESS: fnv1a (metric.rs:256-260) of the records [first, first + n) of a batch's keys into out[first ...] (0 for
 * key None) — the column the slot-range instances pick their records by, computed once instead of once per instance. */
void kto_fnv1a_soa(uint64_t first, uint64_t n, const int32_t *key_len, const uint32_t *key_off, const uint8_t *key_bytes,
                   uint32_t *out)
{
    for (uint64_t i = first; i < first + n; i++)
        out[i] = key_len[i] < 0 ? 0u : (uint32_t)kto_fnv1a(key_bytes + key_off[i], (size_t)key_len[i]);
}

/* ------------------------------------------------------------------------ */
/* Additive analytics — NO reference counterpart (see kta_oracle.h).          */
/* ------------------------------------------------------------------------ */
struct kto_analytics {
    int32_t P;
    uint64_t key_hist[34], val_hist[34];
    int64_t *min_ms, *max_ms;
    uint64_t *smallest, *largest;
    uint8_t *seen, *live;
};

static unsigned size_bucket(int64_t len)
{
    if (len < 0) return 0;
    if (len == 0) return 1;
    unsigned b = 0;
    while ((len >> (b + 1)) != 0) b++;
    return 2 + b;
}

kto_analytics *kto_analytics_new(int32_t P)
{
    kto_analytics *a = (kto_analytics *)calloc(1, sizeof(*a));
    a->P = P;
    a->min_ms = (int64_t *)calloc((size_t)P, 8);
    a->max_ms = (int64_t *)calloc((size_t)P, 8);
    a->smallest = (uint64_t *)calloc((size_t)P, 8);
    a->largest = (uint64_t *)calloc((size_t)P, 8);
    a->seen = (uint8_t *)calloc((size_t)P, 1);
    a->live = (uint8_t *)calloc((size_t)P, 1);
    return a;
}

void kto_analytics_free(kto_analytics *a)
{
    if (!a) return;
    free(a->min_ms); free(a->max_ms); free(a->smallest); free(a->largest); free(a->seen); free(a->live);
    free(a);
}

void kto_analytics_run_soa(kto_analytics *a, uint64_t n, const int32_t *part, const int32_t *key_len,
                           const int32_t *val_len, const int64_t *ts_ms)
{
    for (uint64_t i = 0; i < n; i++) {
        const int32_t p = part[i];
        if (p < 0 || p >= a->P) continue;
        const int64_t ts = ts_ms[i] == -1 ? 0 : ts_ms[i]; /* as metric.rs:209 */
        a->key_hist[size_bucket(key_len[i])]++;
        a->val_hist[size_bucket(val_len[i])]++;
        if (!a->seen[p] || ts < a->min_ms[p]) a->min_ms[p] = ts;
        if (!a->seen[p] || ts > a->max_ms[p]) a->max_ms[p] = ts;
        a->seen[p] = 1;
        if (val_len[i] >= 0) {
            const uint64_t sz = (uint64_t)(key_len[i] > 0 ? key_len[i] : 0) + (uint64_t)val_len[i];
            if (!a->live[p] || sz < a->smallest[p]) a->smallest[p] = sz;
            if (!a->live[p] || sz > a->largest[p]) a->largest[p] = sz;
            a->live[p] = 1;
        }
    }
}

void kto_analytics_export(const kto_analytics *a, uint64_t *key_hist, uint64_t *val_hist, int64_t *min_ts_sec,
                          int64_t *max_ts_sec, uint64_t *smallest, uint64_t *largest)
{
    memcpy(key_hist, a->key_hist, sizeof a->key_hist);
    memcpy(val_hist, a->val_hist, sizeof a->val_hist);
    for (int32_t p = 0; p < a->P; p++) {
        min_ts_sec[p] = a->seen[p] ? a->min_ms[p] / 1000 : INT64_MAX;
        max_ts_sec[p] = a->seen[p] ? a->max_ms[p] / 1000 : INT64_MIN;
        smallest[p] = a->live[p] ? a->smallest[p] : UINT64_MAX;
        largest[p] = a->live[p] ? a->largest[p] : 0;
    }
}
