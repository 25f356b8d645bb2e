/*
 * kta_synth.h — the synthetic topic: a counter-based record generator shared, bit for bit,
 * by the host (C/C++) and the device (HIP).  The reference ships no data, no fixtures and
 * no generator; this stands in for "a Kafka topic consumed start to end"
 * (src/kafka.rs:92-135) with record i being the i-th polled message.
 *
 * Record i depends only on (spec, i): integer arithmetic only (no libm), so that the host
 * and device produce identical columns.  The per-record fields are exactly what the
 * reference handlers read from a BorrowedMessage (src/metric.rs:208-209, 218, 233).
 */
#ifndef KTA_SYNTH_H
#define KTA_SYNTH_H

#include <stdint.h>
#include "kta_hip.h"

#if defined(__HIPCC__)
#define KTA_HD __host__ __device__ static inline
#else
#define KTA_HD static inline
#endif

#ifdef __cplusplus
extern "C" {
#endif

enum {
    KTA_PART_RANDOM = 0,     /* partition = hash(i) % local partitions (fully interleaved)  */
    KTA_PART_KEY_AFFINE = 1, /* keyed records: key_id % local partitions (Kafka's default
                                partitioner shape); unkeyed: random                          */
    KTA_PART_RUNS = 2        /* runs of `part_run_len` consecutive records share a partition
                                (librdkafka delivers per-partition fetch batches)            */
};
enum { KTA_VAL_FIXED = 0, KTA_VAL_EXP = 1 };

#define KTA_SYNTH_MAX_KEY_LENS 8

typedef struct kta_synth_spec {
    uint64_t seed;
    uint32_t n_partitions;       /* P of the whole topic                                   */
    uint32_t shard_index;        /* this shard holds partitions p with p % shard_count ==   */
    uint32_t shard_count;        /*   shard_index (1 shard: 0 / 1)                          */
    uint32_t part_mode;          /* KTA_PART_*                                              */
    uint32_t part_run_len;       /* KTA_PART_RUNS                                           */
    uint32_t key_null_permille;  /* records with key None                                  */
    uint32_t key_empty_permille; /* key ids whose key is Some(&[])                          */
    uint32_t n_key_lens;         /* 1..8 */
    uint32_t key_lens[KTA_SYNTH_MAX_KEY_LENS]; /* key length = key_lens[hash(key_id) % n]  */
    uint64_t n_distinct_keys;    /* D; 0 => key_id = i (all distinct)                       */
    uint32_t tombstone_permille; /* records with payload None                              */
    uint32_t val_empty_permille; /* records with payload Some(&[])                          */
    uint32_t val_mode;           /* KTA_VAL_*                                               */
    uint32_t val_mean;           /* fixed length, or mean of the exponential-like law       */
    uint32_t val_cap;            /* upper clamp for KTA_VAL_EXP                             */
    uint32_t ts_missing_permille;/* records whose timestamp is -1 (not available)           */
    int64_t ts_base_ms;
    uint32_t ts_step_us;         /* ramp: ts advances by this per record                    */
    uint32_t ts_jitter_ms;       /* uniform +- jitter                                       */
} kta_synth_spec;

/* splitmix64 finaliser */
KTA_HD uint64_t kta_mix64(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

/* independent stream `s` of record/key `i` */
KTA_HD uint64_t kta_rng(uint64_t seed, uint64_t i, uint32_t s)
{
    return kta_mix64(kta_mix64(seed ^ (i * 0xD1B54A32D192ED03ull)) + (uint64_t)s);
}

KTA_HD uint32_t kta_clz32(uint32_t x)
{
    return x ? (uint32_t)__builtin_clz(x) : 32u;
}

KTA_HD uint32_t kta_synth_local_partitions(const kta_synth_spec *sp)
{
    /* partitions p in [0,P) with p % shard_count == shard_index */
    if (sp->shard_index >= sp->n_partitions) return 0;
    return (sp->n_partitions - sp->shard_index + sp->shard_count - 1) / sp->shard_count;
}

/* key id of record i, or -1 when the record has no key */
KTA_HD int64_t kta_synth_key_id(const kta_synth_spec *sp, uint64_t i)
{
    uint64_t r0 = kta_rng(sp->seed, i, 0);
    if ((uint32_t)(r0 % 1000u) < sp->key_null_permille) return -1;
    uint64_t id = sp->n_distinct_keys ? ((r0 >> 10) % sp->n_distinct_keys) : i;
    return (int64_t)(id & 0x7FFFFFFFFFFFFFFFull);
}

/* key length is a function of the key id (the same key always has the same bytes) */
KTA_HD int32_t kta_synth_key_len(const kta_synth_spec *sp, uint64_t key_id)
{
    uint64_t kr = kta_rng(sp->seed ^ 0x6b65795f6c656eull, key_id, 1);
    if ((uint32_t)(kr % 1000u) < sp->key_empty_permille) return 0;
    return (int32_t)sp->key_lens[(kr >> 10) % sp->n_key_lens];
}

/* byte j of key `key_id`: little-endian words, word 0 is the key id itself */
KTA_HD uint8_t kta_synth_key_byte(const kta_synth_spec *sp, uint64_t key_id, uint32_t j)
{
    uint32_t w = j >> 3;
    uint64_t word = w == 0 ? key_id : kta_rng(sp->seed ^ 0x6b65795f627974ull, key_id, w);
    return (uint8_t)(word >> (8u * (j & 7u)));
}

KTA_HD uint64_t kta_synth_key_word(const kta_synth_spec *sp, uint64_t key_id, uint32_t w)
{
    return w == 0 ? key_id : kta_rng(sp->seed ^ 0x6b65795f627974ull, key_id, w);
}

/* everything but the key bytes / key offset */
KTA_HD void kta_synth_record(const kta_synth_spec *sp, uint64_t i, int32_t *partition,
                             int32_t *key_len, int32_t *val_len, int64_t *ts_ms)
{
    uint32_t n_local = kta_synth_local_partitions(sp);
    int64_t kid = kta_synth_key_id(sp, i);
    *key_len = kid < 0 ? -1 : kta_synth_key_len(sp, (uint64_t)kid);

    uint64_t r1 = kta_rng(sp->seed, i, 2);
    uint32_t lp;
    if (sp->part_mode == KTA_PART_KEY_AFFINE && kid >= 0)
        lp = (uint32_t)((uint64_t)kid % n_local);
    else if (sp->part_mode == KTA_PART_RUNS)
        lp = (uint32_t)(kta_rng(sp->seed, i / (sp->part_run_len ? sp->part_run_len : 1u), 3) % n_local);
    else
        lp = (uint32_t)(r1 % n_local);
    *partition = (int32_t)(sp->shard_index + sp->shard_count * lp);

    uint64_t r2 = kta_rng(sp->seed, i, 4);
    uint32_t sel = (uint32_t)(r2 % 1000u);
    if (sel < sp->tombstone_permille) {
        *val_len = -1;
    } else if (sel < sp->tombstone_permille + sp->val_empty_permille) {
        *val_len = 0;
    } else if (sp->val_mode == KTA_VAL_FIXED) {
        *val_len = (int32_t)sp->val_mean;
    } else {
        /* exponential-like, integer only: -ln(U) ~ ln2 * (geometric + uniform fraction).
         * E[lz*256 + f8] = 383.5, and 171/65536 ~ 1/383.5, so the mean is ~val_mean. */
        uint32_t hi = (uint32_t)(r2 >> 32);
        uint32_t lz = kta_clz32(hi);
        uint32_t f8 = (uint32_t)(r2 >> 16) & 0xFFu;
        uint64_t v = ((uint64_t)sp->val_mean * (uint64_t)(lz * 256u + f8) * 171ull) >> 16;
        if (v > sp->val_cap) v = sp->val_cap;
        *val_len = (int32_t)v;
    }

    uint64_t r3 = kta_rng(sp->seed, i, 5);
    if ((uint32_t)((r3 >> 40) % 1000u) < sp->ts_missing_permille) {
        *ts_ms = -1;
    } else {
        int64_t t = sp->ts_base_ms + (int64_t)((i * (uint64_t)sp->ts_step_us) / 1000u);
        if (sp->ts_jitter_ms) {
            uint32_t span = 2u * sp->ts_jitter_ms + 1u;
            t += (int64_t)(uint32_t)(r3 % span) - (int64_t)sp->ts_jitter_ms;
        }
        *ts_ms = t;
    }
}

/* ---- library entry points (libkta_hip.so) ------------------------------------------ */
/* Fill host columns with records [first, first+n): key_off batch-local and packed in record
 * order; key_off/key_bytes/seq may be NULL (then only the four metric columns are written).
 * *n_key_bytes receives the bytes written (or needed). */
int kta_synth_fill_host(const kta_synth_spec *spec, uint64_t first, uint64_t n,
                        const kta_batch *host_cols, uint64_t *n_key_bytes);
/* Same on the device (HBM-resident batch), asynchronous on the context's compute stream
 * except for the key-byte total, which is read back. */
int kta_synth_fill_device(kta_ctx *ctx, const kta_synth_spec *spec, uint64_t first, uint64_t n,
                          const kta_batch *device_cols, uint64_t *n_key_bytes);
/* Named presets for the BASELINE.json configs: "c1".."c5" (BASELINE.md §4). */
int kta_synth_preset(const char *name, kta_synth_spec *out, uint64_t *n_records);

#ifdef __cplusplus
}
#endif
#endif /* KTA_SYNTH_H */
