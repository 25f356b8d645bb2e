/*
 * kta_kafka.h — Kafka record-batch (message format v2, magic 2) decode on the GPU: raw record
 * sets (what a Fetch response carries per partition, and byte-for-byte what a broker's
 * `*.log` segment file contains) -> the struct-of-arrays columns of kta_hip.h.
 *
 * This is the step BEFORE the hot path.  In the reference it happens inside librdkafka
 * (rdkafka-sys 3.0.0+1.6.0 -> librdkafka 1.6.0, `Cargo.lock:612-613`; C source not in the
 * reference tree): `consumer.poll()` (src/kafka.rs:93) hands out one decoded message at a time and
 * the handlers read partition / timestamp / key / payload from it (src/metric.rs:208-209, 218, 233).
 * Here the host only walks the 61-byte batch headers; the varint-framed records are parsed on the
 * device straight into the columns the metric kernels consume, so the PCIe link carries the raw
 * log once and no per-message host work remains.
 *
 * Format (Apache Kafka protocol guide, "Record Batch", KIP-98):
 *   batch : baseOffset i64 | batchLength i32 | partitionLeaderEpoch i32 | magic i8 (=2) | crc u32 |
 *           attributes i16 | lastOffsetDelta i32 | baseTimestamp i64 | maxTimestamp i64 |
 *           producerId i64 | producerEpoch i16 | baseSequence i32 | recordsCount i32 | records
 *           (big endian; 61 bytes of header; the batch occupies 12 + batchLength bytes)
 *   record: length varint | attributes i8 | timestampDelta varlong | offsetDelta varint |
 *           keyLength varint (-1 = null) | key | valueLength varint (-1 = null) | value |
 *           headersCount varint | headers (keyLen varint, key, valueLen varint (-1 null), value)
 *           (varints are zig-zag base-128)
 *   attributes: bits 0-2 compression codec, bit 3 timestamp type (1 = LogAppendTime: every record
 *           carries maxTimestamp), bit 4 transactional, bit 5 control batch (never delivered to the
 *           application), bit 6 delete horizon.
 *
 * What a consumer delivers per record (and what is written to the columns):
 *   partition = the record set's partition, timestamp = baseTimestamp + timestampDelta (or
 *   maxTimestamp for LogAppendTime), key/payload None iff the length is -1.
 * Compression: Snappy batches (codec 2; bare blocks as librdkafka writes them and the snappy-java
 * stream framing of the Java clients) and LZ4 batches (codec 3, LZ4 frame format, linked or
 * independent blocks) are inflated on the device by wave-cooperative kernels, gzip (codec 1) and zstd
 * (codec 4) batches by one lane per batch; batches of an unknown codec and magic 0/1 message sets are
 * counted, never silently mis-decoded.  CRCs are verified only on request (kta_kafka_set_check_crcs) — librdkafka's
 * `check.crcs` defaults to false and the reference does not set it (src/kafka.rs:28-36).
 */
#ifndef KTA_KAFKA_H
#define KTA_KAFKA_H

#include <stdint.h>
#include "kta_hip.h"

struct kta_synth_spec;

#ifdef __cplusplus
extern "C" {
#endif

#define KTA_KAFKA_BATCH_HEADER 61

/* flags of a batch descriptor */
#define KTA_KB_LOG_APPEND_TIME 1u /* attributes bit 3 */
#define KTA_KB_TRANSACTIONAL 2u   /* attributes bit 4 */
#define KTA_KB_SNAPPY 4u          /* records are Snappy compressed (codec 2): inflated on the device */
#define KTA_KB_LZ4 8u             /* records are LZ4 compressed (codec 3, LZ4 frame format): likewise */
#define KTA_KB_GZIP 16u           /* records are gzip compressed (codec 1): likewise                  */
#define KTA_KB_ZSTD 32u           /* records are zstd compressed (codec 4): likewise                  */
/* status of a batch after the device passes */
#define KTA_KB_BAD_CRC 1u         /* CRC-32C mismatch (only with kta_kafka_set_check_crcs)           */
#define KTA_KB_BAD_FRAMING 2u     /* records overran the batch                                      */

typedef struct kta_kafka_batch_desc {
    uint64_t byte_off;    /* offset of the batch (its baseOffset field) in the blob          */
    uint64_t record_base; /* index of its first record in the output columns               */
    uint32_t crc;         /* the batch's stored CRC-32C (over attributes .. end of batch)           */
    uint32_t status;      /* device: 0 ok, KTA_KB_BAD_CRC / KTA_KB_BAD_FRAMING after decoding       */
    uint64_t payload_off; /* where the records are parsed from: byte_off + 61 for an uncompressed    */
    uint64_t payload_end; /*   batch, else this batch's slice of the inflate area (same buffer)      */
    int64_t base_offset;  /* Kafka offset of the first record                                */
    int64_t base_ts_ms;
    int64_t max_ts_ms;
    uint32_t batch_bytes; /* 12 + batchLength                                               */
    int32_t partition;
    int32_t n_records;
    uint32_t flags;
    uint64_t scratch_end; /* gzip, zstd: [payload_end, scratch_end) is the decoder's scratch in the inflate area */
} kta_kafka_batch_desc;

typedef struct kta_kafka_index_stats {
    uint64_t n_batches;          /* descriptors written                                     */
    uint64_t n_records;          /* sum of their recordsCount                               */
    uint64_t n_control_batches;  /* skipped: never delivered to the application             */
    uint64_t n_compressed;       /* skipped: batches of an unknown codec (5..7)             */
    uint64_t n_snappy;           /* Snappy batches (codec 2): inflated on the device         */
    uint64_t n_lz4;              /* LZ4 batches (codec 3): inflated on the device            */
    uint64_t inflate_bytes;      /* bytes of inflate area the descriptors use               */
    uint64_t n_old_magic;        /* skipped: magic 0/1 message sets                         */
    uint64_t trailing_bytes;     /* bytes after the last complete batch (partial fetch tail)*/
    uint64_t bytes_consumed;
    uint64_t n_gzip;             /* gzip batches (codec 1): inflated on the device           */
    uint64_t n_zstd;             /* zstd batches (codec 4): inflated on the device           */
    int64_t first_offset;        /* baseOffset of the first v2 batch, control batches included            */
    int64_t next_offset;         /* max(baseOffset + lastOffsetDelta + 1): the log end offset / high watermark */
    uint64_t any_offsets;        /* 0: no v2 batch, the two offsets are meaningless                       */
} kta_kafka_index_stats;

/* Host: walk the batch headers of one record set.  `record_base_start` is the output index of the
 * set's first record (lets several sets share one output batch).  Snappy batches get a slice of the
 * inflate area, which starts at `inflate_offset` of the SAME device buffer as the blob (so keys stay
 * zero-copy); stats->inflate_bytes tells how much of it is used.  Returns KTA_ERR_CAPACITY when `cap`
 * descriptors do not suffice (stats->n_batches then holds the number needed). */
int kta_kafka_index_host(const uint8_t *bytes, uint64_t len, int32_t partition, uint64_t blob_offset,
                         uint64_t record_base_start, uint64_t inflate_offset, kta_kafka_batch_desc *descs,
                         uint64_t cap, kta_kafka_index_stats *stats);
/* Snappy inflate of one batch payload on the host (bare block or snappy-java stream framing): the
 * same code the device runs.  Returns the bytes produced or -1. */
int64_t kta_snappy_inflate_host(const uint8_t *src, uint64_t n, uint8_t *dst, uint64_t cap);
/* LZ4 frame inflate on the host (the same code the device runs).  Returns the bytes produced or -1. */
int64_t kta_lz4_inflate_host(const uint8_t *src, uint64_t n, uint8_t *dst, uint64_t cap);
/* zstd (one or more frames) inflate on the host (the same code the device runs).  Returns the bytes produced or -1. */
int64_t kta_zstd_inflate_host(const uint8_t *src, uint64_t n, uint8_t *dst, uint64_t cap);
/* The same with the table layout of the device's wave-per-batch kernel, where the Huffman table lies over the
 * sequence tables and what must outlive a block goes through a spill (csrc/kta_zstd.h, ZsWorkSmall): the host's
 * execution of that text, for tests.  Returns the bytes produced or -1. */
int64_t kta_zstd_inflate_host_small(const uint8_t *src, uint64_t n, uint8_t *dst, uint64_t cap);
/* gzip (one member) inflate on the host: the code the device runs, stage by stage (Huffman decoding into
 * literals + match tokens, then the copies).  Returns the bytes produced or -1. */
int64_t kta_gzip_inflate_host(const uint8_t *src, uint64_t n, uint8_t *dst, uint64_t cap);
/* The single-pass form (the device's one-lane-per-batch kernel, kta_kafka_set_variant 1). */
int64_t kta_gzip_inflate_lane_host(const uint8_t *src, uint64_t n, uint8_t *dst, uint64_t cap);

/* The record decode kernel's rounds on the host (csrc/kta_records.h: the chain over the length prefixes and the parse
 * of one record are the device's own code): a group of `lanes` lanes (a divisor of 64) per batch, `window` bytes per
 * round (a multiple of 16 * lanes), at most `per_round` records per round (a multiple of `lanes`).  `blob` holds the
 * bytes the descriptors point into (payload_end <= blob_len); key_off may be NULL; offsets are relative to `blob`.
 * Test infrastructure for the CPU suite — the product decodes on the device (kta_kafka_decode_device). */
int kta_kafka_decode_rounds_host(const uint8_t *blob, uint64_t blob_len, const kta_kafka_batch_desc *descs,
                                 uint64_t n_batches, uint32_t lanes, uint32_t window, uint32_t per_round,
                                 int32_t *partition, int32_t *key_len, int32_t *val_len, int64_t *ts_ms,
                                 uint32_t *key_off, uint64_t *n_key_bytes, uint64_t *n_bad_batches);

/* Device: parse the records of `n_batches` indexed batches out of `blob_device` (16-byte aligned: that is all
 * correctness asks for; 128-byte alignment, as hipMalloc and the blob ring give it, is a performance recommendation
 * only — the kernel's window loads then fall on whole lines — readable for 64 bytes past `blob_len`) into the device columns `out` (capacity >= total records).
 * Keys are ZERO-COPY: when out->key_off is set, key_off[i] is the offset of record i's key inside the
 * blob, so the caller passes `blob_device` itself as `key_bytes` when submitting the columns
 * (out->key_bytes is ignored; the blob must then be < 4 GiB and stay alive until the kernels ran).
 * `descs_host` is copied to the device.  *n_key_bytes receives the total key bytes seen;
 * *n_bad_batches the batches whose records overran the batch (their remaining records are written
 * as unkeyed tombstones on partition -1 so that they are reported, not counted). */
int kta_kafka_decode_device(kta_ctx *ctx, const uint8_t *blob_device, uint64_t blob_len,
                            const kta_kafka_batch_desc *descs_host, uint64_t n_batches, uint64_t n_records,
                            const kta_batch *out, uint64_t *n_key_bytes, uint64_t *n_bad_batches);

/* ---- the raw-log pipeline -----------------------------------------------------------------
 * The fetcher writes each Fetch response's record set (or a chunk of a `*.log` segment, cut anywhere)
 * straight into a PINNED staging blob; submit indexes the batch headers on the host, sends the
 * bytes over PCIe on the copy stream and decodes + accumulates on the compute stream while the next
 * blob is being filled (ring of stages).  stats->bytes_consumed tells how many bytes were whole
 * batches; the caller carries the remaining tail (a partial batch) into the next blob. */
int kta_kafka_configure(kta_ctx *ctx, uint64_t blob_capacity, int n_stages);  /* before first acquire; 0 = default (256 MiB, 3) */
int kta_kafka_blob_acquire(kta_ctx *ctx, uint8_t **host_ptr, uint64_t *capacity);
int kta_kafka_blob_submit(kta_ctx *ctx, uint64_t len, int32_t partition, kta_kafka_index_stats *stats);

/* Compressed batches of a blob are inflated in groups whose slices of the inflate area span at most
 * `bytes` (0 = default, 1 GiB): the slices are sized by per-batch bounds, so this caps the device memory a
 * blob of many small compressed batches can ask for and keeps key offsets inside 32 bits.  A single batch
 * larger than the limit still goes through alone. */
int kta_kafka_set_inflate_limit(kta_ctx *ctx, uint64_t bytes);

/* Convenience for hosts that hold the raw bytes in ordinary memory: memcpy into the staging ring
 * (chunked at batch boundaries) and submit (records get consecutive sequence numbers). */
int kta_kafka_consume(kta_ctx *ctx, const uint8_t *bytes, uint64_t len, int32_t partition,
                      kta_kafka_index_stats *stats);

/* Producer side, for benchmarks and fixtures: encode records [first, first+n) of the synthetic
 * topic (include/kta_synth.h) as v2 record batches of `records_per_batch` records, exactly as a
 * broker segment would hold them (valid CRC-32C, zero-filled values of the right length, the
 * spec's partition ignored: a record set belongs to ONE partition, chosen at consume time).
 * `out` may be NULL to size the buffer; *len receives the bytes written / needed. */
int kta_kafka_encode_synth_host(const struct kta_synth_spec *spec, uint64_t first, uint64_t n,
                                uint32_t records_per_batch, uint8_t *out, uint64_t cap, uint64_t *len);
/* As above with a codec: 0 = none, 2 = Snappy (bare block), 3 = LZ4 (one frame of linked 64 KiB blocks),
 * both from a small greedy compressor; values are then filled with a 24-byte periodic pseudo-random
 * pattern so that the stream has real copies.  0x100 = uncompressed batches with that same value
 * pattern (what a caller compresses with another codec, e.g. bench.py with zlib for gzip).  | 0x200: the
 * values are words, numbers and punctuation instead (skewed bytes and many short copies, as JSON
 * payloads have them: Huffman-coded literals in zstd, long code tables in gzip). */
int kta_kafka_encode_synth_host_ex(const struct kta_synth_spec *spec, uint64_t first, uint64_t n,
                                   uint32_t records_per_batch, int codec, uint8_t *out, uint64_t cap,
                                   uint64_t *len);

/* librdkafka's `check.crcs` (default false; the reference forwards user options, src/kafka.rs:38-42):
 * when enabled, every batch's CRC-32C (Castagnoli, over the bytes from `attributes` to the end of
 * the batch) is verified on the device before decoding; a batch that fails is not delivered — its
 * records are written with partition -1 (counted as bad-partition records) and it is counted in
 * *n_bad_batches.  kta_kafka_crc_errors returns the number of CRC failures since the context was
 * created. */
int kta_kafka_set_check_crcs(kta_ctx *ctx, int enable);
int kta_kafka_crc_errors(kta_ctx *ctx, uint64_t *n);
/* CRC-32C of host bytes (the same tables the device uses; check value of "123456789": 0xE3069283). */
uint32_t kta_crc32c_host(const uint8_t *bytes, uint64_t len);

/* Decode kernel choice of this context: 0 = automatic (default: by the number of batches in the call and their mean
 * size), 1 = one lane per batch (kept for comparison; also selects the lane-per-batch inflate kernels of all four
 * codecs instead of the wave-cooperative / two-stage ones), 2 = one wave per batch with 8 KiB LDS windows (calls with
 * fewer than 2048 batches), 10 = four batches per wave, 3 KiB windows, 16 records per round (batches below 28 KiB),
 * 11 = two batches per wave, 8 KiB windows, 32 records per round (batches of 28 KiB and more).  Any other value:
 * KTA_ERR_INVALID (3 ... 9 and 12 ... 14 were geometries that lost the side-by-side timing of round 5 and were deleted:
 * profiles/r05_decode_geometries.jsonl). */
int kta_kafka_set_variant(kta_ctx *ctx, int variant);

/* Average duration (ms) of the decode kernel since the previous call ([1]; [0] is reserved, -1);
 * launches[] the counts.  Needs kta_set_timing(ctx, 1). */
int kta_kafka_time_stats(kta_ctx *ctx, float avg_ms[2], uint64_t launches[2]);

#ifdef __cplusplus
}
#endif
#endif /* KTA_KAFKA_H */
