/*
 * kta_kafka_oracle.c — CPU ORACLE for the Kafka record-batch v2 decode (TEST INFRASTRUCTURE).
 *
 * In the reference this step is not in the tree: `consumer.poll()` (src/kafka.rs:93) returns
 * messages already decoded by librdkafka (rdkafka-sys 3.0.0+1.6.0 => librdkafka 1.6.0,
 * Cargo.lock:612-613; the C file there is rdkafka_msgset_reader.c, v2 path).  This file restates the
 * PUBLISHED wire format (Apache Kafka protocol guide, "Record Batch"; KIP-98) and what a consumer
 * delivers per record, as a plain sequential decoder.  Pinned by: round trips against an independent
 * Python ENCODER (tests/kafka_format.py) whose inputs are the expected outputs, and a committed
 * byte-level fixture (tests/golden/kafka_v2_recordset.*).  Parity with librdkafka itself is
 * unpinned (not installed in the image).
 */
#include <stdint.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

int64_t kto_snappy_inflate(const uint8_t *src, uint64_t n, uint8_t *out, uint64_t cap);
int64_t kto_lz4_inflate(const uint8_t *src, uint64_t n, uint8_t *out, uint64_t cap);
int64_t kto_gzip_inflate(const uint8_t *src, uint64_t n, uint8_t *out, uint64_t cap);

typedef struct {
    uint64_t control_batches, compressed_batches, old_magic_batches, trailing_bytes, bad_batches, batches;
} kto_kafka_stats;

static uint64_t rd_be(const uint8_t *p, int n)
{
    uint64_t v = 0;
    for (int i = 0; i < n; i++) v = (v << 8) | p[i];
    return v;
}

/* zig-zag varint; returns bytes consumed (0 on overrun) */
static size_t rd_varint(const uint8_t *p, const uint8_t *end, int64_t *out)
{
    uint64_t v = 0;
    size_t i = 0;
    for (;;) {
        if (p + i >= end || i >= 10) return 0;
        uint8_t b = p[i];
        v |= (uint64_t)(b & 0x7f) << (7 * i);
        i++;
        if (!(b & 0x80)) break;
    }
    *out = (int64_t)(v >> 1) ^ -(int64_t)(v & 1);
    return i;
}

/* Decode one record set.  Output arrays may be NULL (count only).  Returns the number of records. */
int64_t kto_kafka_decode(const uint8_t *bytes, uint64_t len, int32_t partition, int32_t *part, int32_t *klen,
                         int32_t *vlen, int64_t *ts_ms, int64_t *offsets, uint32_t *koff, uint8_t *kbytes,
                         uint64_t *n_key_bytes, kto_kafka_stats *st)
{
    uint64_t pos = 0, n = 0, kb = 0;
    memset(st, 0, sizeof *st);
    while (pos + 12 <= len) {
        int32_t batch_length = (int32_t)rd_be(bytes + pos + 8, 4);
        if (batch_length < 49) break;
        uint64_t total = 12 + (uint64_t)batch_length;
        if (pos + total > len) break;
        const uint8_t *b = bytes + pos;
        const uint8_t *bend = b + total;
        if (b[16] != 2) { st->old_magic_batches++; pos += total; continue; }
        uint16_t attrs = (uint16_t)rd_be(b + 21, 2);
        if (attrs & 0x20) { st->control_batches++; pos += total; continue; }
        if ((attrs & 0x07) > 3) { st->compressed_batches++; pos += total; continue; }   /* zstd arrives here only when the
                                                                                        * test could not transcode it (tests/oracle_c.py) */
        int64_t base_offset = (int64_t)rd_be(b, 8);
        int64_t base_ts = (int64_t)rd_be(b + 27, 8), max_ts = (int64_t)rd_be(b + 35, 8);
        int32_t count = (int32_t)rd_be(b + 57, 4);
        if (count <= 0) { pos += total; continue; }
        st->batches++;
        const uint8_t *p = b + 61;
        int bad = 0;
        /* a record takes at least 7 bytes: a batch announcing more records than its payload can hold is
         * corrupt as a whole (the product clamps the count the same way and delivers none of them).  The
         * payload bound depends on the header alone: the batch's own bytes when uncompressed, the codec's
         * maximum expansion of them otherwise (gzip 1032x, Snappy 22x, LZ4 255x) -- so a compressed batch
         * whose stream turns out corrupt keeps its announced count, every record flagged. */
        {
            uint64_t clen = total - 61, codec = attrs & 0x07;
            uint64_t payload = codec == 0 ? clen : codec == 1 ? clen * 1032 + 64 : codec == 2 ? clen * 22 + 64 : clen * 255 + 64;
            if ((uint64_t)count > payload / 7 + 1) {
                count = (int32_t)(payload / 7 + 1);
                bad = 1;
            }
        }
        uint8_t *inflated = NULL;
        if ((attrs & 0x07) != 0) { /* gzip / Snappy / LZ4: the records section is compressed as a whole */
            uint64_t cap = (total - 61) * 1100 + (1u << 22);
            inflated = (uint8_t *)malloc((size_t)cap);
            int64_t got = (attrs & 0x07) == 2 ? kto_snappy_inflate(b + 61, total - 61, inflated, cap)
                        : (attrs & 0x07) == 3 ? kto_lz4_inflate(b + 61, total - 61, inflated, cap)
                                              : kto_gzip_inflate(b + 61, total - 61, inflated, cap);
            if (got < 0) { bad = 1; got = 0; }
            p = inflated;
            bend = inflated + got;
        }
        for (int32_t j = 0; j < count; j++) {
            int64_t rlen, tsd, offd, kl, vl;
            size_t c;
            if (bad || p >= bend || !(c = rd_varint(p, bend, &rlen)) || rlen < 0 || p + c + rlen > bend) bad = 1;
            if (!bad) {
                const uint8_t *q = p + c, *rend = q + rlen;
                q += 1; /* record attributes */
                size_t c1 = rd_varint(q, rend, &tsd); q += c1;
                size_t c2 = c1 ? rd_varint(q, rend, &offd) : 0; q += c2;
                size_t c3 = c2 ? rd_varint(q, rend, &kl) : 0; q += c3;
                if (!c3 || kl < -1 || (kl > 0 && q + kl > rend)) bad = 1;
                const uint8_t *kp = q;
                if (!bad && kl > 0) q += kl;
                size_t c4 = bad ? 0 : rd_varint(q, rend, &vl);
                if (!bad && (!c4 || vl < -1 || q + c4 + (vl > 0 ? vl : 0) > rend)) bad = 1;
                if (!bad) {
                    if (part) {
                        part[n] = partition; klen[n] = (int32_t)kl; vlen[n] = (int32_t)vl;
                        ts_ms[n] = (attrs & 0x08) ? max_ts : base_ts + tsd;
                        if (offsets) offsets[n] = base_offset + offd;
                        if (koff) koff[n] = (uint32_t)kb;
                        if (kbytes && kl > 0) memcpy(kbytes + kb, kp, (size_t)kl);
                    }
                    if (kl > 0) kb += (uint64_t)kl;
                    p = rend; /* headers are skipped */
                    n++;
                    continue;
                }
            }
            /* corrupt batch: the product marks the remaining records with partition -1 */
            if (part) {
                part[n] = -1; klen[n] = -1; vlen[n] = -1; ts_ms[n] = -1;
                if (offsets) offsets[n] = -1;
                if (koff) koff[n] = (uint32_t)kb;
            }
            n++;
        }
        if (bad) st->bad_batches++;
        free(inflated);
        pos += total;
    }
    st->trailing_bytes = len - pos;
    if (n_key_bytes) *n_key_bytes = kb;
    return (int64_t)n;
}

/* CRC-32C (Castagnoli), bit-at-a-time on purpose (no tables shared with the product): reflected
 * polynomial 0x82F63B78, init and final xor 0xFFFFFFFF.  Published check value: "123456789" -> 0xE3069283.
 * A Kafka v2 batch stores it at offset 17, computed over the bytes from offset 21 to the end. */
uint32_t kto_crc32c(const uint8_t *p, uint64_t n)
{
    uint32_t c = 0xFFFFFFFFu;
    for (uint64_t i = 0; i < n; i++) {
        c ^= p[i];
        for (int k = 0; k < 8; k++) c = (c & 1u) ? (c >> 1) ^ 0x82F63B78u : c >> 1;
    }
    return c ^ 0xFFFFFFFFu;
}

/* 1 if the v2 batch starting at `batch` (12 + batchLength bytes) carries a correct CRC */
int kto_kafka_batch_crc_ok(const uint8_t *batch, uint64_t total)
{
    uint32_t stored = (uint32_t)rd_be(batch + 17, 4);
    return stored == kto_crc32c(batch + 21, total - 21);
}

/* ---- Snappy (Kafka codec 2): independent sequential inflater --------------------------------------
 * Restates google/snappy's format_description.txt and snappy-java's stream framing (magic
 * "\x82SNAPPY\0", two big-endian version words, then [u32 BE length][block]...).  Written against the
 * format text, not against the product's kta_snappy.h. */
static int64_t kto_snappy_block(const uint8_t *src, uint64_t n, uint8_t *out, uint64_t cap)
{
    uint64_t ulen = 0, i = 0, o = 0;
    int shift = 0;
    for (;;) {
        if (i >= n || shift > 28) return -1;
        uint8_t b = src[i++];
        ulen |= (uint64_t)(b & 0x7f) << shift;
        shift += 7;
        if (!(b & 0x80)) break;
    }
    if (ulen > cap) return -1;
    while (i < n) {
        uint8_t tag = src[i++];
        switch (tag & 3) {
        case 0: {
            uint64_t l = (uint64_t)(tag >> 2) + 1;
            if (l > 60) {
                int extra = (int)l - 60;
                if (i + (uint64_t)extra > n) return -1;
                l = 0;
                for (int k = extra - 1; k >= 0; k--) l = (l << 8) | src[i + (uint64_t)k];
                l += 1;
                i += (uint64_t)extra;
            }
            if (i + l > n || o + l > ulen) return -1;
            memcpy(out + o, src + i, (size_t)l);
            i += l; o += l;
            break;
        }
        default: {
            uint64_t l, off;
            if ((tag & 3) == 1) {
                if (i >= n) return -1;
                l = ((tag >> 2) & 7) + 4;
                off = ((uint64_t)(tag & 0xe0) << 3) | src[i];
                i += 1;
            } else if ((tag & 3) == 2) {
                if (i + 2 > n) return -1;
                l = (tag >> 2) + 1;
                off = src[i] | ((uint64_t)src[i + 1] << 8);
                i += 2;
            } else {
                if (i + 4 > n) return -1;
                l = (tag >> 2) + 1;
                off = src[i] | ((uint64_t)src[i + 1] << 8) | ((uint64_t)src[i + 2] << 16) | ((uint64_t)src[i + 3] << 24);
                i += 4;
            }
            if (off == 0 || off > o || o + l > ulen) return -1;
            while (l--) { out[o] = out[o - off]; o++; }
            break;
        }
        }
    }
    return o == ulen ? (int64_t)o : -1;
}

int64_t kto_snappy_inflate(const uint8_t *src, uint64_t n, uint8_t *out, uint64_t cap)
{
    static const uint8_t magic[8] = {0x82, 'S', 'N', 'A', 'P', 'P', 'Y', 0};
    if (n >= 16 && memcmp(src, magic, 8) == 0) {
        uint64_t i = 16, o = 0;
        while (i + 4 <= n) {
            uint64_t clen = rd_be(src + i, 4);
            i += 4;
            if (clen == 0 || i + clen > n) return -1;
            int64_t got = kto_snappy_block(src + i, clen, out + o, cap - o);
            if (got < 0) return -1;
            o += (uint64_t)got;
            i += clen;
        }
        return i == n ? (int64_t)o : -1;
    }
    return kto_snappy_block(src, n, out, cap);
}

/* ---- LZ4 (Kafka codec 3, LZ4 frame format): independent sequential inflater ------------------------
 * Restates lz4_Frame_format.md and lz4_Block_format.md (lz4 v1.9 doc/).  Checksums are skipped.
 * Written against the format text, not against the product's kta_lz4.h. */
int64_t kto_lz4_inflate(const uint8_t *src, uint64_t n, uint8_t *out, uint64_t cap)
{
    if (n < 7 || src[0] != 0x04 || src[1] != 0x22 || src[2] != 0x4d || src[3] != 0x18) return -1;
    uint8_t flg = src[4], bd = src[5];
    if ((flg & 0xc0) != 0x40) return -1;
    int bsid = (bd >> 4) & 7;
    if (bsid < 4) return -1;
    uint64_t bmax = (uint64_t)1 << (8 + 2 * bsid);
    uint64_t i = 6;
    if (flg & 0x08) i += 8; /* content size */
    if (flg & 0x01) i += 4; /* dictionary id */
    i += 1;                 /* header checksum */
    uint64_t o = 0;
    for (;;) {
        if (i + 4 > n) return -1;
        uint32_t w = (uint32_t)src[i] | ((uint32_t)src[i + 1] << 8) | ((uint32_t)src[i + 2] << 16) | ((uint32_t)src[i + 3] << 24);
        i += 4;
        if (w == 0) break; /* end mark */
        uint64_t sz = w & 0x7fffffffu;
        if (sz > bmax || i + sz > n) return -1;
        if (w & 0x80000000u) {
            if (o + sz > cap) return -1;
            memcpy(out + o, src + i, (size_t)sz);
            o += sz;
        } else {
            const uint8_t *q = src + i, *qe = q + sz;
            while (q < qe) {
                uint8_t tok = *q++;
                uint64_t ll = tok >> 4;
                if (ll == 15) { uint8_t b; do { if (q >= qe) return -1; b = *q++; ll += b; } while (b == 255); }
                if ((uint64_t)(qe - q) < ll || o + ll > cap) return -1;
                memcpy(out + o, q, (size_t)ll);
                q += ll; o += ll;
                if (q == qe) break;
                if (qe - q < 2) return -1;
                uint64_t off = (uint64_t)q[0] | ((uint64_t)q[1] << 8);
                q += 2;
                uint64_t ml = tok & 15;
                if (ml == 15) { uint8_t b; do { if (q >= qe) return -1; b = *q++; ml += b; } while (b == 255); }
                ml += 4;
                if (off == 0 || off > o || o + ml > cap) return -1;
                while (ml--) { out[o] = out[o - off]; o++; }
            }
        }
        i += sz;
        if (flg & 0x10) i += 4; /* block checksum */
    }
    return (int64_t)o;
}


/* ---- gzip (Kafka codec 1) -----------------------------------------------------------------------
 * Not a restatement: the oracle hands the member to zlib itself (inflateInit2 with windowBits 15 + 16 =
 * gzip wrapper, header and CRC-32 / ISIZE trailer checked by zlib), the library every Kafka client uses
 * for this codec.  Like the product, exactly one member is accepted. */
#include <zlib.h>
int64_t kto_gzip_inflate(const uint8_t *src, uint64_t n, uint8_t *out, uint64_t cap)
{
    z_stream z;
    memset(&z, 0, sizeof z);
    if (inflateInit2(&z, 15 + 16) != Z_OK) return -1;
    z.next_in = (Bytef *)src;
    z.avail_in = (uInt)n;
    z.next_out = out;
    z.avail_out = (uInt)(cap > 0xFFFFFFFFu ? 0xFFFFFFFFu : cap);
    int rc = inflate(&z, Z_FINISH);
    int64_t got = (rc == Z_STREAM_END && z.avail_in == 0) ? (int64_t)z.total_out : -1;
    inflateEnd(&z);
    return got;
}
