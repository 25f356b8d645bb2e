// kafka_encode.cpp — the PRODUCER side of include/kta_kafka.h (host only): encodes the synthetic topic
// as Kafka v2 record batches (optionally Snappy / LZ4 compressed) exactly as a broker segment holds
// them, for benchmarks, fixtures and topic files; plus the host CRC-32C.  Nothing here is on the
// measured path: the consumer side (index, inflate, CRC check, decode) is csrc/kta_kafka.hip.
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "kta_kafka.h"
#include "kta_synth.h"

namespace {

uint32_t g_crc_table[256];
bool g_crc_ready = false;

uint32_t crc32c(const uint8_t *p, size_t n)
{
    if (!g_crc_ready) {
        for (uint32_t i = 0; i < 256; i++) {
            uint32_t c = i;
            for (int k = 0; k < 8; k++) c = (c & 1) ? (c >> 1) ^ 0x82F63B78u : c >> 1;
            g_crc_table[i] = c;
        }
        g_crc_ready = true;
    }
    uint32_t c = 0xFFFFFFFFu;
    for (size_t i = 0; i < n; i++) c = g_crc_table[(c ^ p[i]) & 0xFF] ^ (c >> 8);
    return c ^ 0xFFFFFFFFu;
}

size_t put_varint(uint8_t *dst, int64_t v)
{
    uint64_t z = ((uint64_t)v << 1) ^ (uint64_t)(v >> 63);
    size_t i = 0;
    while (z >= 0x80) {
        if (dst) dst[i] = (uint8_t)(z | 0x80);
        z >>= 7;
        i++;
    }
    if (dst) dst[i] = (uint8_t)z;
    return i + 1;
}

void put_be(uint8_t *p, uint64_t v, int n)
{
    for (int i = n - 1; i >= 0; i--) { p[i] = (uint8_t)v; v >>= 8; }
}

} // namespace

namespace {

// greedy Snappy compressor (bare block) for benchmark / fixture data: 4-byte hash matcher
void snappy_compress(const std::vector<uint8_t> &in, std::vector<uint8_t> &out)
{
    out.clear();
    uint64_t n = in.size(), v = n;
    while (v >= 0x80) { out.push_back((uint8_t)(v | 0x80)); v >>= 7; }
    out.push_back((uint8_t)v);
    std::vector<uint32_t> table(1u << 14, 0xFFFFFFFFu);
    auto emit_literal = [&](uint64_t from, uint64_t to) {
        while (from < to) {
            const uint64_t l = to - from < 65536 ? to - from : 65536;
            if (l <= 60) out.push_back((uint8_t)((l - 1) << 2));
            else if (l <= 256) { out.push_back(60 << 2); out.push_back((uint8_t)(l - 1)); }
            else { out.push_back(61 << 2); out.push_back((uint8_t)((l - 1) & 0xFF)); out.push_back((uint8_t)((l - 1) >> 8)); }
            out.insert(out.end(), in.begin() + from, in.begin() + from + l);
            from += l;
        }
    };
    uint64_t i = 0, lit = 0;
    while (i + 4 <= n) {
        uint32_t w;
        memcpy(&w, &in[i], 4);
        const uint32_t h = (w * 0x1e35a7bdu) >> 18;
        const uint32_t cand = table[h];
        table[h] = (uint32_t)i;
        if (cand != 0xFFFFFFFFu && i - cand < 65536 && memcmp(&in[cand], &in[i], 4) == 0) {
            uint64_t m = 4;
            while (i + m < n && in[cand + m] == in[i + m]) m++;
            emit_literal(lit, i);
            uint64_t left = m;
            const uint64_t off = i - cand;
            while (left) {   // copies of at most 64 bytes with a 2-byte offset
                uint64_t l = left < 64 ? left : 64;
                if (left - l > 0 && left - l < 4) l = left - 4;   // keep the remainder >= 4
                out.push_back((uint8_t)(2 | ((l - 1) << 2)));
                out.push_back((uint8_t)(off & 0xFF));
                out.push_back((uint8_t)(off >> 8));
                left -= l;
            }
            i += m;
            lit = i;
        } else {
            i++;
        }
    }
    emit_literal(lit, n);
}

// greedy LZ4 compressor: one frame (version 1, linked blocks of at most 64 KiB, no checksums, header
// checksum byte left 0 — decoders that verify it are not the target of benchmark data)
void lz4_compress_frame(const std::vector<uint8_t> &in, std::vector<uint8_t> &out)
{
    out.clear();
    const uint8_t hdr[7] = {0x04, 0x22, 0x4D, 0x18, 0x40, 0x40, 0x00};
    out.insert(out.end(), hdr, hdr + 7);
    auto put_len = [&](uint64_t v) {
        while (v >= 255) { out.push_back(255); v -= 255; }
        out.push_back((uint8_t)v);
    };
    std::vector<uint32_t> table(1u << 14);
    const uint64_t n = in.size();
    for (uint64_t b0 = 0; b0 < n; b0 += 65536) {
        const uint64_t bend = b0 + 65536 < n ? b0 + 65536 : n;
        const size_t size_at = out.size();
        out.insert(out.end(), 4, 0);
        std::fill(table.begin(), table.end(), 0xFFFFFFFFu);
        uint64_t i = b0, lit = b0;
        const uint64_t limit = bend > 12 + b0 ? bend - 12 : b0;
        while (i < limit) {
            uint32_t w;
            memcpy(&w, &in[i], 4);
            const uint32_t h = (w * 2654435761u) >> 18;
            const uint32_t cand = table[h];
            table[h] = (uint32_t)i;
            if (cand != 0xFFFFFFFFu && i - cand <= 65535 && memcmp(&in[cand], &in[i], 4) == 0) {
                uint64_t m = 4;
                while (i + m < bend - 5 && in[cand + m] == in[i + m]) m++;
                const uint64_t ll = i - lit;
                out.push_back((uint8_t)(((ll < 15 ? ll : 15) << 4) | (m - 4 < 15 ? m - 4 : 15)));
                if (ll >= 15) put_len(ll - 15);
                out.insert(out.end(), in.begin() + lit, in.begin() + i);
                out.push_back((uint8_t)((i - cand) & 0xFF));
                out.push_back((uint8_t)((i - cand) >> 8));
                if (m - 4 >= 15) put_len(m - 4 - 15);
                i += m;
                lit = i;
            } else {
                i++;
            }
        }
        const uint64_t ll = bend - lit;
        out.push_back((uint8_t)((ll < 15 ? ll : 15) << 4));
        if (ll >= 15) put_len(ll - 15);
        out.insert(out.end(), in.begin() + lit, in.begin() + bend);
        const uint32_t sz = (uint32_t)(out.size() - size_at - 4);
        out[size_at] = (uint8_t)sz; out[size_at + 1] = (uint8_t)(sz >> 8);
        out[size_at + 2] = (uint8_t)(sz >> 16); out[size_at + 3] = (uint8_t)(sz >> 24);
    }
    out.insert(out.end(), 4, 0); // end mark
}

} // namespace

extern "C" int kta_kafka_encode_synth_host(const kta_synth_spec *spec, uint64_t first, uint64_t n, uint32_t records_per_batch,
                                uint8_t *out, uint64_t cap, uint64_t *len)
{
    return kta_kafka_encode_synth_host_ex(spec, first, n, records_per_batch, 0, out, cap, len);
}

extern "C" int kta_kafka_encode_synth_host_ex(const kta_synth_spec *spec, uint64_t first, uint64_t n, uint32_t records_per_batch,
                                   int codec, uint8_t *out, uint64_t cap, uint64_t *len)
{
    const bool patterned = codec != 0;   // 0x100: uncompressed, but with the value pattern of the compressed forms
    const bool text = (codec & 0x200) != 0;   // 0x200: values of words and punctuation (skewed bytes, many short copies) instead
    codec &= 0xFF;
    if (!spec || !len || records_per_batch == 0 || (codec != 0 && codec != 2 && codec != 3)) return KTA_ERR_INVALID;
    std::vector<uint8_t> packed;
    uint64_t pos = 0;
    bool fits = true;
    std::vector<uint8_t> rec; // one batch's records
    for (uint64_t b0 = 0; b0 < n; b0 += records_per_batch) {
        const uint32_t cnt = (uint32_t)((n - b0) < records_per_batch ? (n - b0) : records_per_batch);
        rec.clear();
        int64_t base_ts = 0, max_ts = INT64_MIN;
        for (uint32_t j = 0; j < cnt; j++) {
            int32_t p, kl, vl;
            int64_t ts;
            kta_synth_record(spec, first + b0 + j, &p, &kl, &vl, &ts);
            if (j == 0) base_ts = ts;
            if (ts > max_ts) max_ts = ts;
            uint8_t hdr[48];
            size_t h = 0;
            hdr[h++] = 0;                                   // record attributes
            h += put_varint(hdr + h, ts - base_ts);
            h += put_varint(hdr + h, (int64_t)j);
            h += put_varint(hdr + h, kl);
            const size_t klb = kl > 0 ? (size_t)kl : 0, vlb = vl > 0 ? (size_t)vl : 0;
            uint8_t vh[12];
            const size_t vhn = put_varint(vh, vl);
            const size_t body = h + klb + vhn + vlb + 1;    // + headersCount (0)
            uint8_t lh[12];
            const size_t lhn = put_varint(lh, (int64_t)body);
            const size_t at = rec.size();
            rec.resize(at + lhn + body, 0);                 // values stay zero-filled
            uint8_t *q = rec.data() + at;
            memcpy(q, lh, lhn); q += lhn;
            memcpy(q, hdr, h); q += h;
            if (klb) {
                const uint64_t kid = (uint64_t)kta_synth_key_id(spec, first + b0 + j);
                for (size_t x = 0; x < klb; x++) q[x] = kta_synth_key_byte(spec, kid, (uint32_t)x);
                q += klb;
            }
            memcpy(q, vh, vhn);                             // value bytes + the 0 headersCount are already zero
            if (text && vlb) {                              // words from a small vocabulary, numbers, JSON-like punctuation
                static const char *const words[32] = {"user", "action", "click", "view", "purchase", "session", "timestamp", "amount",
                                                      "currency", "EUR", "USD", "status", "ok", "error", "retry", "region", "eu-west",
                                                      "us-east", "device", "mobile", "desktop", "browser", "version", "payload", "items",
                                                      "price", "quantity", "id", "name", "true", "false", "null"};
                uint64_t h = kta_mix64(first + b0 + j);
                size_t x = 0;
                while (x < vlb) {
                    h = kta_mix64(h);
                    const char *w = words[h & 31];
                    if (((h >> 5) & 7) == 0) {              // a number now and then
                        char num[24];
                        const int nn = snprintf(num, sizeof num, "%u", (unsigned)((h >> 8) % 100000));
                        for (int c = 0; c < nn && x < vlb; c++) q[vhn + x++] = (uint8_t)num[c];
                    } else {
                        for (; *w && x < vlb; w++) q[vhn + x++] = (uint8_t)*w;
                    }
                    static const char seps[8] = {'"', ':', ',', ' ', '{', '}', '"', '_'};
                    if (x < vlb) q[vhn + x++] = (uint8_t)seps[(h >> 24) & 7];
                }
            } else if (patterned && vlb) {                  // a periodic pattern: compressible, but with real copies
                const uint64_t seed = kta_mix64(first + b0 + j);
                for (size_t x = 0; x < vlb; x++) q[vhn + x] = (uint8_t)(kta_mix64(seed + (x % 24)) >> 7);
            }
        }
        if (codec != 0) {
            if (codec == 2) snappy_compress(rec, packed);
            else lz4_compress_frame(rec, packed);
            rec.swap(packed);
        }
        const uint64_t total = KTA_KAFKA_BATCH_HEADER + rec.size();
        if (out && pos + total <= cap) {
            uint8_t *h = out + pos;
            put_be(h, first + b0, 8);                       // baseOffset
            put_be(h + 8, total - 12, 4);                   // batchLength
            put_be(h + 12, 0, 4);                           // partitionLeaderEpoch
            h[16] = 2;                                      // magic
            put_be(h + 21, (uint64_t)codec, 2);             // attributes: CreateTime, codec
            put_be(h + 23, cnt - 1, 4);                     // lastOffsetDelta
            put_be(h + 27, (uint64_t)base_ts, 8);
            put_be(h + 35, (uint64_t)max_ts, 8);
            put_be(h + 43, ~0ull, 8);                       // producerId -1
            put_be(h + 51, 0xFFFF, 2);                      // producerEpoch -1
            put_be(h + 53, 0xFFFFFFFFull, 4);               // baseSequence -1
            put_be(h + 57, cnt, 4);                         // recordsCount
            memcpy(h + KTA_KAFKA_BATCH_HEADER, rec.data(), rec.size());
            put_be(h + 17, crc32c(h + 21, total - 21), 4);
        } else if (out) {
            fits = false;
        }
        pos += total;
    }
    *len = pos;
    return (!out || fits) ? KTA_OK : KTA_ERR_CAPACITY;
}


extern "C" uint32_t kta_crc32c_host(const uint8_t *bytes, uint64_t len) { return bytes ? crc32c(bytes, (size_t)len) : 0u; }
