// kta_snappy.h — Snappy inflate for compressed Kafka record batches (attributes codec 2).
// Same code on the host (index: sizes; CPU tests) and on the device (one lane per batch).
//
// Format (google/snappy format_description.txt): a block is a little-endian base-128 preamble with
// the uncompressed length, then elements tagged by the low 2 bits of their first byte:
//   00 literal   len-1 in the upper 6 bits (< 60) or in the next 1..4 bytes (60..63)
//   01 copy      len = 4 + ((tag >> 2) & 7), offset = ((tag >> 5) << 8) | next byte
//   10 copy      len = 1 + (tag >> 2), offset = next 2 bytes (LE)
//   11 copy      len = 1 + (tag >> 2), offset = next 4 bytes (LE)
// Copies may overlap their own output (offset < len repeats a pattern).
// Kafka's Java clients wrap the blocks in snappy-java's stream framing ("\x82SNAPPY\0", version, compat
// version, then [u32 BE length][block]...); librdkafka writes one bare block.  Both are accepted.
#pragma once

#include <stdint.h>

#if defined(__HIPCC__)
#define KTA_SNAPPY_HD __host__ __device__ inline
#else
#define KTA_SNAPPY_HD inline
#endif

namespace kta {

KTA_SNAPPY_HD bool snappy_is_xerial(const uint8_t *p, uint64_t n)
{
    return n >= 16 && p[0] == 0x82 && p[1] == 'S' && p[2] == 'N' && p[3] == 'A' && p[4] == 'P' && p[5] == 'P' &&
           p[6] == 'Y' && p[7] == 0;
}

// preamble of one block: returns bytes consumed (0 on error)
KTA_SNAPPY_HD uint32_t snappy_preamble(const uint8_t *p, uint64_t n, uint64_t *len)
{
    uint64_t v = 0;
    for (uint32_t i = 0; i < 5 && i < n; i++) {
        v |= (uint64_t)(p[i] & 0x7F) << (7 * i);
        if (!(p[i] & 0x80)) {
            *len = v;
            return i + 1;
        }
    }
    return 0;
}

// Uncompressed size of a batch payload (bare block or xerial stream); -1 if malformed.
KTA_SNAPPY_HD int64_t snappy_uncompressed_len(const uint8_t *p, uint64_t n)
{
    uint64_t len = 0;
    if (!snappy_is_xerial(p, n)) return snappy_preamble(p, n, &len) ? (int64_t)len : -1;
    uint64_t pos = 16, total = 0;
    while (pos + 4 <= n) {
        const uint64_t clen = ((uint64_t)p[pos] << 24) | ((uint64_t)p[pos + 1] << 16) | ((uint64_t)p[pos + 2] << 8) | p[pos + 3];
        pos += 4;
        if (clen == 0 || pos + clen > n) return -1;
        if (!snappy_preamble(p + pos, clen, &len)) return -1;
        total += len;
        pos += clen;
    }
    return pos == n ? (int64_t)total : -1;
}

// Inflate one block into dst[0, cap); returns bytes produced or -1.
KTA_SNAPPY_HD int64_t snappy_inflate_block(const uint8_t *p, uint64_t n, uint8_t *dst, uint64_t cap)
{
    uint64_t want = 0;
    const uint32_t pre = snappy_preamble(p, n, &want);
    if (!pre || want > cap) return -1;
    uint64_t ip = pre, op = 0;
    while (ip < n) {
        const uint32_t tag = p[ip++];
        uint64_t len, off;
        if ((tag & 3u) == 0u) {                       // literal
            len = tag >> 2;
            if (len >= 60) {
                const uint32_t nb = (uint32_t)len - 59;
                if (ip + nb > n) return -1;
                len = 0;
                for (uint32_t k = 0; k < nb; k++) len |= (uint64_t)p[ip + k] << (8 * k);
                ip += nb;
            }
            len += 1;
            if (ip + len > n || op + len > want) return -1;
            for (uint64_t k = 0; k < len; k++) dst[op + k] = p[ip + k];
            ip += len;
            op += len;
            continue;
        }
        if ((tag & 3u) == 1u) {
            if (ip + 1 > n) return -1;
            len = 4 + ((tag >> 2) & 7u);
            off = ((uint64_t)(tag >> 5) << 8) | p[ip];
            ip += 1;
        } else if ((tag & 3u) == 2u) {
            if (ip + 2 > n) return -1;
            len = 1 + (tag >> 2);
            off = (uint64_t)p[ip] | ((uint64_t)p[ip + 1] << 8);
            ip += 2;
        } else {
            if (ip + 4 > n) return -1;
            len = 1 + (tag >> 2);
            off = (uint64_t)p[ip] | ((uint64_t)p[ip + 1] << 8) | ((uint64_t)p[ip + 2] << 16) | ((uint64_t)p[ip + 3] << 24);
            ip += 4;
        }
        if (off == 0 || off > op || op + len > want) return -1;
        for (uint64_t k = 0; k < len; k++) dst[op + k] = dst[op - off + k];   // byte order makes overlap repeat
        op += len;
    }
    return op == want ? (int64_t)op : -1;
}

// Inflate a batch payload (bare block or xerial stream); returns bytes produced or -1.
KTA_SNAPPY_HD int64_t snappy_inflate(const uint8_t *p, uint64_t n, uint8_t *dst, uint64_t cap)
{
    if (!snappy_is_xerial(p, n)) return snappy_inflate_block(p, n, dst, cap);
    uint64_t pos = 16, op = 0;
    while (pos + 4 <= n) {
        const uint64_t clen = ((uint64_t)p[pos] << 24) | ((uint64_t)p[pos + 1] << 16) | ((uint64_t)p[pos + 2] << 8) | p[pos + 3];
        pos += 4;
        if (clen == 0 || pos + clen > n) return -1;
        const int64_t got = snappy_inflate_block(p + pos, clen, dst + op, cap - op);
        if (got < 0) return -1;
        op += (uint64_t)got;
        pos += clen;
    }
    return pos == n ? (int64_t)op : -1;
}

} // namespace kta
