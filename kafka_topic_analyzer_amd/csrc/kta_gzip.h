// kta_gzip.h — gzip inflate for compressed Kafka record batches (attributes codec 1).
// Same code on the host (index: size; CPU tests against zlib's output) and on the device (one lane per
// batch: DEFLATE is bit-serial, and a fetch carries thousands of independent batches).
//
// Container (RFC 1952): 1f 8b | CM = 8 | FLG | MTIME 4 | XFL | OS | [FEXTRA: u16 len + bytes] |
//   [FNAME: zero terminated] | [FCOMMENT: zero terminated] | [FHCRC: 2] | DEFLATE stream | CRC-32 4 |
//   ISIZE 4 (uncompressed length mod 2^32, little endian).  Kafka producers write ONE member per
//   batch (librdkafka: deflateInit2 with windowBits 15+16; Java: GZIPOutputStream); a second member
//   makes ISIZE disagree with the output and the batch is reported, not mis-decoded.  The CRC-32 is not
//   verified (the batch CRC-32C covers the compressed bytes).
// DEFLATE (RFC 1951): blocks of BFINAL 1 bit | BTYPE 2 bits: 00 stored (LEN, ~LEN, bytes), 01 fixed
//   Huffman codes, 10 dynamic codes (HLIT, HDIST, HCLEN, code-length code in the order 16 17 18 0 8 7 9 6
//   10 5 11 4 12 3 13 2 14 1 15, run-length coded code lengths), 11 invalid.  Codes are canonical and are
//   packed starting from the least significant bit; they are decoded here the way section 3.2.2
//   describes them: per code length the number of codes and the symbols in code order.
//   Literal/length symbols 257..285 and distance symbols 0..29 carry extra bits; their bases follow the
//   closed forms below instead of tables.  Matches reach at most 32 KiB back and may overlap themselves.
#pragma once

#include <stdint.h>

#if defined(__HIPCC__)
#define KTA_GZIP_HD __host__ __device__ inline
#else
#define KTA_GZIP_HD inline
#endif

namespace kta {

struct GzBits {
    const uint8_t *p;
    uint64_t n, pos;
    uint64_t hold;
    uint32_t bits;
    bool overrun;   // read past the end of the input
};

KTA_GZIP_HD uint32_t gz_get(GzBits &b, uint32_t k)   // k <= 16
{
    while (b.bits < k) {
        if (b.pos < b.n) b.hold |= (uint64_t)b.p[b.pos++] << b.bits;
        else b.overrun = true;
        b.bits += 8;
    }
    const uint32_t v = (uint32_t)(b.hold & ((1u << k) - 1u));
    b.hold >>= k;
    b.bits -= k;
    return v;
}

// canonical code: count[len] codes of each length, symbols ordered by (length, symbol value)
template <int NSYM> struct GzHuff {
    uint16_t count[16];
    uint16_t symbol[NSYM];
};

// Returns 0 for a complete code, > 0 for an incomplete one (left-over code space), < 0 if over-subscribed.
template <int NSYM> KTA_GZIP_HD int gz_build(GzHuff<NSYM> &h, const uint8_t *lengths, int n)
{
    for (int l = 0; l < 16; l++) h.count[l] = 0;
    for (int s = 0; s < n; s++) h.count[lengths[s]]++;
    if (h.count[0] == n) return 0;   // no codes at all: legal for the distance code of a literal-only block
    int left = 1;
    for (int l = 1; l < 16; l++) {
        left = (left << 1) - (int)h.count[l];
        if (left < 0) return left;
    }
    uint16_t offs[16];
    offs[1] = 0;
    for (int l = 1; l < 15; l++) offs[l + 1] = (uint16_t)(offs[l] + h.count[l]);
    for (int s = 0; s < n; s++)
        if (lengths[s]) h.symbol[offs[lengths[s]]++] = (uint16_t)s;
    return left;
}

// One symbol: walk the code lengths; `first` is the first code of the current length, `index` the
// position of its symbol.  -1: no such code (or the input ran out).
template <int NSYM> KTA_GZIP_HD int gz_decode(GzBits &b, const GzHuff<NSYM> &h)
{
    int code = 0, first = 0, index = 0;
    for (int l = 1; l < 16; l++) {
        code |= (int)gz_get(b, 1);
        const int c = h.count[l];
        if (code - c < first) return h.symbol[index + (code - first)];
        index += c;
        first = (first + c) << 1;
        code <<= 1;
    }
    return -1;
}

// Skips the member header; returns the offset of the DEFLATE stream or 0 if this is not a gzip member.
KTA_GZIP_HD uint64_t gzip_header(const uint8_t *p, uint64_t n)
{
    if (n < 18 || p[0] != 0x1f || p[1] != 0x8b || p[2] != 8 || (p[3] & 0xE0)) return 0;
    const uint32_t flg = p[3];
    uint64_t pos = 10;
    if (flg & 4u) {                                   // FEXTRA
        if (pos + 2 > n) return 0;
        pos += 2 + ((uint64_t)p[pos] | ((uint64_t)p[pos + 1] << 8));
    }
    for (uint32_t f = 8u; f <= 16u; f <<= 1)          // FNAME, FCOMMENT: zero terminated
        if (flg & f) {
            while (pos < n && p[pos]) pos++;
            pos++;
        }
    if (flg & 2u) pos += 2;                           // FHCRC
    return pos + 8 <= n ? pos : 0;
}

// Uncompressed size the trailer announces; -1 if this is not a gzip member.
KTA_GZIP_HD int64_t gzip_uncompressed_len(const uint8_t *p, uint64_t n)
{
    if (!gzip_header(p, n)) return -1;
    return (int64_t)((uint64_t)p[n - 4] | ((uint64_t)p[n - 3] << 8) | ((uint64_t)p[n - 2] << 16) | ((uint64_t)p[n - 1] << 24));
}

// The symbols of one compressed block.  Returns false on malformed input.
KTA_GZIP_HD bool gz_codes(GzBits &b, const GzHuff<288> &lencode, const GzHuff<30> &distcode, uint8_t *dst, uint64_t &op,
                          uint64_t cap)
{
    while (true) {
        int sym = gz_decode(b, lencode);
        if (sym < 0 || b.overrun) return false;
        if (sym < 256) {
            if (op >= cap) return false;
            dst[op++] = (uint8_t)sym;
            continue;
        }
        if (sym == 256) return true;                  // end of block
        if (sym > 285) return false;
        uint32_t len;
        if (sym < 265) len = 3u + (uint32_t)(sym - 257);
        else if (sym == 285) len = 258u;
        else {
            const uint32_t k = (uint32_t)sym - 261u, e = k >> 2;               // 265..284: 1..5 extra bits
            len = 3u + ((4u + (k & 3u)) << e) + gz_get(b, e);
        }
        const int ds = gz_decode(b, distcode);
        if (ds < 0 || ds > 29) return false;
        uint32_t dist;
        if (ds < 4) dist = 1u + (uint32_t)ds;
        else {
            const uint32_t e = ((uint32_t)ds >> 1) - 1u;                       // 1..13 extra bits
            dist = 1u + ((2u + ((uint32_t)ds & 1u)) << e) + gz_get(b, e);
        }
        if (b.overrun || dist > op || op + len > cap) return false;
        for (uint32_t k = 0; k < len; k++) dst[op + k] = dst[op - dist + k];    // may overlap itself
        op += len;
    }
}

// Inflates one gzip member into dst[0 .. cap).  Returns the bytes produced or -1.
KTA_GZIP_HD int64_t gzip_inflate(const uint8_t *src, uint64_t n, uint8_t *dst, uint64_t cap)
{
    const uint64_t start = gzip_header(src, n);
    if (!start) return -1;
    GzBits b{src, n - 8, start, 0, 0, false};         // the trailer is not part of the DEFLATE stream
    GzHuff<288> lencode;
    GzHuff<30> distcode;
    uint8_t lengths[320];
    uint64_t op = 0;
    uint32_t last;
    do {
        last = gz_get(b, 1);
        const uint32_t type = gz_get(b, 2);
        if (b.overrun) return -1;
        if (type == 0) {                              // stored: byte aligned LEN, ~LEN, bytes
            b.hold = 0;
            b.bits = 0;
            if (b.pos + 4 > b.n) return -1;
            const uint32_t len = (uint32_t)src[b.pos] | ((uint32_t)src[b.pos + 1] << 8);
            const uint32_t nlen = (uint32_t)src[b.pos + 2] | ((uint32_t)src[b.pos + 3] << 8);
            b.pos += 4;
            if ((len ^ 0xFFFFu) != nlen || b.pos + len > b.n || op + len > cap) return -1;
            for (uint32_t k = 0; k < len; k++) dst[op + k] = src[b.pos + k];
            op += len;
            b.pos += len;
        } else if (type == 1) {                       // fixed codes (RFC 1951 3.2.6)
            for (int s = 0; s < 288; s++) lengths[s] = (uint8_t)(s < 144 ? 8 : (s < 256 ? 9 : (s < 280 ? 7 : 8)));
            (void)gz_build(lencode, lengths, 288);
            for (int s = 0; s < 30; s++) lengths[s] = 5;
            (void)gz_build(distcode, lengths, 30);
            if (!gz_codes(b, lencode, distcode, dst, op, cap)) return -1;
        } else if (type == 2) {                       // dynamic codes (3.2.7)
            const uint32_t nlen = gz_get(b, 5) + 257, ndist = gz_get(b, 5) + 1, ncode = gz_get(b, 4) + 4;
            if (nlen > 286 || ndist > 30) return -1;
            const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
            for (uint32_t i = 0; i < 19; i++) lengths[order[i]] = (uint8_t)(i < ncode ? gz_get(b, 3) : 0);
            GzHuff<30> &clcode = distcode;            // the code-length code (19 symbols) borrows the storage
            if (gz_build(clcode, lengths, 19) != 0) return -1;
            uint32_t i = 0;
            while (i < nlen + ndist) {
                const int sym = gz_decode(b, clcode);
                if (sym < 0 || b.overrun) return -1;
                if (sym < 16) {
                    lengths[i++] = (uint8_t)sym;
                    continue;
                }
                uint32_t prev = 0, rep;
                if (sym == 16) {
                    if (i == 0) return -1;
                    prev = lengths[i - 1];
                    rep = 3 + gz_get(b, 2);
                } else if (sym == 17) rep = 3 + gz_get(b, 3);
                else rep = 11 + gz_get(b, 7);
                if (i + rep > nlen + ndist) return -1;
                while (rep--) lengths[i++] = (uint8_t)prev;
            }
            if (lengths[256] == 0) return -1;         // no end-of-block code
            int left = gz_build(lencode, lengths, (int)nlen);
            if (left < 0 || (left > 0 && (uint32_t)(lencode.count[0] + lencode.count[1]) != nlen)) return -1;   // incomplete: only a single 1-bit code
            left = gz_build(distcode, lengths + nlen, (int)ndist);
            if (left < 0 || (left > 0 && (uint32_t)(distcode.count[0] + distcode.count[1]) != ndist)) return -1;
            if (!gz_codes(b, lencode, distcode, dst, op, cap)) return -1;
        } else {
            return -1;
        }
    } while (!last);
    const uint64_t isize = (uint64_t)src[n - 4] | ((uint64_t)src[n - 3] << 8) | ((uint64_t)src[n - 2] << 16) | ((uint64_t)src[n - 1] << 24);
    if (b.pos != b.n || (op & 0xFFFFFFFFull) != isize) return -1;   // a second member, trailing bytes, corrupt trailer
    return (int64_t)op;
}

}  // namespace kta
