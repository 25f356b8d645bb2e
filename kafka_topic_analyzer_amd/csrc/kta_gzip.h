// kta_gzip.h — gzip inflate for compressed Kafka record batches (attributes codec 1).
// Same code on the host (index: size; CPU tests against zlib's output) and on the device (one lane per
// batch: DEFLATE is bit-serial, and a fetch carries thousands of independent batches).
//
// Container (RFC 1952): 1f 8b | CM = 8 | FLG | MTIME 4 | XFL | OS | [FEXTRA: u16 len + bytes] |
//   [FNAME: zero terminated] | [FCOMMENT: zero terminated] | [FHCRC: 2] | DEFLATE stream | CRC-32 4 |
//   ISIZE 4 (uncompressed length mod 2^32, little endian).  Kafka producers write ONE member per
//   batch (librdkafka: deflateInit2 with windowBits 15+16; Java: GZIPOutputStream); a second member
//   is reported, not mis-decoded.  The CRC-32 is not verified (the batch CRC-32C covers the compressed
//   bytes).
// DEFLATE (RFC 1951): blocks of BFINAL 1 bit | BTYPE 2 bits: 00 stored (LEN, ~LEN, bytes), 01 fixed
//   Huffman codes, 10 dynamic codes (HLIT, HDIST, HCLEN, code-length code in the order 16 17 18 0 8 7 9 6
//   10 5 11 4 12 3 13 2 14 1 15, run-length coded code lengths), 11 invalid.  Codes are canonical
//   (section 3.2.2) and packed starting from the least significant bit of the stream.
//   Literal/length symbols 257..285 and distance symbols 0..29 carry extra bits; their bases follow the
//   closed forms below instead of tables.  Matches reach at most 32 KiB back and may overlap themselves.
//
// Decoding a symbol: peek 15 bits, reverse them (the first stream bit becomes the most significant), and
// find the code length l with  rev15 < limit[l],  limit[l] = (first code of length l + number of codes of
// length l) << (15 - l)  — the canonical codes of increasing length occupy increasing, adjacent ranges of
// the 15-bit space.  The symbol is  symbol[base[l] + (rev15 >> (15 - l))],  base[l] = index of the first
// symbol of length l minus its first code.  limit/base are 2 x 16 small integers that the device keeps in
// registers (statically indexed); everything indexed by data (symbol lists, code lengths, counts) lives in
// a caller-provided work area of GZ_WORK u16 words with a stride, which is LDS on the device
// (element i of lane t at work[i * stride + t]) and a local array on the host (stride 1).
#pragma once

#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define KTA_GZIP_HD __host__ __device__ inline __attribute__((always_inline))   // one call site per kernel: the LDS
                                                                              // address space of `work` is then known
#else
#define KTA_GZIP_HD inline
#endif

namespace kta {

// work area layout (u16 words): counts | literal/length symbols | distance symbols | code lengths
constexpr uint32_t GZ_W_COUNT = 0;      // 16
constexpr uint32_t GZ_W_LSYM = 16;      // 288
constexpr uint32_t GZ_W_DSYM = 304;     // 32
constexpr uint32_t GZ_W_LENS = 336;     // 320
constexpr uint32_t GZ_WORK = 656;

struct GzBits {
    const uint8_t *p;
    uint64_t n, pos;     // pos: next byte to load into `hold` (may run past n: zeros)
    uint64_t hold;
    uint32_t bits;
    bool overrun;        // consumed bits that were not there
};

// at least 32 valid bits in hold (zeros past the end of the input)
KTA_GZIP_HD void gz_refill(GzBits &b)
{
    if (b.bits >= 32) return;
    if (b.pos + 4 <= b.n) {
        uint32_t w;
        __builtin_memcpy(&w, b.p + b.pos, 4);
        b.hold |= (uint64_t)w << b.bits;
        b.pos += 4;
        b.bits += 32;
        return;
    }
    while (b.bits < 32) {
        if (b.pos < b.n) b.hold |= (uint64_t)b.p[b.pos] << b.bits;
        b.pos++;
        b.bits += 8;
    }
}

KTA_GZIP_HD void gz_drop(GzBits &b, uint32_t k)
{
    b.hold >>= k;
    b.bits -= k;
    // bytes beyond the input were counted in pos: some were consumed iff fewer than that many bits remain
    if (b.pos > b.n && b.bits < 8 * (uint32_t)(b.pos - b.n)) b.overrun = true;
}

KTA_GZIP_HD uint32_t gz_get(GzBits &b, uint32_t k)   // k <= 16
{
    gz_refill(b);
    const uint32_t v = (uint32_t)b.hold & ((1u << k) - 1u);
    gz_drop(b, k);
    return v;
}

struct GzCode {            // statically indexed only: registers on the device
    uint32_t limit[16];    // limit[l] for l = 1..15 (limit[0] unused)
    int32_t base[16];
};

KTA_GZIP_HD uint32_t gz_rev15(uint32_t v)
{
#if defined(__clang__)
    return __builtin_bitreverse32(v) >> 17;
#else
    v = ((v & 0x5555u) << 1) | ((v >> 1) & 0x5555u);
    v = ((v & 0x3333u) << 2) | ((v >> 2) & 0x3333u);
    v = ((v & 0x0F0Fu) << 4) | ((v >> 4) & 0x0F0Fu);
    v = ((v & 0x00FFu) << 8) | ((v >> 8) & 0x00FFu);   // 16-bit reversal
    return v >> 1;
#endif
}

// Builds the code of `n` symbols whose lengths are at work[(lens + s) * stride]; the symbols go to
// work[(syms + i) * stride].  Returns 0 for a complete code, > 0 for an incomplete one, < 0 if over-subscribed.
KTA_GZIP_HD int gz_build(GzCode &c, uint16_t *work, uint32_t stride, uint32_t lens, uint32_t syms, int n)
{
    uint16_t *count = work + GZ_W_COUNT * stride;
    for (int l = 0; l < 16; l++) count[l * stride] = 0;
    for (int s = 0; s < n; s++) count[(work[(lens + s) * stride] & 15u) * stride]++;
    int left = 1;
    uint32_t first = 0, index = 0;
    bool over = false;
#pragma unroll
    for (int l = 1; l < 16; l++) {
        const uint32_t cnt = count[l * stride];
        if (!over) {
            left = (left << 1) - (int)cnt;
            if (left < 0) over = true;
        }
        c.limit[l] = over ? 0u : (first + cnt) << (15 - l);
        c.base[l] = (int32_t)index - (int32_t)first;
        count[l * stride] = (uint16_t)index;           // becomes the running offset of this length's symbols
        index += cnt;
        first = (first + cnt) << 1;
    }
    if (over) return -1;
    for (int s = 0; s < n; s++) {
        const uint32_t l = work[(lens + s) * stride] & 15u;
        if (l) work[(syms + count[l * stride]++) * stride] = (uint16_t)s;
    }
    return index == 0 ? 0 : left;                       // no codes at all: legal for an unused distance code
}

// One symbol; -1 if the bits are no code of `c`.
KTA_GZIP_HD int gz_decode(GzBits &b, const GzCode &c, const uint16_t *work, uint32_t stride, uint32_t syms)
{
    gz_refill(b);
    const uint32_t rev = gz_rev15((uint32_t)b.hold & 0x7FFFu);
    uint32_t l = 0;
    int32_t idx = 0;
#pragma unroll
    for (int k = 15; k >= 1; k--)
        if (rev < c.limit[k]) {                        // the smallest such length wins (limits never decrease)
            l = (uint32_t)k;
            idx = c.base[k] + (int32_t)(rev >> (15 - k));
        }
    if (!l) return -1;
    gz_drop(b, l);
    return work[(syms + (uint32_t)idx) * stride];
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

// dst[op .. op+len) = dst[op-dist ..): 8 bytes per step when the source does not overlap the step, a
// register-held period otherwise (dist < 8), so a copy costs few dependent memory round trips.
KTA_GZIP_HD void gz_copy_match(uint8_t *dst, uint64_t op, uint32_t dist, uint32_t len)
{
    uint32_t k = 0;
    if (dist >= 8) {
        for (; k + 8 <= len; k += 8) {
            uint64_t w;
            __builtin_memcpy(&w, dst + op - dist + k, 8);
            __builtin_memcpy(dst + op + k, &w, 8);
        }
        for (; k < len; k++) dst[op + k] = dst[op - dist + k];
        return;
    }
    uint64_t pat = 0;
    for (uint32_t i = 0; i < dist; i++) pat |= (uint64_t)dst[op - dist + i] << (8 * i);
    uint32_t ph = 0;
    for (; k < len; k++) {
        dst[op + k] = (uint8_t)(pat >> (8 * ph));
        ph = ph + 1 == dist ? 0 : ph + 1;
    }
}

// The symbols of one compressed block.  Returns false on malformed input.
KTA_GZIP_HD bool gz_codes(GzBits &b, const GzCode &lencode, const GzCode &distcode, const uint16_t *work, uint32_t stride,
                          uint8_t *dst, uint64_t &op, uint64_t cap)
{
    while (true) {
        const int sym = gz_decode(b, lencode, work, stride, GZ_W_LSYM);
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
        const int ds = gz_decode(b, distcode, work, stride, GZ_W_DSYM);
        if (ds < 0 || ds > 29) return false;
        uint32_t dist;
        if (ds < 4) dist = 1u + (uint32_t)ds;
        else {
            const uint32_t e = ((uint32_t)ds >> 1) - 1u;                       // 1..13 extra bits
            dist = 1u + ((2u + ((uint32_t)ds & 1u)) << e) + gz_get(b, e);
        }
        if (b.overrun || dist > op || op + len > cap) return false;
        gz_copy_match(dst, op, dist, len);
        op += len;
    }
}

// Inflates one gzip member into dst[0 .. cap).  `work`: GZ_WORK u16 words with `stride` (see above).
// Returns the bytes produced or -1.
KTA_GZIP_HD int64_t gzip_inflate(const uint8_t *src, uint64_t n, uint8_t *dst, uint64_t cap, uint16_t *work, uint32_t stride)
{
    const uint64_t start = gzip_header(src, n);
    if (!start) return -1;
    GzBits b{src + start, n - 8 - start, 0, 0, 0, false};   // the trailer is not part of the DEFLATE stream
    GzCode lencode, distcode;
    uint64_t op = 0;
    uint32_t last;
    do {
        last = gz_get(b, 1);
        const uint32_t type = gz_get(b, 2);
        if (b.overrun) return -1;
        if (type == 0) {                              // stored: byte aligned LEN, ~LEN, bytes
            const uint64_t at = b.pos - b.bits / 8;   // whole bytes still in `hold` are handed back
            b.hold = 0;
            b.bits = 0;
            if (at + 4 > b.n) return -1;
            const uint32_t len = (uint32_t)b.p[at] | ((uint32_t)b.p[at + 1] << 8);
            const uint32_t nlen = (uint32_t)b.p[at + 2] | ((uint32_t)b.p[at + 3] << 8);
            if ((len ^ 0xFFFFu) != nlen || at + 4 + len > b.n || op + len > cap) return -1;
            for (uint32_t k = 0; k < len; k++) dst[op + k] = b.p[at + 4 + k];
            op += len;
            b.pos = at + 4 + len;
        } else if (type == 1) {                       // fixed codes (RFC 1951 3.2.6)
            for (uint32_t s = 0; s < 288; s++)
                work[(GZ_W_LENS + s) * stride] = (uint16_t)(s < 144 ? 8 : (s < 256 ? 9 : (s < 280 ? 7 : 8)));
            (void)gz_build(lencode, work, stride, GZ_W_LENS, GZ_W_LSYM, 288);
            for (uint32_t s = 0; s < 30; s++) work[(GZ_W_LENS + s) * stride] = 5;
            (void)gz_build(distcode, work, stride, GZ_W_LENS, GZ_W_DSYM, 30);
            if (!gz_codes(b, lencode, distcode, work, stride, dst, op, cap)) return -1;
        } else if (type == 2) {                       // dynamic codes (3.2.7)
            const uint32_t nlen = gz_get(b, 5) + 257, ndist = gz_get(b, 5) + 1, ncode = gz_get(b, 4) + 4;
            if (nlen > 286 || ndist > 30) return -1;
            const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
            for (uint32_t i = 0; i < 19; i++)
                work[(GZ_W_LENS + order[i]) * stride] = (uint16_t)(i < ncode ? gz_get(b, 3) : 0);
            GzCode &clcode = distcode;                // the code-length code (19 symbols) borrows the storage
            if (gz_build(clcode, work, stride, GZ_W_LENS, GZ_W_DSYM, 19) != 0) return -1;
            uint32_t i = 0;
            while (i < nlen + ndist) {
                const int sym = gz_decode(b, clcode, work, stride, GZ_W_DSYM);
                if (sym < 0 || b.overrun) return -1;
                if (sym < 16) {
                    work[(GZ_W_LENS + i++) * stride] = (uint16_t)sym;
                    continue;
                }
                uint32_t prev = 0, rep;
                if (sym == 16) {
                    if (i == 0) return -1;
                    prev = work[(GZ_W_LENS + i - 1) * stride];
                    rep = 3 + gz_get(b, 2);
                } else if (sym == 17) rep = 3 + gz_get(b, 3);
                else rep = 11 + gz_get(b, 7);
                if (i + rep > nlen + ndist) return -1;
                while (rep--) work[(GZ_W_LENS + i++) * stride] = (uint16_t)prev;
            }
            if (work[(GZ_W_LENS + 256) * stride] == 0) return -1;      // no end-of-block code
            // an incomplete code is legal only as a single 1-bit code
            uint32_t zeros = 0, ones = 0;
            for (uint32_t s = 0; s < nlen; s++) {
                const uint32_t l = work[(GZ_W_LENS + s) * stride];
                zeros += l == 0;
                ones += l == 1;
            }
            int left = gz_build(lencode, work, stride, GZ_W_LENS, GZ_W_LSYM, (int)nlen);
            if (left < 0 || (left > 0 && zeros + ones != nlen)) return -1;
            zeros = ones = 0;
            for (uint32_t s = 0; s < ndist; s++) {
                const uint32_t l = work[(GZ_W_LENS + nlen + s) * stride];
                zeros += l == 0;
                ones += l == 1;
            }
            left = gz_build(distcode, work, stride, GZ_W_LENS + nlen, GZ_W_DSYM, (int)ndist);
            if (left < 0 || (left > 0 && zeros + ones != ndist)) return -1;
            if (!gz_codes(b, lencode, distcode, work, stride, dst, op, cap)) return -1;
        } else {
            return -1;
        }
    } while (!last);
    // the stream must end right before the trailer (else: a second member / trailing bytes) and produce ISIZE
    const uint64_t used = b.pos - b.bits / 8;
    const uint64_t isize = (uint64_t)src[n - 4] | ((uint64_t)src[n - 3] << 8) | ((uint64_t)src[n - 2] << 16) | ((uint64_t)src[n - 1] << 24);
    if (b.overrun || used != b.n || (op & 0xFFFFFFFFull) != isize) return -1;
    return (int64_t)op;
}

}  // namespace kta
