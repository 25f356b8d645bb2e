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
    uint32_t n, pos;     // pos: next byte to load into `hold` (may run past n: zeros); members are < 2 GiB
    uint64_t hold;
    uint32_t bits;
    bool overrun;        // consumed bits that were not there
};

// The two-stage tokenizer on the device reads the stream through a per-lane LDS window (GZ_WIN bytes, word k
// of this lane at win[k * wstride]): a refill is then an LDS read, and the memory round trips are paid once
// per window by all lanes of a wave together (gz_topup_all) instead of once per word — and the loop that
// stores literals holds no memory loads whose arrival it would have to wait for.
constexpr uint32_t GZ_WIN = 128;

struct GzBitsWin : GzBits {
    uint32_t *win;
    uint32_t wstride;
    uint32_t wbase;      // stream offset of the window's first byte; ~0: nothing loaded
};

KTA_GZIP_HD void gz_reset_window(GzBits &) {}
KTA_GZIP_HD void gz_reset_window(GzBitsWin &b) { b.wbase = ~0u; }

// continue reading at byte `pos` of the stream (also the initial state)
template <class Bits>
KTA_GZIP_HD void gz_seek(Bits &b, uint64_t pos)
{
    b.pos = (uint32_t)pos;
    b.hold = 0;
    b.bits = 0;
    gz_reset_window(b);
}

// the window starts at the current position: words [pos, pos + GZ_WIN) of the stream, clamped to its last word
KTA_GZIP_HD void gz_topup(GzBitsWin &b)
{
    if (b.n < 4) return;
    uint32_t w[GZ_WIN / 4];
#pragma unroll
    for (uint32_t k = 0; k < GZ_WIN / 4; k++) {
        const uint32_t at = b.pos + 4 * k;
        __builtin_memcpy(&w[k], b.p + (at + 4 <= b.n ? at : b.n - 4), 4);
    }
#pragma unroll
    for (uint32_t k = 0; k < GZ_WIN / 4; k++) b.win[k * b.wstride] = w[k];
    b.wbase = b.pos;
}

// the four bytes at pos (pos + 4 <= n)
KTA_GZIP_HD uint32_t gz_word(GzBits &b)
{
    uint32_t w;
    __builtin_memcpy(&w, b.p + b.pos, 4);
    return w;
}

KTA_GZIP_HD uint32_t gz_word(GzBitsWin &b)
{
    if (b.wbase == ~0u || b.pos < b.wbase || b.pos - b.wbase >= GZ_WIN || ((b.pos - b.wbase) & 3u)) gz_topup(b);
    return b.win[((b.pos - b.wbase) >> 2) * b.wstride];
}

// The tokenizer's symbol loop keeps this invariant at the top of every iteration: a lane with a whole word left
// in its stream (pos + 4 <= n) has the bytes [pos, pos + 16) inside its window — enough for the (at most two)
// word refills of one iteration, which then read the window unchecked (gz_word_ready).  On the device the
// check is a vote: when the window of any lane of the wave runs low, all of them take a fresh one — one
// memory round trip for the wave instead of one per lane at its own moment.
KTA_GZIP_HD void gz_topup_all(GzBits &) {}
KTA_GZIP_HD void gz_topup_all(GzBitsWin &b)
{
    const bool low = b.pos + 4 <= b.n && (b.wbase == ~0u || b.pos < b.wbase || b.pos - b.wbase > GZ_WIN - 16);
#if defined(__HIP_DEVICE_COMPILE__)
    if (__builtin_amdgcn_ballot_w64(low)) gz_topup(b);
#else
    if (low) gz_topup(b);
#endif
}

// the four bytes at pos, for a caller that holds the invariant above (anything if pos + 4 > n: not used then)
KTA_GZIP_HD uint32_t gz_word_ready(GzBits &b)
{
    uint32_t w = 0;
    if (b.pos + 4 <= b.n) __builtin_memcpy(&w, b.p + b.pos, 4);
    return w;
}

KTA_GZIP_HD uint32_t gz_word_ready(GzBitsWin &b)
{
    return b.win[(((b.pos - b.wbase) >> 2) & (GZ_WIN / 4 - 1)) * b.wstride];
}

// at least 32 valid bits in hold (zeros past the end of the input)
template <class Bits>
KTA_GZIP_HD void gz_refill(Bits &b)
{
    if (b.bits >= 32) return;
    if (b.pos + 4 <= b.n) {
        b.hold |= (uint64_t)gz_word(b) << b.bits;
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
    if (b.pos > b.n && b.bits < 8 * (b.pos - b.n)) b.overrun = true;
}

template <class Bits>
KTA_GZIP_HD uint32_t gz_get(Bits &b, uint32_t k)   // k <= 16
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
template <class Bits>
KTA_GZIP_HD int gz_decode(Bits &b, const GzCode &c, const uint16_t *work, uint32_t stride, uint32_t syms)
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
template <class Bits>
KTA_GZIP_HD bool gz_codes(Bits &b, const GzCode &lencode, const GzCode &distcode, const uint16_t *work, uint32_t stride,
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
    if (!start || n > 0x7FFFFFFFull) return -1;
    GzBits b;
    b.p = src + start;
    b.n = (uint32_t)(n - 8 - start);                       // the trailer is not part of the DEFLATE stream
    b.overrun = false;
    gz_seek(b, 0);
    GzCode lencode, distcode;
    uint64_t op = 0;
    uint32_t last;
    do {
        last = gz_get(b, 1);
        const uint32_t type = gz_get(b, 2);
        if (b.overrun) return -1;
        if (type == 0) {                              // stored: byte aligned LEN, ~LEN, bytes
            const uint64_t at = b.pos - b.bits / 8;   // whole bytes still in `hold` are handed back
            if (at + 4 > b.n) return -1;
            const uint32_t len = (uint32_t)b.p[at] | ((uint32_t)b.p[at + 1] << 8);
            const uint32_t nlen = (uint32_t)b.p[at + 2] | ((uint32_t)b.p[at + 3] << 8);
            if ((len ^ 0xFFFFu) != nlen || at + 4 + len > b.n || op + len > cap) return -1;
            for (uint32_t k = 0; k < len; k++) dst[op + k] = b.p[at + 4 + k];
            op += len;
            gz_seek(b, at + 4 + len);
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

// ---- two-stage inflate ------------------------------------------------------------------------------------
// The device splits DEFLATE where its two kinds of work separate:
//   stage 1, gzip_tokenize   the bit-serial part, one lane per batch: Huffman decoding.  Literals are stored at
//                            their final positions of dst; a match is not executed but recorded as a token.
//   stage 2                  the copies: one wave per batch executes the tokens 64 bytes per step from an LDS
//                            mirror of the output (kafka_gzip_apply in kta_kafka.hip; gz_apply_tokens below is
//                            the same thing sequentially, for the host tests).
// Token (u32):  literals before the match (0..254, 255 = "255 literals and no match yet") | match length
// (0 = none, else 3..258) << 8 | (distance - 1) << 17.  Every match produces >= 3 bytes and every escape token
// stands for 255 literals, so a member of `out` bytes needs at most out/3 + out/255 + 1 tokens
// (gz_token_bound); the literals after the last match need no token.
//
// Symbols are decoded through primary lookup tables indexed with the next GZ_LBITS / GZ_DBITS stream bits
// (entry: symbol << 4 | code length; 0 = the code is longer than the index, or no code at all: the
// canonical search over the longer lengths takes over).  Work area (u16 words, strided like GZ_WORK): the counts of the
// code being built | literal/length symbols | distance symbols | literal/length table | distance table.  The
// code lengths of a block live in the literal/length table's words until its codes are built (the distance
// code first, then the literal/length code, whose table fill is the last reader of nothing).
constexpr uint32_t GZ_LBITS = 9, GZ_DBITS = 6;
constexpr uint32_t GZ2_W_LSYM = 16;      // 288
constexpr uint32_t GZ2_W_DSYM = 304;     // 32
constexpr uint32_t GZ2_W_LTAB = 336;     // 512 (code lengths: the first 320)
constexpr uint32_t GZ2_W_DTAB = 848;     // 64
constexpr uint32_t GZ2_WORK = 912;

KTA_GZIP_HD uint64_t gz_token_bound(uint64_t out) { return out / 3 + out / 255 + 1; }
// The wave tokenizer (kta_gzip_wave.h) closes the literal run of each of its 64 segments with a token of its own, once per
// region it decodes (a block, or a 7 KiB window's half of one): room for that many more in a member of `clen` compressed
// bytes.  (A member of more regions than this — many tiny blocks — is left to the lane tokenizer, which needs none.)
KTA_GZIP_HD uint64_t gz_closing_tokens(uint64_t clen) { return 64 * (2 + clen / 2048); }

struct GzTokens {
    uint32_t *tok;
    uint64_t n, cap;
    uint64_t run;        // literals since the last token
    bool full;
};

KTA_GZIP_HD void gz_emit(GzTokens &t, uint32_t len, uint32_t dist)
{
    while (t.run >= 255u) {
        if (t.n >= t.cap) { t.full = true; return; }
        t.tok[t.n++] = 255u;
        t.run -= 255u;
    }
    if (t.n >= t.cap) { t.full = true; return; }
    t.tok[t.n++] = (uint32_t)t.run | (len << 8) | ((dist - 1u) << 17);
    t.run = 0;
}

// Fills the primary table of the code gz_build just built (its symbols at work[syms..], the per-length end
// offsets into them still in the count words): entries of `pbits` index bits at work[tab..].
KTA_GZIP_HD void gz_fill(uint16_t *work, uint32_t stride, uint32_t syms, uint32_t tab, uint32_t pbits)
{
    const uint32_t size = 1u << pbits;
    for (uint32_t i = 0; i < size; i++) work[(tab + i) * stride] = 0;
    uint32_t first = 0, idx = 0;
    for (uint32_t l = 1; l <= pbits; l++) {
        const uint32_t end = work[(GZ_W_COUNT + l) * stride];        // symbols of lengths 1..l
        for (; idx < end; idx++, first++) {
            const uint32_t entry = ((uint32_t)work[(syms + idx) * stride] << 4) | l;
            // the stream carries a code starting from its most significant bit: index = the code, bit-reversed
#if defined(__clang__)
            const uint32_t r = __builtin_bitreverse32(first) >> (32 - l);
#else
            uint32_t r = 0;
            for (uint32_t k = 0; k < l; k++) r |= ((first >> k) & 1u) << (l - 1 - k);
#endif
            for (uint32_t j = r; j < size; j += 1u << l) work[(tab + j) * stride] = (uint16_t)entry;
        }
        first <<= 1;
    }
}

// hold >>= k with the over-read flag, without a branch
KTA_GZIP_HD void gz_take(GzBits &b, uint32_t k)
{
    b.hold >>= k;
    b.bits -= k;
    b.overrun = b.overrun | (b.pos > b.n && b.bits < 8u * (b.pos - b.n));
}

// gz_refill for a caller inside the symbol loop (window unchecked, see gz_topup_all)
template <class Bits>
KTA_GZIP_HD void gz_refill_ready(Bits &b)
{
    if (b.bits >= 32) return;
    if (b.pos + 4 <= b.n) {
        b.hold |= (uint64_t)gz_word_ready(b) << b.bits;
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

// One symbol from the bits in hold (>= 15 of them, or the zeros of the stream's end) through the primary table
// of PBITS index bits; a code longer than the index (entry 0) is found by the canonical search over the
// remaining lengths only.  -1 if the bits are no code of `c`.
template <uint32_t PBITS>
KTA_GZIP_HD int gz_lookup(GzBits &b, const GzCode &c, const uint16_t *work, uint32_t stride, uint32_t tab, uint32_t syms)
{
    const uint32_t e = work[(tab + ((uint32_t)b.hold & ((1u << PBITS) - 1u))) * stride];
    uint32_t l = e & 15u;
    int sym = (int)(e >> 4);
    if (!l) {
        const uint32_t rev = gz_rev15((uint32_t)b.hold & 0x7FFFu);
        int32_t idx = 0;
#pragma unroll
        for (int k = 15; k > (int)PBITS; k--)
            if (rev < c.limit[k]) {                        // the smallest such length wins (limits never decrease)
                l = (uint32_t)k;
                idx = c.base[k] + (int32_t)(rev >> (15 - k));
            }
        if (!l) return -1;
        sym = work[(syms + (uint32_t)idx) * stride];
    }
    gz_take(b, l);
    return sym;
}

// One symbol of any kind, for the lanes the literal path below could not serve: a literal with a long code or
// at the stream's end, a match (-> token), the end of the block.  Returns the loop state: 0 go on, 1 end of
// block, 2 malformed.
template <class Bits>
KTA_GZIP_HD uint32_t gz_step(Bits &b, const GzCode &lencode, const GzCode &distcode, const uint16_t *work, uint32_t stride,
                             uint8_t *dst, uint32_t &op, uint32_t cap, GzTokens &t)
{
    gz_refill_ready(b);                               // >= 32 bits: a length code and its extra bits are <= 20
    const int sym = gz_lookup<GZ_LBITS>(b, lencode, work, stride, GZ2_W_LTAB, GZ2_W_LSYM);
    if (sym < 0 || sym > 285 || b.overrun) return 2;
    if (sym < 256) {
        if (op >= cap) return 2;
        dst[op++] = (uint8_t)sym;
        t.run++;
        return 0;
    }
    if (sym == 256) return 1;
    const uint32_t k = (uint32_t)sym - 257u;          // 0..28
    uint32_t eb = 0, len = 3u + k;                    // 257..264: 3..10
    if (k >= 8) {
        eb = (k - 4u) >> 2;                           // 265..284: 1..5 extra bits
        len = 3u + ((4u + (k & 3u)) << eb);
    }
    if (k == 28) {                                    // 285
        eb = 0;
        len = 258u;
    }
    len += (uint32_t)b.hold & ((1u << eb) - 1u);
    gz_take(b, eb);
    gz_refill_ready(b);                               // >= 32 bits: a distance code and its extra bits are <= 28
    const int ds = gz_lookup<GZ_DBITS>(b, distcode, work, stride, GZ2_W_DTAB, GZ2_W_DSYM);
    if (ds < 0 || ds > 29) return 2;
    uint32_t eb2 = 0, dist = 1u + (uint32_t)ds;
    if (ds >= 4) {
        eb2 = ((uint32_t)ds >> 1) - 1u;               // 1..13 extra bits
        dist = 1u + ((2u + ((uint32_t)ds & 1u)) << eb2);
    }
    dist += (uint32_t)b.hold & ((1u << eb2) - 1u);
    gz_take(b, eb2);
    if (b.overrun || dist > op || len > cap - op) return 2;
    gz_emit(t, len, dist);
    if (t.full) return 2;
    op += len;
    return 0;
}

// The symbols of one compressed block, tokenized.  Returns false on malformed input (or a full token area).
// One loop with one exit.  At the top of an iteration hold has >= 32 bits (or the stream is at its end); the
// iteration reads the next stream word and the table entry of the next symbol together (one LDS round
// trip), and then is the literal path — one byte stored, the word shifted in if hold ran low, no branch in
// between — or, for the lanes of the wave that need it, gz_step.
template <class Bits>
KTA_GZIP_HD bool gz_codes_tok(Bits &b, const GzCode &lencode, const GzCode &distcode, const uint16_t *work, uint32_t stride,
                              uint8_t *dst, uint64_t &op_io, uint64_t cap_io, GzTokens &t)
{
    uint32_t op = (uint32_t)op_io;
    const uint32_t cap = (uint32_t)cap_io;
    uint32_t state = 0;                               // 0 decoding, 1 end of block, 2 malformed
    gz_topup_all(b);
    gz_refill_ready(b);
    while (state == 0) {
        gz_topup_all(b);
        const uint32_t w = gz_word_ready(b);          // (used only if this iteration ends with a refill)
        const uint32_t e = work[(GZ2_W_LTAB + ((uint32_t)b.hold & ((1u << GZ_LBITS) - 1u))) * stride];
        const uint32_t l = e & 15u;
        // a literal whose code the table resolves, all of its bits real, room for it: the common case
        const bool fast = l != 0 && e < (256u << 4) && op < cap && b.bits >= 32 && b.pos <= b.n;
        if (fast) {
            dst[op++] = (uint8_t)(e >> 4);
            t.run++;
            b.hold >>= l;
            b.bits -= l;
            const bool need = b.bits < 32 && b.pos + 4 <= b.n;
            b.hold |= need ? (uint64_t)w << (b.bits & 31u) : 0ull;
            b.pos += need ? 4u : 0u;
            b.bits += need ? 32u : 0u;
        } else {
            state = gz_step(b, lencode, distcode, work, stride, dst, op, cap, t);
            gz_refill_ready(b);
        }
    }
    op_io = op;
    return state == 1;
}

// Stage 1 of one gzip member: literals into dst[0 .. cap), matches into tok[0 .. tok_cap) (*n_tok of them).
// `work`: GZ2_WORK u16 words with `stride`; `b`: a bit reader (GzBits: straight from memory; GzBitsWin with its
// window fields set: through the window).  Returns the bytes the member produces or -1.
template <class Bits>
KTA_GZIP_HD int64_t gzip_tokenize(Bits &b, const uint8_t *src, uint64_t n, uint8_t *dst, uint64_t cap, uint32_t *tok,
                                  uint64_t tok_cap, uint64_t *n_tok, uint16_t *work, uint32_t stride)
{
    const uint64_t start = gzip_header(src, n);
    if (!start || n > 0x7FFFFFFFull || cap > 0x7FFFFFFFull) return -1;
    b.p = src + start;
    b.n = (uint32_t)(n - 8 - start);
    b.overrun = false;
    gz_seek(b, 0);
    GzCode lencode, distcode;
    GzTokens t{tok, 0, tok_cap, 0, false};
    uint64_t op = 0;
    uint32_t last;
    do {
        last = gz_get(b, 1);
        const uint32_t type = gz_get(b, 2);
        if (b.overrun) return -1;
        if (type == 0) {                              // stored: its bytes are literals
            const uint64_t at = b.pos - b.bits / 8;
            if (at + 4 > b.n) return -1;
            const uint32_t len = (uint32_t)b.p[at] | ((uint32_t)b.p[at + 1] << 8);
            const uint32_t nlen = (uint32_t)b.p[at + 2] | ((uint32_t)b.p[at + 3] << 8);
            if ((len ^ 0xFFFFu) != nlen || at + 4 + len > b.n || op + len > cap) return -1;
            uint32_t k = 0;
            for (; k + 8 <= len; k += 8) {
                uint64_t w;
                __builtin_memcpy(&w, b.p + at + 4 + k, 8);
                __builtin_memcpy(dst + op + k, &w, 8);
            }
            for (; k < len; k++) dst[op + k] = b.p[at + 4 + k];
            op += len;
            t.run += len;
            gz_seek(b, at + 4 + len);
            continue;
        }
        if (type == 3) return -1;
        if (type == 1) {                              // fixed codes (RFC 1951 3.2.6)
            for (uint32_t s = 0; s < 288; s++)
                work[(GZ2_W_LTAB + s) * stride] = (uint16_t)(s < 144 ? 8 : (s < 256 ? 9 : (s < 280 ? 7 : 8)));
            for (uint32_t s = 0; s < 30; s++) work[(GZ2_W_LTAB + 288 + s) * stride] = 5;
            (void)gz_build(distcode, work, stride, GZ2_W_LTAB + 288, GZ2_W_DSYM, 30);
            gz_fill(work, stride, GZ2_W_DSYM, GZ2_W_DTAB, GZ_DBITS);
            (void)gz_build(lencode, work, stride, GZ2_W_LTAB, GZ2_W_LSYM, 288);
            gz_fill(work, stride, GZ2_W_LSYM, GZ2_W_LTAB, GZ_LBITS);
        } else {                                      // dynamic codes (3.2.7)
            const uint32_t nlen = gz_get(b, 5) + 257, ndist = gz_get(b, 5) + 1, ncode = gz_get(b, 4) + 4;
            if (nlen > 286 || ndist > 30) return -1;
            const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
            for (uint32_t i = 0; i < 19; i++)
                work[(GZ2_W_LTAB + order[i]) * stride] = (uint16_t)(i < ncode ? gz_get(b, 3) : 0);
            GzCode &clcode = distcode;                // the code-length code (19 symbols) borrows the storage
            if (gz_build(clcode, work, stride, GZ2_W_LTAB, GZ2_W_DSYM, 19) != 0) return -1;
            uint32_t i = 0;
            while (i < nlen + ndist) {
                const int sym = gz_decode(b, clcode, work, stride, GZ2_W_DSYM);
                if (sym < 0 || b.overrun) return -1;
                if (sym < 16) {
                    work[(GZ2_W_LTAB + i++) * stride] = (uint16_t)sym;
                    continue;
                }
                uint32_t prev = 0, rep;
                if (sym == 16) {
                    if (i == 0) return -1;
                    prev = work[(GZ2_W_LTAB + i - 1) * stride];
                    rep = 3 + gz_get(b, 2);
                } else if (sym == 17) rep = 3 + gz_get(b, 3);
                else rep = 11 + gz_get(b, 7);
                if (i + rep > nlen + ndist) return -1;
                while (rep--) work[(GZ2_W_LTAB + i++) * stride] = (uint16_t)prev;
            }
            if (work[(GZ2_W_LTAB + 256) * stride] == 0) return -1;     // no end-of-block code
            // an incomplete code is legal only as a single 1-bit code
            uint32_t zeros = 0, ones = 0;
            for (uint32_t s = 0; s < ndist; s++) {
                const uint32_t l = work[(GZ2_W_LTAB + nlen + s) * stride];
                zeros += l == 0;
                ones += l == 1;
            }
            int left = gz_build(distcode, work, stride, GZ2_W_LTAB + nlen, GZ2_W_DSYM, (int)ndist);
            if (left < 0 || (left > 0 && zeros + ones != ndist)) return -1;
            gz_fill(work, stride, GZ2_W_DSYM, GZ2_W_DTAB, GZ_DBITS);
            zeros = ones = 0;
            for (uint32_t s = 0; s < nlen; s++) {
                const uint32_t l = work[(GZ2_W_LTAB + s) * stride];
                zeros += l == 0;
                ones += l == 1;
            }
            left = gz_build(lencode, work, stride, GZ2_W_LTAB, GZ2_W_LSYM, (int)nlen);
            if (left < 0 || (left > 0 && zeros + ones != nlen)) return -1;
            gz_fill(work, stride, GZ2_W_LSYM, GZ2_W_LTAB, GZ_LBITS);
        }
        if (!gz_codes_tok(b, lencode, distcode, work, stride, dst, op, cap, t)) return -1;
    } while (!last);
    const uint64_t used = b.pos - b.bits / 8;
    const uint64_t isize = (uint64_t)src[n - 4] | ((uint64_t)src[n - 3] << 8) | ((uint64_t)src[n - 2] << 16) | ((uint64_t)src[n - 1] << 24);
    if (b.overrun || used != b.n || (op & 0xFFFFFFFFull) != isize) return -1;
    *n_tok = t.n;
    return (int64_t)op;
}

// Stage 2, sequentially (host tests; the device runs kafka_gzip_apply): executes the tokens over dst[0 .. total).
// Returns false if a token points outside the output (cannot happen for tokens gzip_tokenize accepted).
KTA_GZIP_HD bool gz_apply_tokens(uint8_t *dst, uint64_t total, const uint32_t *tok, uint64_t n_tok)
{
    uint64_t op = 0;
    for (uint64_t i = 0; i < n_tok; i++) {
        const uint32_t tk = tok[i], len = (tk >> 8) & 511u, dist = (tk >> 17) + 1u;
        op += tk & 255u;
        if (!len) continue;
        if (dist > op || op + len > total) return false;
        for (uint32_t k = 0; k < len; k++) dst[op + k] = dst[op - dist + k];
        op += len;
    }
    return op <= total;
}

}  // namespace kta
