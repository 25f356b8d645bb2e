// kta_zstd.h — Zstandard inflate for compressed Kafka record batches (attributes codec 4, KIP-110).
// Same code on the host (index: size bound, literal scratch; CPU tests against libzstd's output) and on
// the device.  Written from the format specification (RFC 8878); checksums are not verified (the batch
// CRC-32C covers the compressed bytes), dictionaries are not supported (Kafka does not use them),
// skippable frames are refused.
//
// The decoder is written against two small policies, so that one text serves three executions:
//   S, where the compressed bytes come from:  byte(at)
//        ZsMem         plain memory (the host; the device's one-lane-per-batch kernel; one lane of a wave on its
//                      own Huffman stream)
//        (kta_kafka.hip: a wave's LDS window on the batch, refilled by all 64 lanes together)
//   O, where the output goes:  lit_src / lit_buf / lit_rle / match / huf_streams, position op
//        ZsOutMem      plain memory, byte by byte
//        (kta_kafka.hip: one wave moving 64 bytes per step through an LDS mirror of the output)
// In the wave form every lane runs the parsing redundantly (it is uniform across the wave: same bytes, same
// tables in LDS, same decisions) and the policies do the work that has width: the window refills, the copies,
// the four Huffman streams on four lanes.
//
// Frame: magic FD2FB528 (LE) | Frame_Header_Descriptor | [Window_Descriptor] | [Dictionary_ID] |
//   [Frame_Content_Size] | blocks | [checksum 4].  Block: 3-byte LE header (last 1 bit, type 2 bits:
//   raw / RLE / compressed / reserved, size 21 bits).  A compressed block is a literals section (raw, RLE,
//   or Huffman coded in 1 or 4 backward bit streams; the Huffman weights are direct 4-bit values or FSE
//   coded) and a sequences section (count, three FSE tables for literal-length / offset / match-length
//   codes — predefined, RLE, described, or repeated from the previous block — and one backward bit
//   stream of interleaved FSE states and extra bits).  A sequence copies `literal_length` literals, then
//   `match_length` bytes from `offset` back; offset values 1..3 name the three most recent offsets.
//
// Backward bit streams: the last byte carries a final 1 marker; bits are consumed from just below it
// towards the first byte.  Reading before the first byte yields zeros (the format relies on it for the
// last state refills), and the final position is checked.
#pragma once

#include <stdint.h>

#if defined(__HIPCC__)
#define KTA_ZSTD_HD __host__ __device__ inline __attribute__((always_inline))   // inlined into the kernel: the address
                                                                              // spaces (LDS tables, window) are then known
#else
#define KTA_ZSTD_HD inline
#endif

namespace kta {

constexpr uint32_t ZS_BLOCK_MAX = 128u << 10;

// Per-batch scratch of the decoder (plain memory: global on the device).  FSE entries are
// symbol | nbits << 8 | base << 16; Huffman entries symbol | nbits << 8.
struct ZsWork {
    static constexpr bool kHufShares = false;
    uint32_t ll[1 << 9], of[1 << 8], ml[1 << 9];
    uint32_t wfse[1 << 6];     // FSE table of the Huffman weights
    uint16_t huf[1 << 11];
    int16_t norm[64];          // normalized counts of the table being built
    uint16_t next[64];         // per-symbol state counters while building
    uint16_t first[64];        // (the wave source's table build: where a symbol's run of cells begins)
    uint8_t weights[256];
    uint8_t ll_log, of_log, ml_log, huf_log;
    uint8_t have_ll, have_of, have_ml, have_huf;
    KTA_ZSTD_HD uint16_t *huf_table() { return huf; }
    KTA_ZSTD_HD const uint16_t *huf_table() const { return huf; }
};

// The same scratch where it is scarce (the wave kernel's LDS: what a wave needs there decides how many waves a CU holds,
// and the waves are what the kernel's rate goes with).  A block is done with its Huffman table — the literals are all
// decoded — before it builds or uses its sequence tables, so the 4 KiB table lies over them; what has to outlive that
// goes through a ZsSpill in plain memory (zs_block): the sequence tables of the block before (a "repeat" mode may name
// them) around the literals of a Huffman-coded block, the Huffman table for a later block that brings no tree.  A frame
// of one block — a Kafka batch of the usual size — never spills.
struct ZsWorkSmall {
    static constexpr bool kHufShares = true;
    uint32_t ll[1 << 9], of[1 << 8], ml[1 << 9];      // contiguous: seq_words() of them
    uint32_t wfse[1 << 6];
    int16_t norm[64];
    uint16_t next[64];
    uint16_t first[64];
    uint8_t weights[256];
    uint8_t ll_log, of_log, ml_log, huf_log;
    uint8_t have_ll, have_of, have_ml, have_huf;
    KTA_ZSTD_HD uint16_t *huf_table() { return reinterpret_cast<uint16_t *>(ll); }
    KTA_ZSTD_HD const uint16_t *huf_table() const { return reinterpret_cast<const uint16_t *>(ll); }
};

struct ZsSpill {
    uint32_t seq[(1 << 9) + (1 << 8) + (1 << 9)];      // ll, of, ml as they lie in the work struct
    uint32_t huf[1 << 10];                             // the Huffman table (2^11 entries of 16 bits)
};
static_assert(sizeof(ZsSpill::seq) >= sizeof(ZsSpill::huf), "the Huffman table lies inside the sequence tables");

KTA_ZSTD_HD uint32_t zs_highbit(uint32_t v)   // floor(log2(v)), v > 0
{
    return 31u - (uint32_t)__builtin_clz(v);
}

// ---- byte source: plain memory ------------------------------------------------------------------------------
struct ZsMem {
    const uint8_t *p;
    KTA_ZSTD_HD uint32_t byte(uint32_t at) { return p[at]; }
    // a value every lane of a wave source holds alike (read from an LDS table, say): a wave source moves it to the
    // scalar unit, which keeps the uniform parsing out of the vector registers; nothing to do here
    KTA_ZSTD_HD uint32_t uni(uint32_t x) { return x; }
    // the eight bytes [first, first + 8) of the slice [base, base + n), little endian; bytes outside the slice are zero
    KTA_ZSTD_HD uint64_t le64(uint32_t base, uint32_t n, int32_t first)
    {
        uint64_t c = 0;
        for (int i = 0; i < 8; i++) {
            const int32_t k = first + i;
            if (k >= 0 && (uint32_t)k < n) c |= (uint64_t)p[base + (uint32_t)k] << (8 * i);
        }
        return c;
    }
    KTA_ZSTD_HD const uint8_t *memory() const { return p; }     // (a wave source hands out its global pointer here)
};

// ---- forward (LSB first) bit reader: FSE table descriptions --------------------------------------------
struct ZsFwd {
    uint32_t base, n;   // the bytes [base, base + n) of the source
    uint32_t bit;       // next bit
    bool bad;
    uint64_t cont;      // the stream bits [lo, lo + 64), lo a multiple of 8 (zeros past the end): one load per 6+ bytes
    uint32_t lo;
    bool loaded;
};

KTA_ZSTD_HD ZsFwd zs_fwd_init(uint32_t base, uint32_t n)
{
    ZsFwd f;
    f.base = base;
    f.n = n;
    f.bit = 0;
    f.bad = false;
    f.cont = 0;
    f.lo = 0;
    f.loaded = false;
    return f;
}

template <class S>
KTA_ZSTD_HD uint32_t zs_fwd(S &src, ZsFwd &f, uint32_t nb)   // nb <= 16
{
    if (nb == 0) return 0;
    if (f.bit + nb > 8 * f.n) { f.bad = true; return 0; }
    if (!f.loaded || f.bit < f.lo || f.bit + nb > f.lo + 64) {
        f.lo = f.bit & ~7ull;
        f.cont = src.le64(f.base, f.n, (int32_t)(f.lo >> 3));
        f.loaded = true;
    }
    const uint32_t v = (uint32_t)(f.cont >> (uint32_t)(f.bit - f.lo)) & ((1u << nb) - 1u);
    f.bit += nb;
    return v;
}

// ---- backward bit reader ---------------------------------------------------------------------------------
// A 64-bit container holds the stream bits [lo, lo + 64), lo a multiple of 8 (bytes before the stream's first
// are zeros); it is reloaded, eight bytes at once, when a read leaves it — about once per sequence or per
// eight Huffman symbols instead of once per field.
struct ZsBack {
    uint32_t base, n;
    int32_t off;        // bits [0, off) are unread; may go negative (zeros)
    uint64_t cont;
    int32_t lo;         // bit index of the container's bit 0; off + 1 .. : nothing loaded
    bool loaded;
};

template <class S>
KTA_ZSTD_HD bool zs_back_init(S &src, ZsBack &b, uint32_t base, uint32_t n)
{
    if (n == 0) return false;
    const uint32_t last = src.byte(base + n - 1);
    if (last == 0) return false;
    b.base = base;
    b.n = n;
    b.off = (int32_t)(8 * (n - 1)) + (int32_t)zs_highbit(last);
    b.cont = 0;
    b.lo = 0;
    b.loaded = false;
    return true;
}

template <class S>
KTA_ZSTD_HD uint32_t zs_back(S &src, ZsBack &b, uint32_t nb)   // nb <= 32
{
    b.off -= (int32_t)nb;
    if (nb == 0) return 0;
    const int32_t at = b.off;                        // the bits [at, at + nb)
    if (!b.loaded || at < b.lo || at + (int32_t)nb > b.lo + 64) {
        // the container ends at the byte boundary at or above at + nb: reads go downwards from here
        const int32_t byte_hi = (at + (int32_t)nb + 7) >> 3, byte_lo = byte_hi - 8;
        b.cont = src.le64(b.base, b.n, byte_lo);
        b.lo = byte_lo * 8;
        b.loaded = true;
    }
    return (b.cont >> (uint32_t)(at - b.lo)) & ((1ull << nb) - 1ull);
}

// The same reader for a caller that reads several fields in a row (a sequence: up to 31 + 16 + 16 extra bits, then up to
// 9 + 9 + 8 state bits): zs_back_fill loads the container so that the unread bits end in its top byte — 57 of them at least
// are then in it (zeros below the stream's first byte) —, zs_back_take hands out the next nb without looking: the caller
// counts, and fills again before the 57 run out.  (zs_back's own checks — is the field inside the container, 64-bit
// compares and all — were ~ 25 scalar instructions per field, six fields per sequence; the wave kernel is bound by the
// scalar unit's issue rate.)
template <class S>
KTA_ZSTD_HD void zs_back_fill(S &src, ZsBack &b)
{
    const int32_t byte_lo = ((b.off + 7) >> 3) - 8;
    b.cont = src.le64(b.base, b.n, byte_lo);
    b.lo = byte_lo * 8;
    b.loaded = true;
}

KTA_ZSTD_HD uint32_t zs_back_take(ZsBack &b, uint32_t nb)   // nb <= 31; at most 57 bits between two fills
{
    b.off -= (int32_t)nb;
    return (uint32_t)(b.cont >> ((uint32_t)(b.off - b.lo) & 63u)) & ((1u << nb) - 1u);   // (& 63: nb = 0 right after a fill)
}

// ---- FSE ----------------------------------------------------------------------------------------------------
// Reads a table description (normalized counts) into w.norm; returns the accuracy log, 0 on error.
template <class W, class S>
KTA_ZSTD_HD uint32_t zs_read_norm(S &src, ZsFwd &f, W &w, uint32_t max_log, uint32_t max_sym, uint32_t *n_sym)
{
    const uint32_t log = 5 + zs_fwd(src, f, 4);
    if (f.bad || log > max_log) return 0;
    int32_t remaining = 1 << log;
    uint32_t s = 0;
    while (remaining > 0 && s <= max_sym) {
        const uint32_t bits = zs_highbit((uint32_t)remaining + 1) + 1;
        uint32_t val = zs_fwd(src, f, bits);
        if (f.bad) return 0;
        const uint32_t lower = (1u << (bits - 1)) - 1u;
        const uint32_t threshold = (1u << bits) - 1u - ((uint32_t)remaining + 1u);
        if ((val & lower) < threshold) {            // small values take one bit less
            f.bit -= 1;
            val &= lower;
        } else if (val > lower) {
            val -= threshold;
        }
        const int32_t proba = (int32_t)val - 1;     // -1: "less than one"
        remaining -= proba < 0 ? 1 : proba;
        w.norm[s++] = (int16_t)proba;
        if (proba == 0) {                           // runs of zero probabilities: 2-bit repeat counts
            uint32_t rep = zs_fwd(src, f, 2);
            while (true) {
                for (uint32_t i = 0; i < rep && s <= max_sym; i++) w.norm[s++] = 0;
                if (rep != 3) break;
                rep = zs_fwd(src, f, 2);
                if (f.bad) return 0;
            }
        }
    }
    if (f.bad || remaining != 0 || s > max_sym + 1) return 0;
    f.bit = (f.bit + 7) & ~7ull;                    // the description ends on a byte boundary
    *n_sym = s;
    return log;
}

// Spreads w.norm[0 .. n_sym) into the decoding table `t` of 1 << log entries.
template <class W, class S>
KTA_ZSTD_HD bool zs_build_fse(S &src, W &w, uint32_t *t, uint32_t log, uint32_t n_sym)
{
    const uint32_t size = 1u << log, mask = size - 1u;
    uint32_t high = size;
    for (uint32_t s = 0; s < n_sym; s++)
        if ((int16_t)src.uni((uint16_t)w.norm[s]) == -1) {
            t[--high] = s;
            w.next[s] = 1;
        }
    const uint32_t step = (size >> 1) + (size >> 3) + 3u;
    uint32_t pos = 0;
    for (uint32_t s = 0; s < n_sym; s++) {
        const int32_t cnt = (int16_t)src.uni((uint16_t)w.norm[s]);
        if (cnt <= 0) continue;
        w.next[s] = (uint16_t)cnt;
        for (int32_t i = 0; i < cnt; i++) {
            t[pos] = s;
            do pos = (pos + step) & mask;
            while (pos >= high);
        }
    }
    if (pos != 0) return false;
    for (uint32_t i = 0; i < size; i++) {
        const uint32_t s = src.uni(t[i]), x = src.uni(w.next[s]);
        w.next[s] = (uint16_t)(x + 1);
        const uint32_t nb = log - zs_highbit(x);
        t[i] = s | (nb << 8) | (((x << nb) - size) << 16);
    }
    return true;
}

// A source may bring its own way of building the table — the wave source does (kta_kafka.hip: all 64 lanes at it; the loops
// above are ~2 500 dependent LDS round trips for three tables of 2^9 cells when every lane of a wave walks them alike) —;
// every other source takes the loops above, which are the format's own words (RFC 8878 4.1.1).
template <class W, class S>
KTA_ZSTD_HD auto zs_build_fse_any(S &src, W &w, uint32_t *t, uint32_t log, uint32_t n_sym, int) -> decltype(src.build_fse(w, t, log, n_sym))
{
    return src.build_fse(w, t, log, n_sym);
}
template <class W, class S>
KTA_ZSTD_HD bool zs_build_fse_any(S &src, W &w, uint32_t *t, uint32_t log, uint32_t n_sym, long)
{
    return zs_build_fse(src, w, t, log, n_sym);
}

KTA_ZSTD_HD void zs_build_rle(uint32_t *t, uint32_t sym) { t[0] = sym; }   // log 0: one state, no bits

// Predefined distributions (RFC 8878 3.1.1.3.2.2.1-3): literal lengths and match lengths accuracy 6, offsets 5.
template <class W>
KTA_ZSTD_HD void zs_default_norm(W &w, int which)
{
    if (which == 0) {   // literal length codes 0..35
        const int8_t d[36] = {4, 3, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 1, 1, 1, 2, 2, 2, 2, 2, 2, 2, 2, 2, 3, 2, 1, 1, 1, 1, 1, -1, -1, -1, -1};
        for (int i = 0; i < 36; i++) w.norm[i] = d[i];
    } else if (which == 1) {   // offset codes 0..28
        const int8_t d[29] = {1, 1, 1, 1, 1, 1, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1};
        for (int i = 0; i < 29; i++) w.norm[i] = d[i];
    } else {   // match length codes 0..52
        const int8_t d[53] = {1, 4, 3, 2, 2, 2, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1,
                              1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1, -1, -1};
        for (int i = 0; i < 53; i++) w.norm[i] = d[i];
    }
}

// One of the three sequence tables.  mode: 0 predefined, 1 RLE, 2 described, 3 repeat.
template <class W, class S>
KTA_ZSTD_HD bool zs_seq_table(W &w, int which, uint32_t mode, S &src, uint32_t base, uint32_t n, uint32_t *pos)
{
    uint32_t *t = which == 0 ? w.ll : (which == 1 ? w.of : w.ml);
    uint8_t &log = which == 0 ? w.ll_log : (which == 1 ? w.of_log : w.ml_log);
    uint8_t &have = which == 0 ? w.have_ll : (which == 1 ? w.have_of : w.have_ml);
    const uint32_t max_log = which == 1 ? 8 : 9, max_sym = which == 0 ? 35 : (which == 1 ? 31 : 52);
    if (mode == 0) {
        zs_default_norm(w, which);
        log = which == 1 ? 5 : 6;
        have = 1;
        return zs_build_fse_any(src, w, t, log, which == 0 ? 36 : (which == 1 ? 29 : 53), 0);
    }
    if (mode == 1) {
        if (*pos >= n || src.byte(base + *pos) > max_sym) return false;
        zs_build_rle(t, src.byte(base + (*pos)++));
        log = 0;
        have = 1;
        return true;
    }
    if (mode == 2) {
        ZsFwd f = zs_fwd_init(base + *pos, n - *pos);
        uint32_t n_sym = 0;
        const uint32_t l = zs_read_norm(src, f, w, max_log, max_sym, &n_sym);
        if (!l) return false;
        *pos += f.bit >> 3;
        log = (uint8_t)l;
        have = 1;
        return zs_build_fse_any(src, w, t, l, n_sym, 0);
    }
    return have != 0;   // repeat: the previous block's table
}

// ---- Huffman ------------------------------------------------------------------------------------------------
// Tree description at p[0 .. n): fills w.huf / w.huf_log; returns the bytes consumed, 0 on error.
template <class W, class S>
KTA_ZSTD_HD uint32_t zs_read_huffman(W &w, S &src, uint32_t base, uint32_t n)
{
    if (n < 1) return 0;
    const uint32_t hb = src.byte(base);
    uint32_t n_w = 0;
    uint32_t used;
    if (hb >= 128) {                                  // direct: 4 bits per weight
        n_w = hb - 127;
        used = 1 + (n_w + 1) / 2;
        if (used > n) return 0;
        for (uint32_t i = 0; i < n_w; i++) {
            const uint32_t pair = src.byte(base + 1 + i / 2);
            w.weights[i] = (uint8_t)((i & 1) ? (pair & 15) : (pair >> 4));
        }
    } else {                                          // FSE coded weights, two interleaved states
        used = 1 + hb;
        if (hb == 0 || used > n) return 0;
        ZsFwd f = zs_fwd_init(base + 1, hb);
        uint32_t n_sym = 0;
        const uint32_t log = zs_read_norm(src, f, w, 6, 12, &n_sym);   // weights 0..12 (max code length 11 + 1)
        if (!log || !zs_build_fse_any(src, w, w.wfse, log, n_sym, 0)) return 0;
        const uint32_t at = f.bit >> 3;
        if (at >= hb) return 0;
        ZsBack b;
        if (!zs_back_init(src, b, base + 1 + at, hb - at)) return 0;
        uint32_t s1 = (uint32_t)zs_back(src, b, log), s2 = (uint32_t)zs_back(src, b, log);
        if (b.off < 0) return 0;
        while (true) {
            if (n_w >= 254) return 0;
            const uint32_t e1 = src.uni(w.wfse[s1]);
            w.weights[n_w++] = (uint8_t)(e1 & 0xFF);
            s1 = (e1 >> 16) + (uint32_t)zs_back(src, b, (e1 >> 8) & 0xFF);
            if (b.off < 0) { w.weights[n_w++] = (uint8_t)(src.uni(w.wfse[s2]) & 0xFF); break; }
            const uint32_t e2 = src.uni(w.wfse[s2]);
            w.weights[n_w++] = (uint8_t)(e2 & 0xFF);
            s2 = (e2 >> 16) + (uint32_t)zs_back(src, b, (e2 >> 8) & 0xFF);
            if (b.off < 0) { w.weights[n_w++] = (uint8_t)(src.uni(w.wfse[s1]) & 0xFF); break; }
        }
    }
    // the last weight is implied: the code space must add up to a power of two
    uint32_t sum = 0;
    for (uint32_t i = 0; i < n_w; i++) {
        const uint32_t wt = src.uni(w.weights[i]);
        if (wt > 11) return 0;
        sum += wt ? 1u << (wt - 1) : 0u;
    }
    if (sum == 0) return 0;
    const uint32_t log = zs_highbit(sum) + 1;
    if (log > 11) return 0;
    const uint32_t left = (1u << log) - sum;
    if (left & (left - 1)) return 0;                  // not a power of two
    w.weights[n_w++] = (uint8_t)(zs_highbit(left) + 1);
    // table: the symbols of weight 1 (longest codes) come first, each weight's symbols in symbol order
    uint32_t rank_start[13];
    uint32_t at = 0;
    for (uint32_t wt = 1; wt <= log; wt++) {
        rank_start[wt] = at;
        for (uint32_t s = 0; s < n_w; s++)
            if (src.uni(w.weights[s]) == wt) at += 1u << (wt - 1);
    }
    if (at != (1u << log)) return 0;
    uint16_t *huf = w.huf_table();
    for (uint32_t s = 0; s < n_w; s++) {
        const uint32_t wt = src.uni(w.weights[s]);
        if (!wt) continue;
        const uint32_t len = 1u << (wt - 1), nb = log + 1 - wt;
        for (uint32_t i = 0; i < len; i++) huf[rank_start[wt] + i] = (uint16_t)(s | (nb << 8));
        rank_start[wt] += len;
    }
    w.huf_log = (uint8_t)log;
    w.have_huf = 1;
    return used;
}

template <class W, class S>
KTA_ZSTD_HD bool zs_huf_stream(const W &w, S &src, uint32_t base, uint32_t n, uint8_t *out, uint32_t count)
{
    ZsBack b;
    if (!zs_back_init(src, b, base, n)) return false;
    const uint32_t log = w.huf_log, mask = (1u << log) - 1u;
    const uint16_t *huf = w.huf_table();
    uint32_t state = (uint32_t)zs_back(src, b, log);
    for (uint32_t i = 0; i < count; i++) {
        if (b.off <= -(int32_t)log) return false;    // more symbols wanted than the stream holds
        const uint32_t e = huf[state], nb = e >> 8;
        out[i] = (uint8_t)e;
        state = ((state << nb) | (uint32_t)zs_back(src, b, nb)) & mask;
    }
    return b.off == -(int32_t)log;                    // every bit consumed, none invented
}

// ---- literals and sequences ------------------------------------------------------------------------------
struct ZsLit {
    const uint8_t *p;   // raw / decoded literals (nullptr for RLE)
    uint32_t n;
    uint8_t rle;
};

// Parses the literals section header at p: type, regenerated and compressed sizes, header bytes (0 = error).
template <class S>
KTA_ZSTD_HD uint32_t zs_lit_header(S &src, uint32_t base, uint32_t n, uint32_t *type, uint32_t *regen, uint32_t *comp,
                                   uint32_t *streams)
{
    if (n < 1) return 0;
    const uint32_t b0 = src.byte(base), sf = (b0 >> 2) & 3;
    const uint32_t p1 = n > 1 ? src.byte(base + 1) : 0, p2 = n > 2 ? src.byte(base + 2) : 0;
    const uint32_t p3 = n > 3 ? src.byte(base + 3) : 0, p4 = n > 4 ? src.byte(base + 4) : 0;
    *type = b0 & 3;
    *comp = 0;
    *streams = 1;
    if (*type < 2) {                                  // raw / RLE
        if (sf == 0 || sf == 2) { *regen = b0 >> 3; return 1; }
        if (sf == 1) { if (n < 2) return 0; *regen = (b0 >> 4) | (p1 << 4); return 2; }
        if (n < 3) return 0;
        *regen = (b0 >> 4) | (p1 << 4) | (p2 << 12);
        return 3;
    }
    *streams = sf == 0 ? 1 : 4;
    if (sf < 2) {                                     // 10 + 10 bits
        if (n < 3) return 0;
        const uint32_t v = (b0 >> 4) | (p1 << 4) | (p2 << 12);
        *regen = v & 0x3FF;
        *comp = v >> 10;
        return 3;
    }
    if (sf == 2) {                                    // 14 + 14 bits
        if (n < 4) return 0;
        const uint32_t v = (b0 >> 4) | (p1 << 4) | (p2 << 12) | (p3 << 20);
        *regen = v & 0x3FFF;
        *comp = v >> 14;
        return 4;
    }
    if (n < 5) return 0;                              // 18 + 18 bits
    const uint64_t v = (b0 >> 4) | ((uint64_t)p1 << 4) | ((uint64_t)p2 << 12) | ((uint64_t)p3 << 20) | ((uint64_t)p4 << 28);
    *regen = (uint32_t)(v & 0x3FFFF);
    *comp = (uint32_t)(v >> 18);
    return 5;
}

// ---- output sink: plain memory ----------------------------------------------------------------------------------
// The callers have checked every bound (room in dst, literals left, offsets inside the frame); a sink moves bytes.
struct ZsOutMem {
    uint8_t *dst;
    uint32_t op;

    template <class S>
    KTA_ZSTD_HD void lit_src(S &src, uint32_t at, uint32_t cnt)      // literals that sit in the compressed bytes
    {
        for (uint32_t k = 0; k < cnt; k++) dst[op + k] = (uint8_t)src.byte(at + k);
        op += cnt;
    }
    KTA_ZSTD_HD void lit_buf(const uint8_t *lit, uint32_t cnt)         // Huffman-decoded literals
    {
        for (uint32_t k = 0; k < cnt; k++) dst[op + k] = lit[k];
        op += cnt;
    }
    KTA_ZSTD_HD void lit_rle(uint8_t v, uint32_t cnt)
    {
        for (uint32_t k = 0; k < cnt; k++) dst[op + k] = v;
        op += cnt;
    }
    KTA_ZSTD_HD void match(uint32_t dist, uint32_t len)                // may overlap itself
    {
        uint32_t k = 0;
        if (dist >= 8)
            for (; k + 8 <= len; k += 8) {
                uint64_t v;
                __builtin_memcpy(&v, dst + op - dist + k, 8);
                __builtin_memcpy(dst + op + k, &v, 8);
            }
        for (; k < len; k++) dst[op + k] = dst[op - dist + k];
        op += len;
    }
    // a ZsWorkSmall's tables to and from their ZsSpill
    KTA_ZSTD_HD void spill_words(uint32_t *plain, const uint32_t *work, uint32_t n) { for (uint32_t i = 0; i < n; i++) plain[i] = work[i]; }
    KTA_ZSTD_HD void unspill_words(uint32_t *work, const uint32_t *plain, uint32_t n) { for (uint32_t i = 0; i < n; i++) work[i] = plain[i]; }
    // the Huffman streams of a literals section (1 or 4; stream i: n[i] bytes at at[i], count[i] symbols, to
    // out + the counts before it): one after the other
    template <class W, class S>
    KTA_ZSTD_HD bool huf_streams(const W &w, S &src, uint32_t streams, const uint32_t at[4], const uint32_t n[4],
                                 const uint32_t count[4], uint8_t *out)
    {
        uint32_t done = 0;
        for (uint32_t i = 0; i < streams; i++) {
            if (!zs_huf_stream(w, src, at[i], n[i], out + done, count[i])) return false;
            done += count[i];
        }
        return true;
    }
};

// The literals of a block being handed out to its sequences.
struct ZsLits {
    uint32_t type, regen;
    uint32_t at;            // literals handed out so far
    uint32_t src_at;        // raw: where they start in the source
    uint8_t rle;
    uint8_t *buf;           // Huffman coded: decoded here
    uint32_t block_start, cap;
};

template <class S, class O>
KTA_ZSTD_HD bool zs_put_literals(ZsLits &l, S &src, O &out, uint32_t cnt)
{
    if (cnt > (uint32_t)l.regen - l.at || out.op + cnt > l.cap || out.op + cnt - l.block_start > ZS_BLOCK_MAX) return false;
    if (l.type == 0) out.lit_src(src, l.src_at + l.at, cnt);
    else if (l.type == 1) out.lit_rle(l.rle, cnt);
    else out.lit_buf(l.buf + l.at, cnt);
    l.at += cnt;
    return true;
}

// One compressed block: the bytes [base, base + n) of src, appended through `out`.  `cap`: room of the whole
// output; `frame_start`: output position where the frame began (offsets may not reach before it).
// `spill`, `last` (the frame's last block): see ZsWorkSmall — a work struct with a table of its own never looks at them.
template <class W, class S, class O>
KTA_ZSTD_HD bool zs_block(W &w, S &src, uint32_t base, uint32_t n, O &out, uint32_t cap, uint32_t frame_start,
                          uint32_t rep[3], uint8_t *lit_buf, uint32_t lit_cap, ZsSpill *spill, bool last)
{
    const uint32_t block_start = out.op;
    uint32_t type, regen, comp, streams;
    const uint32_t hdr = zs_lit_header(src, base, n, &type, &regen, &comp, &streams);
    if (!hdr || regen > ZS_BLOCK_MAX) return false;
    uint32_t pos = hdr;
    uint32_t lit_src_at = 0;                          // type 0: where the raw literals start in src
    uint8_t lit_rle = 0;
    if (type == 0) {                                  // raw: used in place
        if (pos + regen > n) return false;
        lit_src_at = base + pos;
        pos += regen;
    } else if (type == 1) {                           // RLE
        if (pos + 1 > n) return false;
        lit_rle = (uint8_t)src.byte(base + pos++);
    } else {                                          // Huffman coded (2) / with the previous tree (3)
        if (pos + comp > n || regen > lit_cap) return false;
        if (type == 3 && !w.have_huf) return false;
        bool seq_out = false;                         // the sequence tables wait in the spill while the Huffman table is in their place
        if (W::kHufShares) {
            if (!spill) return false;                 // (a batch without scratch has no Huffman-coded literals to decode into either)
            seq_out = (w.have_ll | w.have_of | w.have_ml) != 0;
            if (seq_out) out.spill_words(spill->seq, w.ll, (uint32_t)(sizeof(spill->seq) / 4));
            if (type == 3) out.unspill_words(reinterpret_cast<uint32_t *>(w.huf_table()), spill->huf, (uint32_t)(sizeof(spill->huf) / 4));
        }
        uint32_t q = base + pos, qn = comp;
        if (type == 2) {
            const uint32_t used = zs_read_huffman(w, src, q, qn);
            if (!used) return false;
            q += used;
            qn -= used;
        }
        uint32_t at[4] = {q, 0, 0, 0}, len[4] = {qn, 0, 0, 0}, count[4] = {regen, 0, 0, 0};
        if (streams == 4) {
            if (qn < 6) return false;
            const uint32_t s1 = (uint32_t)src.byte(q) | ((uint32_t)src.byte(q + 1) << 8);
            const uint32_t s2 = (uint32_t)src.byte(q + 2) | ((uint32_t)src.byte(q + 3) << 8);
            const uint32_t s3 = (uint32_t)src.byte(q + 4) | ((uint32_t)src.byte(q + 5) << 8);
            if (6 + s1 + s2 + s3 > qn) return false;
            const uint32_t each = ((uint32_t)regen + 3) / 4;
            if (3 * each > regen) return false;
            at[0] = q + 6;
            at[1] = at[0] + s1;
            at[2] = at[1] + s2;
            at[3] = at[2] + s3;
            len[0] = s1;
            len[1] = s2;
            len[2] = s3;
            len[3] = qn - 6 - s1 - s2 - s3;
            count[0] = count[1] = count[2] = each;
            count[3] = regen - 3 * each;
        }
        if (!out.huf_streams(w, src, streams, at, len, count, lit_buf)) return false;
        if (W::kHufShares) {
            if (type == 2 && !last) out.spill_words(spill->huf, reinterpret_cast<const uint32_t *>(w.huf_table()), (uint32_t)(sizeof(spill->huf) / 4));
            if (seq_out) out.unspill_words(w.ll, spill->seq, (uint32_t)(sizeof(spill->seq) / 4));
        }
        pos += comp;
    }
    // sequences section
    if (pos >= n) return false;
    uint32_t n_seq = src.byte(base + pos++);
    if (n_seq >= 128) {
        if (n_seq < 255) {
            if (pos >= n) return false;
            n_seq = ((n_seq - 128) << 8) + src.byte(base + pos++);
        } else {
            if (pos + 2 > n) return false;
            n_seq = src.byte(base + pos) + (src.byte(base + pos + 1) << 8) + 0x7F00u;
            pos += 2;
        }
    }
    ZsLits lits{type, regen, 0, lit_src_at, lit_rle, lit_buf, block_start, cap};
    if (n_seq) {
        if (pos >= n) return false;
        const uint32_t modes = src.byte(base + pos++);
        if (modes & 3u) return false;                 // reserved bits
        if (!zs_seq_table(w, 0, modes >> 6, src, base, n, &pos) || !zs_seq_table(w, 1, (modes >> 4) & 3u, src, base, n, &pos) ||
            !zs_seq_table(w, 2, (modes >> 2) & 3u, src, base, n, &pos))
            return false;
        if (pos >= n) return false;
        ZsBack b;
        if (!zs_back_init(src, b, base + pos, n - pos)) return false;
        // Match length and literal length codes: baseline | extra bits << 24 (RFC 8878 3.1.1.3.2.1.1).  Match lengths: codes
        // 0..31 are 3 + code, then baselines 35, 37, 39, 41, 43, 47, 51, 59, 67, 83, 99, 131, 259, ... with 1, 1, 1, 1, 2, 2, 3, 3,
        // 4, 4, 5, 7, 8, ..., 16 extra bits; literal lengths: 0..15 are the code, then 16, 18, 20, 22, 24, 28, 32, 40, 48, 64,
        // 128, ..., 65536 with 1, 1, 1, 1, 2, 2, 3, 3, 4, 6, 7, ..., 16.  (Tables, not the closed forms: on the device a
        // scalar load each instead of ~ 30 scalar compare-and-select instructions.)
        static constexpr uint32_t kMl[53] = {0x3, 0x4, 0x5, 0x6, 0x7, 0x8, 0x9, 0xA, 0xB, 0xC, 0xD, 0xE, 0xF, 0x10, 0x11, 0x12, 0x13, 0x14, 0x15, 0x16, 0x17, 0x18, 0x19, 0x1A, 0x1B, 0x1C, 0x1D, 0x1E, 0x1F, 0x20, 0x21, 0x22, 0x1000023, 0x1000025, 0x1000027, 0x1000029, 0x200002B, 0x200002F, 0x3000033, 0x300003B, 0x4000043, 0x4000053, 0x5000063, 0x7000083, 0x8000103, 0x9000203, 0xA000403, 0xB000803, 0xC001003, 0xD002003, 0xE004003, 0xF008003, 0x10010003};
        static constexpr uint32_t kLl[36] = {0x0, 0x1, 0x2, 0x3, 0x4, 0x5, 0x6, 0x7, 0x8, 0x9, 0xA, 0xB, 0xC, 0xD, 0xE, 0xF, 0x1000010, 0x1000012, 0x1000014, 0x1000016, 0x2000018, 0x200001C, 0x3000020, 0x3000028, 0x4000030, 0x6000040, 0x7000080, 0x8000100, 0x9000200, 0xA000400, 0xB000800, 0xC001000, 0xD002000, 0xE004000, 0xF008000, 0x10010000};
        zs_back_fill(src, b);
        uint32_t sl = zs_back_take(b, w.ll_log), so = zs_back_take(b, w.of_log), sm = zs_back_take(b, w.ml_log);    // <= 26 bits
        if (b.off < 0) return false;
        for (uint32_t i = 0; i < n_seq; i++) {
            const uint32_t el = src.uni(w.ll[sl]), eo = src.uni(w.of[so]), em = src.uni(w.ml[sm]);
            const uint32_t lc = el & 0xFF, oc = eo & 0xFF, mc = em & 0xFF;
            if (oc > 31 || mc > 52 || lc > 35) return false;
            const uint32_t mt = kMl[mc], lt = kLl[lc];
            const uint32_t ml_bits = mt >> 24, ll_bits = lt >> 24;
            // the sequence's fields: offset (<= 31 bits), match length (<= 16), literal length (<= 16), then the three states
            // (<= 9 + 9 + 8); a fill is good for 57, and what the sequence before left of its container (b.off - b.lo bits)
            // often serves this one too
            if (b.off - b.lo < (int32_t)(oc + ml_bits)) zs_back_fill(src, b);
            const uint32_t ov = (1u << oc) + zs_back_take(b, oc);
            const uint32_t mlen = (mt & 0xFFFFFFu) + zs_back_take(b, ml_bits);
            if (b.off - b.lo < (int32_t)(ll_bits + 26)) zs_back_fill(src, b);
            const uint32_t llen = (lt & 0xFFFFFFu) + zs_back_take(b, ll_bits);
            if (i + 1 < n_seq) {                      // state updates: literal length, match length, offset
                sl = (el >> 16) + zs_back_take(b, (el >> 8) & 0xFF);
                sm = (em >> 16) + zs_back_take(b, (em >> 8) & 0xFF);
                so = (eo >> 16) + zs_back_take(b, (eo >> 8) & 0xFF);
            }
            if (b.off < 0) return false;
            uint32_t offset;
            if (ov > 3) {
                offset = ov - 3;
                rep[2] = rep[1];
                rep[1] = rep[0];
                rep[0] = offset;
            } else {
                uint32_t idx = (uint32_t)ov - 1 + (llen == 0 ? 1u : 0u);
                if (idx == 0) offset = rep[0];
                else {
                    offset = idx < 3 ? (idx == 1 ? rep[1] : rep[2]) : rep[0] - 1;
                    if (idx > 1) rep[2] = rep[1];
                    rep[1] = rep[0];
                    rep[0] = offset;
                }
            }
            if (!zs_put_literals(lits, src, out, llen)) return false;
            if (offset == 0 || offset > out.op - frame_start || out.op + mlen > cap || out.op + mlen - block_start > ZS_BLOCK_MAX)
                return false;
            out.match(offset, mlen);
        }
        if (b.off != 0) return false;                 // the stream is consumed exactly
    } else if (pos != n) {
        return false;
    }
    return zs_put_literals(lits, src, out, (uint32_t)regen - lits.at);    // the literals after the last sequence
}

struct ZsFrame {
    uint32_t header;       // bytes of magic + frame header
    uint64_t window;       // window size (content size for single-segment frames)
    uint64_t content;      // frame content size, ~0 if absent
    bool checksum;
};

template <class S>
KTA_ZSTD_HD bool zs_frame_header(S &src, uint32_t base, uint32_t n, ZsFrame *f)
{
    if (n < 6 || src.byte(base) != 0x28 || src.byte(base + 1) != 0xB5 || src.byte(base + 2) != 0x2F || src.byte(base + 3) != 0xFD)
        return false;
    const uint32_t fhd = src.byte(base + 4), fcs_flag = fhd >> 6, single = (fhd >> 5) & 1, did = fhd & 3;
    if (fhd & 0x08) return false;                     // reserved bit
    if (did) return false;                            // dictionaries are not supported
    uint32_t pos = 5;
    f->window = 0;
    if (!single) {
        if (pos >= n) return false;
        const uint32_t wd = src.byte(base + pos++), wlog = 10 + (wd >> 3);
        if (wlog > 31) return false;
        f->window = (1ull << wlog) + ((1ull << wlog) >> 3) * (wd & 7);
    }
    const uint32_t fcs_bytes = fcs_flag == 0 ? single : (1u << fcs_flag);
    if (pos + fcs_bytes > n) return false;
    f->content = ~0ull;
    if (fcs_bytes) {
        uint64_t v = 0;
        for (uint32_t i = 0; i < fcs_bytes; i++) v |= (uint64_t)src.byte(base + pos + i) << (8 * i);
        if (fcs_bytes == 2) v += 256;
        f->content = v;
        pos += fcs_bytes;
    }
    if (single) f->window = f->content;
    f->checksum = (fhd >> 2) & 1;
    f->header = pos;
    return true;
}

// Walks the frames and block headers of a batch payload.  *bound: an upper bound of the inflated size (the
// content sizes where present, else blocks x block maximum); *lit: the largest regenerated size of a
// Huffman-coded literals section (the scratch the decoder needs).  false if the framing is malformed.
KTA_ZSTD_HD bool zstd_scan(const uint8_t *p, uint64_t n, uint64_t *bound, uint64_t *lit)
{
    uint64_t pos = 0, total = 0, max_lit = 0;
    if (n == 0 || n >= (1ull << 31)) return false;     // (the decoder's positions are 32 bits wide)
    ZsMem src{p};
    while (pos < n) {
        ZsFrame f;
        if (!zs_frame_header(src, pos, n - pos, &f)) return false;
        pos += f.header;
        const uint64_t block_max = f.window < ZS_BLOCK_MAX ? f.window : ZS_BLOCK_MAX;
        uint64_t frame_bound = 0;
        while (true) {
            if (pos + 3 > n) return false;
            const uint32_t h = (uint32_t)p[pos] | ((uint32_t)p[pos + 1] << 8) | ((uint32_t)p[pos + 2] << 16);
            pos += 3;
            const uint32_t type = (h >> 1) & 3, size = h >> 3;
            if (type == 3 || size > ZS_BLOCK_MAX) return false;
            if (type == 1) {                          // RLE: one byte, `size` copies
                if (pos + 1 > n) return false;
                pos += 1;
                frame_bound += size;
            } else {
                if (pos + size > n) return false;
                if (type == 2) {
                    uint32_t lt, regen, comp, streams;
                    if (!zs_lit_header(src, pos, size, &lt, &regen, &comp, &streams)) return false;
                    if (lt >= 2 && regen > max_lit) max_lit = regen;
                    frame_bound += block_max ? block_max : ZS_BLOCK_MAX;
                } else {
                    frame_bound += size;
                }
                pos += size;
            }
            if (h & 1) break;
        }
        if (f.checksum) {
            if (pos + 4 > n) return false;
            pos += 4;
        }
        total += f.content < frame_bound ? f.content : frame_bound;   // (an absent content size is ~0)
    }
    *bound = total;
    *lit = max_lit;
    return true;
}

// Inflates the frames of one batch payload — the bytes [0, n) of src — through `out` (room for cap bytes).
// `lit`: scratch of at least the size zstd_scan reported.  Returns the bytes produced or -1.
template <class W, class S, class O>
KTA_ZSTD_HD int64_t zstd_inflate_t(S &src, uint32_t n, O &out, uint32_t cap, W *w, uint8_t *lit, uint32_t lit_cap, ZsSpill *spill = nullptr)
{
    uint32_t pos = 0;
    if (n == 0) return -1;
    while (pos < n) {
        ZsFrame f;
        if (!zs_frame_header(src, pos, n - pos, &f)) return -1;
        pos += f.header;
        const uint32_t frame_start = out.op;
        uint32_t rep[3] = {1, 4, 8};
        w->have_ll = w->have_of = w->have_ml = w->have_huf = 0;
        while (true) {
            if (pos + 3 > n) return -1;
            const uint32_t h = src.byte(pos) | (src.byte(pos + 1) << 8) | (src.byte(pos + 2) << 16);
            pos += 3;
            const uint32_t type = (h >> 1) & 3, size = h >> 3;
            if (type == 3 || size > ZS_BLOCK_MAX) return -1;
            if (type == 0) {
                if (pos + size > n || out.op + size > cap) return -1;
                out.lit_src(src, pos, size);
                pos += size;
            } else if (type == 1) {
                if (pos + 1 > n || out.op + size > cap) return -1;
                out.lit_rle((uint8_t)src.byte(pos), size);
                pos += 1;
            } else {
                if (pos + size > n) return -1;
                if (!zs_block(*w, src, pos, size, out, cap, frame_start, rep, lit, lit_cap, spill, (h & 1) != 0)) return -1;
                pos += size;
            }
            if (h & 1) break;
        }
        if (f.checksum) {
            if (pos + 4 > n) return -1;
            pos += 4;
        }
        if (f.content != ~0ull && out.op - frame_start != f.content) return -1;
    }
    return (int64_t)out.op;
}

// ... from plain memory into plain memory (the host; one lane of the device's lane-per-batch kernel)
KTA_ZSTD_HD int64_t zstd_inflate(const uint8_t *src, uint64_t n, uint8_t *dst, uint64_t cap, ZsWork *w, uint8_t *lit,
                                 uint64_t lit_cap)
{
    if (n >= (1ull << 31) || cap >= (1ull << 31)) return -1;
    ZsMem s{src};
    ZsOutMem o{dst, 0};
    return zstd_inflate_t(s, (uint32_t)n, o, (uint32_t)cap, w, lit, (uint32_t)(lit_cap < (1ull << 31) ? lit_cap : (1ull << 31) - 1));
}

// ... the same with the small work struct and its spill: the host's execution of what the wave kernel runs
KTA_ZSTD_HD int64_t zstd_inflate_small(const uint8_t *src, uint64_t n, uint8_t *dst, uint64_t cap, ZsWorkSmall *w, ZsSpill *spill,
                                       uint8_t *lit, uint64_t lit_cap)
{
    if (n >= (1ull << 31) || cap >= (1ull << 31)) return -1;
    ZsMem s{src};
    ZsOutMem o{dst, 0};
    return zstd_inflate_t(s, (uint32_t)n, o, (uint32_t)cap, w, lit, (uint32_t)(lit_cap < (1ull << 31) ? lit_cap : (1ull << 31) - 1), spill);
}

}  // namespace kta
