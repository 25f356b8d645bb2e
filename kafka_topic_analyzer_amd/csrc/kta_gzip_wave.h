// kta_gzip_wave.h — stage 1 of the gzip inflate (csrc/kta_gzip.h: Huffman decoding into literals + match tokens) with one
// WAVE per batch: the symbols of ONE DEFLATE block decoded by 64 lanes at once.  (Included by kta_kafka.hip and, compiled
// for the host over tests/native/wave_emu.h, by tests/native/gzip_wave_emu.cpp: the CPU suite runs this text.)
//
// DEFLATE is bit-serial: where symbol k + 1 begins is known once symbol k is decoded, which is why rounds 3-5 gave a batch
// to one LANE (kafka_gzip_tokenize<L>) — ~ 2 500 symbols one after the other at ~ 1 us each, and a GPU with as many busy
// lanes as the fetch has batches.  But Huffman codes synchronise: a decoder started at a wrong bit reads garbage for a few
// symbols and then, almost always, falls into step with the real sequence of symbol starts — and from there on
// decodes exactly what the in-order decoder decodes.  So (as pugz does on CPU threads, and the "self-synchronising"
// GPU Huffman decoders do):
//   1. the block's bits [p0, pend) are cut into 64 segments; lane i decodes from its segment's first bit b_i, as if a
//      symbol began there, until it has passed the segment's end, and keeps where it stopped: e_i (plus what it counted on
//      the way: output bytes, tokens).  Lane 0 starts at a real symbol start: e_0 is real;
//   2. repeat: lane i looks at e_(i-1).  If that is where it started from, nothing to do; else it decodes its segment
//      again from there.  The lanes whose whole chain of predecessors started where the one before stopped are CONFIRMED
//      — they decoded what the in-order decoder decodes — and every repetition confirms at least one more lane, in
//      practice nearly all of them: a lane that had synchronised inside its segment stops at the same e_i as before, and
//      its successor keeps its result.  The loop ends when the confirmed lanes reach the end-of-block symbol, an error, or
//      the last segment;
//   3. exclusive prefix sums of the confirmed lanes' output bytes and token counts give every lane its place in the output
//      and in the token list; one more decode of the segment, from the confirmed start, writes literals and tokens there.
// A segment's literals that no match follows are closed by a token of their own (run, no match), so a lane's tokens do
// not depend on its neighbours'; stage 2 (kafka_gzip_apply; gz_apply_tokens) reads such a token as "advance".
//
// The compressed bytes come through an LDS window (7 KiB: the usual batch whole; a longer member in several windows), so do
// the code tables, which all lanes share — one batch per wave — and build together (gw_build: ballots rank the symbols of
// every code length, a lane fills its entries of the lookup table by decoding their index).  Everything a member may
// hold that this kernel does not do itself — members of 128 MiB and more, a token area too small for the extra closing
// tokens — and every stream it finds malformed is LEFT to the lane kernel, which runs afterwards over the
// batches marked kGwNotDone and gives the verdict: a batch this kernel finishes is one the lane kernel would have accepted
// with the same bytes (the tests hold both against zlib and against each other on damaged streams).
#pragma once

#ifndef KTA_READLANE
#define KTA_READLANE(v, l) ((uint32_t)__builtin_amdgcn_readlane((int)(v), (int)(l)))
#endif
#ifndef KTA_BALLOT64    // the lanes (of those that call it together) whose predicate holds.  (wave_emu.h: a meeting point.)
#define KTA_BALLOT64(p) ((uint64_t)__builtin_amdgcn_ballot_w64(p))
#endif
#ifndef KTA_SHFL_UP     // lane - off's value; a lane below `off` keeps its own.  (wave_emu.h: a meeting point.)
#define KTA_SHFL_UP(v, off) __shfl_up((v), (off))
#endif
#ifndef KTA_UNI         // a value every lane holds alike, moved to a scalar register (the emulator: the value)
#define KTA_UNI(v) ((uint32_t)__builtin_amdgcn_readfirstlane((int)(v)))
#endif

constexpr uint32_t kGwWinBytes = 7168;             // the LDS window on the stream: seven 16-byte units per lane
constexpr uint32_t kGwMarginBits = 128;            // a decoder never starts a symbol this close to the window's end
constexpr uint32_t kGwHeaderBits = 8192;           // a block header (code lengths) is parsed with this much window ahead
// index bits of the lookup tables (literal/length, distance, code-length code).  (10 bits for the literal/length code — 1 KiB
// more table, paid with a 6 KiB window, which the bench's 5-6.4 KB batches then need twice — measured slower: 1.92 -> 1.97 ms.)
constexpr uint32_t kGwLBits = 9, kGwDBits = 6, kGwCBits = 7;
constexpr uint32_t kGwNotDone = 0xFFFFFFFFu;       // token count word: this batch is the lane kernel's
constexpr uint32_t kGwMaxStream = 1u << 27;        // longer members: the lane kernel (bit positions stay far below 2^31)
constexpr uint32_t kGwMinSegBits = 64;
constexpr uint32_t kGwLitsPerRound = 4;            // (1: 1.63 ms, 2: 1.51, 4: 1.445, 8: 1.46 for 16 667 batches of 16 KiB, inflate + decode)

struct GwShared {
    uint32_t win[kGwWinBytes / 4 + 8];             // + 8 words: a peek reads the word of its bit and the next
    uint16_t ltab[1u << kGwLBits];                 // literal/length code: symbol << 4 | code length (0: longer than the index, or none)
    uint16_t dtab[1u << kGwDBits];                 // distance code
    uint16_t ctab[1u << kGwCBits];                 // code-length code (complete: its codes are at most 7 bits)
    uint16_t lsym[288];                            // symbols in canonical order
    uint16_t dsym[32];                             // (the code-length code's too, while a header is parsed)
    uint32_t lcode[16], dcode[16];                 // per code length: limit (bits 0-15) | base (bits 16-31, signed) — kta_gzip.h, GzCode
    uint8_t lens[320];                             // the block's code lengths: literal/length, then distance
};
static_assert(sizeof(GwShared) <= 10240, "sixteen waves per CU");

struct GwWindow {
    const uint8_t *buffer;     // the fetch buffer
    uint64_t g0;               // absolute offset of the DEFLATE stream's first byte
    uint64_t last_unit;        // absolute offset of the last 16-byte unit that holds bytes of this batch
    int32_t wbit0;             // stream bit position of the window's first bit (negative: the window begins before the stream)
    uint32_t lane;
};

// the window begins at the 16-byte unit that holds stream bit p
__device__ __forceinline__ void gw_load_window(GwShared &sh, GwWindow &w, uint32_t p)
{
    const uint64_t wabs = (w.g0 + (p >> 3)) & ~15ull;
    w.wbit0 = (int32_t)(8 * (int64_t)(wabs - w.g0));
    __syncthreads();                                   // (every lane is done with the window that was)
    uint4 v[kGwWinBytes / 1024];
#pragma unroll
    for (uint32_t k = 0; k < kGwWinBytes / 1024; k++) {        // in flight together
        const uint64_t a = wabs + (uint64_t)(k * 64 + w.lane) * 16;
        v[k] = *reinterpret_cast<const uint4 *>(w.buffer + (a < w.last_unit ? a : w.last_unit));
    }
    uint4 *win4 = reinterpret_cast<uint4 *>(sh.win);
#pragma unroll
    for (uint32_t k = 0; k < kGwWinBytes / 1024; k++) win4[k * 64 + w.lane] = v[k];
    if (w.lane < 8) sh.win[kGwWinBytes / 4 + w.lane] = 0;
    __syncthreads();
}

// the stream's 32 bits from p on (p inside the window, kGwMarginBits before its end at most): a code and its extra bits are
// at most 28 (a distance code of 15 bits with 13 extra bits)
__device__ __forceinline__ uint32_t gw_peek(const GwShared &sh, int32_t wbit0, uint32_t p)
{
    const uint32_t q = (uint32_t)((int32_t)p - wbit0), wd = q >> 5;
    const uint32_t lo = sh.win[wd], hi = sh.win[wd + 1];
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_alignbit(hi, lo, q & 31u);
#else
    return (uint32_t)((((uint64_t)hi << 32) | lo) >> (q & 31u));
#endif
}

__device__ __forceinline__ uint32_t gw_bitrev32(uint32_t v)
{
#if defined(__clang__)
    return __builtin_bitreverse32(v);
#else
    v = ((v & 0x55555555u) << 1) | ((v >> 1) & 0x55555555u);
    v = ((v & 0x33333333u) << 2) | ((v >> 2) & 0x33333333u);
    v = ((v & 0x0F0F0F0Fu) << 4) | ((v >> 4) & 0x0F0F0F0Fu);
    return __builtin_bswap32(v);
#endif
}
__device__ __forceinline__ uint32_t gw_rev15(uint32_t v) { return gw_bitrev32(v) >> 17; }

struct GwBuilt {
    int left;              // code space left (0: complete; < 0: over-subscribed)
    uint32_t used;         // symbols with a code
    uint32_t short01;      // symbols of length 0 or 1
};

// The canonical code of the n symbols whose lengths are lens[0 .. n) (<= MAXL): symbols in code order -> syms, the lookup
// table of PBITS index bits -> tab, limit | base per length -> code (if not null).  All 64 lanes; CHUNKS * 64 >= n.
template <uint32_t PBITS, uint32_t MAXL, uint32_t CHUNKS>
__device__ __forceinline__ GwBuilt gw_build(uint32_t lane, const uint8_t *lens, uint32_t n, uint16_t *syms, uint16_t *tab, uint32_t *code)
{
    const uint64_t below = (1ull << lane) - 1ull;
    uint32_t mylen[CHUNKS], rank[CHUNKS], myoff[CHUNKS];
#pragma unroll
    for (uint32_t c = 0; c < CHUNKS; c++) {
        const uint32_t s = c * 64 + lane;
        mylen[c] = s < n ? (uint32_t)lens[s] & 15u : 0u;
        rank[c] = 0;
        myoff[c] = 0;
    }
    uint32_t cnt[MAXL + 1];
    cnt[0] = 0;
#pragma unroll
    for (uint32_t l = 1; l <= MAXL; l++) {
        uint32_t t = 0;
#pragma unroll
        for (uint32_t c = 0; c < CHUNKS; c++) {
            const uint64_t m = KTA_BALLOT64(mylen[c] == l);
            if (mylen[c] == l) rank[c] = t + (uint32_t)__builtin_popcountll(m & below);
            t += (uint32_t)__builtin_popcountll(m);
        }
        cnt[l] = t;
    }
    GwBuilt r;
    int left = 1;
    uint32_t first = 0, index = 0;
    bool over = false;
    uint32_t limit[MAXL + 1];
    int32_t base[MAXL + 1];
#pragma unroll
    for (uint32_t l = 1; l <= MAXL; l++) {
        const uint32_t c_l = cnt[l];
        if (!over) {
            left = (left << 1) - (int)c_l;
            if (left < 0) over = true;
        }
        limit[l] = over ? 0u : (first + c_l) << (15 - l);
        base[l] = (int32_t)index - (int32_t)first;
#pragma unroll
        for (uint32_t c = 0; c < CHUNKS; c++)
            if (mylen[c] == l) myoff[c] = index;
        index += c_l;
        first = (first + c_l) << 1;
    }
    r.left = over ? -1 : left;
    r.used = index;
    r.short01 = n - index + cnt[1];
    __syncthreads();                                   // (syms, tab, code: nobody reads the ones of the block before any more)
    if (over) return r;
#pragma unroll
    for (uint32_t c = 0; c < CHUNKS; c++)
        if (mylen[c]) syms[myoff[c] + rank[c]] = (uint16_t)(c * 64 + lane);
    if (code && lane == 0) {
#pragma unroll
        for (uint32_t l = 1; l <= MAXL; l++) code[l] = (limit[l] & 0xFFFFu) | ((uint32_t)(uint16_t)(int16_t)base[l] << 16);
    }
    __syncthreads();
#pragma unroll
    for (uint32_t j = 0; j < ((1u << PBITS) + 63) / 64; j++) {
        const uint32_t i = j * 64 + lane;
        const uint32_t x15 = (gw_bitrev32(i) >> (32 - PBITS)) << (15 - PBITS);   // the index bits as the code's first bits
        uint32_t l = 0;
        int32_t idx = 0;
#pragma unroll
        for (uint32_t k = (PBITS < MAXL ? PBITS : MAXL); k >= 1; k--)
            if (x15 < limit[k]) {                      // the smallest such length wins
                l = k;
                idx = base[k] + (int32_t)(x15 >> (15 - k));
            }
        if (i < (1u << PBITS)) tab[i] = l ? (uint16_t)(((uint32_t)syms[(uint32_t)idx] << 4) | l) : (uint16_t)0;
    }
    __syncthreads();
    return r;
}

// The limit | base words of the code lengths the lookup table does not resolve (PBITS + 1 .. 15), in registers: every lane
// holds the same ones.
template <uint32_t PBITS>
struct GwLong {
    uint32_t cw[15 - PBITS];
    __device__ __forceinline__ void load(const uint32_t *code)
    {
#pragma unroll
        for (uint32_t k = PBITS + 1; k <= 15; k++) cw[k - PBITS - 1] = KTA_UNI(code[k]);
    }
};

// one symbol of a code through its lookup table; the canonical search over the longer lengths for what the table does not
// resolve.  (One literal in a hundred of the bench's batches has a code of 10 .. 13 bits, so one lane in 64 takes the search in
// every other round; a second table for those codes — they fill the top of the 15-bit code space, 64 .. 96 entries at a
// resolution of four strings — instead of the six compare-and-select steps was built and measured: 1.445 -> 1.433 ms, not kept.)  v: the stream's bits (>= 15 real or zeros).  Returns the length, 0 if the bits are no code; the symbol -> *sym.
template <uint32_t PBITS>
__device__ __forceinline__ uint32_t gw_symbol(uint32_t v, const uint16_t *tab, const uint16_t *syms, uint32_t n_syms, const GwLong<PBITS> &lg,
                                              uint32_t *sym)
{
    const uint32_t e = tab[v & ((1u << PBITS) - 1u)];
    uint32_t l = e & 15u;
    *sym = e >> 4;
    if (!l) {
        const uint32_t rev = gw_rev15(v & 0x7FFFu);
        int32_t idx = 0;
#pragma unroll
        for (uint32_t k = 15; k > PBITS; k--) {
            const uint32_t cw = lg.cw[k - PBITS - 1];
            if (rev < (cw & 0xFFFFu)) {
                l = k;
                idx = (int32_t)(int16_t)(cw >> 16) + (int32_t)(rev >> (15 - k));
            }
        }
        if (!l || (uint32_t)idx >= n_syms) return 0;
        *sym = syms[(uint32_t)idx];
    }
    return l;
}

struct GwSeg {
    uint32_t end;          // where the decoder stopped (a symbol start, or the bit after the end-of-block symbol)
    uint32_t out;          // output bytes of the symbols it decoded
    uint32_t ntok;         // tokens they make (matches, 255-literal escapes, the closing token)
    uint32_t flags;        // 1: stopped at the end of the block; 2: stopped at bits that are no symbol (or do not fit)
};

// Decodes the symbols that begin in [p, limit).  WRITE: literals to dst[op ..], tokens to tok[ti ..] (what the counting
// decode of the same range announced), matches checked against the output (dist <= op, room up to cap).
// A round of the loop: up to kGwLitsPerRound literals, then the symbol that is none, if one came up.  (One symbol of either kind
// per round pays for the match's ~ 50 instructions whenever one of the 64 lanes has one — nearly every round, though few of the
// symbols are matches; an inner loop that takes literals as long as they come makes the wave wait, round after round, for
// the lane with the longest run: 1.12 -> 1.84 ms.)
template <bool WRITE>
__device__ __forceinline__ GwSeg gw_decode_segment(const GwShared &sh, int32_t wbit0, uint32_t p, uint32_t limit, uint32_t n_lsym,
                                                   uint32_t n_dsym, const GwLong<kGwLBits> &llong, const GwLong<kGwDBits> &dlong, uint8_t *dst,
                                                   uint32_t op, uint32_t cap, uint32_t *tok, uint32_t ti)
{
    GwSeg r{p, 0, 0, 0};
    const uint32_t op0 = op;
    uint32_t run = 0;
    while (p < limit) {
        uint32_t v = 0, sym = 0, l = 0;
        bool lit = true;                               // no symbol that is not a literal waits in (sym, l, v)
#pragma unroll
        for (uint32_t j = 0; j < kGwLitsPerRound; j++) {
            if (lit && p < limit) {
                // (a peek brings 32 bits and a code is at most 15: every second literal finds its bits behind the one before)
                v = (j & 1u) ? v >> l : gw_peek(sh, wbit0, p);
                l = gw_symbol<kGwLBits>(v, sh.ltab, sh.lsym, n_lsym, llong, &sym);
                lit = l != 0 && sym < 256;
                if (WRITE && lit && op >= cap) {
                    lit = false;
                    l = 0;
                }
                if (lit) {
                    if (WRITE) dst[op] = (uint8_t)sym;
                    p += l;
                    op++;
                    run++;
                }
            }
        }
        if (lit) continue;
        if (!l) { r.flags = 2; break; }
        p += l;
        if (sym == 256) { r.flags = 1; break; }
        if (sym > 285) { r.flags = 2; break; }
        v = gw_peek(sh, wbit0, p);                     // the length's extra bits (<= 5)
        const uint32_t k = sym - 257u;                 // 0..28
        uint32_t eb = 0, len = 3u + k;                 // 257..264: 3..10
        if (k >= 8) {
            eb = (k - 4u) >> 2;                        // 265..284: 1..5 extra bits
            len = 3u + ((4u + (k & 3u)) << eb);
        }
        if (k == 28) {                                 // 285
            eb = 0;
            len = 258u;
        }
        len += v & ((1u << eb) - 1u);
        p += eb;
        v = gw_peek(sh, wbit0, p);
        uint32_t ds;
        l = gw_symbol<kGwDBits>(v, sh.dtab, sh.dsym, n_dsym, dlong, &ds);
        if (!l || ds > 29) { r.flags = 2; break; }
        v >>= l;
        uint32_t eb2 = 0, dist = 1u + ds;
        if (ds >= 4) {
            eb2 = (ds >> 1) - 1u;                      // 1..13 extra bits
            dist = 1u + ((2u + (ds & 1u)) << eb2);
        }
        dist += v & ((1u << eb2) - 1u);
        p += l + eb2;
        const uint32_t esc = run / 255u;
        if (WRITE) {
            if (dist > op || len > cap - op) { r.flags = 2; break; }
            for (uint32_t i = 0; i < esc; i++) tok[ti + i] = 255u;
            tok[ti + esc] = (run - esc * 255u) | (len << 8) | ((dist - 1u) << 17);
        }
        ti += esc + 1;
        r.ntok += esc + 1;
        run = 0;
        op += len;
    }
    if (run) {                                         // the literals no match of this segment follows: a token of their own
        const uint32_t esc = run / 255u, rest = run - esc * 255u;
        if (WRITE) {
            for (uint32_t i = 0; i < esc; i++) tok[ti + i] = 255u;
            if (rest) tok[ti + esc] = rest;
        }
        r.ntok += esc + (rest ? 1u : 0u);
    }
    r.end = p;
    r.out = op - op0;
    return r;
}

__device__ __forceinline__ uint32_t gw_scan_excl(uint32_t lane, uint32_t v, uint32_t *total)   // exclusive prefix sum over the wave
{
    uint32_t incl = v;
#pragma unroll
    for (uint32_t off = 1; off < 64; off <<= 1) {
        const uint32_t up = KTA_SHFL_UP(incl, off);
        if (lane >= off) incl += up;
    }
    *total = KTA_READLANE(incl, 63);
    return incl - v;
}

// Stage 1 of one gzip member by one wave.  The member: buffer[src_off, src_off + n); its output slice dst[0 .. cap) (cap: what
// the trailer announced); tokens -> tok[0 .. tok_cap).  Returns the number of tokens, or kGwNotDone (nothing it wrote counts).
__device__ __forceinline__ uint32_t gw_tokenize_member(GwShared &sh, const uint8_t *buffer, uint64_t src_off, uint64_t n, uint8_t *dst,
                                                       uint64_t cap64, uint32_t *tok, uint64_t tok_cap64, uint32_t lane)
{
    const uint8_t *src = buffer + src_off;
    const uint64_t start = kta::gzip_header(src, n);
    if (!start || n - 8 - start >= kGwMaxStream || cap64 > 0x7FFFFFFFull) return kGwNotDone;
    const uint32_t nbits = 8u * (uint32_t)(n - 8 - start);             // the DEFLATE stream's bits
    const uint32_t cap = (uint32_t)cap64, tok_cap = tok_cap64 > 0x7FFFFFFFull ? 0x7FFFFFFFu : (uint32_t)tok_cap64;
    GwWindow w{buffer, src_off + start, ((src_off + n + 15) & ~15ull) - 16, 0, lane};
    gw_load_window(sh, w, 0);
    uint32_t p = 0, op = 0, nt = 0;
    uint32_t last_block;
    do {
        // ---- block header (every lane alike) ----
        const uint32_t wend = (uint32_t)(w.wbit0 + (int32_t)(8 * kGwWinBytes));        // stream bit behind the window
        if (wend < nbits && p + kGwHeaderBits + kGwMarginBits > wend) gw_load_window(sh, w, p);
        if (p + 3 > nbits) return kGwNotDone;
        uint32_t v = gw_peek(sh, w.wbit0, p);
        last_block = KTA_UNI(v & 1u);
        const uint32_t type = KTA_UNI(((uint32_t)v >> 1) & 3u);
        p += 3;
        uint32_t nlen = 0, ndist = 0;
        if (type == 1) {                               // fixed codes (RFC 1951 3.2.6)
            nlen = 288;
            ndist = 30;
            __syncthreads();
            for (uint32_t s = lane; s < 288; s += 64) sh.lens[s] = (uint8_t)(s < 144 ? 8 : (s < 256 ? 9 : (s < 280 ? 7 : 8)));
            if (lane < 30) sh.lens[288 + lane] = 5;
            __syncthreads();
        } else if (type == 2) {                        // dynamic codes (3.2.7)
            if (p + 14 > nbits) return kGwNotDone;
            v = gw_peek(sh, w.wbit0, p);
            nlen = KTA_UNI(((uint32_t)v & 31u) + 257u);
            ndist = KTA_UNI((((uint32_t)v >> 5) & 31u) + 1u);
            const uint32_t ncode = KTA_UNI((((uint32_t)v >> 10) & 15u) + 4u);
            p += 14;
            if (nlen > 286 || ndist > 30 || p + 3 * ncode > nbits) return kGwNotDone;
            // the code-length code: lane i reads the i-th 3-bit length
            __syncthreads();
            if (lane < 19) sh.lens[lane] = 0;
            __syncthreads();
            if (lane < ncode) {
                const uint32_t order = lane < 3 ? 16u + lane : (lane == 3 ? 0u : ((lane & 1u) ? 8u - ((lane - 3u) >> 1) : 7u + ((lane - 2u) >> 1)));
                sh.lens[order] = (uint8_t)(gw_peek(sh, w.wbit0, p + 3 * lane) & 7u);
            }
            p += 3 * ncode;
            __syncthreads();
            const GwBuilt cb = gw_build<kGwCBits, 7, 1>(lane, sh.lens, 19, sh.dsym, sh.ctab, nullptr);
            if (cb.left != 0) return kGwNotDone;      // (the lane kernel refuses an incomplete code-length code too)
            // The code lengths: a sequence of symbols of the code-length code (a length; "repeat the one before 3-6 times";
            // "3-10 zeros"; "11-138 zeros"), each where the one before ends — walked by every lane alike that was ~ 150 rounds
            // of two dependent LDS round trips, a quarter of the kernel's time.  So 64 bits at a time: lane j decodes the symbol
            // that would begin at bit p + j, the chain of the real ones (0, then where each ends) is followed with readlanes,
            // a prefix sum over the chain's lanes gives every symbol its place among the lengths, and "the one before" is
            // the value of the nearest chain lane below that carries one.
            uint32_t i = 0, prev = 0;
            bool bad = false;
            const uint32_t need = nlen + ndist;
            uint8_t *vals = reinterpret_cast<uint8_t *>(sh.dsym);      // (the code-length code's symbols are done with)
            while (i < need) {
                if (p > nbits) { bad = true; break; }
                const uint32_t vj = gw_peek(sh, w.wbit0, p + lane);
                const uint32_t e = sh.ctab[vj & 127u];
                const uint32_t l = e & 15u, sym = e >> 4, x = vj >> l;
                uint32_t adv = 0, cnt = 0;                 // bits the symbol takes (0: no symbol), lengths it stands for
                if (l) {
                    if (sym < 16) { adv = l; cnt = 1; }
                    else if (sym == 16) { adv = l + 2; cnt = 3 + (x & 3u); }
                    else if (sym == 17) { adv = l + 3; cnt = 3 + (x & 7u); }
                    else { adv = l + 7; cnt = 11 + (x & 127u); }
                }
                uint64_t chain = 0;
                for (uint32_t j = 0; j < 64;) {
                    chain |= 1ull << j;
                    const uint32_t a = KTA_READLANE(adv, j);
                    if (!a) break;                         // (an error if this symbol is one of those wanted: below)
                    j += a;
                }
                const bool on = ((chain >> lane) & 1ull) != 0;
                uint32_t total;
                const uint32_t excl = gw_scan_excl(lane, on ? cnt : 0u, &total);
                const uint32_t remaining = need - i;
                const bool used = on && excl < remaining;
                const bool err = used && (adv == 0 || cnt > remaining - excl || (sym == 16 && i + excl == 0));
                if (KTA_BALLOT64(err)) { bad = true; break; }
                // the nearest used lane at or below this one that carries a value (is no "repeat"): its number + 1, 0 if none
                uint32_t src = used && sym != 16 ? lane + 1 : 0u;
#pragma unroll
                for (uint32_t off = 1; off < 64; off <<= 1) {
                    const uint32_t up = KTA_SHFL_UP(src, off);
                    if (lane >= off && up > src) src = up;
                }
                __syncthreads();
                vals[lane] = (uint8_t)(sym < 16 ? sym : 0u);
                __syncthreads();
                const uint32_t val = src ? (uint32_t)vals[src - 1] : prev;
                if (used)
                    for (uint32_t k = 0; k < cnt; k++) sh.lens[i + excl + k] = (uint8_t)val;
                const uint64_t um = KTA_BALLOT64(used);    // (lane 0 at least: it is on the chain, and a length is wanted)
                const uint32_t jl = 63u - (uint32_t)__builtin_clzll(um);
                p += jl + KTA_READLANE(adv, jl);
                i += KTA_READLANE(excl + cnt, jl);
                prev = KTA_READLANE(val, jl);
            }
            if (bad || p > nbits) return kGwNotDone;
            __syncthreads();
            if (sh.lens[256] == 0) return kGwNotDone; // no end-of-block code
        } else if (type == 0) {                        // stored: LEN, ~LEN at the next byte boundary, then LEN bytes — literals
            const uint32_t at = (p + 7u) & ~7u;
            if (at + 32 > nbits) return kGwNotDone;
            v = gw_peek(sh, w.wbit0, at);
            const uint32_t len = KTA_UNI(v & 0xFFFFu), nlen16 = KTA_UNI(v >> 16);
            const uint32_t esc = len / 255u, rest = len - esc * 255u, ntk = esc + (rest ? 1u : 0u);
            if ((len ^ 0xFFFFu) != nlen16 || at + 32 + 8 * len > nbits || len > cap - op || ntk > tok_cap - nt) return kGwNotDone;
            const uint8_t *from = buffer + w.g0 + ((at + 32) >> 3);
            for (uint32_t k = lane; k < len; k += 64) dst[op + k] = from[k];
            for (uint32_t k = lane; k < esc; k += 64) tok[nt + k] = 255u;
            if (rest && lane == 0) tok[nt + esc] = rest;
            op += len;
            nt += ntk;
            p = at + 32 + 8 * len;
        } else {
            return kGwNotDone;                         // the invalid type: the lane kernel says so
        }
        if (type != 0) {
            // an incomplete code is legal only as a single 1-bit code (kta_gzip.h: gzip_tokenize)
            const GwBuilt db = gw_build<kGwDBits, 15, 1>(lane, sh.lens + nlen, ndist, sh.dsym, sh.dtab, sh.dcode);
            if (type == 2 && (db.left < 0 || (db.left > 0 && db.short01 != ndist))) return kGwNotDone;
            const GwBuilt lb = gw_build<kGwLBits, 15, 5>(lane, sh.lens, nlen, sh.lsym, sh.ltab, sh.lcode);
            if (type == 2 && (lb.left < 0 || (lb.left > 0 && lb.short01 != nlen))) return kGwNotDone;
            // ---- the block's symbols: region after region of the window ----
            const uint32_t n_lsym = lb.used, n_dsym = db.used;
            GwLong<kGwLBits> llong;
            GwLong<kGwDBits> dlong;
            llong.load(sh.lcode);
            dlong.load(sh.dcode);
            bool eob = false;
            while (!eob) {
                uint32_t wend2 = (uint32_t)(w.wbit0 + (int32_t)(8 * kGwWinBytes));
                if (wend2 - kGwMarginBits < nbits && p + 8 * (kGwWinBytes / 2) > wend2) {   // past the window's middle and more stream behind it
                    gw_load_window(sh, w, p);
                    wend2 = (uint32_t)(w.wbit0 + (int32_t)(8 * kGwWinBytes));
                }
                const uint32_t pend = nbits < wend2 - kGwMarginBits ? nbits : wend2 - kGwMarginBits;
                if (p >= pend) return kGwNotDone;     // the stream ends inside the block
                uint32_t seg = (pend - p + 63u) / 64u;
                seg = seg < kGwMinSegBits ? kGwMinSegBits : seg;
                const uint32_t b = p + lane * seg;
                const bool active = b < pend;
                const uint32_t lim = active ? (b + seg < pend ? b + seg : pend) : b;
                const uint32_t n_active = (uint32_t)__builtin_popcountll(KTA_BALLOT64(active));
#ifdef KTA_GW_STATS
                if (lane == 0) { gw_stats[0]++; gw_stats[3] += n_active; }
#endif
                uint32_t from = b;
                GwSeg r{b, 0, 0, 0};
                if (active) r = gw_decode_segment<false>(sh, w.wbit0, from, lim, n_lsym, n_dsym, llong, dlong, nullptr, 0, 0, nullptr, 0);
                uint32_t K;                            // the lanes [0, K) are confirmed
                for (;;) {
                    uint32_t prev_end = KTA_SHFL_UP(r.end, 1);
                    if (lane == 0) prev_end = p;
                    const bool linked = from == prev_end;
                    const uint64_t unlinked = KTA_BALLOT64(active && !linked), stopped = KTA_BALLOT64(active && r.flags != 0);
                    const uint32_t first_unlinked = unlinked ? (uint32_t)__builtin_ctzll(unlinked) : 64u;
                    const uint32_t first_stopped = stopped ? (uint32_t)__builtin_ctzll(stopped) : 64u;
                    if (first_stopped < first_unlinked) { K = first_stopped + 1; break; }
                    if (first_unlinked >= n_active) { K = n_active; break; }
#ifdef KTA_GW_STATS
                    if (lane == 0) { gw_stats[1]++; gw_stats[2] += (uint32_t)__builtin_popcountll(unlinked); }
#endif
                    if (active && !linked) {
                        from = prev_end;
                        r = gw_decode_segment<false>(sh, w.wbit0, from, lim, n_lsym, n_dsym, llong, dlong, nullptr, 0, 0, nullptr, 0);
                    }
                }
                K = KTA_UNI(K);
                const uint32_t stop_flags = KTA_READLANE(r.flags, K - 1), region_end = KTA_READLANE(r.end, K - 1);
                if (stop_flags & 2u) return kGwNotDone;
                if (!stop_flags && pend == nbits) return kGwNotDone;                  // the stream ends without an end-of-block symbol
                const bool mine = lane < K;
                uint32_t tot_out, tot_tok;
                const uint32_t my_op = gw_scan_excl(lane, mine ? r.out : 0u, &tot_out);
                const uint32_t my_ti = gw_scan_excl(lane, mine ? r.ntok : 0u, &tot_tok);
                if (tot_out > cap - op || tot_tok > tok_cap - nt) return kGwNotDone;
                GwSeg wr{0, 0, 0, 0};
                if (mine) wr = gw_decode_segment<true>(sh, w.wbit0, from, lim, n_lsym, n_dsym, llong, dlong, dst, op + my_op, cap, tok, nt + my_ti);
                if (KTA_BALLOT64(mine && (wr.flags & 2u))) return kGwNotDone;         // a match that reaches before the output's first byte
                op += tot_out;
                nt += tot_tok;
                p = region_end;
                eob = (stop_flags & 1u) != 0;
            }
        }
        if (p > nbits) return kGwNotDone;
    } while (!last_block);
    if (((p + 7u) >> 3) != (nbits >> 3) || op != cap) return kGwNotDone;            // every byte of the stream used; the size the trailer told
    return nt;
}

// One wave per batch.  The batch's token area (kta_kafka.hip: behind its slice of the inflate area): [u32 count | u32 0 | tokens].
__device__ __forceinline__ void gzip_tokenize_wave(GwShared &sh, uint8_t *buffer, kta_kafka_batch_desc *descs, uint64_t n_batches)
{
    const uint32_t lane = threadIdx.x;
    const uint64_t b = blockIdx.x;
    if (b >= n_batches) return;
    const kta_kafka_batch_desc d = descs[b];
    if (!(d.flags & KTA_KB_GZIP) || d.status) return;
    const uint64_t n = (uint64_t)d.batch_bytes - KTA_KAFKA_BATCH_HEADER, cap = d.payload_end - d.payload_off;
    const uint64_t scratch = (d.payload_end + 63) & ~63ull;
    if (d.scratch_end < scratch + 16) return;                  // no token area (an empty member): the lane kernel's
    uint32_t *tok = reinterpret_cast<uint32_t *>(buffer + scratch);
    const uint32_t got = gw_tokenize_member(sh, buffer, d.byte_off + KTA_KAFKA_BATCH_HEADER, n, buffer + d.payload_off, cap, tok + 2,
                                            (d.scratch_end - scratch - 8) / 4, lane);
    if (lane == 0) tok[0] = got;
}
