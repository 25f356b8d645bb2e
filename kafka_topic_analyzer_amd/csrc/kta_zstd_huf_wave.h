// kta_zstd_huf_wave.h — the Huffman-coded literals of a zstd block (RFC 8878 4.2: one or four streams, read backwards)
// decoded by all 64 lanes of the wave that inflates the batch.  (Included by kta_kafka.hip and, compiled for the host over
// tests/native/wave_emu.h, by tests/native/zstd_huf_emu.cpp: the CPU suite runs this text.)
//
// Rounds 2-6 gave each stream to one LANE, which read it byte by byte from global memory: a batch of JSON-like values —
// what zstd meets in a Kafka topic, and what bench.py's patterned values are not — took ~ 1 ms of dependent memory round
// trips, and the kernel inflated 9 GB/s of such a log where it does 70 of the bench's.  A Huffman stream is what
// kta_gzip_wave.h's symbols are without the matches, so the same scheme serves: the stream's bits, 2 KiB at a time in the
// LDS window the sequence reader does not need yet, are cut into 64 segments; every lane decodes its segment from a
// speculative start, the lanes whose predecessor ended where they began are confirmed (a lane that is not decodes again from
// there; prefix codes fall into step within a few symbols), a prefix sum of the confirmed lanes' symbol counts gives every
// lane its place in the literals, and a last decode writes them.  Everything counts downwards: bit `q` is the number of
// unread bits, a symbol's code are the bits just below q, the stream ends at q = 0.
#pragma once

#ifndef KTA_READLANE
#define KTA_READLANE(v, l) ((uint32_t)__builtin_amdgcn_readlane((int)(v), (int)(l)))
#endif
#ifndef KTA_BALLOT64
#define KTA_BALLOT64(p) ((uint64_t)__builtin_amdgcn_ballot_w64(p))
#endif
#ifndef KTA_SHFL_UP
#define KTA_SHFL_UP(v, off) __shfl_up((v), (off))
#endif
#ifndef KTA_UNI
#define KTA_UNI(v) ((uint32_t)__builtin_amdgcn_readfirstlane((int)(v)))
#endif

constexpr uint32_t kZhWinBytes = 2048;             // the wave kernel's source window (kZsWin)
constexpr uint32_t kZhMinSegBits = 32;
constexpr uint32_t kZhLowMarginBits = 64;          // a segment never ends this close to the window's first bit: a peek reads 32 bits below q

struct ZhSeg {
    int32_t end;           // unread bits where the decoder stopped (<= its limit; < 0: it read bits the stream does not have)
    uint32_t cnt;          // symbols it decoded
};

// the 32 stream bits below q (q >= 1; bits below the stream's first read as zeros), q + wofs = the bit's index in the window
__device__ __forceinline__ uint32_t zh_peek(const uint32_t *win, int32_t wofs, int32_t q)
{
    const uint32_t t = (uint32_t)(q + wofs) - 1u, wd = t >> 5, sh = (t & 31u) + 1u;     // the top bit wanted: bit t of the window
    const uint32_t hi = win[wd], lo = win[wd - 1];
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t v = sh == 32 ? hi : __builtin_amdgcn_alignbit(hi, lo, sh);
#else
    uint32_t v = (uint32_t)((((uint64_t)hi << 32) | lo) >> sh);
#endif
    if (q < 32) v &= ~0u << (32 - q);
    return v;
}

// Decodes the symbols whose codes begin at q = from, ..., as long as q > limit.  WRITE: the symbols to out[0 ..].
template <bool WRITE>
__device__ __forceinline__ ZhSeg zh_decode_segment(const uint32_t *win, int32_t wofs, const uint16_t *huf, uint32_t log, int32_t from,
                                                   int32_t limit, uint8_t *out)
{
    int32_t q = from;
    uint32_t cnt = 0;
    while (q > limit) {
        const uint32_t e = huf[zh_peek(win, wofs, q) >> (32 - log)];
        if (WRITE) out[cnt] = (uint8_t)e;
        q -= (int32_t)(e >> 8);                        // (every entry of a table zs_read_huffman built has a length >= 1)
        cnt++;
    }
    return ZhSeg{q, cnt};
}

__device__ __forceinline__ uint32_t zh_scan_excl(uint32_t lane, uint32_t v, uint32_t *total)
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

// One stream: the n bytes at buffer[abs ..) hold `count` symbols -> out[0 .. count).  win: kZhWinBytes / 4 + 1 words of LDS
// (this function's while it runs); lo_bound: the lowest address (16-byte aligned) the wave may load from.
__device__ __forceinline__ bool zh_stream(uint32_t *win, const uint8_t *buffer, uint64_t abs, uint64_t lo_bound, uint32_t n, uint32_t count,
                                          const uint16_t *huf, uint32_t log, uint8_t *out, uint32_t lane)
{
    if (n == 0 || n > (1u << 27)) return false;
    const uint32_t last = KTA_UNI(buffer[abs + n - 1]);
    if (last == 0) return false;
    int32_t p = (int32_t)(8 * (n - 1)) + (int32_t)(31 - __builtin_clz(last));       // unread bits (below the padding marker)
    uint32_t produced = 0;
    while (p > 0) {
        // the window ends at the 16-byte boundary at or above the byte of bit p - 1
        const uint64_t top = (abs + (uint64_t)((p + 7) >> 3) + 15) & ~15ull;
        const uint64_t wabs = top >= lo_bound + kZhWinBytes ? top - kZhWinBytes : lo_bound;
        __syncthreads();
#pragma unroll
        for (uint32_t k = 0; k < kZhWinBytes / 1024; k++) {
            const uint64_t a = wabs + (uint64_t)(k * 64 + lane) * 16;
            const uint4 v = *reinterpret_cast<const uint4 *>(buffer + (a < top ? a : top - 16));
            reinterpret_cast<uint4 *>(win)[k * 64 + lane] = v;
        }
        __syncthreads();
        const int32_t wofs = (int32_t)(8 * ((int64_t)abs - (int64_t)wabs));          // window bit of the stream's bit 0
        int32_t plow = (int32_t)kZhLowMarginBits - wofs;                             // lowest q a segment of this window may end at
        plow = plow < 0 ? 0 : plow;
        if (p <= plow) return false;                   // (cannot happen: a window is 2 KiB and ends above p)
        uint32_t seg = (uint32_t)(p - plow + 63) / 64u;
        seg = seg < kZhMinSegBits ? kZhMinSegBits : seg;
        const int32_t b = p - (int32_t)(lane * seg);
        const bool active = b > plow;
        const int32_t lim = active ? (b - (int32_t)seg > plow ? b - (int32_t)seg : plow) : b;
        const uint32_t n_active = (uint32_t)__builtin_popcountll(KTA_BALLOT64(active));
        int32_t from = b;
        ZhSeg r{b, 0};
        if (active) r = zh_decode_segment<false>(win, wofs, huf, log, from, lim, nullptr);
        for (;;) {
            int32_t prev_end = (int32_t)KTA_SHFL_UP((uint32_t)r.end, 1);
            if (lane == 0) prev_end = p;
            const bool linked = from == prev_end;
            const uint64_t unlinked = KTA_BALLOT64(active && !linked);
            if (!unlinked) break;
            if (active && !linked) {
                from = prev_end;
                r = zh_decode_segment<false>(win, wofs, huf, log, from, lim, nullptr);
            }
        }
        uint32_t total;
        const uint32_t excl = zh_scan_excl(lane, active ? r.cnt : 0u, &total);
        const int32_t region_end = (int32_t)KTA_READLANE((uint32_t)r.end, n_active - 1);
        if (total > count - produced || region_end < 0) return false;               // more symbols than announced; bits the stream does not have
        if (active) (void)zh_decode_segment<true>(win, wofs, huf, log, from, lim, out + produced + excl);
        produced += total;
        p = region_end;
    }
    return p == 0 && produced == count;
}

// The streams of one literals section (1 or 4): stream i is the n[i] bytes at buffer[src0 + at[i] ..) and holds count[i] symbols;
// the literals -> out[0 .. count[0] + ... ).  Every lane of the wave calls this with the same arguments.
__device__ __forceinline__ bool zh_streams(uint32_t *win, const uint8_t *buffer, uint64_t src0, uint32_t streams, const uint32_t at[4],
                                           const uint32_t n[4], const uint32_t count[4], const uint16_t *huf, uint32_t log, uint8_t *out,
                                           uint32_t lane)
{
    const uint64_t lo_bound = src0 & ~15ull;
    uint32_t done = 0;
    bool ok = true;
    for (uint32_t s = 0; s < streams && ok; s++) {
        if (n[s] > (1u << 27) || count[s] > (1u << 27)) return false;
        ok = zh_stream(win, buffer, src0 + at[s], lo_bound, (uint32_t)n[s], (uint32_t)count[s], huf, log, out + done, lane);
        done += count[s];
    }
    return ok;
}
