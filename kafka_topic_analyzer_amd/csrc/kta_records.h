// kta_records.h — the two per-record steps of the wave-cooperative record decode (kafka_decode_coop, kta_kafka.hip),
// written against the bytes of one WINDOW of a batch: the leader's chain over the records' length prefixes and a
// lane's parse of one record (Kafka message format v2: length varint | attributes i8 | timestampDelta varlong |
// offsetDelta varint | keyLength varint | key | valueLength varint | value | headers; zig-zag base-128 varints).
//
// Same code on the device (the window lies in LDS) and on the host: kta_kafka_decode_rounds_host runs the kernel's
// rounds with the lanes one after the other and every byte behind the window's valid part poisoned, so the CPU tests
// pin these functions — and the way they are gated by `limit` — against the oracle without a GPU.
//
// Everything here is relative to the window base: `limit` is the number of valid bytes in the window; the dwords that
// hold a byte below `limit` may be read whole, and reads may run up to 16 bytes past the window (the caller's buffer
// has that much room), but nothing read at or beyond `limit` may decide anything.
#pragma once

#include <stdint.h>

#if defined(__HIPCC__)
#define KTA_REC_HD __host__ __device__ inline __attribute__((always_inline))
#else
#define KTA_REC_HD inline
#endif

namespace kta {
namespace rec {

// Where a window of W bytes that has to hold the byte at `pos` begins: on a 128-byte line for the windows the dispatcher uses
// (the loads are non-temporal — the log is read once — and a window that begins inside a line would fetch that line twice), on a
// 16-byte block for the small windows of the tests.  Host statement and kernel agree on it, so they agree on the rounds.
// line_windows(W): ONE predicate for both — a geometry whose windows are whole KiB takes the wave-wide KiB loads of the kernel
// (kta_decode_coop.h) AND bases on 128-byte lines; every other window size gets the per-group 16-byte loads and 16-byte bases.
constexpr bool line_windows(uint32_t W) { return W % 1024u == 0u; }
KTA_REC_HD uint64_t window_base(uint64_t pos, uint32_t W) { return pos & ~(line_windows(W) ? 127ull : 15ull); }

// The four bytes at byte offset (sh & 3) of the dword pair lo, hi.
KTA_REC_HD uint32_t bytes4(uint32_t lo, uint32_t hi, uint32_t sh)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_alignbyte(hi, lo, sh & 3u);
#else
    return (uint32_t)((((uint64_t)hi << 32) | lo) >> (8u * (sh & 3u)));
#endif
}

KTA_REC_HD long long unzigzag32(uint32_t v) { return (long long)(int32_t)((v >> 1) ^ (0u - (v & 1u))); }

// A varint of one to four bytes at the start of w: its length and its (still zig-zag) value.  False — with a length
// of zero — when none of the four bytes ends it.
KTA_REC_HD bool varint4(uint32_t w, uint32_t &nb, uint32_t &v)
{
    const uint32_t stop = ~w & 0x80808080u;                 // bit 7 of every byte that ends a varint
    nb = (uint32_t)__builtin_ffs((int)stop) >> 3;           // the first of them: bit 7, 15, 23 or 31
    const uint32_t x = w & (stop - 1u) & 0x7F7F7F7Fu;       // the payload bits below it, seven to a byte
    const uint32_t y = x - ((x >> 1) & 0x3F803F80u);        // two halves of fourteen bits, sixteen apart
    v = (y & 0x3FFFu) | ((y >> 2) & 0x0FFFC000u);
    return stop != 0u;
}

// A varint of one or two bytes at the start of w — the length prefix of every record below 8 KiB: its length and
// its (still zig-zag) value.  False when the second byte does not end it.
KTA_REC_HD bool varint2(uint32_t w, uint32_t &nb, uint32_t &v)
{
    const uint32_t two = (w >> 7) & 1u;                     // the first byte goes on
    nb = 1u + two;
    v = (w & 0x7Fu) | ((w >> 1) & 0x3F80u & (0u - two));
    return (w & 0x8080u) != 0x8080u;
}

// Zig-zag varlong at byte offset `off` of the window, byte by byte (any length, the window's edge): the cold path.
KTA_REC_HD bool window_varlong(const uint8_t *win, uint32_t &off, uint32_t limit, long long &out)
{
    unsigned long long v = 0;
    for (uint32_t shift = 0; shift < 70; shift += 7) {
        if (off >= limit) return false;
        const uint32_t b = win[off++];
        v |= (unsigned long long)(b & 0x7Fu) << (shift < 64 ? shift : 63);
        if (!(b & 0x80u)) {
            out = (long long)(v >> 1) ^ -(long long)(v & 1ull);
            return true;
        }
    }
    return false;
}

// The four bytes at byte offset `off` of the window.  On the device ONE LDS read at the bytes' own address (gfx950's LDS takes
// unaligned dword reads: the compiler emits ds_read_b32 for the 4-byte copy), on the host the two dwords that hold them.
KTA_REC_HD uint32_t window_bytes4(const uint32_t *w32, uint32_t off)
{
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t v;
    __builtin_memcpy(&v, reinterpret_cast<const uint8_t *>(w32) + off, 4);
    return v;
#else
    const uint32_t i = off >> 2;
    return bytes4(w32[i], w32[i + 1], off);
#endif
}

// The leader's chain: from `cur`, follow the length prefixes while they are ordinary — one or two bytes (a record below
// 8 KiB; a longer one does not fit a window, so the round ends with it anyway), all four bytes at `cur` inside the
// window — and publish the record starts; at most `want` records in all (k counts them).
// Stops at the window's edge (fewer than four bytes left: the next window begins at `cur`), at `want`, or — returning
// true — in front of a length of three bytes or more, which the caller takes the long way.  The chain does not judge
// the lengths: a negative one or one that overruns the batch leaves garbage starts behind it, all of them inside
// [0, 2^15), and the lane that parses the record reports it (parse_record and its caller), which condemns the batch.
// One step is an LDS round trip and some twenty operations on the leader's lane alone; it is what a batch costs in
// serial time, so nothing that can wait for the parse is done here.
KTA_REC_HD bool chain(const uint32_t *w32, uint32_t limit, uint32_t want, uint32_t *starts, uint32_t &k, uint32_t &cur)
{
    if (limit < 4u) return false;
    const uint32_t last = limit - 4u;                // a step reads the four bytes at `cur`
    constexpr uint32_t LONG = 0x80000000u;           // `cur` in front of a long one: beyond every `last`
    bool go = (k < want) & (cur <= last);
    while (go) {                                     // one block, one branch: the leaders of a wave run it in step
        uint32_t nb, v;
        const bool ordinary = varint2(window_bytes4(w32, cur), nb, v);
        starts[k] = cur;                             // (k < want: inside the array; it stays only if k moves on)
        k += ordinary ? 1u : 0u;
        cur = ordinary ? cur + nb + (v >> 1) : LONG; // < 2^13 + 2 + 2^13: no overflow, whatever the bytes
        go = (k < want) & (cur <= last);
    }
    if (cur != LONG) return false;
    cur = starts[k];                                 // back in front of the long one
    return true;
}

enum : uint32_t {
    REC_OK = 0,
    REC_INCOMPLETE = 1,     // the header does not lie inside the window: the next window starts at this record
    REC_BAD = 2,            // negative length, key or value overrun the record
    REC_VALUE_LENGTH_OUTSIDE = 3   // the value length lies behind the window (a key of the window's size): `after` = where
};

struct Record {
    long long ts_delta, key_len, val_len;   // key_len, val_len: -1 = null
    uint32_t key;                           // position of the key
    uint32_t after;                         // position behind the value length (REC_VALUE_LENGTH_OUTSIDE: of the value length)
};

// The last word on a record once its value length is known: the value must end inside the record.
KTA_REC_HD uint32_t value_fits(long long val_len, uint64_t after, uint64_t rec_end)
{
    return val_len >= -1 && after + (uint64_t)(val_len > 0 ? val_len : 0) <= rec_end ? REC_OK : REC_BAD;
}

// One record: `start` is where its length prefix begins, `rec_end` where the next record begins (from the chain; the
// caller has checked it against the batch's end).  The ordinary header — a length of up to three bytes, then the
// attributes byte and three varints of up to four bytes and eight bytes in all — lies in the four dwords at `start`
// (ONE round trip) and is taken apart in registers; the value length costs a second round trip.  Everything else
// (longer varints, the window's edge) goes byte by byte.
KTA_REC_HD uint32_t parse_record(const uint8_t *win, uint32_t start, uint32_t rec_end, uint32_t limit, Record &r)
{
    const uint32_t *w32 = reinterpret_cast<const uint32_t *>(win);
    uint32_t off = 0;
    bool head = false;
    {
#if defined(__HIP_DEVICE_COMPILE__)
        uint32_t a012[3];                            // the twelve bytes at `start`, at their own address (see window_bytes4)
        __builtin_memcpy(a012, win + start, 12);
        const uint32_t a0 = a012[0], a1 = a012[1], a2 = a012[2];
#else
        const uint32_t i0 = start >> 2;
        const uint32_t d0 = w32[i0], d1 = w32[i0 + 1], d2 = w32[i0 + 2], d3 = w32[i0 + 3];
        const uint32_t a0 = bytes4(d0, d1, start), a1 = bytes4(d1, d2, start), a2 = bytes4(d2, d3, start);
#endif
        uint32_t nbl, len;
        bool ok = varint4(a0, nbl, len) && nbl <= 3u && !(len & 1u);           // (a negative length: the long way says so)
        const uint32_t skip = 8u * ((nbl <= 3u ? nbl : 3u) + 1u);           // the length and the attributes byte: 16..32 bits
        unsigned long long x = (uint32_t)((((unsigned long long)a1 << 32) | a0) >> skip) |
                               ((((unsigned long long)a2 << 32) | a1) >> skip << 32);
        uint32_t used = 0, val[3];
#pragma unroll
        for (int f = 0; f < 3; f++) {
            uint32_t nb;
            ok = varint4((uint32_t)x, nb, val[f]) && ok;
            x >>= 8u * nb;
            used += nb;
        }
        const uint32_t body = start + (skip >> 3);
        if (ok && used <= 8u && body + used <= limit) {
            r.ts_delta = unzigzag32(val[0]);
            r.key_len = unzigzag32(val[2]);
            off = body + used;
            head = true;
        }
    }
    if (!head) {
        long long len, offset_delta;
        off = start;
        if (!window_varlong(win, off, limit, len)) return REC_INCOMPLETE;
        if (len < 0) return REC_BAD;
        off += 1;                                                           // the attributes byte
        if (!(window_varlong(win, off, limit, r.ts_delta) && window_varlong(win, off, limit, offset_delta) &&
              window_varlong(win, off, limit, r.key_len)))
            return REC_INCOMPLETE;
    }
    r.key = off;
    if (r.key_len < -1 || r.key_len > 0x7FFFFFFFll) return REC_BAD;
    const uint32_t voff = off + (r.key_len > 0 ? (uint32_t)r.key_len : 0u);   // < 2^32: off <= limit, a window
    if (voff >= rec_end) return REC_BAD;                                    // the key overruns the record
    if (voff < limit) {
        uint32_t nb, v;
        if (varint4(window_bytes4(w32, voff), nb, v) && voff + nb <= limit) {
            r.val_len = unzigzag32(v);
            r.after = voff + nb;
            return value_fits(r.val_len, r.after, rec_end);
        }
        uint32_t o = voff;
        if (window_varlong(win, o, limit, r.val_len)) {
            r.after = o;
            return value_fits(r.val_len, r.after, rec_end);
        }
    }
    r.after = voff;
    return REC_VALUE_LENGTH_OUTSIDE;
}

} // namespace rec
} // namespace kta
