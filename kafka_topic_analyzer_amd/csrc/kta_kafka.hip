// kta_kafka.hip — Kafka record-batch v2 decode (include/kta_kafka.h): host header index + gfx950 kernels
// that turn raw record sets into the struct-of-arrays columns the metric kernels consume.
//
//   host    kta_kafka_index_host      walks the 61-byte batch headers, sizes the inflate slices
//   device  kafka_crc32c              optional CRC-32C check (librdkafka's check.crcs), one wave per batch
//           kafka_snappy_inflate_coop / kafka_lz4_inflate_coop   wave-cooperative inflate (64 bytes per step)
//           kafka_gzip_tokenize<L> + kafka_gzip_apply            gzip in two stages: Huffman decoding one lane per
//                                                                batch (literals in place, matches as tokens),
//                                                                then the copies one wave per batch
//           kafka_zstd_inflate_coop                              zstd one wave per batch (uniform parsing, LDS tables,
//                                                                64-byte copy steps, Huffman streams on four lanes)
//           kafka_decode_coop<G, W, R>  record parse: G batches per wave through LDS windows (kta_decode_coop.h)
//           kafka_decode / kafka_inflate_lane / kafka_gzip_inflate / kafka_zstd_inflate
//                                                                one-lane-per-batch forms kept for comparison
//                                                                (kta_kafka_set_variant 1)
//   host    kta_kafka_blob_acquire / _submit / kta_kafka_consume   pinned staging ring -> PCIe -> the above ->
//                                                                  kta_submit_device (both handlers)
//
// Batches are independent (each carries its own base timestamp and record count); the records inside a
// batch form a linked list (length prefixes).  Value bytes are never needed — the reference's handlers
// only take their length (src/metric.rs:233-245) — and keys stay where they are (key_off points into the
// raw blob / inflate area).
#include "../../include/kta_kafka.h"
#include "../../include/kta_synth.h"
#include "kta_snappy.h"
#include "kta_lz4.h"
#include "kta_gzip.h"
#include "kta_zstd.h"
#include "kta_records.h"

#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

hipStream_t kta_internal_stream(kta_ctx *ctx);
int kta_internal_device(kta_ctx *ctx);
void kta_internal_set_error(kta_ctx *ctx, const char *msg);
void **kta_internal_ext_slot(kta_ctx *ctx, void (*free_fn)(void *));
uint64_t kta_internal_take_seq(kta_ctx *ctx, uint64_t n);
bool kta_internal_timing(kta_ctx *ctx);
bool kta_internal_count_alive(kta_ctx *ctx);
hipStream_t kta_internal_copy_stream(kta_ctx *ctx);

namespace {

constexpr int kLanesPerBlock = 64; // one wave per workgroup: spreads few batches over many CUs
constexpr uint64_t kMaxBatchInflate = 512ull << 20;   // inflated size one batch may claim

// Automatic choice.  Every geometry the library could dispatch at the start of round 5 was timed side by side on one
// box (profiles/r05_decode_geometries.jsonl: 2 M records as ~2 / ~16 / ~134 KiB batches, kernel ms):
//                          2 KiB    16 KiB   134 KiB
//   <4, 3 KiB, 16>   (10)  0.130    0.146    0.231      <4, 2 KiB, 16>: 0.129 / 0.152 / 0.286;  <8, 1 KiB, 16>: 0.133 / 0.180 / 0.476
//   <2, 8 KiB, 32>   (11)  0.253    0.141    0.149      128 / 64 records per round: 0.147 / 0.147 and 0.163 / 0.161;
//                                                       with the next window prefetched into registers: 0.155 / 0.148 and 0.160 / 0.158
//   <1, 8 KiB, 256>  (2)   0.283    0.142    0.154      <4, 8 KiB, 128>: 0.306 / 0.194 / 0.205
// Every geometry sat on one plateau of 3.7-4.2 TB/s, and the round's last calls found out why by switching parts of the kernel off
// (profiles/r05_decode_switches.jsonl): the time is the windows' way through LDS (6.0-6.7 TB/s on its own) plus the column stores;
// chain and parse hide behind other waves' loads.  What moved it: non-temporal window loads on 128-byte bases and a KiB of one
// window per load instruction (kta_decode_coop.h) — <4, 3 KiB, 16> 0.137 -> 0.120 ms at 16 KiB batches (4 M records: 0.248 -> 0.217,
// 0.282 -> 0.235 on a slower box), <2, 8 KiB, 32> 0.139 -> 0.136; 134 KiB batches 0.141 -> 0.137.  With it four batches per wave
// pay up to about 30 KiB batches (8 KiB batches 0.124 / 0.146 ms, 16 KiB 0.125 / 0.138, 32 KiB 0.138 / 0.136; few large batches are
// bound by the serial chain of a batch: 134 KiB 0.23 / 0.137).  The losers' instantiations, the prefetching forms, the staged and the
// deferred column stores were deleted with their timings kept.
int decode_variant_for(int forced, uint64_t n_batches, uint64_t blob_len)
{
    if (forced) return forced;
    if (n_batches < 2048) return 2;            // few batches: one wave each
    const uint64_t mean = blob_len / n_batches;
    return mean < 28672 ? 10 : 11;
}

#include "kta_decode_coop.h"   // Reader, read_varlong, pin, kafka_decode_coop<G, W, R>
#include "kta_gzip_wave.h"     // gzip_tokenize_wave: stage 1 of the gzip inflate, one wave per batch
#include "kta_zstd_huf_wave.h" // zh_streams: the Huffman-coded literals of a zstd block, by the whole wave

// Walk one batch.  WRITE = false: only total the key bytes.  Returns false if the records overrun
// the batch (corrupt / truncated batch).
template <bool WRITE>
__device__ __forceinline__ bool walk_batch(const uint4 *words, const kta_kafka_batch_desc &d, bool want_keys,
                                           int32_t *part, int32_t *klen_out, int32_t *vlen_out, int64_t *ts_out,
                                           uint32_t *koff_out, uint64_t blob_base, uint64_t *seq_out,
                                           uint64_t seq_base, uint64_t *key_total)
{
    Reader r{words, d.payload_off, ~0ull, make_uint4(0, 0, 0, 0)};
    const uint64_t batch_end = d.payload_end;
    uint64_t kb = 0;
    bool ok = true;
    int32_t j = 0;
    for (; j < d.n_records; j++) {
        if (r.pos >= batch_end) { ok = false; break; }
        const long long len = read_varlong(r);
        const uint64_t rec_end = r.pos + (uint64_t)len;
        if (len < 0 || rec_end > batch_end) { ok = false; break; }
        (void)r.byte();                                  // record attributes (unused)
        const long long ts_delta = read_varlong(r);
        (void)read_varlong(r);                           // offsetDelta
        const long long kl = read_varlong(r);            // -1 = null key
        const uint64_t key_pos = r.pos;
        if (kl > 0) r.pos += (uint64_t)kl;
        if (kl < -1 || r.pos > rec_end) { ok = false; break; }
        const long long vl = read_varlong(r);            // -1 = null value (tombstone)
        if (vl < -1 || r.pos + (uint64_t)(vl > 0 ? vl : 0) > rec_end) { ok = false; break; }
        if (WRITE) {
            const uint64_t i = d.record_base + (uint64_t)j;
            part[i] = d.partition;
            klen_out[i] = (int32_t)kl;
            vlen_out[i] = (int32_t)vl;
            ts_out[i] = (d.flags & KTA_KB_LOG_APPEND_TIME) ? d.max_ts_ms : d.base_ts_ms + ts_delta;
            if (seq_out) seq_out[i] = seq_base + i;
            // zero-copy keys: key_off points at the key inside the raw blob (key_bytes == the blob)
            if (want_keys) koff_out[i] = (uint32_t)(kl > 0 ? key_pos - blob_base : 0);
        }
        kb += kl > 0 ? (uint64_t)kl : 0;
        r.pos = rec_end;                                 // skip the value and the headers
    }
    if (WRITE && !ok) {
        // make the damage visible downstream: the remaining records carry partition -1
        for (; j < d.n_records; j++) {
            const uint64_t i = d.record_base + (uint64_t)j;
            part[i] = -1; klen_out[i] = -1; vlen_out[i] = -1; ts_out[i] = -1;
            if (seq_out) seq_out[i] = seq_base + i;
            if (want_keys) koff_out[i] = 0u;
        }
    }
    *key_total = kb;
    return ok;
}

__global__ __launch_bounds__(kLanesPerBlock) void kafka_decode(const uint4 *words, const kta_kafka_batch_desc *descs,
                                                               uint64_t n_batches, int want_keys, int32_t *part,
                                                               int32_t *klen, int32_t *vlen, int64_t *ts, uint32_t *koff,
                                                               uint64_t blob_base, uint64_t *seq, uint64_t seq_base,
                                                               unsigned long long *n_bad, unsigned long long *n_keyb)
{
    const uint64_t b = (uint64_t)blockIdx.x * kLanesPerBlock + threadIdx.x;
    if (b >= n_batches) return;
    kta_kafka_batch_desc d = descs[b];
    if (d.status) d.payload_end = d.payload_off;   // failed check.crcs / inflate: not delivered, walk nothing
    uint64_t kb = 0;
    const bool ok = walk_batch<true>(words, d, want_keys != 0, part, klen, vlen, ts, koff, blob_base, seq, seq_base, &kb);
    if (!ok) atomicAdd(n_bad, 1ull);
    if (n_keyb && kb) atomicAdd(n_keyb, (unsigned long long)kb);
}

// ---- per-context state for kta_kafka_consume / timing ---------------------------------------------
// ---- Snappy inflate: one lane per compressed batch ------------------------------------------------
// (kta_snappy.h; the same function is compiled for the host and checked there against the oracle.)
// The inflated records land in the batch's slice of the inflate area, which is part of the same
// device buffer as the raw blob, so key offsets stay valid for the zero-copy alive-key pass.
__global__ __launch_bounds__(kLanesPerBlock) void kafka_inflate_lane(uint8_t *buffer, kta_kafka_batch_desc *descs,
                                                                      uint64_t n_batches, uint32_t codecs)
{
    const uint64_t b = (uint64_t)blockIdx.x * kLanesPerBlock + threadIdx.x;
    if (b >= n_batches) return;
    const kta_kafka_batch_desc d = descs[b];
    if (!(d.flags & codecs) || d.status) return;       // `codecs`: the KTA_KB_* codecs this launch handles
    const uint8_t *src = buffer + d.byte_off + KTA_KAFKA_BATCH_HEADER;
    const uint64_t n = (uint64_t)d.batch_bytes - KTA_KAFKA_BATCH_HEADER, cap = d.payload_end - d.payload_off;
    if (d.flags & KTA_KB_SNAPPY) {
        const int64_t got = kta::snappy_inflate(src, n, buffer + d.payload_off, cap);
        if (got < 0 || (uint64_t)got != cap) descs[b].status = KTA_KB_BAD_FRAMING;
    } else {
        const int64_t got = kta::lz4_inflate(src, n, buffer + d.payload_off, cap);   // cap is only a bound
        if (got < 0) descs[b].status = KTA_KB_BAD_FRAMING;
        else descs[b].payload_end = d.payload_off + (uint64_t)got;
    }
}

// ---- gzip inflate: one lane per batch, L batches per workgroup --------------------------------------
// DEFLATE is bit-serial inside a member; the parallelism is across the batches of a fetch.  Everything
// the decoder indexes with data (symbol lists, code lengths) is in LDS, lane-interleaved; the code
// limits stay in registers (csrc/kta_gzip.h).  L = 16 leaves lanes of a wave unused on purpose: the work
// is a chain of dependent memory accesses and the LDS work area (1.3 KiB per batch) bounds the batches
// in flight per CU either way, so more, emptier waves hide more latency (measured: 16 lanes 12 ms, 64
// lanes 40 ms for 16 667 batches of 16 KiB).
constexpr uint32_t kGzipLanes = 16;
template <uint32_t L>
__global__ __launch_bounds__(L) void kafka_gzip_inflate(uint8_t *buffer, kta_kafka_batch_desc *descs, uint64_t n_batches)
{
    __shared__ uint16_t s_work[kta::GZ_WORK * L];
    const uint64_t b = (uint64_t)blockIdx.x * L + threadIdx.x;
    if (b >= n_batches) return;
    const kta_kafka_batch_desc d = descs[b];
    if (!(d.flags & KTA_KB_GZIP) || d.status) return;
    const uint8_t *src = buffer + d.byte_off + KTA_KAFKA_BATCH_HEADER;
    const uint64_t n = (uint64_t)d.batch_bytes - KTA_KAFKA_BATCH_HEADER, cap = d.payload_end - d.payload_off;
    const int64_t got = kta::gzip_inflate(src, n, buffer + d.payload_off, cap, s_work + threadIdx.x, L);
    if (got < 0 || (uint64_t)got != cap) descs[b].status = KTA_KB_BAD_FRAMING;   // the trailer told the size
}

// ---- gzip inflate in two stages (csrc/kta_gzip.h) -----------------------------------------------------
// Stage 1: Huffman decoding.  kafka_gzip_tokenize_wave (kta_gzip_wave.h; round 6) gives a batch to one WAVE, whose 64 lanes
// decode a block's symbols from speculative starts that synchronise; what it leaves — stored blocks, streams it finds
// malformed — kafka_gzip_tokenize<L> does afterwards: one LANE per batch, L batches per workgroup, the lookup tables of a lane
// (1.9 KiB) in LDS, lane-interleaved.  Literals go straight to their final bytes of the batch's slice, matches become tokens in
// the batch's scratch (the host index placed it behind the slice: [u32 count, u32 0, tokens]).  The lane kernel has few lanes
// per wave on purpose: the lanes of a wave sit in different branches (literal / match / table build) most of the time, so a
// wave costs the sum of its lanes' paths, and the LDS tables bound the batches in flight per CU either way (measured on
// 16 667 batches of 16 KiB, when it was the only stage 1: 8 lanes 4.41 ms, 16 lanes 4.71 ms for inflate + decode).
// Stage 2, kafka_gzip_apply: one WAVE per batch executes the tokens.  The output is processed in 2 KiB
// chunks through an LDS ring of the last 4 KiB: a chunk is loaded with its literals in place, the matches
// that start in it are copied 64 bytes per step inside LDS (a dependent step costs LDS latency, not a
// memory round trip), and the chunk is written back in whole 16-byte units.  A match that reaches further
// back than the ring reads the written-back output (behind a fence, past L1).
constexpr uint32_t kGzTokLanes = 8;
// (16 KiB / 4 KiB until round 6: ten waves per CU, each waiting on its own chain of LDS round trips — 0.67 ms for 16 667 batches
// of 16 KiB; 8 KiB / 4 KiB 0.46; 4 KiB / 2 KiB 0.40: more waves, and the matches behind the ring read L2 at little cost)
constexpr uint32_t kLzRing = 4096, kLzChunk = 2048;

template <uint32_t L>
__global__ __launch_bounds__(L) void kafka_gzip_tokenize(uint8_t *buffer, kta_kafka_batch_desc *descs, uint64_t n_batches, int only_left)
{
    __shared__ uint16_t s_work[kta::GZ2_WORK * L];
    __shared__ uint32_t s_win[kta::GZ_WIN / 4 * L];            // the lanes' windows on their streams
    const uint64_t b = (uint64_t)blockIdx.x * L + threadIdx.x;
    if (b >= n_batches) return;
    const kta_kafka_batch_desc d = descs[b];
    if (!(d.flags & KTA_KB_GZIP) || d.status) return;
    const uint8_t *src = buffer + d.byte_off + KTA_KAFKA_BATCH_HEADER;
    const uint64_t n = (uint64_t)d.batch_bytes - KTA_KAFKA_BATCH_HEADER, cap = d.payload_end - d.payload_off;
    const uint64_t scratch = (d.payload_end + 63) & ~63ull;
    const bool has_tokens = d.scratch_end >= scratch + 16;          // (an empty member has none and needs none)
    if (cap && !has_tokens) { descs[b].status = KTA_KB_BAD_FRAMING; return; }
    uint32_t *tok = reinterpret_cast<uint32_t *>(buffer + scratch);
    if (only_left && has_tokens && tok[0] != kGwNotDone) return;    // kafka_gzip_tokenize_wave has done this one
    uint64_t n_tok = 0;
    kta::GzBitsWin bits;
    bits.win = s_win + threadIdx.x;
    bits.wstride = L;
    const int64_t got = kta::gzip_tokenize(bits, src, n, buffer + d.payload_off, cap, tok + 2,
                                           has_tokens ? (d.scratch_end - scratch - 8) / 4 : 0, &n_tok, s_work + threadIdx.x, L);
    if (got < 0 || (uint64_t)got != cap) descs[b].status = KTA_KB_BAD_FRAMING;   // the trailer told the size
    else if (has_tokens) tok[0] = (uint32_t)n_tok;
}

// Stage 1 with one WAVE per batch (kta_gzip_wave.h): the block's symbols decoded by 64 lanes from speculative starts that
// synchronise.  What it leaves (kGwNotDone in the token count) kafka_gzip_tokenize<L> does afterwards.
__global__ __launch_bounds__(64) KTA_WAVES_PER_EU(4, 8) void kafka_gzip_tokenize_wave(uint8_t *buffer, kta_kafka_batch_desc *descs, uint64_t n_batches)
{
    __shared__ GwShared s_gw;
    gzip_tokenize_wave(s_gw, buffer, descs, n_batches);
}

// The last kLzRing bytes of one batch's output, mirrored in LDS and moved in chunks (one wave; see above).
struct LzWindow {
    uint4 *ring4;          // LDS, kLzRing bytes
    uint8_t *dst;          // the batch's slice (64-byte aligned)
    uint32_t total;        // bytes of output (< 2 GiB)
    uint32_t c0;           // the chunk [c0, c0 + kLzChunk) is in the ring and not yet written back
    uint32_t lane;

    __device__ __forceinline__ uint8_t *ring() const { return reinterpret_cast<uint8_t *>(ring4); }
    __device__ __forceinline__ void load(uint32_t at)           // chunk `at` (literals in place) -> ring
    {
        static_assert(kLzChunk % 1024 == 0 && kLzRing >= kLzChunk + 64, "whole 16-byte units per lane; the step before a chunk is in the ring");
        c0 = at;
        const uint32_t last = ((total + 15u) & ~15u) - 16u;      // last 16-byte unit of the slice (total > 0)
        static_assert(kLzChunk == 2048, "two 16-byte units per lane");
        const uint32_t o0 = at + lane * 16u, o1 = o0 + 1024u;
        const uint4 v0 = *reinterpret_cast<const uint4 *>(dst + (o0 < last ? o0 : last));   // in flight together
        const uint4 v1 = *reinterpret_cast<const uint4 *>(dst + (o1 < last ? o1 : last));
        ring4[(o0 & (kLzRing - 1)) >> 4] = v0;
        ring4[(o1 & (kLzRing - 1)) >> 4] = v1;
        __syncthreads();
    }
    __device__ __forceinline__ void advance()                   // write the chunk back, take the next one
    {
        __syncthreads();
#pragma unroll
        for (uint32_t k = 0; k < kLzChunk / 1024; k++) {
            const uint32_t off = c0 + (k * 64 + lane) * 16u;
            if (off < total) *reinterpret_cast<uint4 *>(dst + off) = ring4[(off & (kLzRing - 1)) >> 4];
        }
        if (c0 + kLzChunk < total) load(c0 + kLzChunk);
        else c0 += kLzChunk;
    }
    // out[op .. op + len) = out[op - dist ..), op inside or after the current chunk; dist <= op, op + len <= total
    __device__ __forceinline__ void match(uint32_t op, uint32_t dist, uint32_t len)
    {
        while (len) {
            while (op >= c0 + kLzChunk) advance();
            uint32_t n = c0 + kLzChunk - op;
            n = n < len ? n : len;
            const bool in_ring = op - dist + kLzRing >= c0 + kLzChunk;   // the whole source is in the ring
            // The n bytes of the match in this chunk at once, 64 per step with no step waiting for another: byte i is the byte
            // i mod dist of the `dist` bytes before op, and those are final.  (Until round 6 a step copied from what the step
            // before had written — a chain of LDS round trips per match — behind ~ 45 scalar instructions of loop control, and
            // the kernel was bound by the scalar unit's issue rate: 12 K scalar instructions per batch.)
            const bool periodic = dist < n;
            const float inv = periodic ? __frcp_rn((float)dist) * 0.99999976f : 0.0f;   // (a hair below 1 / dist: the quotient is never too big)
            if (!in_ring) __threadfence_block();       // further back than the ring: written back already (kLzRing >= chunk + 64) —
                                                       // the wave's own earlier stores have reached L2: a wait, no cache maintenance
            for (uint32_t i0 = 0; i0 < n; i0 += 64) {
                const uint32_t i = i0 + lane;
                uint32_t m = i;
                if (periodic) {
                    m = i - (uint32_t)((float)i * inv) * dist;
                    m = m >= dist ? m - dist : m;
                }
                if (i < n) {
                    const uint32_t s = op - dist + m;
                    const uint8_t v = in_ring ? ring()[s & (kLzRing - 1)] : __hip_atomic_load(dst + s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ring()[(op + i) & (kLzRing - 1)] = v;
                }
            }
            op += n;
            len -= n;
        }
    }
    __device__ __forceinline__ void finish()                    // everything up to `total` written back
    {
        while (c0 < total) advance();
    }
};

__global__ __launch_bounds__(64) void kafka_gzip_apply(uint8_t *buffer, kta_kafka_batch_desc *descs, uint64_t n_batches)
{
    __shared__ uint4 s_ring[kLzRing / 16];
    const uint32_t lane = threadIdx.x;
    const uint64_t b = blockIdx.x;
    if (b >= n_batches) return;
    const kta_kafka_batch_desc d = descs[b];
    if (!(d.flags & KTA_KB_GZIP) || d.status || d.payload_end == d.payload_off) return;
    const uint32_t total = (uint32_t)(d.payload_end - d.payload_off);          // (the index caps a batch at 512 MiB)
    const uint32_t *tok = reinterpret_cast<const uint32_t *>(buffer + ((d.payload_end + 63) & ~63ull));
    const uint32_t n_tok = tok[0];
    if (n_tok == 0) return;                                    // literals only: stage 1 wrote everything
    LzWindow w{s_ring, buffer + d.payload_off, total, 0, lane};
    w.load(0);
    uint32_t op = 0;
    bool bad = false;
    // 64 tokens at a time, a lane each: a prefix sum of what the tokens advance gives every match its place, and then the
    // matches are copied in ROUNDS — in a round every lane whose match is short, lies in the chunk and copies from bytes
    // that no match still waiting in front of it will write (its source ends before the first waiting match begins: the
    // literals are all in place already) copies it by itself, byte by byte inside LDS; the first waiting match is always
    // one of them, or, if it is long or crosses the chunk's end, the wave copies it together as before (LzWindow::match).
    // Values of words and punctuation make ~ 1 500 matches of ~ 10 bytes per batch; one after the other, ~ 70 instructions
    // each, they were 2.8 ms of the 4.4 a million such records took.
    constexpr uint32_t kShort = 32;
    for (uint32_t t0 = 0; t0 < n_tok && !bad; t0 += 64) {
        const uint32_t tk = t0 + lane < n_tok ? tok[2 + t0 + lane] : 0u;
        const uint32_t run = tk & 255u, len = (tk >> 8) & 511u, dist = (tk >> 17) + 1u;
        uint32_t incl = run + len;
#pragma unroll
        for (uint32_t off = 1; off < 64; off <<= 1) {
            const uint32_t up = __shfl_up(incl, off);
            if (lane >= off) incl += up;
        }
        const uint32_t group = __builtin_amdgcn_readlane(incl, 63);
        const uint32_t mop = op + incl - len;                              // my match's first byte
        if (__builtin_amdgcn_ballot_w64(len && (mop > total || dist > mop || len > total - mop))) {   // (stage 1 accepts only tokens inside the output)
            bad = true;
            break;
        }
        const uint32_t src_end = mop - dist + (len < dist ? len : dist);   // (the bytes a self-overlapping match reads before its own)
        unsigned long long pending = __builtin_amdgcn_ballot_w64(len != 0);
        while (pending) {
            const uint32_t first = (uint32_t)__builtin_ctzll(pending);
            const uint32_t f_op = __builtin_amdgcn_readlane(mop, first), f_len = __builtin_amdgcn_readlane(len, first);
            const uint32_t f_dist = __builtin_amdgcn_readlane(dist, first);
            while (f_op >= w.c0 + kLzChunk) w.advance();                   // (everything before the first waiting match is final)
            const uint32_t cend = w.c0 + kLzChunk;
            if (f_len > kShort || f_op + f_len > cend || f_op - f_dist + kLzRing < cend) {
                w.match(f_op, f_dist, f_len);
                pending &= ~(1ull << first);
                continue;
            }
            const bool now = ((pending >> lane) & 1ull) && len <= kShort && mop + len <= cend && mop - dist + kLzRing >= cend &&
                             (lane == first || src_end <= f_op);
            if (now)
                for (uint32_t k = 0; k < len; k++) w.ring()[(mop + k) & (kLzRing - 1)] = w.ring()[(mop - dist + k) & (kLzRing - 1)];
            pending &= ~__builtin_amdgcn_ballot_w64(now);
        }
        op += group;
    }
    w.finish();
    if (bad && lane == 0) descs[b].status = KTA_KB_BAD_FRAMING;
}

// ---- zstd inflate: one lane per batch -------------------------------------------------------------
// csrc/kta_zstd.h; the decoder's tables (10.6 KiB) and the Huffman-decoded literals of a block live in the
// batch's scratch, which the host index placed right behind its slice of the inflate area.
constexpr uint64_t kZstdWorkBytes = (sizeof(kta::ZsWork) + 63) & ~63ull;

__global__ __launch_bounds__(kGzipLanes) void kafka_zstd_inflate(uint8_t *buffer, kta_kafka_batch_desc *descs, uint64_t n_batches)
{
    const uint64_t b = (uint64_t)blockIdx.x * kGzipLanes + threadIdx.x;
    if (b >= n_batches) return;
    const kta_kafka_batch_desc d = descs[b];
    if (!(d.flags & KTA_KB_ZSTD) || d.status) return;
    const uint8_t *src = buffer + d.byte_off + KTA_KAFKA_BATCH_HEADER;
    const uint64_t n = (uint64_t)d.batch_bytes - KTA_KAFKA_BATCH_HEADER, cap = d.payload_end - d.payload_off;
    const uint64_t scratch = (d.payload_end + 63) & ~63ull;          // 64-byte aligned: the tables are u32
    if (d.scratch_end < scratch + kZstdWorkBytes) { descs[b].status = KTA_KB_BAD_FRAMING; return; }
    kta::ZsWork *w = reinterpret_cast<kta::ZsWork *>(buffer + scratch);
    const int64_t got = kta::zstd_inflate(src, n, buffer + d.payload_off, cap, w, buffer + scratch + kZstdWorkBytes,
                                          d.scratch_end - scratch - kZstdWorkBytes);
    if (got < 0) descs[b].status = KTA_KB_BAD_FRAMING;
    else descs[b].payload_end = d.payload_off + (uint64_t)got;       // the slice was sized by a bound
}

// ---- zstd inflate, wave-cooperative: one wave per batch ------------------------------------------------
// csrc/kta_zstd.h run by a whole wave: every lane parses the frame / block / table descriptions and decodes
// the sequences redundantly (uniform work: the compressed bytes come through an LDS window that all 64 lanes
// refill together, the FSE / Huffman tables are in LDS), and the parts with width use it — literals and
// matches move 64 bytes per step through an LDS mirror of the last 8 KiB of output (further back: the
// written output, behind a fence), the four Huffman streams of a literals section run on four lanes.
// The policy methods are force-inlined: a call would pass `this` through memory, and the LDS address
// spaces of the window, the ring and the tables would be lost (flat accesses, private-memory traffic).
constexpr uint32_t kZsWin = 2048;                  // (1 KiB — a thirteenth wave per CU — measured the same: round 6)
constexpr int32_t kZsNoWindowRel = INT32_MIN / 2;  // (at - kZsNoWindowRel is huge for every position `at` of a batch: "not in the window")

struct ZsWaveSrc {                 // byte source: the batch payload behind an LDS window
    const uint8_t *buffer;
    uint64_t src0;                 // the payload is buffer[src0 .. src0 + n)
    uint32_t n;
    uint4 *win4;
    int32_t wlo;                   // the window's first byte, relative to src0 (>= -15: it begins on a 16-byte boundary of the
                                   // buffer); kZsNoWindowRel: empty.  (Relative and 32 bits wide: "is it in the window" is two
                                   // scalar instructions, not the eight of a 64-bit compare of buffer offsets, and a sequence asks twice.)
    uint32_t lane;

    __device__ __forceinline__ const uint8_t *memory() const { return buffer + src0; }
    __device__ __forceinline__ uint32_t uni(uint32_t x) const { return (uint32_t)__builtin_amdgcn_readfirstlane((int)x); }
    __device__ __forceinline__ bool inside(uint32_t at) const { return (uint32_t)((int32_t)at - wlo) < kZsWin; }
    __device__ __forceinline__ void fetch(uint32_t at, uint32_t need)   // make [at, at + need) readable through the window
    {
        __syncthreads();
        const int32_t mis = (int32_t)((uint32_t)src0 & 15u);            // src0 - mis is a 16-byte boundary
        int32_t base = (int32_t)((at + (uint32_t)mis) & ~15u) - mis;
        if (wlo != kZsNoWindowRel && (int32_t)at < wlo) {               // reading backwards (bit streams): the window ends just above
            const int32_t end = (int32_t)((at + need + (uint32_t)mis + 15u) & ~15u) - mis;
            base = end - (int32_t)kZsWin >= -mis ? end - (int32_t)kZsWin : -mis;
        }
        wlo = base;
        const int32_t last = (int32_t)((n + (uint32_t)mis + 15u) & ~15u) - 16 - mis;     // last readable unit of the batch
        static_assert(kZsWin == 2048, "two 16-byte units per lane");
        const int32_t a0 = wlo + (int32_t)lane * 16, a1 = a0 + 1024;
        const uint8_t *m = buffer + src0;
        const uint4 v0 = *reinterpret_cast<const uint4 *>(m + (a0 < last ? a0 : last));    // in flight together
        const uint4 v1 = *reinterpret_cast<const uint4 *>(m + (a1 < last ? a1 : last));
        win4[lane] = v0;
        win4[lane + 64] = v1;
        __syncthreads();
    }
    __device__ __forceinline__ uint32_t byte(uint32_t at)          // `at` is the same in every lane
    {
        if (!inside(at)) fetch(at, 1);
        return uni(reinterpret_cast<const uint8_t *>(win4)[(int32_t)at - wlo]);
    }
    // The FSE decoding table of w.norm[0 .. n_sym) (kta_zstd.h: zs_build_fse, RFC 8878 4.1.1) with all 64 lanes: the spec's
    // loops — spread the symbols over the cells with a stride, then walk the cells in order handing every symbol its states —
    // are two chains of dependent LDS accesses, 2^log long, which a wave that executes them lane-uniformly pays at full LDS
    // latency each: three tables of 2^9 cells were most of the 58 k instructions and 0.35 ms a batch took (rounds 2-5).
    //   * the symbols "less than one" (norm -1) take the cells from the top, one each, in symbol order: a ballot;
    //   * the stride is odd, so raw step r lands on cell (r x step) mod size, every cell once; the cells below `high` are
    //     kept, in the order of r, and kept cell number j belongs to the symbol whose run of counts holds j (prefix sums of
    //     the counts, one symbol per lane; a cell finds its symbol by bisection);
    //   * cell i's state number is count(symbol) + the cells of that symbol below i: 64 cells at a time in ascending order,
    //     a lane counts its symbol among the lower lanes (64 readlanes) on top of what the chunks before left in w.next.
    template <class W>
    __device__ __forceinline__ bool build_fse(W &w, uint32_t *t, uint32_t log, uint32_t n_sym)
    {
        const uint32_t size = 1u << log, mask = size - 1u;
        __syncthreads();                                   // (w.norm was written lane-uniformly: every lane's writes are in)
        const int32_t cnt = lane < n_sym ? (int32_t)w.norm[lane] : 0;        // lane = symbol (at most 53 of them)
        const bool less = cnt == -1;
        const unsigned long long mless = __builtin_amdgcn_ballot_w64(less);
        const uint32_t n_less = (uint32_t)__popcll(mless), high = size - n_less;
        const uint32_t k_less = __builtin_amdgcn_mbcnt_hi((uint32_t)(mless >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mless, 0u));
        if (less) t[size - 1u - k_less] = lane;
        // exclusive prefix sums of the positive counts -> w.next[symbol] (the run of kept cells of the symbol begins there)
        uint32_t run = cnt > 0 ? (uint32_t)cnt : 0u, incl = run;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t up = __shfl_up(incl, off);
            if (lane >= (uint32_t)off) incl += up;
        }
        const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        if (total + n_less != size) return false;          // (the counts do not fill the table: zs_build_fse ends with pos != 0)
        uint16_t *first = w.first;
        first[lane] = (uint16_t)(incl - run);
        __syncthreads();
        const uint32_t step = (size >> 1) + (size >> 3) + 3u;
        uint32_t kept_before = 0;                          // (wave-uniform) kept cells of the chunks before this one
        for (uint32_t r0 = 0; r0 < size; r0 += 64u) {
            const uint32_t r = r0 + lane, pos = (r * step) & mask;
            const bool keep = r < size && pos < high;
            const unsigned long long mk = __builtin_amdgcn_ballot_w64(keep);
            const uint32_t j = kept_before + __builtin_amdgcn_mbcnt_hi((uint32_t)(mk >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mk, 0u));
            kept_before += (uint32_t)__popcll(mk);
            // the last symbol whose run begins at or below j (runs of length 0 begin where the next one does: take the last)
            uint32_t lo = 0, hi = n_sym;                   // answer in [lo, hi)
#pragma unroll 1
            while (hi - lo > 1u) {
                const uint32_t mid = (lo + hi) >> 1;
                if ((uint32_t)first[mid] <= j) lo = mid; else hi = mid;
            }
            if (keep) t[pos] = lo;
        }
        __syncthreads();
        // states: next[symbol] = its count (1 for the less-than-one symbols), then the cells in ascending order
        w.next[lane] = (uint16_t)(less ? 1u : run);
        __syncthreads();
        for (uint32_t i0 = 0; i0 < size; i0 += 64u) {
            const uint32_t i = i0 + lane;
            const bool on = i < size;
            const uint32_t sym = on ? t[i] : 0xFFFFu;
            uint32_t below = 0, same = 0;
#pragma unroll 8
            for (uint32_t k = 0; k < 64u; k++) {
                const uint32_t sk = (uint32_t)__builtin_amdgcn_readlane((int)sym, (int)k);
                same += sk == sym ? 1u : 0u;
                below += sk == sym && k < lane ? 1u : 0u;
            }
            const uint32_t x = on ? (uint32_t)w.next[sym] + below : 1u;
            __syncthreads();                               // (every lane has read its symbol's counter)
            if (on && below + 1u == same) w.next[sym] = (uint16_t)(x + 1u);   // the chunk's last cell of the symbol leaves the counter behind it
            if (on) {
                const uint32_t nb = log - kta::zs_highbit(x);
                t[i] = sym | (nb << 8) | (((x << nb) - size) << 16);
            }
            __syncthreads();
        }
        return true;
    }
    // the eight bytes [first, first + 8) of the slice [base, base + len), little endian; outside the slice: zeros
    __device__ __forceinline__ uint64_t le64(uint32_t base, uint32_t len, int32_t first)
    {
        const int32_t lo = first < 0 ? 0 : first, hi = first + 8 < (int32_t)len ? first + 8 : (int32_t)len;
        if (lo >= hi) return 0;
        const uint32_t a0 = base + (uint32_t)lo, a1 = base + (uint32_t)hi - 1;
        if (!inside(a0) || !inside(a1)) fetch(a0, (uint32_t)(hi - lo));
        if (hi - lo == 8) {
            // all eight bytes inside the slice (every container of a sequence's fields but the stream's first): three aligned
            // words and two funnel shifts, one LDS round trip — the byte-by-byte form below is eight of them behind
            // compares, ~ 150 instructions per container, and a sequence takes one or two
            const uint32_t o = (uint32_t)((int32_t)a0 - wlo), wd = o >> 2, sh = (o & 3u) * 8u;
            const uint32_t *w32 = reinterpret_cast<const uint32_t *>(win4);
            const uint32_t x0 = w32[wd], x1 = w32[wd + 1], x2 = w32[wd + 2 < kZsWin / 4 ? wd + 2 : kZsWin / 4 - 1];   // (x2 counts only if sh > 0: then wd + 2 is inside)
            return (uint64_t)uni(__builtin_amdgcn_alignbit(x1, x0, sh)) | ((uint64_t)uni(__builtin_amdgcn_alignbit(x2, x1, sh)) << 32);
        }
        const uint8_t *w = reinterpret_cast<const uint8_t *>(win4) + ((int32_t)base - wlo);   // (may lie before the window: indexed with k below)
        uint64_t c = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int32_t k = first + i;
            if (k >= lo && k < hi) c |= (uint64_t)w[k] << (8 * i);
        }
        return (uint64_t)uni((uint32_t)c) | ((uint64_t)uni((uint32_t)(c >> 32)) << 32);
    }
};

template <uint32_t kZsRing>
struct ZsOutWave {                 // output sink: 64 bytes per step, the last kZsRing bytes mirrored in LDS
    uint8_t *dst;
    uint32_t op;                   // (a batch inflates to less than 2 GiB: the index caps it at 512 MiB)
    uint8_t *ring;
    uint32_t lane;
    // (Round 6 measured the kernel without its literal copies (- 0.43 of 2.59 ms) and without its match copies (- 1.03), then
    // tried what those numbers suggested: far matches without any fence (s_waitcnt vmcnt(63): what lies behind the ring is
    // more than 63 stores old) — kept, worth nothing measurable —, every match served from the ring (- 0.08 ms: the far
    // path is not it), the 256 bytes behind a literal run requested ahead and dealt with ds_bpermute (+ 0.16 ms: dropped).
    // What a batch costs is its ~ 50 k instructions at the ~ 11 cycles each that two waves per SIMD leave exposed; the
    // copies are a third of them.  More waves — under 128 registers and 10 KiB of LDS a wave — was the way: kafka_zstd_inflate_coop.)

    __device__ __forceinline__ void put(uint32_t i, uint8_t v)
    {
        dst[op + i] = v;
        ring[(op + i) & (kZsRing - 1)] = v;
    }
    // (The copies count in 32 bits from pointers formed once per call: a batch of JSON-like values runs ~ 1 400 sequences of a
    // few bytes each through these, at 16 waves per CU the kernel is bound by the instructions it issues, and 64-bit index
    // arithmetic per byte-lane was a third of them.)
    template <bool PAST_L1>
    __device__ __forceinline__ void lit_from(const uint8_t *p, uint32_t cnt)
    {
        uint8_t *d = dst + op;
        const uint32_t ro = op;
        if (cnt <= 64) {                                               // the usual run: one step
            if (lane < cnt) {
                const uint8_t v = PAST_L1 ? __hip_atomic_load(p + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : p[lane];
                d[lane] = v;
                ring[(ro + lane) & (kZsRing - 1)] = v;
            }
        } else {
            for (uint32_t i0 = 0; i0 < cnt; i0 += 256) {               // four loads in flight per lane
                uint8_t v[4];
#pragma unroll
                for (uint32_t k = 0; k < 4; k++) {
                    const uint32_t i = i0 + lane + 64 * k;
                    v[k] = i < cnt ? (PAST_L1 ? __hip_atomic_load(p + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : p[i]) : (uint8_t)0;
                }
#pragma unroll
                for (uint32_t k = 0; k < 4; k++) {
                    const uint32_t i = i0 + lane + 64 * k;
                    if (i < cnt) {
                        d[i] = v[k];
                        ring[(ro + i) & (kZsRing - 1)] = v[k];
                    }
                }
            }
        }
        op += cnt;
    }
    template <class S>
    __device__ __forceinline__ void lit_src(S &src, uint32_t at, uint32_t cnt) { lit_from<false>(src.memory() + at, cnt); }
    // (the Huffman streams' literals were written by lanes of this wave: read past L1)
    __device__ __forceinline__ void lit_buf(const uint8_t *lit, uint32_t cnt) { lit_from<true>(lit, cnt); }
    __device__ __forceinline__ void lit_rle(uint8_t v, uint32_t cnt)
    {
        for (uint32_t i = lane; i < cnt; i += 64) put(i, v);
        op += cnt;
    }
    __device__ __forceinline__ void match(uint32_t dist, uint32_t len)
    {
        uint8_t *d = dst + op;
        const uint32_t ro = op;                                        // (the ring is indexed modulo its size)
        if (dist <= kZsRing) {
            const uint32_t ph = dist < 64 ? lane % dist : lane;
            for (uint32_t i0 = 0; i0 < len; i0 += 64) {
                const uint32_t i = i0 + lane;
                // dist >= 64: this step's sources were written before it; dist < 64: the period just before this step
                // (not the one before the whole match: a match longer than the ring has overwritten that one)
                const uint32_t s = ro + i0 - dist + ph;
                if (i < len) {
                    const uint8_t v = ring[s & (kZsRing - 1)];
                    d[i] = v;
                    ring[(ro + i) & (kZsRing - 1)] = v;
                }
            }
        } else {
            // The bytes behind the ring were stored at least kZsRing / 64 store instructions ago (a store moves at most 64
            // bytes), and a wave's memory operations complete in order: once all but its last few are done, those stores
            // have reached L2, where the load — agent scope: past L1 — finds them.  No fence: the workgroup-scope fence
            // waits for EVERY earlier store, the literals and matches just written, a memory round trip per far match
            // (and rounds 2-5's agent-scope fence wrote the L2 back on top of it).
            static_assert(kZsRing >= 1024, "what lies behind the ring is at least 16 stores old");
            const uint8_t *from = d - dist;
            for (uint32_t i0 = 0; i0 < len; i0 += 64) {
                const uint32_t i = i0 + lane;
                if (kZsRing >= 64 * 64) asm volatile("s_waitcnt vmcnt(63)" ::: "memory");
                else if (kZsRing >= 32 * 64) asm volatile("s_waitcnt vmcnt(31)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
                if (i < len) {
                    const uint8_t v = __hip_atomic_load(from + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    d[i] = v;
                    ring[(ro + i) & (kZsRing - 1)] = v;
                }
            }
        }
        op += len;
    }
    // The tables of the work struct (LDS) to and from the batch's spill (global): every lane moves the same words both ways,
    // so what it reads back is what it wrote itself.  (Rare: a frame of several blocks; kta_zstd.h, ZsWorkSmall.)
    __device__ __forceinline__ void spill_words(uint32_t *plain, const uint32_t *work, uint32_t n)
    {
        __syncthreads();
        for (uint32_t i = lane; i < n; i += 64) plain[i] = work[i];
    }
    __device__ __forceinline__ void unspill_words(uint32_t *work, const uint32_t *plain, uint32_t n)
    {
        __syncthreads();
        for (uint32_t i = lane; i < n; i += 64) work[i] = plain[i];
        __syncthreads();
    }
    // the Huffman streams of a literals section: all 64 lanes on one stream at a time, through the source's LDS window (which
    // the sequences' reader does not need yet; kta_zstd_huf_wave.h); the decoded literals are read back by all lanes (lit_buf)
    template <class W, class S>
    __device__ __forceinline__ bool huf_streams(const W &w, S &src, uint32_t streams, const uint32_t at[4],
                                                const uint32_t n[4], const uint32_t count[4], uint8_t *out)
    {
        const bool ok = zh_streams(reinterpret_cast<uint32_t *>(src.win4), src.buffer, src.src0, streams, at, n, count, w.huf_table(),
                                   w.huf_log, out, lane);
        src.wlo = kZsNoWindowRel;                          // (the window holds a stream's bytes now)
        __threadfence_block();                             // (the lanes' literals are read back by the wave's other lanes, past L1)
        return ok;
    }
};

template <uint32_t kZsRing>
__device__ __forceinline__ void zstd_inflate_wave(uint8_t *buffer, kta_kafka_batch_desc *descs, uint64_t n_batches, kta::ZsWorkSmall *s_w,
                                                  uint4 *s_win, uint8_t *s_ring)
{
    const uint32_t lane = threadIdx.x;
    const uint64_t b = blockIdx.x;
    if (b >= n_batches) return;
    const kta_kafka_batch_desc d = descs[b];
    if (!(d.flags & KTA_KB_ZSTD) || d.status) return;
    const uint64_t src0 = d.byte_off + KTA_KAFKA_BATCH_HEADER, n = (uint64_t)d.batch_bytes - KTA_KAFKA_BATCH_HEADER;
    const uint64_t cap = d.payload_end - d.payload_off;
    const uint64_t spill = (d.payload_end + 63) & ~63ull;                        // where the lane kernel has its tables
    const uint64_t scratch = spill + kZstdWorkBytes;                             // the literals of a Huffman-coded section
    static_assert(sizeof(kta::ZsSpill) <= kZstdWorkBytes, "the spill lies in the lane kernel's table space");
    const uint64_t lit_cap = d.scratch_end > scratch ? d.scratch_end - scratch : 0;
    ZsWaveSrc src{buffer, src0, (uint32_t)n, s_win, kZsNoWindowRel, lane};
    ZsOutWave<kZsRing> out{buffer + d.payload_off, 0, s_ring, lane};
    if (n >= (1ull << 31) || cap >= (1ull << 31)) {        // (the decoder's positions are 32 bits wide; the index caps a batch at 512 MiB)
        if (lane == 0) descs[b].status = KTA_KB_BAD_FRAMING;
        return;
    }
    const int64_t got = kta::zstd_inflate_t(src, (uint32_t)n, out, (uint32_t)cap, s_w, buffer + scratch,
                                             (uint32_t)(lit_cap < (1ull << 31) ? lit_cap : (1ull << 31) - 1),
                                             d.scratch_end >= scratch ? reinterpret_cast<kta::ZsSpill *>(buffer + spill) : nullptr);
    if (lane == 0) {
        if (got < 0) descs[b].status = KTA_KB_BAD_FRAMING;
        else descs[b].payload_end = d.payload_off + (uint64_t)got;   // the slice was sized by a bound
    }
}

__global__ __launch_bounds__(64) KTA_WAVES_PER_EU(4, 8) void kafka_zstd_inflate_coop(uint8_t *buffer, kta_kafka_batch_desc *descs, uint64_t n_batches)
{
    // The wave is bound by latencies, so what the kernel delivers goes with the waves a CU holds, and those go with the LDS a
    // wave takes (160 KiB a CU, handed out in 1280-byte pieces) until the registers bind at four waves per SIMD (120 VGPRs):
    //   rounds 2-5  tables 10 KiB + window 2 KiB + an 8 KiB mirror of the output's last bytes   8 waves per CU   2.42 ms
    //   round 6     the mirror 1 KiB: matches that reach behind it read the output from L2,
    //               which costs little (every match served from the mirror: - 3 %)             12 (11)           1.95 ms
    //   round 6     the Huffman table over the sequence tables (kta_zstd.h, ZsWorkSmall: a block
    //               is done with the one before it builds the others; 6 KiB of tables), which
    //               leaves room for a 2 KiB mirror in 8 pieces = 10 240 bytes                     16               1.56 ms
    // for 1 M records in 16 KiB batches, 96 MB compressed (1 KiB mirror with the small tables: 1.60 ms).
    constexpr uint32_t kRing = 2048;
    __shared__ kta::ZsWorkSmall s_w;
    __shared__ uint4 s_win[kZsWin / 16];
    __shared__ uint8_t s_ring[kRing];
    static_assert(sizeof(kta::ZsWorkSmall) + kZsWin + kRing <= 10240, "sixteen waves per CU");
    zstd_inflate_wave<kRing>(buffer, descs, n_batches, &s_w, s_win, s_ring);
}

// ---- Snappy inflate, wave-cooperative: one wave per compressed batch -------------------------------
// Elements are inherently sequential, but each one moves up to 64 bytes: the wave parses the tag
// uniformly (compressed stream staged in an LDS window, so a tag costs LDS latency, not an L2 round
// trip) and executes every literal / copy with all 64 lanes.  The last 2 KiB of output are mirrored in
// an LDS ring: a copy whose offset reaches further back reads the global output instead (after a fence).
// (Rounds 2-5: a 4 KiB window and a 16 KiB ring — 20 KiB of LDS, eight waves per CU, each waiting on its own chain of round
// trips: Snappy 1.59 ms, LZ4 1.27 ms for 16 667 batches of 16 KiB.  Round 6, after the zstd and gzip kernels had shown what the
// occupancy is worth and how little the copies behind the ring cost: 4 KiB / 4 KiB 0.94 / 0.82 ms, 4 / 2 0.83 / 0.74,
// 2 KiB / 2 KiB — 32 waves per CU, all a CU holds — 0.79 / 0.70 ms, 2 / 1 0.82 / 0.72.)
// (The LZ4 kernel then took its parse to the scalar unit and its positions to 32 bits — 0.69 -> 0.52 ms —; the same rewrite of the
// Snappy kernel measured 0.78 -> 0.83 ms with the scalar parse and 0.78 -> 0.77 with 32-bit positions alone: not kept.  The two
// differ in where they were bound: LZ4 issued 14 K scalar and 9 K vector instructions per batch, Snappy 7.5 K and 16 K.)
constexpr uint32_t kSnapWin = 2048;
constexpr uint32_t kSnapRing = 2048;

__global__ __launch_bounds__(64) void kafka_snappy_inflate_coop(uint8_t *buffer, kta_kafka_batch_desc *descs,
                                                                 uint64_t n_batches)
{
    __shared__ uint4 s_in[kSnapWin / 16];
    __shared__ uint8_t s_ring[kSnapRing];
    const uint8_t *win = reinterpret_cast<const uint8_t *>(s_in);
    const uint32_t lane = threadIdx.x;
    const uint64_t b = blockIdx.x;
    if (b >= n_batches) return;
    const kta_kafka_batch_desc d = descs[b];
    if (!(d.flags & KTA_KB_SNAPPY) || d.status) return;
    const uint64_t src0 = d.byte_off + KTA_KAFKA_BATCH_HEADER;     // absolute offsets into `buffer`
    const uint64_t src_end = d.byte_off + d.batch_bytes;
    uint8_t *dst = buffer + d.payload_off;
    const uint64_t cap = d.payload_end - d.payload_off;
    const uint4 *blocks = reinterpret_cast<const uint4 *>(buffer);
    uint64_t wbase = ~0ull;                                          // absolute, 16-byte aligned
    // make [at, at + need) readable through `win`; everything is wave-uniform
    auto fetch = [&](uint64_t at, uint32_t need) {
        if (wbase != ~0ull && at >= wbase && at + need <= wbase + kSnapWin) return;
        __syncthreads();
        wbase = at & ~15ull;
        const uint64_t last = ((src_end + 15) & ~15ull) - 16;          // last readable block of the batch
        uint4 stage[kSnapWin / 1024];                                  // loads in flight together, then LDS
#pragma unroll
        for (uint32_t u = 0; u < kSnapWin / 1024; u++) {
            const uint64_t a = wbase + lane * 16 + u * 1024;
            stage[u] = blocks[(a < last ? a : last) >> 4];
        }
#pragma unroll
        for (uint32_t u = 0; u < kSnapWin / 1024; u++) pin(stage[u]);
#pragma unroll
        for (uint32_t u = 0; u < kSnapWin / 1024; u++) s_in[lane + 64 * u] = stage[u];
        __syncthreads();
    };
    auto in_byte = [&](uint64_t at) -> uint32_t { return win[at - wbase]; };

    uint64_t pos = src0, op = 0;
    bool bad = false;
    fetch(pos, 16);
    bool xerial = src_end - src0 >= 16;
    if (xerial) {
        const uint8_t magic[8] = {0x82, 'S', 'N', 'A', 'P', 'P', 'Y', 0};
        for (int k = 0; k < 8; k++) xerial = xerial && in_byte(pos + k) == magic[k];
    }
    if (xerial) pos += 16;
    while (!bad && pos < src_end) {                                  // one iteration per block
        uint64_t bend = src_end;
        if (xerial) {
            if (pos + 4 > src_end) { bad = true; break; }
            fetch(pos, 4);
            const uint64_t clen = ((uint64_t)in_byte(pos) << 24) | ((uint64_t)in_byte(pos + 1) << 16) |
                                  ((uint64_t)in_byte(pos + 2) << 8) | in_byte(pos + 3);
            pos += 4;
            bend = pos + clen;
            if (clen == 0 || bend > src_end) { bad = true; break; }
        }
        // preamble: uncompressed length of this block (little-endian base-128, at most 5 bytes)
        uint64_t want = 0;
        uint32_t pre = 0;
        bool terminated = false;
        fetch(pos, 8);
        while (pre < 5 && pos + pre < bend && !terminated) {
            const uint32_t byte = in_byte(pos + pre);
            want |= (uint64_t)(byte & 0x7Fu) << (7 * pre);
            terminated = !(byte & 0x80u);
            pre++;
        }
        if (!terminated || op + want > cap) { bad = true; break; }
        const uint64_t block_out_end = op + want;
        uint64_t ip = pos + pre;
        while (ip < bend) {                                          // one iteration per element
            fetch(ip, 8);
            const uint32_t tag = in_byte(ip);
            ip++;
            if ((tag & 3u) == 0u) {                                  // literal
                uint64_t len = tag >> 2;
                if (len >= 60) {
                    const uint32_t nb = (uint32_t)len - 59;
                    if (ip + nb > bend) { bad = true; break; }
                    len = 0;
                    for (uint32_t k = 0; k < nb; k++) len |= (uint64_t)in_byte(ip + k) << (8 * k);
                    ip += nb;
                }
                len += 1;
                if (ip + len > bend || op + len > block_out_end) { bad = true; break; }
                for (uint64_t i = lane; i < len; i += 64) {          // straight from the compressed stream
                    const uint8_t v = buffer[ip + i];
                    dst[op + i] = v;
                    s_ring[(op + i) & (kSnapRing - 1)] = v;
                }
                ip += len;
                op += len;
                continue;
            }
            uint32_t len;
            uint64_t off;
            if ((tag & 3u) == 1u) {
                if (ip + 1 > bend) { bad = true; break; }
                len = 4 + ((tag >> 2) & 7u);
                off = ((uint64_t)(tag >> 5) << 8) | in_byte(ip);
                ip += 1;
            } else if ((tag & 3u) == 2u) {
                if (ip + 2 > bend) { bad = true; break; }
                len = 1 + (tag >> 2);
                off = (uint64_t)in_byte(ip) | ((uint64_t)in_byte(ip + 1) << 8);
                ip += 2;
            } else {
                if (ip + 4 > bend) { bad = true; break; }
                len = 1 + (tag >> 2);
                off = (uint64_t)in_byte(ip) | ((uint64_t)in_byte(ip + 1) << 8) | ((uint64_t)in_byte(ip + 2) << 16) |
                      ((uint64_t)in_byte(ip + 3) << 24);
                ip += 4;
            }
            if (off == 0 || off > op || op + len > block_out_end) { bad = true; break; }
            // len <= 64: one step.  Source index repeats with period `off` when the copy overlaps itself.
            uint8_t v = 0;
            if (off <= kSnapRing) {
                if (lane < len) v = s_ring[(op - off + (off < len ? lane % (uint32_t)off : lane)) & (kSnapRing - 1)];
            } else {
                __threadfence_block();                               // this wave's earlier output stores have reached L2
                if (lane < len)                                      // off > ring >= len: no overlap; read past L1
                    v = __hip_atomic_load(dst + (op - off + lane), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (lane < len) {                                        // (all reads above precede these writes)
                dst[op + lane] = v;
                s_ring[(op + lane) & (kSnapRing - 1)] = v;
            }
            op += len;
        }
        if (bad || op != block_out_end) { bad = true; break; }
        pos = bend;
    }
    if ((bad || op != cap) && lane == 0) descs[b].status = KTA_KB_BAD_FRAMING;
}

// ---- LZ4 inflate, wave-cooperative (same scheme as the Snappy kernel; grammar of kta_lz4.h) ---------
__global__ __launch_bounds__(64) void kafka_lz4_inflate_coop(uint8_t *buffer, kta_kafka_batch_desc *descs,
                                                              uint64_t n_batches)
{
    __shared__ uint4 s_in[kSnapWin / 16];
    __shared__ uint8_t s_ring[kSnapRing];
    const uint8_t *win = reinterpret_cast<const uint8_t *>(s_in);
    const uint32_t lane = threadIdx.x;
    const uint64_t b = blockIdx.x;
    if (b >= n_batches) return;
    const kta_kafka_batch_desc d = descs[b];
    if (!(d.flags & KTA_KB_LZ4) || d.status) return;
    // Positions are relative to the batch's first payload byte and 32 bits wide, and what the wave parses — tokens, lengths,
    // offsets — is moved to scalar registers as it is read (every lane reads the same byte): until round 6 the sequence loop
    // computed in 64-bit VECTOR arithmetic on values all lanes held alike and branched through the exec mask, and at the 32
    // waves per CU the kernel has now it is bound by the instructions it issues.
    const uint8_t *src = buffer + d.byte_off + KTA_KAFKA_BATCH_HEADER;
    const uint32_t src_len = (uint32_t)d.batch_bytes - KTA_KAFKA_BATCH_HEADER;
    uint8_t *dst = buffer + d.payload_off;
    const uint64_t cap64 = d.payload_end - d.payload_off;             // a bound: blocks x block maximum size
    const uint32_t cap = cap64 < 0x7FFFFFFFull ? (uint32_t)cap64 : 0x7FFFFFFFu;
    const int32_t mis = (int32_t)((d.byte_off + KTA_KAFKA_BATCH_HEADER) & 15ull);    // src - mis is a 16-byte boundary
    const int32_t last = (int32_t)((src_len + (uint32_t)mis + 15u) & ~15u) - 16 - mis;   // last readable unit of the batch
    int32_t wbase = INT32_MIN / 2;                                    // the window's first byte (relative; >= -15)
    auto fetch = [&](uint32_t at, uint32_t need) {
        if ((uint32_t)((int32_t)at - wbase) < kSnapWin && (uint32_t)((int32_t)(at + need) - wbase) <= kSnapWin) return;
        __syncthreads();
        wbase = (int32_t)((at + (uint32_t)mis) & ~15u) - mis;
        uint4 stage[kSnapWin / 1024];                                  // loads in flight together, then LDS
#pragma unroll
        for (uint32_t u = 0; u < kSnapWin / 1024; u++) {
            const int32_t a = wbase + (int32_t)(lane * 16 + u * 1024);
            stage[u] = *reinterpret_cast<const uint4 *>(src + (a < last ? a : last));
        }
#pragma unroll
        for (uint32_t u = 0; u < kSnapWin / 1024; u++) pin(stage[u]);
#pragma unroll
        for (uint32_t u = 0; u < kSnapWin / 1024; u++) s_in[lane + 64 * u] = stage[u];
        __syncthreads();
    };
    auto in_byte = [&](uint32_t at) -> uint32_t { return (uint32_t)__builtin_amdgcn_readfirstlane((int)win[(int32_t)at - wbase]); };
    // `n` bytes from p (memory) to the output at op
    auto copy_in = [&](const uint8_t *p, uint32_t op, uint32_t n) {
        uint8_t *o = dst + op;
        for (uint32_t i = lane; i < n; i += 64) {
            const uint8_t v = p[i];
            o[i] = v;
            s_ring[(op + i) & (kSnapRing - 1)] = v;
        }
    };
    // copy `len` bytes of earlier output, `off` back, to the current position (64 bytes per step)
    auto copy_match = [&](uint32_t op, uint32_t off, uint32_t len) {
        uint8_t *o = dst + op;
        if (off <= kSnapRing) {
            // off >= 64: a step's sources were written before it; off < 64: the period just before the step
            // (not the one before the whole match: a match longer than the ring has overwritten that one)
            const uint32_t ph = off < 64 ? lane % off : lane;
            for (uint32_t i0 = 0; i0 < len; i0 += 64) {
                const uint32_t i = i0 + lane;
                if (i < len) {
                    const uint8_t v = s_ring[(op + i0 - off + ph) & (kSnapRing - 1)];
                    o[i] = v;
                    s_ring[(op + i) & (kSnapRing - 1)] = v;
                }
            }
        } else {
            const uint8_t *from = o - off;
            for (uint32_t i0 = 0; i0 < len; i0 += 64) {
                const uint32_t i = i0 + lane;
                __threadfence_block();
                if (i < len) {
                    const uint8_t v = __hip_atomic_load(from + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    o[i] = v;
                    s_ring[(op + i) & (kSnapRing - 1)] = v;
                }
            }
        }
    };

    uint32_t op = 0;
    bool bad = src_len < 7 || src_len >= 0x7FFFFFF0u;
    uint32_t pos = 0, block_max = 0;
    bool block_checksum = false;
    if (!bad) {
        fetch(pos, 16);
        bad = !(in_byte(pos) == 0x04 && in_byte(pos + 1) == 0x22 && in_byte(pos + 2) == 0x4D && in_byte(pos + 3) == 0x18);
        const uint32_t flg = in_byte(pos + 4), bd = in_byte(pos + 5), bs = (bd >> 4) & 7u;
        bad = bad || (flg >> 6) != 1u || bs < 4;
        block_max = 1u << (8 + 2 * bs);
        block_checksum = (flg & 0x10u) != 0;
        pos += 6 + ((flg & 0x08u) ? 8 : 0) + ((flg & 0x01u) ? 4 : 0) + 1;
    }
    while (!bad) {                                                    // one iteration per block
        if (pos + 4 > src_len) { bad = true; break; }
        fetch(pos, 4);
        const uint32_t w = in_byte(pos) | (in_byte(pos + 1) << 8) | (in_byte(pos + 2) << 16) | (in_byte(pos + 3) << 24);
        pos += 4;
        if (w == 0) break;                                            // end mark
        const uint32_t sz = w & 0x7FFFFFFFu;
        if (sz > block_max || sz > src_len - pos) { bad = true; break; }
        const uint32_t bend = pos + sz;
        if (w & 0x80000000u) {                                        // stored block
            if (sz > cap - op) { bad = true; break; }
            copy_in(src + pos, op, sz);
            op += sz;
        } else {
            uint32_t ip = pos;
            while (ip < bend) {                                       // one iteration per sequence
                fetch(ip, 8);
                const uint32_t token = in_byte(ip);
                ip++;
                uint32_t lit = token >> 4;
                if (lit == 15) {
                    uint32_t byte;
                    do {
                        if (ip >= bend) { bad = true; break; }
                        fetch(ip, 1);
                        byte = in_byte(ip);
                        ip++;
                        lit += byte;
                    } while (byte == 255);
                    if (bad) break;
                }
                if (lit > bend - ip || lit > cap - op) { bad = true; break; }
                copy_in(src + ip, op, lit);                           // literals straight from the compressed stream
                ip += lit;
                op += lit;
                if (ip == bend) break;                                // the last sequence has no match
                if (ip + 2 > bend) { bad = true; break; }
                fetch(ip, 8);
                const uint32_t off = in_byte(ip) | (in_byte(ip + 1) << 8);
                ip += 2;
                uint32_t ml = token & 15u;
                if (ml == 15) {
                    uint32_t byte;
                    do {
                        if (ip >= bend) { bad = true; break; }
                        fetch(ip, 1);
                        byte = in_byte(ip);
                        ip++;
                        ml += byte;
                    } while (byte == 255);
                    if (bad) break;
                }
                ml += 4;
                if (off == 0 || off > op || ml > cap - op) { bad = true; break; }
                copy_match(op, off, ml);
                op += ml;
            }
            if (bad) break;
        }
        pos = bend + (block_checksum ? 4 : 0);
    }
    if (lane == 0) {
        if (bad) descs[b].status = KTA_KB_BAD_FRAMING;
        else descs[b].payload_end = d.payload_off + op;               // the slice was sized by a bound
    }
}

// ---- CRC-32C (Castagnoli, reflected polynomial 0x82F63B78) on the device -------------------------
// One wave per batch.  The bytes [start, end) are cut on the absolute 64-byte grid; per 4 KiB window
// every lane reduces its 64-byte chunk with slicing-by-4 tables held in LDS (raw register update, no
// init / final xor), then the chunks are stitched with the linearity of CRCs:
//     state' = Z_len(state) ^ XOR_i Z_after(i)(c_i)
// where Z_n (advance the register over n zero bytes) is a GF(2)[x] multiplication mod P by the
// precomputed constant x^(8n) — a table of 4097 constants built once on the host.
constexpr uint32_t kCrcPoly = 0x82F63B78u;
constexpr uint32_t kCrcWindow = 4096;

struct CrcTables {
    uint32_t slice[4][256];          // slicing-by-4
    uint32_t zshift[kCrcWindow + 1]; // zshift[n]: multiplier that advances the register over n zero bytes
};

// (a * b) mod P in the reflected representation (bit 31 = x^0)
__host__ __device__ inline uint32_t gf_mul(uint32_t a, uint32_t b)
{
    uint32_t r = 0;
    for (int i = 0; i < 32; i++) {
        if (a & 0x80000000u) r ^= b;
        a <<= 1;
        b = (b >> 1) ^ ((b & 1u) ? kCrcPoly : 0u);
    }
    return r;
}

void build_crc_tables(CrcTables &t)
{
    for (uint32_t i = 0; i < 256; i++) {
        uint32_t c = i;
        for (int k = 0; k < 8; k++) c = (c & 1u) ? (c >> 1) ^ kCrcPoly : c >> 1;
        t.slice[0][i] = c;
    }
    for (uint32_t i = 0; i < 256; i++)
        for (int s = 1; s < 4; s++) t.slice[s][i] = (t.slice[s - 1][i] >> 8) ^ t.slice[0][t.slice[s - 1][i] & 0xFFu];
    // zshift[0] = 1 (x^0 = bit 31); advancing a register value r over one zero byte is r -> (r >> 8) ^ T0[r & 0xff],
    // a linear map, i.e. multiplication by x^8: zshift[n+1] = zshift[n] * x^8
    uint32_t x8 = 0x80000000u;
    for (int k = 0; k < 8; k++) x8 = (x8 >> 1) ^ ((x8 & 1u) ? kCrcPoly : 0u);
    t.zshift[0] = 0x80000000u;
    for (uint32_t n = 0; n < kCrcWindow; n++) t.zshift[n + 1] = gf_mul(t.zshift[n], x8);
}

__global__ __launch_bounds__(64) void kafka_crc32c(const uint4 *blocks, kta_kafka_batch_desc *descs, uint64_t n_batches,
                                                   const CrcTables *tables, unsigned long long *n_crc_bad)
{
    __shared__ uint32_t s_t[4][256];
    const uint32_t lane = threadIdx.x;
    for (uint32_t i = lane; i < 1024; i += 64) (&s_t[0][0])[i] = (&tables->slice[0][0])[i];
    __syncthreads();
    const uint64_t b = blockIdx.x;
    if (b >= n_batches) return;
    const kta_kafka_batch_desc d = descs[b];
    const uint64_t start = d.byte_off + 21, end = d.byte_off + d.batch_bytes;   // attributes .. end of batch
    uint32_t state = 0xFFFFFFFFu;
    for (uint64_t wbase = start & ~63ull; wbase < end; wbase += kCrcWindow) {   // wave-uniform
        const uint64_t cb = wbase + 64ull * lane;                               // this lane's 64-byte chunk
        const uint64_t lo = cb > start ? cb : start;
        const uint64_t hi = cb + 64 < end ? cb + 64 : end;
        uint32_t c = 0;
        if (lo < hi) {
            const uint4 *p = blocks + (cb >> 4);
            uint32_t w[16];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const uint4 v = p[q];
                w[4 * q] = v.x; w[4 * q + 1] = v.y; w[4 * q + 2] = v.z; w[4 * q + 3] = v.w;
            }
            if (lo == cb && hi == cb + 64) {                                    // full chunk: slicing-by-4
#pragma unroll
                for (int k = 0; k < 16; k++) {
                    c ^= w[k];
                    c = s_t[3][c & 0xFFu] ^ s_t[2][(c >> 8) & 0xFFu] ^ s_t[1][(c >> 16) & 0xFFu] ^ s_t[0][c >> 24];
                }
            } else {                                                            // first / last chunk: bytewise
                for (uint32_t o = (uint32_t)(lo - cb); o < (uint32_t)(hi - cb); o++) {
                    const uint32_t byte = (w[o >> 2] >> ((o & 3u) * 8u)) & 0xFFu;
                    c = (c >> 8) ^ s_t[0][(c ^ byte) & 0xFFu];
                }
            }
        }
        const uint64_t wlo = wbase > start ? wbase : start;
        const uint64_t whi = wbase + kCrcWindow < end ? wbase + kCrcWindow : end;
        const uint32_t after = lo < hi ? (uint32_t)(whi - hi) : 0u;             // bytes of the window behind my chunk
        uint32_t part = (lo < hi && c) ? gf_mul(c, tables->zshift[after]) : 0u;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) part ^= __shfl_xor(part, off);
        state = gf_mul(state, tables->zshift[(uint32_t)(whi - wlo)]) ^ part;
    }
    if (lane == 0 && (state ^ 0xFFFFFFFFu) != d.crc) {
        descs[b].status = KTA_KB_BAD_CRC;
        atomicAdd(n_crc_bad, 1ull);
    }
}

// One stage of the raw-log pipeline: a pinned host blob the fetcher fills, its device copy, the
// pinned batch index and the decoded columns.
struct BlobStage {
    uint8_t *h_blob = nullptr, *d_blob = nullptr;
    uint64_t cap = 0;      // host blob capacity
    uint64_t d_cap = 0;    // device buffer capacity: the blob plus the inflate area of compressed batches
    kta_kafka_batch_desc *h_descs = nullptr; // pinned: the H2D copy of the index is truly asynchronous
    uint64_t desc_cap = 0;
    kta_batch out{};
    uint64_t out_cap = 0;
    hipEvent_t copied = nullptr, done = nullptr;
    bool busy = false;
};

struct KafkaState {
    // The descriptors of a decode call travel on the COPY stream into one of two device buffers that take turns, so that
    // the upload of call k + 1 (5.9 MB for 66 667 batches; from pageable memory a staged, host-blocking copy) runs beside the
    // kernels of call k instead of in front of its own: round 5's bench line had a step of 0.396 ms around a 0.210 ms kernel,
    // now 0.23-0.25 around 0.21-0.22.  (Descriptors in pinned memory — the upload a DMA the host does not wait for — were tried
    // as an API and dropped: a host that enqueues that far ahead of the GPU met 9 ms stalls of the runtime every few calls.)
    kta_kafka_batch_desc *d_descs2[2] = {nullptr, nullptr};
    uint64_t desc_cap2[2] = {0, 0};
    hipEvent_t ev_desc_up[2] = {nullptr, nullptr}, ev_desc_free[2] = {nullptr, nullptr};   // uploaded (copy stream) / no longer read (compute stream)
    bool desc_used[2] = {false, false};
    int desc_slot = 0;
    uint64_t *d_scalars = nullptr; // [0] key-byte total, [1] bad batches
    CrcTables *d_crc_tables = nullptr;
    uint64_t *d_crc_bad = nullptr;  // CRC failures since the context was created
    bool check_crcs = false;
    std::vector<BlobStage> stages;
    uint64_t blob_capacity = 256ull << 20;
    uint64_t inflate_limit = 0;     // kta_kafka_set_inflate_limit: 0 = default (1 GiB per group of batches)
    int variant = 0;                // kta_kafka_set_variant: 0 = automatic, 1 = lane per batch, 2 / 10 / 11 = wave geometries
    int cur = 0;
    bool acquired = false;
    std::vector<hipEvent_t> ev[2];
    size_t ev_used[2] = {0, 0};
    double ms_sum[2] = {0, 0};
    uint64_t ms_cnt[2] = {0, 0};
};

void free_out(kta_batch &o)
{
    if (o.partition) (void)hipFree(o.partition);
    if (o.key_len) (void)hipFree(o.key_len);
    if (o.val_len) (void)hipFree(o.val_len);
    if (o.ts_ms) (void)hipFree(o.ts_ms);
    if (o.key_off) (void)hipFree(o.key_off);
    memset(&o, 0, sizeof(o));
}

void free_state(void *p)
{
    KafkaState *st = static_cast<KafkaState *>(p);
    if (!st) return;
    for (int k = 0; k < 2; k++) {
        if (st->d_descs2[k]) (void)hipFree(st->d_descs2[k]);
        if (st->ev_desc_up[k]) (void)hipEventDestroy(st->ev_desc_up[k]);
        if (st->ev_desc_free[k]) (void)hipEventDestroy(st->ev_desc_free[k]);
    }
    if (st->d_scalars) (void)hipFree(st->d_scalars);
    if (st->d_crc_tables) (void)hipFree(st->d_crc_tables);
    if (st->d_crc_bad) (void)hipFree(st->d_crc_bad);
    for (auto &g : st->stages) {
        if (g.h_blob) (void)hipHostFree(g.h_blob);
        if (g.d_blob) (void)hipFree(g.d_blob);
        if (g.h_descs) (void)hipHostFree(g.h_descs);
        free_out(g.out);
        if (g.copied) (void)hipEventDestroy(g.copied);
        if (g.done) (void)hipEventDestroy(g.done);
    }
    for (auto &v : st->ev)
        for (auto e : v) (void)hipEventDestroy(e);
    delete st;
}

KafkaState *state_of(kta_ctx *ctx)
{
    void **slot = kta_internal_ext_slot(ctx, free_state);
    if (!*slot) *slot = new KafkaState();
    return static_cast<KafkaState *>(*slot);
}

int hip_err(kta_ctx *ctx, hipError_t e, const char *what)
{
    std::string m = std::string(what) + ": " + hipGetErrorString(e);
    kta_internal_set_error(ctx, m.c_str());
    return e == hipErrorOutOfMemory ? KTA_ERR_NOMEM : KTA_ERR_HIP;
}

#define KK(ctx, call)                                            \
    do {                                                         \
        hipError_t e__ = (call);                                 \
        if (e__ != hipSuccess) return hip_err(ctx, e__, #call);  \
    } while (0)

inline uint32_t be32(const uint8_t *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
inline uint64_t be64(const uint8_t *p) { return ((uint64_t)be32(p) << 32) | be32(p + 4); }
inline uint16_t be16(const uint8_t *p) { return (uint16_t)(((uint16_t)p[0] << 8) | p[1]); }

int drain(kta_ctx *ctx, KafkaState *st)
{
    KK(ctx, hipStreamSynchronize(kta_internal_stream(ctx)));
    for (int k = 0; k < 2; k++) {
        for (size_t i = 0; i + 1 < st->ev_used[k]; i += 2) {
            float ms = 0.f;
            KK(ctx, hipEventElapsedTime(&ms, st->ev[k][i], st->ev[k][i + 1]));
            st->ms_sum[k] += ms;
            st->ms_cnt[k]++;
        }
        st->ev_used[k] = 0;
    }
    return KTA_OK;
}

int pair(kta_ctx *ctx, KafkaState *st, int k, hipEvent_t *a, hipEvent_t *b)
{
    if (st->ev_used[k] + 2 > 1024) {
        int rc = drain(ctx, st);
        if (rc != KTA_OK) return rc;
    }
    while (st->ev[k].size() < st->ev_used[k] + 2) {
        hipEvent_t e;
        KK(ctx, hipEventCreate(&e));
        st->ev[k].push_back(e);
    }
    *a = st->ev[k][st->ev_used[k]];
    *b = st->ev[k][st->ev_used[k] + 1];
    st->ev_used[k] += 2;
    return KTA_OK;
}

} // namespace

extern "C" {

int kta_kafka_index_host(const uint8_t *bytes, uint64_t len, int32_t partition, uint64_t blob_offset,
                         uint64_t record_base_start, uint64_t inflate_offset, kta_kafka_batch_desc *descs,
                         uint64_t cap, kta_kafka_index_stats *stats)
{
    if (!bytes || !stats || (cap && !descs)) return KTA_ERR_INVALID;
    memset(stats, 0, sizeof(*stats));
    uint64_t pos = 0, rec = record_base_start, nb = 0, inflate = 0;
    while (pos + 12 <= len) {
        const int32_t batch_length = (int32_t)be32(bytes + pos + 8);
        if (batch_length < KTA_KAFKA_BATCH_HEADER - 12) break;           // not a v2 batch header: stop
        const uint64_t total = 12ull + (uint64_t)batch_length;
        if (pos + total > len) break;                                     // partial batch at the end of a fetch
        // (the walk is a chain of cache misses, one header per batch, 5-6 KB apart: the next header is asked for while this
        // batch is looked at, and the one after it where batches of this size would put it, and the trailer gzip / zstd read)
        __builtin_prefetch(bytes + pos + total);
        __builtin_prefetch(bytes + pos + total + 60);
        if (pos + 2 * total + 64 <= len) __builtin_prefetch(bytes + pos + 2 * total);
        __builtin_prefetch(bytes + pos + total - 8);
        const uint8_t magic = bytes[pos + 16];
        if (magic != 2) {
            stats->n_old_magic++;
        } else {
            // the log's extent as a consumer's watermarks see it: every v2 batch counts, control batches
            // included, and a batch ends at baseOffset + lastOffsetDelta + 1 (compaction may have removed records)
            const int64_t b_first = (int64_t)be64(bytes + pos);
            const int64_t b_next = b_first + (int64_t)(int32_t)be32(bytes + pos + 23) + 1;
            if (!stats->any_offsets || b_first < stats->first_offset) stats->first_offset = b_first;
            if (!stats->any_offsets || b_next > stats->next_offset) stats->next_offset = b_next;
            stats->any_offsets = 1;
            const uint16_t attrs = be16(bytes + pos + 21);
            int32_t count = (int32_t)be32(bytes + pos + 57);
            const uint32_t codec = attrs & 0x07u;
            int64_t inflated = 0;
            if (codec == 2) {
                const uint64_t clen = total - KTA_KAFKA_BATCH_HEADER;
                inflated = kta::snappy_uncompressed_len(bytes + pos + KTA_KAFKA_BATCH_HEADER, clen);
                // a copy element expands at most 3 bytes into 64: anything beyond ~22x is a corrupt preamble
                if (inflated > (int64_t)(clen * 22 + 64)) inflated = -1;
            }
            if (codec == 3) inflated = kta::lz4_inflate_bound(bytes + pos + KTA_KAFKA_BATCH_HEADER, total - KTA_KAFKA_BATCH_HEADER);
            if (codec == 1) {
                const uint64_t clen = total - KTA_KAFKA_BATCH_HEADER;
                inflated = kta::gzip_uncompressed_len(bytes + pos + KTA_KAFKA_BATCH_HEADER, clen);
                if (inflated > (int64_t)(clen * 1032 + 64)) inflated = -1;   // DEFLATE expands at most 1032x
            }
            uint64_t scratch = 0;
            if (codec == 1 && inflated > 0 && inflated <= (int64_t)kMaxBatchInflate)
                scratch = 8 + 4 * (kta::gz_token_bound((uint64_t)inflated) + kta::gz_closing_tokens(total - KTA_KAFKA_BATCH_HEADER));   // the tokens of the two-stage inflate
            if (codec == 4) {
                uint64_t bound = 0, lit = 0;
                if (kta::zstd_scan(bytes + pos + KTA_KAFKA_BATCH_HEADER, total - KTA_KAFKA_BATCH_HEADER, &bound, &lit)) {
                    inflated = (int64_t)bound;
                    scratch = kZstdWorkBytes + lit + 64;                  // tables + literals (+ alignment slack)
                } else {
                    inflated = -1;
                }
            }
            // A batch that claims to inflate to more than this is not decoded (and is reported): brokers cap a
            // batch at message.max.bytes (1 MiB by default, rarely above 100 MiB) before compression ratios,
            // and one absurd bound (an LZ4 frame of RLE blocks, a zstd window) must not make the whole
            // call run out of inflate area.
            if (inflated > (int64_t)kMaxBatchInflate) inflated = -1;
            if (attrs & 0x20) stats->n_control_batches++;                 // control batch: never delivered
            else if (codec > 4) stats->n_compressed++;                    // unknown codecs (5..7): not decoded here
            else if (count > 0) {
                // A record occupies at least 7 bytes (length, attributes, timestamp delta, offset delta, key
                // length, value length, header count).  A header that announces more records than its payload
                // can hold is corrupt: the batch is reported, and the count is clamped so that a forged
                // header cannot ask for billions of output slots.  The payload of an uncompressed batch is
                // its own bytes; of a compressed one, at most the codec's maximum expansion of its bytes
                // (a rule that depends on the header alone, never on the size probe: a batch whose stream
                // is corrupt keeps its announced records, all of them flagged).  zstd has no useful
                // expansion limit (an RLE block is 4 bytes for 128 KiB), so there the frame's own bound
                // is used when the scan found one.
                const uint64_t clen = total - KTA_KAFKA_BATCH_HEADER;
                const uint64_t payload_bytes = codec == 0 ? clen
                                             : codec == 1 ? clen * 1032 + 64
                                             : codec == 2 ? clen * 22 + 64
                                             : codec == 3 ? clen * 255 + 64
                                             : (inflated > 0 ? (uint64_t)inflated : clen * 1032 + 64);
                const bool forged = (uint64_t)count > payload_bytes / 7 + 1;
                if (forged) count = (int32_t)(payload_bytes / 7 + 1);
                if (nb < cap) {
                    kta_kafka_batch_desc &d = descs[nb];
                    d.byte_off = blob_offset + pos;
                    d.record_base = rec;
                    d.crc = be32(bytes + pos + 17);
                    d.status = forged ? KTA_KB_BAD_FRAMING : 0u;
                    d.base_offset = (int64_t)be64(bytes + pos);
                    d.base_ts_ms = (int64_t)be64(bytes + pos + 27);
                    d.max_ts_ms = (int64_t)be64(bytes + pos + 35);
                    d.batch_bytes = (uint32_t)total;
                    d.partition = partition;
                    d.n_records = count;
                    d.flags = ((attrs & 0x08) ? KTA_KB_LOG_APPEND_TIME : 0u) | ((attrs & 0x10) ? KTA_KB_TRANSACTIONAL : 0u);
                    if (codec != 0) {
                        d.flags |= codec == 2 ? KTA_KB_SNAPPY : (codec == 3 ? KTA_KB_LZ4 : (codec == 1 ? KTA_KB_GZIP : KTA_KB_ZSTD));
                        if (inflated < 0) {          // malformed stream: nothing to parse, reported as bad
                            d.status = KTA_KB_BAD_FRAMING;
                            inflated = 0;
                        }
                        d.payload_off = inflate_offset + inflate;
                        d.payload_end = d.payload_off + (uint64_t)inflated;
                        d.scratch_end = inflated > 0 && scratch ? ((d.payload_end + 63) & ~63ull) + scratch : d.payload_end;
                    } else {
                        d.payload_off = d.byte_off + KTA_KAFKA_BATCH_HEADER;
                        d.payload_end = d.byte_off + total;
                        d.scratch_end = d.payload_end;
                    }
                }
                if (codec != 0) {
                    if (codec == 2) stats->n_snappy++;
                    else if (codec == 3) stats->n_lz4++;
                    else if (codec == 1) stats->n_gzip++;
                    else stats->n_zstd++;
                    inflate += ((uint64_t)(inflated > 0 ? inflated : 0) + 63) & ~63ull;   // 64-byte aligned slices
                    if (inflated > 0 && scratch) inflate += (scratch + 63) & ~63ull;

                }
                nb++;
                rec += (uint64_t)count;
            }
        }
        pos += total;
    }
    stats->n_batches = nb;
    stats->n_records = rec - record_base_start;
    stats->bytes_consumed = pos;
    stats->trailing_bytes = len - pos;
    stats->inflate_bytes = inflate;
    return nb > cap ? KTA_ERR_CAPACITY : KTA_OK;
}

int64_t kta_lz4_inflate_host(const uint8_t *src, uint64_t n, uint8_t *dst, uint64_t cap)
{
    if (!src || (!dst && cap)) return -1;
    return kta::lz4_inflate(src, n, dst, cap);
}

int64_t kta_zstd_inflate_host(const uint8_t *src, uint64_t n, uint8_t *dst, uint64_t cap)
{
    if (!src || (!dst && cap)) return -1;
    uint64_t bound = 0, lit = 0;
    if (!kta::zstd_scan(src, n, &bound, &lit)) return -1;
    std::vector<uint8_t> scratch(sizeof(kta::ZsWork) + lit + 64);
    return kta::zstd_inflate(src, n, dst, cap, reinterpret_cast<kta::ZsWork *>(scratch.data()), scratch.data() + sizeof(kta::ZsWork), lit);
}

int64_t kta_zstd_inflate_host_small(const uint8_t *src, uint64_t n, uint8_t *dst, uint64_t cap)
{
    if (!src || (!dst && cap)) return -1;
    uint64_t bound = 0, lit = 0;
    if (!kta::zstd_scan(src, n, &bound, &lit)) return -1;
    std::vector<uint8_t> scratch(lit + 64);
    kta::ZsWorkSmall w;
    kta::ZsSpill spill;
    memset(&spill, 0xA5, sizeof(spill));                 // (nothing may be read from it that was not put there)
    return kta::zstd_inflate_small(src, n, dst, cap, &w, &spill, scratch.data(), lit);
}

int64_t kta_gzip_inflate_host(const uint8_t *src, uint64_t n, uint8_t *dst, uint64_t cap)
{
    if (!src || (!dst && cap)) return -1;
    uint16_t work[kta::GZ2_WORK];
    std::vector<uint32_t> tok(kta::gz_token_bound(cap));
    uint64_t n_tok = 0;
    uint32_t window[kta::GZ_WIN / 4];                 // the device's stream window too (one lane: word stride 1)
    kta::GzBitsWin bits;
    bits.win = window;
    bits.wstride = 1;
    const int64_t got = kta::gzip_tokenize(bits, src, n, dst, cap, tok.data(), tok.size(), &n_tok, work, 1);
    if (got < 0 || !kta::gz_apply_tokens(dst, (uint64_t)got, tok.data(), n_tok)) return -1;
    return got;
}

int64_t kta_gzip_inflate_lane_host(const uint8_t *src, uint64_t n, uint8_t *dst, uint64_t cap)
{
    if (!src || (!dst && cap)) return -1;
    uint16_t work[kta::GZ_WORK];
    return kta::gzip_inflate(src, n, dst, cap, work, 1);
}

int64_t kta_snappy_inflate_host(const uint8_t *src, uint64_t n, uint8_t *dst, uint64_t cap)
{
    if (!src || (!dst && cap)) return -1;
    return kta::snappy_inflate(src, n, dst, cap);
}

// The rounds of kafka_decode_coop on the host: one group of `lanes` lanes per batch, its lanes one after the other
// (inside a phase the kernel's lanes only meet in s_bad and s_first_incomplete), over kta::rec::chain and
// kta::rec::parse_record as the device compiles them.  The window holds the batch's valid bytes and poison behind
// them — what the device finds there is whatever the last window left.
int kta_kafka_decode_rounds_host(const uint8_t *blob, uint64_t blob_len, const kta_kafka_batch_desc *descs,
                                 uint64_t n_batches, uint32_t lanes, uint32_t window, uint32_t per_round,
                                 int32_t *partition, int32_t *key_len, int32_t *val_len, int64_t *ts_ms,
                                 uint32_t *key_off, uint64_t *n_key_bytes, uint64_t *n_bad_batches)
{
    namespace rec = kta::rec;
    if (!blob || (!descs && n_batches) || !partition || !key_len || !val_len || !ts_ms) return KTA_ERR_INVALID;
    if (lanes == 0 || 64 % lanes != 0 || window == 0 || window % (lanes * 16) != 0 || per_round == 0 || per_round % lanes != 0)
        return KTA_ERR_INVALID;
    const uint32_t W = window, R = per_round;
    std::vector<uint32_t> win_words(W / 4 + 4), starts(R + 1);
    uint8_t *win = reinterpret_cast<uint8_t *>(win_words.data());
    uint64_t key_bytes = 0, n_bad = 0;
    for (uint64_t b = 0; b < n_batches; b++) {
        const kta_kafka_batch_desc &d = descs[b];
        const uint64_t end = d.payload_end;
        if (end > blob_len || d.payload_off > end) return KTA_ERR_INVALID;
        const bool append_time = (d.flags & KTA_KB_LOG_APPEND_TIME) != 0;
        const int64_t ts_base = append_time ? d.max_ts_ms : d.base_ts_ms, ts_mask = append_time ? 0 : -1;
        const uint32_t total = (uint32_t)d.n_records;
        uint64_t pos = d.payload_off, kb = 0;
        uint32_t j = 0;
        bool bad = d.status != 0, run = !bad && j < total;
        while (run) {
            if (pos >= end) { bad = true; break; }
            const uint64_t wbase = rec::window_base(pos, W);
            const uint64_t span = ((end + 15) & ~15ull) - wbase, rest = end - wbase;
            const uint32_t wbytes = span < W ? (uint32_t)span : W;
            const bool to_the_end = rest <= wbytes;
            const uint32_t limit = to_the_end ? (uint32_t)rest : wbytes;
            const uint32_t end_rel = rest < 0xF0000000ull ? (uint32_t)rest : 0xF0000000u;
            memset(win, 0xA5 ^ (int)(pos & 0x5A), W + 16);
            memcpy(win, blob + wbase, limit);
            bool s_bad = false;
            uint32_t first_incomplete = R;
            // the leader
            const uint32_t want = total - j < R ? total - j : R;
            uint32_t k = 0, cur = (uint32_t)(pos - wbase);
            uint64_t next = 0;
            if (rec::chain(win_words.data(), limit, want, starts.data(), k, cur)) {
                uint32_t off = cur;
                long long len;
                if (rec::window_varlong(win, off, limit, len)) {
                    const uint64_t rec_end = wbase + off + (uint64_t)len;
                    if (len < 0 || rec_end > end) s_bad = true;
                    else { starts[k++] = cur; next = rec_end; }
                } else if (to_the_end) {
                    s_bad = true;
                }
            }
            if (!next) next = wbase + cur;
            const uint64_t next_rel = next - wbase;
            starts[k] = next_rel < 0xFFFFFFFFull ? (uint32_t)next_rel : 0xFFFFFFFFu;
            const uint32_t found = k;
            // the lanes
            std::vector<uint32_t> my_kl(found, 0);
            for (k = 0; k < found; k++) {
                const uint32_t start = starts[k], rec_end = starts[k + 1];
                if (rec_end > end_rel) { s_bad = true; continue; }
                rec::Record r;
                uint32_t verdict = rec::parse_record(win, start, rec_end, limit, r);
                if (verdict == rec::REC_VALUE_LENGTH_OUTSIDE) {
                    uint64_t at = wbase + r.after;
                    unsigned long long v = 0;
                    for (uint32_t shift = 0; shift < 70; shift += 7) {
                        const uint32_t byte = at < blob_len ? blob[at] : 0u;      // (the device's blob is padded)
                        at++;
                        v |= (unsigned long long)(byte & 0x7Fu) << (shift < 64 ? shift : 63);
                        if (!(byte & 0x80u)) break;
                    }
                    r.val_len = (long long)(v >> 1) ^ -(long long)(v & 1ull);
                    verdict = rec::value_fits(r.val_len, at - wbase, rec_end);
                }
                if (verdict == rec::REC_INCOMPLETE) { if (k < first_incomplete) first_incomplete = k; continue; }
                if (verdict != rec::REC_OK) { s_bad = true; continue; }
                const uint64_t i = d.record_base + j + k;
                partition[i] = d.partition;
                key_len[i] = (int32_t)r.key_len;
                val_len[i] = (int32_t)r.val_len;
                ts_ms[i] = ts_base + (r.ts_delta & ts_mask);
                if (key_off) key_off[i] = r.key_len > 0 ? (uint32_t)wbase + r.key : 0u;
                my_kl[k] = r.key_len > 0 ? (uint32_t)r.key_len : 0u;
            }
            const uint32_t done = first_incomplete < found ? first_incomplete : found;
            if (s_bad || done == 0) { bad = true; break; }
            for (k = 0; k < done; k++) kb += my_kl[k];
            j += done;
            pos = done < found ? wbase + starts[done] : next;
            run = j < total;
        }
        if (bad) {
            for (uint32_t r = j; r < total; r++) {
                const uint64_t i = d.record_base + r;
                partition[i] = -1; key_len[i] = -1; val_len[i] = -1; ts_ms[i] = -1;
                if (key_off) key_off[i] = 0u;
            }
            n_bad++;
        }
        key_bytes += kb;
    }
    if (n_key_bytes) *n_key_bytes = key_bytes;
    if (n_bad_batches) *n_bad_batches = n_bad;
    return KTA_OK;
}

int kta_kafka_decode_device(kta_ctx *ctx, const uint8_t *blob_device, uint64_t blob_len,
                            const kta_kafka_batch_desc *descs_host, uint64_t n_batches, uint64_t n_records,
                            const kta_batch *out, uint64_t *n_key_bytes, uint64_t *n_bad_batches)
{
    if (!ctx || !out) return KTA_ERR_INVALID;
    if (n_key_bytes) *n_key_bytes = 0;
    if (n_bad_batches) *n_bad_batches = 0;
    if (n_batches == 0) return KTA_OK;
    if (!blob_device || !descs_host) return KTA_ERR_INVALID;
    if ((reinterpret_cast<uintptr_t>(blob_device) & 15u) != 0) {
        kta_internal_set_error(ctx, "blob_device must be 16-byte aligned (and padded to a multiple of 16 bytes)");
        return KTA_ERR_INVALID;
    }
    if (n_records > out->capacity) {
        kta_internal_set_error(ctx, "decoded records exceed the output batch capacity");
        return KTA_ERR_CAPACITY;
    }
    KK(ctx, hipSetDevice(kta_internal_device(ctx)));
    hipStream_t s = kta_internal_stream(ctx);
    KafkaState *st = state_of(ctx);
    hipStream_t cs = kta_internal_copy_stream(ctx);
    const int dk = st->desc_slot;
    st->desc_slot ^= 1;
    if (!st->ev_desc_up[dk]) {
        KK(ctx, hipEventCreateWithFlags(&st->ev_desc_up[dk], hipEventDisableTiming));
        KK(ctx, hipEventCreateWithFlags(&st->ev_desc_free[dk], hipEventDisableTiming));
    }
    if (st->desc_cap2[dk] < n_batches) {
        KK(ctx, hipStreamSynchronize(s));
        KK(ctx, hipStreamSynchronize(cs));
        if (st->d_descs2[dk]) (void)hipFree(st->d_descs2[dk]);
        st->d_descs2[dk] = nullptr;
        KK(ctx, hipMalloc((void **)&st->d_descs2[dk], n_batches * sizeof(kta_kafka_batch_desc)));
        st->desc_cap2[dk] = n_batches;
        st->desc_used[dk] = false;
    }
    kta_kafka_batch_desc *const d_descs = st->d_descs2[dk];
    if (!st->d_scalars) {
        KK(ctx, hipMalloc((void **)&st->d_scalars, 2 * sizeof(uint64_t)));
        KK(ctx, hipMemsetAsync(st->d_scalars, 0, 2 * sizeof(uint64_t), s));
    }
    // (the kernels of the call before last read this buffer: the copy waits for them on the device, not the host)
    if (st->desc_used[dk]) KK(ctx, hipStreamWaitEvent(cs, st->ev_desc_free[dk], 0));
    KK(ctx, hipMemcpyAsync(d_descs, descs_host, n_batches * sizeof(kta_kafka_batch_desc), hipMemcpyHostToDevice, cs));
    KK(ctx, hipEventRecord(st->ev_desc_up[dk], cs));
    KK(ctx, hipStreamWaitEvent(s, st->ev_desc_up[dk], 0));
    st->desc_used[dk] = true;
    if (n_bad_batches || n_key_bytes)   // per-call counts wanted: start from zero (otherwise they accumulate)
        KK(ctx, hipMemsetAsync(st->d_scalars, 0, 2 * sizeof(uint64_t), s));
    // Zero-copy keys: key_off[i] is the key's offset inside the raw blob, so the caller passes the
    // blob itself as `key_bytes` to kta_submit_device; nothing is copied.  out->key_bytes is ignored.
    const bool want_keys = out->key_off != nullptr;
    if (want_keys && blob_len >= (1ull << 32)) {
        kta_internal_set_error(ctx, "record set must be < 4 GiB when key offsets are wanted (key_off is u32)");
        return KTA_ERR_CAPACITY;
    }
    const bool timing = kta_internal_timing(ctx);
    const uint32_t grid = (uint32_t)((n_batches + kLanesPerBlock - 1) / kLanesPerBlock);
    const uint4 *words = reinterpret_cast<const uint4 *>(blob_device);
    hipEvent_t a = nullptr, b = nullptr;
    uint64_t scal[2] = {0, 0};
    if (st->check_crcs) {   // librdkafka check.crcs=true: verify every batch before it is decoded
        if (!st->d_crc_tables) {
            CrcTables *host = new CrcTables();
            build_crc_tables(*host);
            hipError_t e = hipMalloc((void **)&st->d_crc_tables, sizeof(CrcTables));
            if (e == hipSuccess) e = hipMemcpy(st->d_crc_tables, host, sizeof(CrcTables), hipMemcpyHostToDevice);
            if (e == hipSuccess) e = hipMalloc((void **)&st->d_crc_bad, sizeof(uint64_t));
            if (e == hipSuccess) e = hipMemset(st->d_crc_bad, 0, sizeof(uint64_t));
            delete host;
            if (e != hipSuccess) return hip_err(ctx, e, "CRC-32C tables");
        }
        if (timing) { int rc = pair(ctx, st, 0, &a, &b); if (rc != KTA_OK) return rc; KK(ctx, hipEventRecord(a, s)); }
        hipLaunchKernelGGL(kafka_crc32c, dim3((uint32_t)n_batches), dim3(64), 0, s, words, d_descs, n_batches,
                           st->d_crc_tables, reinterpret_cast<unsigned long long *>(st->d_crc_bad));
        KK(ctx, hipGetLastError());
        if (timing) KK(ctx, hipEventRecord(b, s));
    }
    bool any_snappy = false, any_lz4 = false, any_gzip = false, any_zstd = false;
    uint64_t buffer_end = blob_len;
    for (uint64_t i = 0; i < n_batches; i++)
        if (descs_host[i].flags & (KTA_KB_SNAPPY | KTA_KB_LZ4 | KTA_KB_GZIP | KTA_KB_ZSTD)) {
            any_snappy = any_snappy || (descs_host[i].flags & KTA_KB_SNAPPY);
            any_lz4 = any_lz4 || (descs_host[i].flags & KTA_KB_LZ4);
            any_gzip = any_gzip || (descs_host[i].flags & KTA_KB_GZIP);
            any_zstd = any_zstd || (descs_host[i].flags & KTA_KB_ZSTD);
            if (descs_host[i].payload_end > buffer_end) buffer_end = descs_host[i].payload_end;
            if (descs_host[i].scratch_end > buffer_end) buffer_end = descs_host[i].scratch_end;
        }
    if (want_keys && buffer_end >= (1ull << 32)) {
        kta_internal_set_error(ctx, "blob + inflate area must be < 4 GiB when key offsets are wanted (key_off is u32)");
        return KTA_ERR_CAPACITY;
    }
    if (any_snappy || any_lz4 || any_gzip || any_zstd) {   // inflate compressed batches into their slices of the same buffer
        uint8_t *buf = const_cast<uint8_t *>(blob_device);
        uint32_t lane_codecs = 0u;
        if (st->variant != 1) {
            if (any_snappy)
                hipLaunchKernelGGL(kafka_snappy_inflate_coop, dim3((uint32_t)n_batches), dim3(64), 0, s, buf, d_descs,
                                   n_batches);
            if (any_lz4)
                hipLaunchKernelGGL(kafka_lz4_inflate_coop, dim3((uint32_t)n_batches), dim3(64), 0, s, buf, d_descs,
                                   n_batches);
        } else {
            lane_codecs |= KTA_KB_SNAPPY | KTA_KB_LZ4;
        }
        if (any_gzip && st->variant != 1) {   // Huffman decoding one lane per batch, then the copies one wave per batch
#ifndef KTA_GZIP_LANE_STAGE1
            hipLaunchKernelGGL(kafka_gzip_tokenize_wave, dim3((uint32_t)n_batches), dim3(64), 0, s, buf, d_descs, n_batches);
            const int only_left = 1;
#else
            const int only_left = 0;
#endif
            hipLaunchKernelGGL((kafka_gzip_tokenize<kGzTokLanes>), dim3((uint32_t)((n_batches + kGzTokLanes - 1) / kGzTokLanes)),
                               dim3(kGzTokLanes), 0, s, buf, d_descs, n_batches, only_left);
            hipLaunchKernelGGL(kafka_gzip_apply, dim3((uint32_t)n_batches), dim3(64), 0, s, buf, d_descs, n_batches);
        } else if (any_gzip) {
            hipLaunchKernelGGL((kafka_gzip_inflate<kGzipLanes>), dim3((uint32_t)((n_batches + kGzipLanes - 1) / kGzipLanes)),
                               dim3(kGzipLanes), 0, s, buf, d_descs, n_batches);
        }
        if (any_zstd && st->variant != 1)
            hipLaunchKernelGGL(kafka_zstd_inflate_coop, dim3((uint32_t)n_batches), dim3(64), 0, s, buf, d_descs, n_batches);
        else if (any_zstd)
            hipLaunchKernelGGL(kafka_zstd_inflate, dim3((uint32_t)((n_batches + kGzipLanes - 1) / kGzipLanes)), dim3(kGzipLanes), 0,
                               s, buf, d_descs, n_batches);
        if (lane_codecs)
            hipLaunchKernelGGL(kafka_inflate_lane, dim3(grid), dim3(kLanesPerBlock), 0, s, buf, d_descs, n_batches,
                               lane_codecs);
        KK(ctx, hipGetLastError());
    }
    if (timing) { int rc = pair(ctx, st, 1, &a, &b); if (rc != KTA_OK) return rc; KK(ctx, hipEventRecord(a, s)); }
    unsigned long long *d_bad = reinterpret_cast<unsigned long long *>(st->d_scalars + 1);
    unsigned long long *d_keyb = n_key_bytes ? reinterpret_cast<unsigned long long *>(st->d_scalars) : nullptr;
    const int wk = want_keys ? 1 : 0;
#define KTA_DECODE_COOP(G, W, R)                                                                                      \
    hipLaunchKernelGGL((kafka_decode_coop<G, W, R>), dim3((uint32_t)((n_batches + (G) - 1) / (G))), dim3(64), 0, s,  \
                       words, d_descs, n_batches, wk, out->partition, out->key_len, out->val_len, out->ts_ms,     \
                       out->key_off, (uint64_t)0, out->seq, (uint64_t)0, d_bad, d_keyb)
    switch (decode_variant_for(st->variant, n_batches, blob_len)) {
    case 1: // one lane per batch (kept for comparison)
        hipLaunchKernelGGL(kafka_decode, dim3(grid), dim3(kLanesPerBlock), 0, s, words, d_descs, n_batches, wk,
                           out->partition, out->key_len, out->val_len, out->ts_ms, out->key_off, (uint64_t)0, out->seq,
                           (uint64_t)0, d_bad, d_keyb);
        break;
    case 2: KTA_DECODE_COOP(1, 8192u, 256u); break;   // one wave per batch: calls with few batches
    case 10: KTA_DECODE_COOP(4, 3072u, 16u); break;   // 16 lanes per batch, 3 KiB windows (~11 records of the 256-byte mean), one parse round
    default: KTA_DECODE_COOP(2, 8192u, 32u); break;   // (11) 32 lanes per batch, 8 KiB windows (~31 records), one parse round
    }
#undef KTA_DECODE_COOP
    KK(ctx, hipGetLastError());
    if (timing) KK(ctx, hipEventRecord(b, s));
    KK(ctx, hipEventRecord(st->ev_desc_free[dk], s));          // the descriptors' buffer is free once these kernels are done
    if (n_bad_batches || n_key_bytes) {
        KK(ctx, hipMemcpyAsync(scal, st->d_scalars, 2 * sizeof(uint64_t), hipMemcpyDeviceToHost, s));
        KK(ctx, hipStreamSynchronize(s));
        if (n_bad_batches) *n_bad_batches = scal[1];
        if (n_key_bytes) *n_key_bytes = scal[0];
    }
    return KTA_OK;
}

int kta_kafka_configure(kta_ctx *ctx, uint64_t blob_capacity, int n_stages)
{
    if (!ctx || n_stages < 0 || n_stages > 16) return KTA_ERR_INVALID;
    KafkaState *st = state_of(ctx);
    if (!st->stages.empty()) {
        kta_internal_set_error(ctx, "kta_kafka_configure must precede the first kta_kafka_blob_acquire");
        return KTA_ERR_INVALID;
    }
    if (blob_capacity) st->blob_capacity = blob_capacity;
    if (st->blob_capacity >= (1ull << 32) - 64) {
        kta_internal_set_error(ctx, "blob capacity must be < 4 GiB (key offsets are u32)");
        return KTA_ERR_INVALID;
    }
    st->stages.resize(n_stages ? (size_t)n_stages : 3);
    return KTA_OK;
}

int kta_kafka_blob_acquire(kta_ctx *ctx, uint8_t **host_ptr, uint64_t *capacity)
{
    if (!ctx || !host_ptr || !capacity) return KTA_ERR_INVALID;
    KK(ctx, hipSetDevice(kta_internal_device(ctx)));
    KafkaState *st = state_of(ctx);
    if (st->stages.empty()) st->stages.resize(3);
    BlobStage &g = st->stages[st->cur];
    if (!g.h_blob) {
        g.cap = st->blob_capacity;
        KK(ctx, hipHostMalloc((void **)&g.h_blob, g.cap + 64, hipHostMallocDefault));
        g.d_cap = g.cap + 128;
        KK(ctx, hipMalloc((void **)&g.d_blob, g.d_cap));
        KK(ctx, hipEventCreateWithFlags(&g.copied, hipEventDisableTiming));
        KK(ctx, hipEventCreateWithFlags(&g.done, hipEventDisableTiming));
    }
    if (g.busy) { // the ring wrapped: the kernels that read this stage's device blob must be done
        KK(ctx, hipEventSynchronize(g.done));
        g.busy = false;
    }
    *host_ptr = g.h_blob;
    *capacity = g.cap;
    st->acquired = true;
    return KTA_OK;
}

int kta_kafka_blob_submit(kta_ctx *ctx, uint64_t len, int32_t partition, kta_kafka_index_stats *stats)
{
    if (!ctx || !stats) return KTA_ERR_INVALID;
    KK(ctx, hipSetDevice(kta_internal_device(ctx)));
    KafkaState *st = state_of(ctx);
    if (!st->acquired) {
        kta_internal_set_error(ctx, "kta_kafka_blob_submit without kta_kafka_blob_acquire");
        return KTA_ERR_INVALID;
    }
    BlobStage &g = st->stages[st->cur];
    if (len > g.cap) {
        kta_internal_set_error(ctx, "record set larger than the blob staging capacity");
        return KTA_ERR_CAPACITY;
    }
    st->acquired = false;
    int rc = kta_flush(ctx); // keep consumption order with per-message records staged earlier
    if (rc != KTA_OK) return rc;
    hipStream_t s = kta_internal_stream(ctx), cs = kta_internal_copy_stream(ctx);
    // 1. host: index the batch headers (pinned descriptor array, grown on demand)
    if (g.desc_cap == 0) {
        g.desc_cap = 4096;
        KK(ctx, hipHostMalloc((void **)&g.h_descs, g.desc_cap * sizeof(kta_kafka_batch_desc), hipHostMallocDefault));
    }
    const uint64_t inflate_at = (len + 127) & ~63ull;   // inflate area: right behind the raw bytes, 64-byte aligned
    rc = kta_kafka_index_host(g.h_blob, len, partition, 0, 0, inflate_at, g.h_descs, g.desc_cap, stats);
    if (rc == KTA_ERR_CAPACITY) {
        (void)hipHostFree(g.h_descs);
        g.desc_cap = stats->n_batches + stats->n_batches / 4;
        KK(ctx, hipHostMalloc((void **)&g.h_descs, g.desc_cap * sizeof(kta_kafka_batch_desc), hipHostMallocDefault));
        rc = kta_kafka_index_host(g.h_blob, len, partition, 0, 0, inflate_at, g.h_descs, g.desc_cap, stats);
    }
    if (rc != KTA_OK) return rc;
    if (stats->n_batches == 0) return KTA_OK;
    const bool keys = kta_internal_count_alive(ctx);
    const uint64_t used = stats->bytes_consumed, nrec = stats->n_records, nb = stats->n_batches;
    // Compressed batches inflate into an area behind the raw bytes.  Its size is a bound per batch (zstd /
    // LZ4 frames without a content size: blocks x block maximum), so a blob of many small batches could ask
    // for far more than it will use — and keys are addressed with 32-bit offsets into this one buffer.  The
    // batches are therefore processed in groups whose inflate span stays under a limit; the groups reuse
    // the area, which is safe because everything of a group (inflate, decode, both handlers) is enqueued on
    // the compute stream before the next group's inflate.
    uint64_t limit = st->inflate_limit ? st->inflate_limit : (1ull << 30);
    const uint64_t addressable = (1ull << 32) - (128ull << 20);
    if (keys && inflate_at + limit > addressable) limit = addressable > inflate_at + (64ull << 20) ? addressable - inflate_at : (64ull << 20);
    auto inflate_lo = [&](uint64_t i) { return g.h_descs[i].payload_off; };   // compressed batches only
    auto is_comp = [&](uint64_t i) { return (g.h_descs[i].flags & (KTA_KB_SNAPPY | KTA_KB_LZ4 | KTA_KB_GZIP | KTA_KB_ZSTD)) != 0; };
    auto inflate_hi = [&](uint64_t i) {
        const uint64_t e = g.h_descs[i].scratch_end > g.h_descs[i].payload_end ? g.h_descs[i].scratch_end : g.h_descs[i].payload_end;
        return (e + 63) & ~63ull;
    };
    std::vector<uint64_t> cut{0};                         // group k = batches [cut[k], cut[k+1])
    uint64_t max_span = 0;
    {
        uint64_t lo = ~0ull, hi = 0;
        for (uint64_t i = 0; i < nb; i++) {
            if (!is_comp(i)) continue;
            if (lo != ~0ull && inflate_hi(i) - lo > limit) {   // would exceed: close the group before batch i
                cut.push_back(i);
                if (hi - lo > max_span) max_span = hi - lo;
                lo = ~0ull;
            }
            if (lo == ~0ull) lo = inflate_lo(i);
            hi = inflate_hi(i);
        }
        if (lo != ~0ull && hi - lo > max_span) max_span = hi - lo;
        cut.push_back(nb);
    }
    if (inflate_at + max_span + 128 > g.d_cap) {   // grow the device buffer (the stage is idle here)
        (void)hipFree(g.d_blob);
        g.d_blob = nullptr;
        g.d_cap = inflate_at + max_span + max_span / 4 + 4096;
        KK(ctx, hipMalloc((void **)&g.d_blob, g.d_cap));
    }
    // 2. decoded columns of this stage (grown on demand; the stage is idle here).  Every group is submitted
    // as its own batch, whose columns must start 16-byte aligned: a group's records start at a multiple of 4.
    const uint64_t ncol = nrec + 4 * (cut.size() - 1);
    if (g.out_cap < ncol || (keys && !g.out.key_off)) {
        free_out(g.out);
        g.out_cap = ncol + ncol / 4 + 1024;
        KK(ctx, hipMalloc((void **)&g.out.partition, g.out_cap * 4 + 16));
        KK(ctx, hipMalloc((void **)&g.out.key_len, g.out_cap * 4 + 16));
        KK(ctx, hipMalloc((void **)&g.out.val_len, g.out_cap * 4 + 16));
        KK(ctx, hipMalloc((void **)&g.out.ts_ms, g.out_cap * 8 + 16));
        if (keys) KK(ctx, hipMalloc((void **)&g.out.key_off, g.out_cap * 4 + 16));
        g.out.capacity = g.out_cap;
    }
    g.out.key_bytes = keys ? g.d_blob : nullptr; // zero-copy: keys are hashed in place in the raw log
    g.out.key_bytes_capacity = keys ? inflate_at + max_span : 0;
    // 3. raw log over PCIe on the copy stream; decode + metric handlers on the compute stream
    KK(ctx, hipMemcpyAsync(g.d_blob, g.h_blob, (used + 63) & ~63ull, hipMemcpyHostToDevice, cs));
    KK(ctx, hipEventRecord(g.copied, cs));
    KK(ctx, hipStreamWaitEvent(s, g.copied, 0));
    const uint64_t base = kta_internal_take_seq(ctx, nrec);
    for (size_t k = 0; k + 1 < cut.size(); k++) {
        const uint64_t b0 = cut[k], b1 = cut[k + 1];
        if (b0 == b1) continue;
        uint64_t shift = 0;                               // this group's inflate slices start at inflate_at
        for (uint64_t i = b0; i < b1; i++)
            if (is_comp(i)) { shift = inflate_lo(i) - inflate_at; break; }
        if (shift)
            for (uint64_t i = b0; i < b1; i++)
                if (is_comp(i)) {
                    g.h_descs[i].payload_off -= shift;
                    g.h_descs[i].payload_end -= shift;
                    g.h_descs[i].scratch_end -= shift;
                }
        const uint64_t r0 = g.h_descs[b0].record_base;                       // consumption index of the group's first record
        const uint64_t r1 = b1 < nb ? g.h_descs[b1].record_base : nrec;
        const uint64_t c0 = (r0 + 4 * k + 3) & ~3ull;                         // its place in the columns: aligned, past group k-1
        if (c0 != r0)
            for (uint64_t i = b0; i < b1; i++) g.h_descs[i].record_base += c0 - r0;
        rc = kta_kafka_decode_device(ctx, g.d_blob, used, g.h_descs + b0, b1 - b0, ncol, &g.out, nullptr, nullptr);
        if (rc != KTA_OK) return rc;
        kta_batch view = g.out;                           // this group's records: [c0, c0 + r1 - r0) of the stage's columns
        view.partition += c0;
        view.key_len += c0;
        view.val_len += c0;
        view.ts_ms += c0;
        if (view.key_off) view.key_off += c0;
        if (view.seq) view.seq += c0;
        view.capacity = r1 - r0;
        rc = kta_submit_device(ctx, &view, r1 - r0, base + r0);
        if (rc != KTA_OK) return rc;
    }
    KK(ctx, hipEventRecord(g.done, s));
    g.busy = true;
    st->cur = (st->cur + 1) % (int)st->stages.size();
    return KTA_OK;
}

int kta_kafka_consume(kta_ctx *ctx, const uint8_t *bytes, uint64_t len, int32_t partition,
                      kta_kafka_index_stats *stats)
{
    if (!ctx || !bytes || !stats) return KTA_ERR_INVALID;
    memset(stats, 0, sizeof(*stats));
    KafkaState *st = state_of(ctx);
    if (st->stages.empty() && len + 64 > st->blob_capacity) {   // first use: size the ring for this caller
        int rc = kta_kafka_configure(ctx, len + len / 4 + 4096, 3);
        if (rc != KTA_OK) return rc;
    }
    uint64_t pos = 0;
    while (pos < len) {   // chunk at batch boundaries when the record set exceeds one staging blob
        uint8_t *dst;
        uint64_t cap;
        int rc = kta_kafka_blob_acquire(ctx, &dst, &cap);
        if (rc != KTA_OK) return rc;
        const uint64_t n = len - pos < cap ? len - pos : cap;
        memcpy(dst, bytes + pos, n);
        kta_kafka_index_stats one;
        rc = kta_kafka_blob_submit(ctx, n, partition, &one);
        if (rc != KTA_OK) return rc;
        stats->n_batches += one.n_batches;
        stats->n_records += one.n_records;
        stats->n_control_batches += one.n_control_batches;
        stats->n_compressed += one.n_compressed;
        stats->n_snappy += one.n_snappy;
        stats->n_lz4 += one.n_lz4;
        stats->n_gzip += one.n_gzip;
        stats->n_zstd += one.n_zstd;
        stats->inflate_bytes += one.inflate_bytes;
        stats->n_old_magic += one.n_old_magic;
        stats->bytes_consumed += one.bytes_consumed;
        if (one.bytes_consumed == 0) {   // not even one whole batch fits / trailing partial batch
            stats->trailing_bytes = len - pos;
            if (n == cap && cap < len - pos) {
                kta_internal_set_error(ctx, "a single record batch exceeds the blob staging capacity");
                return KTA_ERR_CAPACITY;
            }
            return KTA_OK;
        }
        pos += one.bytes_consumed;
        stats->trailing_bytes = len - pos;
    }
    return KTA_OK;
}

int kta_kafka_set_inflate_limit(kta_ctx *ctx, uint64_t bytes)
{
    if (!ctx) return KTA_ERR_INVALID;
    state_of(ctx)->inflate_limit = bytes;
    return KTA_OK;
}

int kta_kafka_set_check_crcs(kta_ctx *ctx, int enable)
{
    if (!ctx) return KTA_ERR_INVALID;
    state_of(ctx)->check_crcs = enable != 0;
    return KTA_OK;
}

int kta_kafka_crc_errors(kta_ctx *ctx, uint64_t *n)
{
    if (!ctx || !n) return KTA_ERR_INVALID;
    KafkaState *st = state_of(ctx);
    *n = 0;
    if (!st->d_crc_bad) return KTA_OK;
    KK(ctx, hipSetDevice(kta_internal_device(ctx)));
    KK(ctx, hipMemcpyAsync(n, st->d_crc_bad, sizeof(uint64_t), hipMemcpyDeviceToHost, kta_internal_stream(ctx)));
    KK(ctx, hipStreamSynchronize(kta_internal_stream(ctx)));
    return KTA_OK;
}

int kta_kafka_set_variant(kta_ctx *ctx, int variant)
{
    if (!ctx || !(variant == 0 || variant == 1 || variant == 2 || variant == 10 || variant == 11)) return KTA_ERR_INVALID;
    state_of(ctx)->variant = variant;
    return KTA_OK;
}

int kta_kafka_time_stats(kta_ctx *ctx, float avg_ms[2], uint64_t launches[2])
{
    if (!ctx || !avg_ms || !launches) return KTA_ERR_INVALID;
    KafkaState *st = state_of(ctx);
    int rc = drain(ctx, st);
    if (rc != KTA_OK) return rc;
    for (int k = 0; k < 2; k++) {
        launches[k] = st->ms_cnt[k];
        avg_ms[k] = st->ms_cnt[k] ? (float)(st->ms_sum[k] / (double)st->ms_cnt[k]) : -1.f;
        st->ms_sum[k] = 0;
        st->ms_cnt[k] = 0;
    }
    return KTA_OK;
}

} // extern "C"
