// kta_decode_coop.h — the wave-cooperative record decode kernel (kafka_decode_coop<G, W, R>) and the blob reader it
// shares with the lane-per-batch walk.  kta_kafka.hip includes this file INSIDE its unnamed namespace (the kernel keeps
// the name the profiles know); it has no includes of its own: the including file has included include/kta_kafka.h and
// kta_records.h (at file scope) and provides the HIP runtime — or, for
// the CPU suite, tests/native/wave_emu.h, which runs the 64 lanes of a workgroup as fibers that meet at
// __syncthreads / __any / __shfl_xor (tests/native/decode_coop_emu.cpp, tests/test_decode_emu.py): the kernel below
// is then the very source the GPU runs, window loads, LDS hand-over and round accounting included.
#pragma once

// ---- device-side byte reader over the blob: aligned 16-byte loads, one block cached -------------
// A record header (length, attributes, timestamp delta, offset delta, key length) is <= 26 bytes, so
// it costs one or two dependent loads instead of one per varint.
struct Reader {
    const uint4 *blocks;
    uint64_t pos;
    uint64_t cached_idx;
    uint4 cached;

    __device__ __forceinline__ uint32_t byte()
    {
        const uint64_t bi = pos >> 4;
        if (bi != cached_idx) {
            cached = blocks[bi];
            cached_idx = bi;
        }
        const uint32_t w = (uint32_t)(pos >> 2) & 3u;
        const uint32_t word = w == 0 ? cached.x : (w == 1 ? cached.y : (w == 2 ? cached.z : cached.w));
        const uint32_t b = (word >> ((uint32_t)(pos & 3u) * 8u)) & 0xFFu;
        pos++;
        return b;
    }
};

// unsigned LEB128 (at most 10 groups), then zig-zag
__device__ __forceinline__ long long read_varlong(Reader &r)
{
    unsigned long long v = 0;
    for (uint32_t shift = 0; shift < 70; shift += 7) {
        const uint32_t b = r.byte();
        v |= (unsigned long long)(b & 0x7Fu) << (shift < 64 ? shift : 63);
        if (!(b & 0x80u)) break;
    }
    return (long long)(v >> 1) ^ -(long long)(v & 1ull);
}

// Keeps a staged load where it was issued: without it the compiler sinks each load into the conditional
// LDS store that consumes it and the window fill becomes load -> wait -> store, one HBM round trip each.
__device__ __forceinline__ void pin(uint4 &v)
{
    asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w));
}

// A block of the log on its way into a window: read once, so past the caches (non-temporal).  Round 5 measured the switch on
// the <4, 3 KiB, 16> kernel (profiles/r05_decode_switches.jsonl, 4 M records in 16 KiB batches): - 9 % alone, - 11 ... - 15 % with
// the window bases on 128-byte lines — the columns the kernel writes then stay in L2 until their lines are full.
__device__ __forceinline__ uint4 load_block(const uint8_t *p)
{
#if defined(__HIP_DEVICE_COMPILE__)
    typedef uint32_t v4u __attribute__((ext_vector_type(4)));
    const v4u v = __builtin_nontemporal_load(reinterpret_cast<const v4u *>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
#else
    return *reinterpret_cast<const uint4 *>(p);
#endif
}

// A value of lane `l` (the same l for every lane) in a scalar register.  (tests/native/wave_emu.h: a meeting point.)
#ifndef KTA_READLANE
#define KTA_READLANE(v, l) ((uint32_t)__builtin_amdgcn_readlane((int)(v), (int)(l)))
#endif

// ---- wave-cooperative decode: G batches per wave ---------------------------------------------------
// The lane-per-batch walk (kafka_decode, kta_kafka.hip) is latency bound (one dependent HBM round trip per varint) and has
// only as many active lanes as there are batches.  Here a group of L = 64/G lanes owns one batch:
//   1. a window of the batch is streamed into LDS with coalesced 16-byte non-temporal loads, all in flight together; the WAVE
//      loads the windows of its G groups one KiB per instruction (every lane takes part in every group's window);
//   2. the group's first lane chains the record length prefixes inside the window (LDS latency) and
//      publishes the record starts (kta::rec::chain).  This step is serial per batch and costs a full wave
//      instruction per operation whatever the number of active lanes, so G > 1 matters: the G leaders of
//      a wave chain their batches in the same instruction stream (G = 1 spends half of the kernel here) —
//      and so the chain does nothing but follow the lengths: judging them is the parse's work;
//   3. the group's lanes parse one record each from LDS (kta::rec::parse_record: the header from one round
//      trip) and write the columns (consecutive indices: coalesced stores); keys stay where they are
//      (key_off points into the blob);
//   4. the next window starts at the first record that did not fit — or, after a large value, at the
//      next record start, so value bytes beyond the window are never loaded.
// Every group runs its own rounds; the wave loops until its last group is done.  What the kernel's time is made of, measured
// by switching its parts off (round 5, profiles/r05_decode_switches.jsonl, r05_sq_decode.txt; 4 M records in 16 KiB batches,
// <4, 3 KiB, 16>): streaming the windows through LDS alone 0.161-0.179 ms (6.0-6.7 TB/s: the floor), the chain adds nothing
// (hidden behind other waves' loads), the parse 0.01 ms, the column stores 0.04-0.08 ms; 20 instructions per record in all,
// an instruction every eighth cycle of a SIMD — the kernel waits for memory, it is not bound by issue any more.
// Occupancy: the windows' LDS bounds it (<4, 3 KiB, 16>: 12.4 KiB per wave, twelve waves per CU at 108 registers; <2, 8 KiB, 32>:
// 16.3 KiB, nine waves at 130); <1, 8 KiB, 256> (8 KiB of windows) is held to 96 registers by the hint below: five waves per SIMD.
#ifndef KTA_WAVES_PER_EU   // (tests/native/wave_emu.h defines it away: a host compiler does not parse the attribute)
#define KTA_WAVES_PER_EU(least, most) __attribute__((amdgpu_waves_per_eu(least, most)))
#endif
template <int G, uint32_t W, uint32_t R>   // batches per wave, window bytes, records per round of a group
__device__ __forceinline__ void decode_coop_rounds(const uint4 *blocks, const kta_kafka_batch_desc *descs,
                                                   uint64_t n_batches, int want_keys, int32_t *part,
                                                   int32_t *klen, int32_t *vlen, int64_t *ts, uint32_t *koff,
                                                   uint64_t blob_base, uint64_t *seq, uint64_t seq_base,
                                                   unsigned long long *n_bad, unsigned long long *n_keyb)
{
    namespace rec = kta::rec;
    constexpr uint32_t L = 64 / G;                // lanes per batch
    constexpr uint32_t NLOAD = W / (L * 16);      // staged 16-byte loads per lane and window
    constexpr uint32_t NT = R / L;                // parse rounds per window
    static_assert(W % (L * 16) == 0 && R % L == 0, "window geometry");
    constexpr bool KIB = kta::rec::line_windows(W);           // the wave loads a KiB of ONE window per instruction (the dispatcher's geometries)
    __shared__ uint4 s_win[G][W / 16 + 1];        // + 1: the register paths read whole dwords up to 16 bytes ahead
    __shared__ uint32_t s_start[G][R + 1];        // record starts relative to the window base; [found]: where the last one ends
    __shared__ uint64_t s_next[G];                // absolute position after the last chained record
    __shared__ uint32_t s_found[G], s_first_incomplete[G], s_bad[G];
    const uint32_t lane = threadIdx.x, g = lane / L, sub = lane % L;
    const uint8_t *win = reinterpret_cast<const uint8_t *>(s_win[g]);
    const uint64_t b = (uint64_t)blockIdx.x * G + g;      // the dispatcher balances the waves
    unsigned long long kb = 0;                            // this lane's share of the key bytes
    uint64_t end = 0, pos = 0, record_base = 0;
    int64_t ts_base = 0, ts_mask = 0;
    uint32_t total = 0, j = 0;                            // j: records finished
    int32_t partition = 0;
    bool bad = false;
    if (b < n_batches) {
        const kta_kafka_batch_desc &d = descs[b];
        end = d.payload_end; pos = d.payload_off; record_base = d.record_base;
        const bool append_time = (d.flags & KTA_KB_LOG_APPEND_TIME) != 0;   // every record carries maxTimestamp
        ts_base = append_time ? d.max_ts_ms : d.base_ts_ms;
        ts_mask = append_time ? 0 : -1;
        total = (uint32_t)d.n_records; partition = d.partition;
        bad = d.status != 0;                              // failed check.crcs or inflate: the batch is not delivered
    }
    bool run = !bad && j < total;                         // uniform inside a group
    while (__any(run)) {
        uint64_t wbase = 0;
        uint32_t limit = 0, end_rel = 0;                  // valid bytes in the window; the batch's end seen from its base
        bool to_the_end = false;                          // the window reaches the end of the batch
        uint64_t load_base = 0;                           // this group's window as the wave loads it: a group that does not run
        uint32_t load_bytes = 16;                         // reads the blob's first block into its own window
        if (run && pos >= end) { bad = true; run = false; }
        if (run) {
            wbase = rec::window_base(pos, W);
            const uint64_t span = ((end + 15) & ~15ull) - wbase, rest = end - wbase;
            const uint32_t wbytes = span < W ? (uint32_t)span : W;         // a multiple of 16, at least 16
            to_the_end = rest <= wbytes;
            limit = to_the_end ? (uint32_t)rest : wbytes;
            end_rel = rest < 0xF0000000ull ? (uint32_t)rest : 0xF0000000u;
            load_base = wbase; load_bytes = wbytes;
            if (sub == 0) { s_bad[g] = 0; s_first_incomplete[g] = R; }
        }
        // All loads of the windows are in flight together; a lane behind a batch's last block loads that block again and stores
        // it where its own would go — bytes at and behind `limit`, which decide nothing.
        if (KIB) {
            // Every lane takes part in every group's window: W / 1024 instructions per group, each a KiB of one window on a
            // 128-byte line (against 256 bytes of each of G windows: - 2 % at 16 KiB batches, - 4 % at 2 KiB, round 5); the
            // window's base and size come from the group's first lane through scalar registers.
            // (A group that does not run this round still takes its instructions: the blob's first block into its own window.
            // Passing it over with a scalar branch on the group's first lane — round 6, ADVICE of round 5 — cost the ordinary
            // case 10-18 %: 0.153 / 0.142 / 0.154 ms instead of 0.130 / 0.127 / 0.139 at 2 / 16 / 134 KiB side by side on one box,
            // the loads no longer one straight run.  Waves whose batches differ much in size pay for their finished groups; the dispatcher deals
            // neighbouring batches to a wave, which are mostly of a size.)
            constexpr uint32_t PER = W / 1024;
            uint4 stage[NLOAD];
#pragma unroll
            for (uint32_t u = 0; u < NLOAD; u++) {
                const uint32_t gg = u / PER, c = u % PER;
                const uint64_t gb = (uint64_t)KTA_READLANE((uint32_t)load_base, gg * L) |
                                    ((uint64_t)KTA_READLANE((uint32_t)(load_base >> 32), gg * L) << 32);
                const uint32_t gbytes = KTA_READLANE(load_bytes, gg * L);
                const uint32_t o = (c * 64 + lane) * 16;
                stage[u] = load_block(reinterpret_cast<const uint8_t *>(blocks) + gb + (o < gbytes - 16 ? o : gbytes - 16));
            }
#pragma unroll
            for (uint32_t u = 0; u < NLOAD; u++) pin(stage[u]);
#pragma unroll
            for (uint32_t u = 0; u < NLOAD; u++) s_win[u / PER][(u % PER) * 64 + lane] = stage[u];
        } else if (run) {                                 // (small windows, the tests': each group its own, 16 bytes per lane)
            const uint8_t *src = reinterpret_cast<const uint8_t *>(blocks) + load_base;
            uint4 stage[NLOAD];
#pragma unroll
            for (uint32_t u = 0; u < NLOAD; u++) {
                const uint32_t o = (sub + u * L) * 16;
                stage[u] = load_block(src + (o < load_bytes - 16 ? o : load_bytes - 16));
            }
#pragma unroll
            for (uint32_t u = 0; u < NLOAD; u++) pin(stage[u]);
#pragma unroll
            for (uint32_t u = 0; u < NLOAD; u++) s_win[g][sub + u * L] = stage[u];
        }
        __syncthreads();
        if (run && sub == 0) {                                                 // chain the length prefixes
            const uint32_t *w32 = reinterpret_cast<const uint32_t *>(win);
            const uint32_t want = total - j < R ? total - j : R;
            uint32_t k = 0, cur = (uint32_t)(pos - wbase);
            uint64_t next = 0;
            if (rec::chain(w32, limit, want, s_start[g], k, cur)) {
                // a length of three bytes or more in front of the chain — a record of 8 KiB or more, or a padded encoding:
                // byte by byte, and the round ends behind this record
                uint32_t off = cur;
                long long len;
                if (rec::window_varlong(win, off, limit, len)) {
                    const uint64_t rec_end = wbase + off + (uint64_t)len;
                    if (len < 0 || rec_end > end) s_bad[g] = 1;
                    else { s_start[g][k++] = cur; next = rec_end; }
                } else if (to_the_end) {
                    s_bad[g] = 1;                                              // ran into the end of the batch
                }                                                              // (else it straddles the window: next round)
            }
            if (!next) next = wbase + cur;
            const uint64_t next_rel = next - wbase;
            s_start[g][k] = next_rel < 0xFFFFFFFFull ? (uint32_t)next_rel : 0xFFFFFFFFu;
            s_found[g] = k;
            s_next[g] = next;
        }
        __syncthreads();
        uint32_t my_kl[NT];
#pragma unroll
        for (uint32_t t = 0; t < NT; t++) my_kl[t] = 0;
        if (run) {
            const uint32_t found = s_found[g];
            const uint32_t key_base = (uint32_t)(wbase - blob_base);           // key offsets are 32 bits wide
#pragma unroll
            for (uint32_t t = 0; t < NT; t++) {                                // one record per lane and round
                const uint32_t k = sub + L * t;
                if (k >= found) continue;
                const uint32_t start = s_start[g][k], rec_end = s_start[g][k + 1];
                if (rec_end > end_rel) { s_bad[g] = 1; continue; }             // the record overruns the batch
                rec::Record r;
                uint32_t verdict = rec::parse_record(win, start, rec_end, limit, r);
                if (verdict == rec::REC_VALUE_LENGTH_OUTSIDE) {                // behind a key of the window's size
                    Reader gr{blocks, wbase + r.after, ~0ull, make_uint4(0, 0, 0, 0)};
                    r.val_len = read_varlong(gr);
                    verdict = rec::value_fits(r.val_len, gr.pos - wbase, rec_end);
                }
                if (verdict == rec::REC_INCOMPLETE) { atomicMin(&s_first_incomplete[g], k); continue; }
                if (verdict != rec::REC_OK) { s_bad[g] = 1; continue; }
                const uint64_t i = record_base + j + k;
                part[i] = partition;
                klen[i] = (int32_t)r.key_len;
                vlen[i] = (int32_t)r.val_len;
                ts[i] = ts_base + (r.ts_delta & ts_mask);
                if (seq) seq[i] = seq_base + i;
                if (want_keys) koff[i] = r.key_len > 0 ? key_base + r.key : 0u;
                my_kl[t] = r.key_len > 0 ? (uint32_t)r.key_len : 0u;
            }
        }
        __syncthreads();
        if (run) {
            const uint32_t found = s_found[g], first_inc = s_first_incomplete[g];
            const uint32_t done = first_inc < found ? first_inc : found;
            if (s_bad[g] || done == 0) {                                       // done == 0: no progress, truncated batch
                bad = true;
                run = false;
            } else {
#pragma unroll
                for (uint32_t t = 0; t < NT; t++)
                    if (sub + L * t < done) kb += my_kl[t];
                j += done;
                pos = done < found ? wbase + s_start[g][done] : s_next[g];     // records >= done are redone
                run = j < total;
            }
        }
        __syncthreads();
    }
    if (bad) {
        for (uint32_t r = j + sub; r < total; r += L) {
            const uint64_t i = record_base + r;
            part[i] = -1; klen[i] = -1; vlen[i] = -1; ts[i] = -1;
            if (seq) seq[i] = seq_base + i;
            if (want_keys) koff[i] = 0u;
        }
        if (sub == 0) atomicAdd(n_bad, 1ull);
    }
    if (n_keyb) {                                  // only when the caller asked for the total (one atomic per wave)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) kb += __shfl_xor(kb, off);
        if (lane == 0 && kb) atomicAdd(n_keyb, kb);
    }
}

template <int G, uint32_t W, uint32_t R>
__global__ __launch_bounds__(64) KTA_WAVES_PER_EU(G * W <= 8192 ? 5 : 1, 8) void kafka_decode_coop(const uint4 *blocks, const kta_kafka_batch_desc *descs,
                                                        uint64_t n_batches, int want_keys, int32_t *part,
                                                        int32_t *klen, int32_t *vlen, int64_t *ts, uint32_t *koff,
                                                        uint64_t blob_base, uint64_t *seq, uint64_t seq_base,
                                                        unsigned long long *n_bad, unsigned long long *n_keyb)
{
    decode_coop_rounds<G, W, R>(blocks, descs, n_batches, want_keys, part, klen, vlen, ts, koff, blob_base, seq, seq_base,
                                       n_bad, n_keyb);
}
