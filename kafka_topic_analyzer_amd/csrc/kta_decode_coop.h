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

// ---- experiment switches (template parameter X of the kernel; X = 0 is the product) ----------------------------
// Timed side by side through kta_kafka_set_variant(100 + X / 200 + X) in a build with -DKTA_DECODE_EXPERIMENTS
// (tools/build_variant.sh); the product build instantiates X = 0 only.
#ifdef KTA_DEC_STATS
unsigned long long kta_dec_stats[4];   // (emulator only) rounds, rounds on prefetched blocks, prefetches not taken
#endif
enum : uint32_t {
    DX_NT_LOAD = 1,      // window loads non-temporal (the log is read once)
    DX_ALIGN128 = 2,     // window bases on 128-byte lines instead of 16-byte blocks
    DX_NT_STORE = 4,     // column stores non-temporal
    DX_LOAD_ONLY = 8,    // ablation: windows streamed through LDS, nothing else (columns are not written)
    DX_NO_STORE = 16,    // ablation: everything but the column stores
    DX_NO_PARSE = 32,    // ablation: loads and chain only
    DX_PF_EARLY = 64,    // the blocks behind the window are sent for before the chain (in registers while the round runs)
    DX_PF_LATE = 128,    // the next window is sent for when the chain knows where it begins, before the parse
    DX_LOAD_1K = 256,    // a load instruction of the wave reads 1 KiB of ONE window (all 64 lanes), not 256 bytes of each of four
    DX_STAGE = 512,      // the columns leave in whole aligned blocks of L records (a lane keeps the record of its slot until the block is full)
    DX_DEFER = 1024      // a round's columns are stored behind the NEXT round's window loads (a whole round to complete before anything waits on vmcnt)
};
#ifndef KTA_READLANE   // (tests/native/wave_emu.h: a meeting point)
#define KTA_READLANE(v, l) ((uint32_t)__builtin_amdgcn_readlane((int)(v), (int)(l)))
#endif

template <bool NT>
__device__ __forceinline__ uint4 load_block(const uint8_t *p)
{
#if defined(__HIP_DEVICE_COMPILE__)
    if (NT) {
        typedef uint32_t v4u __attribute__((ext_vector_type(4)));
        const v4u v = __builtin_nontemporal_load(reinterpret_cast<const v4u *>(p));
        return make_uint4(v.x, v.y, v.z, v.w);
    }
#endif
    return *reinterpret_cast<const uint4 *>(p);
}

template <bool NT, class T>
__device__ __forceinline__ void store_col(T *p, T v)
{
#if defined(__HIP_DEVICE_COMPILE__)
    if (NT) { __builtin_nontemporal_store(v, p); return; }
#endif
    *p = v;
}

// ---- wave-cooperative decode: G batches per wave ---------------------------------------------------
// The lane-per-batch walk (kafka_decode, kta_kafka.hip) is latency bound (one dependent HBM round trip per varint) and has
// only as many active lanes as there are batches.  Here a group of L = 64/G lanes owns one batch:
//   1. the group streams a window of the batch into LDS with coalesced 16-byte loads (all in flight);
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
// Every group runs its own rounds; the wave loops until its last group is done.  The kernel is bound by
// instruction issue (profiles/r04_sq_decode.txt), not by memory: the round is written for few instructions
// on the ordinary record, and everything unusual leaves the straight line.
// Register budget: the geometries the dispatcher picks for batches below 64 KiB (<4, 2 KiB, 16>, <8, 1 KiB, 16>) compile to
// 100 / 102 VGPRs without a hint — four waves per SIMD where their LDS (8.6 / 9.0 KiB per workgroup) admits 4.75 — and to
// 96 with it: no spills, the same instruction counts (the kernel measured in round 4 had 89).  The geometries with more
// than 8 KiB of windows per wave do not fit five waves and are compiled without the hint.
#ifndef KTA_WAVES_PER_EU   // (tests/native/wave_emu.h defines it away: a host compiler does not parse the attribute)
#define KTA_WAVES_PER_EU(least, most) __attribute__((amdgpu_waves_per_eu(least, most)))
#endif
template <int G, uint32_t W, uint32_t R, uint32_t X = 0>   // batches per wave, window bytes, records per round of a group, experiment switches
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
    constexpr bool NTL = (X & DX_NT_LOAD) != 0, NTS = (X & DX_NT_STORE) != 0;
    constexpr bool PFE = (X & DX_PF_EARLY) != 0, PFL = (X & DX_PF_LATE) != 0, PF = PFE || PFL;
    constexpr uint64_t AMASK = (X & DX_ALIGN128) ? 127ull : 15ull;   // a window begins on a block / on a line
    constexpr bool C1K = (X & DX_LOAD_1K) != 0;
    static_assert(!C1K || (W % 1024 == 0 && !PF), "1 KiB loads: whole KiB windows, no prefetch");
    constexpr bool STAGE = (X & DX_STAGE) != 0;
    static_assert(!STAGE || NT == 1, "staged columns: one record per lane and round");
    constexpr bool DEFER = (X & DX_DEFER) != 0;
    static_assert(!DEFER || (NT == 1 && !STAGE), "deferred columns: one record per lane and round");
    static_assert((!(PFE && PFL) && W % 128 == 0) || !(X & (DX_ALIGN128 | DX_PF_EARLY | DX_PF_LATE)), "experiment switches");
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
    // (PF) the blocks sent for during the last round, the base of the window they belong to (none: ~0), and how far into them a
    // round may begin (early form: they lie behind the window, the next record begins somewhere in them)
    constexpr uint32_t OV = W >= 256 ? (W / 4) & ~127u : 16u;
    uint4 ahead[PF ? NLOAD : 1];
#pragma unroll
    for (uint32_t u = 0; u < (PF ? NLOAD : 1); u++) ahead[u] = make_uint4(0, 0, 0, 0);   // (uninitialised, the array stays in scratch)
    uint64_t pf_base = ~0ull;
    // (STAGE) lane `sub` owns slot `sub` of the group's current block of L records (record indices blk .. blk + L - 1, blk a multiple
    // of L): the record it keeps for it, the one it parsed this round
    int32_t c_kl = 0, c_vl = 0, f_kl = 0, f_vl = 0;
    int64_t c_ts = 0, f_ts = 0;
    uint32_t c_ko = 0, f_ko = 0;
    bool have = false;
    bool f_ok = false, pend = false;                      // (DEFER) this round's record parsed; last round's record waits to be stored
    uint64_t p_i = 0;
    while (__any(run)) {
        uint64_t wbase = 0;
        uint32_t limit = 0, end_rel = 0;                  // valid bytes in the window; the batch's end seen from its base
        bool to_the_end = false;                          // the window reaches the end of the batch
        bool fetched = false;                             // (PF) the window is the one the last round sent for
        uint32_t c1k_bytes = 16;                          // (C1K) this group's window as the whole wave loads it: base and bytes
        uint64_t c1k_base = 0;
        if (run && pos >= end) { bad = true; run = false; }
        if (run) {
            wbase = pos & ~AMASK;
            if (PFE) {
                fetched = pf_base <= wbase && wbase - pf_base <= OV;
                if (fetched) wbase = pf_base;
            }
            if (PFL) fetched = pf_base == wbase;
#ifdef KTA_DEC_STATS
            if (sub == 0) { kta_dec_stats[0]++; kta_dec_stats[1] += fetched; kta_dec_stats[2] += pf_base != ~0ull && !fetched; }
#endif
            const uint64_t span = ((end + 15) & ~15ull) - wbase, rest = end - wbase;
            const uint32_t wbytes = span < W ? (uint32_t)span : W;         // a multiple of 16, at least 16
            to_the_end = rest <= wbytes;
            limit = to_the_end ? (uint32_t)rest : wbytes;
            end_rel = rest < 0xF0000000ull ? (uint32_t)rest : 0xF0000000u;
            if (C1K) {
                c1k_base = wbase; c1k_bytes = wbytes;
            } else if (!fetched) {
                // all loads of the window are in flight together; a lane behind the batch's last block loads that block
                // again and stores it where its own would go — bytes at and behind `limit`, which decide nothing
                const uint8_t *src = reinterpret_cast<const uint8_t *>(blocks) + wbase;
                uint4 stage[NLOAD];
#pragma unroll
                for (uint32_t u = 0; u < NLOAD; u++) {
                    const uint32_t o = (sub + u * L) * 16;
                    stage[u] = load_block<NTL>(src + (o < wbytes - 16 ? o : wbytes - 16));
                }
#pragma unroll
                for (uint32_t u = 0; u < NLOAD; u++) pin(stage[u]);
#pragma unroll
                for (uint32_t u = 0; u < NLOAD; u++) s_win[g][sub + u * L] = stage[u];
            } else if (PF) {                              // (its own registers and stores: no copies between the two ways)
#pragma unroll
                for (uint32_t u = 0; u < NLOAD; u++) s_win[g][sub + u * L] = ahead[u];
            }
            if (sub == 0) { s_bad[g] = 0; s_first_incomplete[g] = R; }
        }
        if (DEFER && !C1K) {
            if (pend) {
                store_col<NTS>(&part[p_i], partition);
                store_col<NTS>(&klen[p_i], c_kl);
                store_col<NTS>(&vlen[p_i], c_vl);
                store_col<NTS>(&ts[p_i], c_ts);
                if (seq) store_col<NTS>(&seq[p_i], (uint64_t)(seq_base + p_i));
                if (want_keys) store_col<NTS>(&koff[p_i], c_ko);
            }
            pend = false;
        }
        if (C1K) {
            // every lane takes part in every group's window: W / 1024 instructions of 1 KiB each per group (a group that does not
            // run reads the blob's first block into its own window)
            constexpr uint32_t PER = W / 1024;
            uint4 stage[NLOAD];
#pragma unroll
            for (uint32_t u = 0; u < NLOAD; u++) {
                const uint32_t gg = u / PER, c = u % PER;
                const uint64_t gb = (uint64_t)KTA_READLANE((uint32_t)c1k_base, gg * L) | ((uint64_t)KTA_READLANE((uint32_t)(c1k_base >> 32), gg * L) << 32);
                const uint32_t gbytes = KTA_READLANE(c1k_bytes, gg * L);
                const uint32_t o = (c * 64 + lane) * 16;
                stage[u] = load_block<NTL>(reinterpret_cast<const uint8_t *>(blocks) + gb + (o < gbytes - 16 ? o : gbytes - 16));
            }
#pragma unroll
            for (uint32_t u = 0; u < NLOAD; u++) pin(stage[u]);
#pragma unroll
            for (uint32_t u = 0; u < NLOAD; u++) s_win[u / PER][(u % PER) * 64 + lane] = stage[u];
            if (DEFER) {
                if (pend) {
                    store_col<NTS>(&part[p_i], partition);
                    store_col<NTS>(&klen[p_i], c_kl);
                    store_col<NTS>(&vlen[p_i], c_vl);
                    store_col<NTS>(&ts[p_i], c_ts);
                    if (seq) store_col<NTS>(&seq[p_i], (uint64_t)(seq_base + p_i));
                    if (want_keys) store_col<NTS>(&koff[p_i], c_ko);
                }
                pend = false;
            }
        }
        __syncthreads();
        if (X & DX_LOAD_ONLY) {                           // ablation: the window is in LDS; on to the next one
            if (run) {
                kb += s_win[g][sub].x;
                pos = wbase + limit;
                run = !to_the_end;
            }
            __syncthreads();
            continue;
        }
        if (PFE) {
            // Every lane issues the loads, wanted or not (a group that is done or at its batch's end reads the blob's
            // first block): loads under a condition would merge with the old registers through copies, and a copy
            // waits for its load.
            const bool want = run && !to_the_end;         // (the window is full: W bytes, and the batch goes on behind it)
            pf_base = want ? wbase + W : ~0ull;
            const uint64_t from = want ? pf_base : 0ull, span = want ? ((end + 15) & ~15ull) - pf_base : 16ull;
            const uint32_t wbytes = span < W ? (uint32_t)span : W;             // as the round that takes them computes it
            const uint8_t *src = reinterpret_cast<const uint8_t *>(blocks) + from;
#pragma unroll
            for (uint32_t u = 0; u < NLOAD; u++) {
                const uint32_t o = (sub + u * L) * 16;
                ahead[u] = load_block<NTL>(src + (o < wbytes - 16 ? o : wbytes - 16));
            }
        }
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
        if (PFL) {
            // The chain has said where the next window begins (unless a record of this round turns out incomplete: then the
            // blocks are not taken).  Every lane issues the loads, wanted or not, as above.
            const uint64_t next = run ? s_next[g] : end;
            const bool want = run && next < end && j + s_found[g] < total;
            pf_base = want ? next & ~AMASK : ~0ull;
            const uint64_t from = want ? pf_base : 0ull, span = want ? ((end + 15) & ~15ull) - pf_base : 16ull;
            const uint32_t wbytes = span < W ? (uint32_t)span : W;
            const uint8_t *src = reinterpret_cast<const uint8_t *>(blocks) + from;
#pragma unroll
            for (uint32_t u = 0; u < NLOAD; u++) {
                const uint32_t o = (sub + u * L) * 16;
                ahead[u] = load_block<NTL>(src + (o < wbytes - 16 ? o : wbytes - 16));
            }
        }
        uint32_t my_kl[NT];
#pragma unroll
        for (uint32_t t = 0; t < NT; t++) my_kl[t] = 0;
        f_ok = false;
        if (run && !(X & DX_NO_PARSE)) {
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
                if (X & DX_NO_STORE) {                                         // ablation: the parse stays alive through kb
                    my_kl[t] = (uint32_t)r.key_len + (uint32_t)r.val_len + (uint32_t)r.ts_delta + r.key;
                    continue;
                }
                if (DEFER) f_ok = true;
                if (STAGE || DEFER) {
                    f_kl = (int32_t)r.key_len; f_vl = (int32_t)r.val_len; f_ts = ts_base + (r.ts_delta & ts_mask);
                    f_ko = r.key_len > 0 ? key_base + r.key : 0u;
                    my_kl[t] = r.key_len > 0 ? (uint32_t)r.key_len : 0u;
                    continue;
                }
                store_col<NTS>(&part[i], partition);
                store_col<NTS>(&klen[i], (int32_t)r.key_len);
                store_col<NTS>(&vlen[i], (int32_t)r.val_len);
                store_col<NTS>(&ts[i], (int64_t)(ts_base + (r.ts_delta & ts_mask)));
                if (seq) store_col<NTS>(&seq[i], (uint64_t)(seq_base + i));
                if (want_keys) store_col<NTS>(&koff[i], r.key_len > 0 ? key_base + r.key : 0u);
                my_kl[t] = r.key_len > 0 ? (uint32_t)r.key_len : 0u;
            }
        }
        if (X & DX_NO_PARSE) {
            if (run) my_kl[0] = s_start[g][sub < R ? sub : 0];
        }
        __syncthreads();
        const uint64_t i0 = record_base + j;              // (STAGE) the index of this round's first record
        uint32_t dn = 0;                                  // (STAGE) the records this round delivers
        if (run) {
            const uint32_t found = s_found[g], first_inc = s_first_incomplete[g];
            const uint32_t done = first_inc < found ? first_inc : found;
            if (s_bad[g] || (done == 0 && !(PFE && fetched))) {                // done == 0: no progress, truncated batch
                bad = true;
                run = false;
            } else if (PFE && done == 0) {
                pf_base = ~0ull;            // (PF) the window began too far before this record: the next one begins at it
            } else {
#pragma unroll
                for (uint32_t t = 0; t < NT; t++)
                    if (sub + L * t < done) kb += my_kl[t];
                dn = done;
                j += done;
                pos = done < found ? wbase + s_start[g][done] : s_next[g];     // records >= done are redone
                run = j < total;
            }
        }
        if (DEFER) {
            pend = f_ok && sub < dn;
            p_i = i0 + sub;
            c_kl = f_kl; c_vl = f_vl; c_ts = f_ts; c_ko = f_ko;
        }
        if (STAGE) {
            // The round's records k = 0 .. dn - 1 (lane k parsed record k) go to the slots c, c + 1, ... of the current block and on
            // into the next one: slot `sub` takes record (sub - c) mod L.  A block that fills up leaves as ONE aligned store per
            // column (L records: 64 bytes of a 4-byte column with sixteen lanes per batch); what belongs to the next block is kept.
            const uint32_t c = (uint32_t)i0 & (L - 1u), k_src = (sub - c) & (L - 1u), from = g * L + k_src;
            const int32_t n_kl = __shfl(f_kl, from), n_vl = __shfl(f_vl, from);
            const int64_t n_ts = __shfl(f_ts, from);
            const uint32_t n_ko = __shfl(f_ko, from);
            const bool got = k_src < dn, into_cur = sub >= c;
            if (c + dn >= L) {                            // (uniform inside a group) the current block is complete
                const bool valid = into_cur ? true : have;
                const uint64_t i = (i0 & ~(uint64_t)(L - 1u)) + sub;
                if (valid) {
                    store_col<NTS>(&part[i], partition);
                    store_col<NTS>(&klen[i], into_cur ? n_kl : c_kl);
                    store_col<NTS>(&vlen[i], into_cur ? n_vl : c_vl);
                    store_col<NTS>(&ts[i], into_cur ? n_ts : c_ts);
                    if (seq) store_col<NTS>(&seq[i], (uint64_t)(seq_base + i));
                    if (want_keys) store_col<NTS>(&koff[i], into_cur ? n_ko : c_ko);
                }
                have = !into_cur && got;
                c_kl = n_kl; c_vl = n_vl; c_ts = n_ts; c_ko = n_ko;            // (meaningful where `have`)
            } else if (into_cur && got) {
                have = true;
                c_kl = n_kl; c_vl = n_vl; c_ts = n_ts; c_ko = n_ko;
            }
        }
        __syncthreads();
    }
    if (DEFER && pend) {                                  // the last round's records
        part[p_i] = partition; klen[p_i] = c_kl; vlen[p_i] = c_vl; ts[p_i] = c_ts;
        if (seq) seq[p_i] = seq_base + p_i;
        if (want_keys) koff[p_i] = c_ko;
    }
    if (STAGE && have) {                                  // what is left of the batch's last block
        const uint64_t i = ((record_base + j) & ~(uint64_t)(L - 1u)) + sub;
        part[i] = partition; klen[i] = c_kl; vlen[i] = c_vl; ts[i] = c_ts;
        if (seq) seq[i] = seq_base + i;
        if (want_keys) koff[i] = c_ko;
    }
    if (X & (DX_LOAD_ONLY | DX_NO_STORE | DX_NO_PARSE)) {   // ablations: what they computed is "used", the columns stay as they were
        if (kb == 0x123456789ABCull) part[0] = 1;
        return;
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

template <int G, uint32_t W, uint32_t R, uint32_t X = 0>
__global__ __launch_bounds__(64) KTA_WAVES_PER_EU(G * W <= 8192 ? 5 : 1, 8) void kafka_decode_coop(const uint4 *blocks, const kta_kafka_batch_desc *descs,
                                                        uint64_t n_batches, int want_keys, int32_t *part,
                                                        int32_t *klen, int32_t *vlen, int64_t *ts, uint32_t *koff,
                                                        uint64_t blob_base, uint64_t *seq, uint64_t seq_base,
                                                        unsigned long long *n_bad, unsigned long long *n_keyb)
{
    decode_coop_rounds<G, W, R, X>(blocks, descs, n_batches, want_keys, part, klen, vlen, ts, koff, blob_base, seq, seq_base,
                                       n_bad, n_keyb);
}
