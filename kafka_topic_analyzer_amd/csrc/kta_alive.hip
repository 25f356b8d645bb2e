// kta_alive.hip — the partitioned alive-key pass (gfx950): LogCompactionInMemoryMetrics::handle_message
// (/root/reference/src/metric.rs:288-305) over one batch, as two kernels that keep the random accesses
// on chip.
//
// The table semantics are those of kta_kernels.hip: table[h] = max(table[h], ((seq+1) << 1) | alive), the
// largest sequence number per 32-bit hash slot being the last writer in consumption order, which is what
// sequential BitSet::insert / remove leave behind (metric.rs:273-280).  What changes is how a batch gets
// there.  One memory-side atomic (or even one L2-missing read) per record caps the pass at 20-50 G
// records/s whatever the kernel does, so:
//
//   pass 1  kta_alive_partition   streams the batch once (coalesced), hashes every key (fnv32.rs:92-101)
//           and appends the pair (h, local sequence, alive) to the segment [bucket][workgroup], bucket =
//           the hash's top bits.  Segments are private to a workgroup: no global atomics.  Pairs do not go
//           to memory one by one — a partial 64-byte write costs a memory-side read-modify-write, 28 G/s
//           whatever its size (tools/ubench_scatter.hip) — but through a ring of 16 pairs per bucket in
//           LDS that is flushed in aligned 64-byte blocks (software write combining).
//   pass 2  kta_alive_apply       one workgroup owns one bucket, i.e. one contiguous region of the table.
//           It merges the bucket's pairs in an LDS hash table (last writer per slot: 64-bit LDS max on
//           (h, seq)), so that records superseded inside the batch die in LDS, and then applies the
//           survivors to its region with a plain read / compare / write — it is the region's only writer.
//
// Exactness never depends on sizes or on luck: a record that finds no room in its ring, its segment or
// the LDS table takes the direct path (pre-read + atomicMax, as kta_alive_update_filtered), and max is
// commutative.  The running alive count telescopes exactly as in kta_alive_update_counting.
#include "kta_kernels.h"

// Phase timers for tools/ubench_alive.hip (which includes this file with KTA_ALIVE_PHASES defined): thread 0
// of every workgroup adds the ticks (100 MHz) it spent between marks.  Compiled out of the library.
#ifdef KTA_ALIVE_PHASES
__device__ unsigned long long g_kta_phase[2][8];
#define KTA_PHASE_BEGIN unsigned long long t_ph = wall_clock64()
#define KTA_PHASE(k, i)                                              \
    do {                                                             \
        if (threadIdx.x == 0) {                                      \
            const unsigned long long t_now = wall_clock64();         \
            atomicAdd(&g_kta_phase[k][i], t_now - t_ph);             \
            t_ph = t_now;                                            \
        }                                                            \
    } while (0)
#else
#define KTA_PHASE_BEGIN
#define KTA_PHASE(k, i)
#endif

namespace kta {

namespace {

constexpr uint32_t kFnvInit = 0x811c9dc5u;   // fnv32.rs:80
constexpr uint32_t kFnvMul = 0x811c9dc5u;    // fnv32.rs:97: the multiplier is the offset basis, not the FNV prime

__device__ __forceinline__ uint32_t fnv_byte(uint32_t h, uint32_t b) { return (h ^ b) * kFnvMul; }

__device__ __forceinline__ uint32_t fnv_word(uint32_t h, uint32_t w)
{
    h = fnv_byte(h, w & 0xFFu);
    h = fnv_byte(h, (w >> 8) & 0xFFu);
    h = fnv_byte(h, (w >> 16) & 0xFFu);
    return fnv_byte(h, w >> 24);
}

__device__ __forceinline__ uint32_t fnv_16(uint32_t h, const uint4 &v)
{
    return fnv_word(fnv_word(fnv_word(fnv_word(h, v.x), v.y), v.z), v.w);
}

// FNV of `len` more bytes at k (any alignment), continuing from h.  gfx950 runs in unaligned access mode,
// so the body is 16-byte loads at the key's own address; only the last 1..3 bytes go through aligned
// dwords that overlap the key (never a byte beyond the 4-byte word that holds the key's end).
__device__ __forceinline__ uint32_t fnv32_more(uint32_t h, const uint8_t *k, uint32_t len)
{
    while (len >= 16u) {
        uint4 v;
        __builtin_memcpy(&v, k, 16);
        h = fnv_16(h, v);
        k += 16;
        len -= 16u;
    }
    if (len >= 8u) {
        uint2 v;
        __builtin_memcpy(&v, k, 8);
        h = fnv_word(fnv_word(h, v.x), v.y);
        k += 8;
        len -= 8u;
    }
    if (len >= 4u) {
        uint32_t v;
        __builtin_memcpy(&v, k, 4);
        h = fnv_word(h, v);
        k += 4;
        len -= 4u;
    }
    if (len) {
        const uintptr_t a = reinterpret_cast<uintptr_t>(k);
        const uint32_t *w = reinterpret_cast<const uint32_t *>(a & ~(uintptr_t)3);
        const uint32_t skip = (uint32_t)(a & 3u);
        uint64_t two = w[0];
        if (skip + len > 4u) two |= (uint64_t)w[1] << 32;
        two >>= 8u * skip;
        for (uint32_t j = 0; j < len; j++) {
            h = fnv_byte(h, (uint32_t)two & 0xFFu);
            two >>= 8;
        }
    }
    return h;
}

// The direct path: what kta_alive_update_filtered does for one record.
__device__ __forceinline__ long long direct_update(unsigned long long *table, uint32_t h, unsigned long long v)
{
    const unsigned long long seen = __hip_atomic_load(&table[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (seen >= v) return 0;
    const unsigned long long old = atomicMax(&table[h], v);
    return v > old ? (long long)(v & 1ull) - (long long)(old & 1ull) : 0;
}

// Records that found no room on chip (ring, segment, LDS set) are not sent down the direct path where
// they stand — a wave would wait microseconds for two dependent memory round trips while its workgroup
// waits for it at the next barrier — but parked in a small LDS list that the whole workgroup drains, one
// entry per thread, when it fills up and at the end.
constexpr uint32_t kSpill = 512;

struct SpillList {
    uint32_t n;
    uint32_t h[kSpill];
    unsigned long long v[kSpill];
};

// false: the list is full, the caller runs the direct path itself
__device__ __forceinline__ bool spill_push(SpillList &sp, uint32_t h, unsigned long long v)
{
    const uint32_t k = atomicAdd(&sp.n, 1u);
    if (k >= kSpill) return false;
    sp.h[k] = h;
    sp.v[k] = v;
    return true;
}

// all threads of the workgroup, between two barriers of the caller's
__device__ __forceinline__ long long spill_drain(SpillList &sp, unsigned long long *table)
{
    const uint32_t n = sp.n < kSpill ? sp.n : kSpill;
    long long delta = 0;
    for (uint32_t k = threadIdx.x; k < n; k += blockDim.x) delta += direct_update(table, sp.h[k], sp.v[k]);
    return delta;
}

__device__ __forceinline__ void add_running(long long delta, long long *running, long long *s_w)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) delta += __shfl_xor(delta, off);
    const uint32_t nw = blockDim.x >> 6;
    if ((threadIdx.x & 63u) == 0u) s_w[threadIdx.x >> 6] = delta;
    __syncthreads();
    if (threadIdx.x == 0) {
        long long t = 0;
        for (uint32_t w = 0; w < nw; w++) t += s_w[w];
        if (t) atomicAdd(reinterpret_cast<unsigned long long *>(running), (unsigned long long)t);
    }
}

// ------------------------------------------------------------------------------------------------------
// pass 1: hash + partition with software write combining
// ------------------------------------------------------------------------------------------------------
constexpr int kPartThreads = 1024;                 // one workgroup per CU: the rings take most of the LDS
#ifndef KTA_PART_RECS
#define KTA_PART_RECS 4
#endif
constexpr int kPartRecs = KTA_PART_RECS;           // records per thread per round
constexpr uint32_t kPartRound = kPartThreads * kPartRecs;
constexpr uint32_t kRing = 16;                     // pairs per bucket ring: two 64-byte blocks

struct PartCols {
    int32_t kl[kPartRecs], vl[kPartRecs];
    uint32_t ko[kPartRecs];
};
struct PartKeys {
    uint4 k16[kPartRecs];                          // the 16 bytes at the key's offset
};

__device__ __forceinline__ void load_cols(const AliveColumns &c, uint64_t n, uint64_t base, PartCols &r)
{
#pragma unroll
    // Unconditional loads (the index is clamped, the result masked): a load under a lane predicate becomes
    // a branch, and the compiler then waits with vmcnt(0) wherever the value is used — which would also wait
    // for every prefetch issued in between.
    for (int j = 0; j < kPartRecs; j++) {
        const uint64_t i = base + (uint64_t)j * kPartThreads + threadIdx.x;
        const bool in = i < n;
        const uint64_t ic = in ? i : n - 1;
        const int32_t kl = c.key_len[ic], vl = c.val_len[ic];
        const uint32_t ko = c.key_off[ic];
        r.kl[j] = in ? kl : -1;
        r.vl[j] = in ? vl : -1;
        r.ko[j] = in ? ko : 0u;
    }
}

// The first 16 bytes at every key's offset, whatever the key's length: the bytes past a shorter key are
// loaded and ignored (key_bytes is readable for 16 bytes past its last key: kta_hip.h).
__device__ __forceinline__ void load_keys(const AliveColumns &c, const PartCols &r, PartKeys &k)
{
#pragma unroll
    for (int j = 0; j < kPartRecs; j++)      // unconditional as well: a keyless record loads the blob's first 16 bytes
        __builtin_memcpy(&k.k16[j], c.key_bytes + (r.kl[j] > 0 ? r.ko[j] : 0u), 16);
}

#ifdef KTA_HASH_NOINLINE   /* experiment of tools/ubench_alive.hip: the general-length path out of line (code size) */
__device__ __attribute__((noinline)) uint32_t fnv32_more_ool(uint32_t h, const uint8_t *k, uint32_t len) { return fnv32_more(h, k, len); }
#else
__device__ __forceinline__ uint32_t fnv32_more_ool(uint32_t h, const uint8_t *k, uint32_t len) { return fnv32_more(h, k, len); }
#endif

// FNV of a key whose first 16 bytes are in registers
__device__ __forceinline__ uint32_t fnv32_prefetched(const uint4 &k16, const uint8_t *key, uint32_t len)
{
    if (len == 16u) return fnv_16(kFnvInit, k16);
    if (len > 16u) return fnv32_more_ool(fnv_16(kFnvInit, k16), key + 16, len - 16u);
    const uint32_t w[4] = {k16.x, k16.y, k16.z, k16.w};
    uint32_t h = kFnvInit;
#pragma unroll
    for (uint32_t d = 0; d < 3; d++)
        if (len >= 4u * (d + 1u)) h = fnv_word(h, w[d]);
    const uint32_t q = len >> 2;
    uint32_t tw = q == 0u ? w[0] : (q == 1u ? w[1] : (q == 2u ? w[2] : w[3]));
    for (uint32_t t = len & 3u; t > 0u; t--) {
        h = fnv_byte(h, tw & 0xFFu);
        tw >>= 8;
    }
    return h;
}

// LDS address (in pairs) of position p of bucket b's ring.  A ring is one 128-byte row, so without a twist
// every bucket's entry k would sit in the same two banks; rows are rotated by 2 * (b & 7) entries (an even
// rotation keeps the 16-byte pieces of a block aligned).
__device__ __forceinline__ uint32_t ring_at(uint32_t b, uint32_t p)
{
    return b * kRing + ((p + 2u * (b & 7u)) & (kRing - 1));
}

// Write the completed 64-byte blocks [from, to & ~7) of up to kPartRecs segments from their rings (aligned:
// cap is a multiple of 8), and forget them.
template <int BLOG2>
__device__ __forceinline__ void flush_blocks(const unsigned long long *s_ring, unsigned long long *__restrict__ pairs,
                                             uint32_t W, uint32_t w, uint32_t cap, uint32_t (&pb)[kPartRecs],
                                             uint32_t (&from)[kPartRecs], uint32_t (&to)[kPartRecs])
{
#pragma unroll
    for (int j = 0; j < kPartRecs; j++) {
        unsigned long long *seg = pairs + ((uint64_t)pb[j] * W + w) * cap;
        for (uint32_t blk = from[j]; blk + 8 <= to[j]; blk += 8) {
            ulonglong2 *dst = reinterpret_cast<ulonglong2 *>(seg + blk);
            const ulonglong2 a0 = *reinterpret_cast<const ulonglong2 *>(s_ring + ring_at(pb[j], blk));
            const ulonglong2 a1 = *reinterpret_cast<const ulonglong2 *>(s_ring + ring_at(pb[j], blk + 2));
            const ulonglong2 a2 = *reinterpret_cast<const ulonglong2 *>(s_ring + ring_at(pb[j], blk + 4));
            const ulonglong2 a3 = *reinterpret_cast<const ulonglong2 *>(s_ring + ring_at(pb[j], blk + 6));
#ifdef KTA_DBG_NOFLUSH  /* ablation build of tools/ubench_alive.hip only */
            if (a0.x == 0x1234ull) dst[0] = a1;
#else
            dst[0] = a0;
            dst[1] = a1;
            dst[2] = a2;
            dst[3] = a3;
#endif
        }
        from[j] = to[j] = 0u;
    }
}

// pair = h << 32 | (local index + 1) << 1 | alive       (local index < 2^31 - 1: a pair is never zero)
template <int BLOG2>
__global__ __launch_bounds__(kPartThreads) void kta_alive_partition(AliveColumns c, uint64_t n, uint64_t base_seq,
                                                                    unsigned long long *__restrict__ pairs,
                                                                    uint32_t *__restrict__ counts, uint32_t cap,
                                                                    unsigned long long *__restrict__ table,
                                                                    long long *__restrict__ running)
{
    constexpr uint32_t B = 1u << BLOG2;
    KTA_PHASE_BEGIN;
    // 128-byte aligned whatever the static LDS before it adds up to: a ring is one 128-byte row, read back in
    // 16-byte pieces
    extern __shared__ __attribute__((aligned(128))) unsigned long long s_ring[];   // B x kRing pairs
    uint32_t *s_cnt = reinterpret_cast<uint32_t *>(s_ring + (size_t)B * kRing);   // arrivals of this round
    uint32_t *s_fill = s_cnt + B;                                        // pairs accepted into the segment so far
    __shared__ long long s_w[kPartThreads / 64];
    __shared__ SpillList s_spill;
    for (uint32_t b = threadIdx.x; b < B; b += kPartThreads) {
        s_cnt[b] = 0;
        s_fill[b] = 0;
    }
    if (threadIdx.x == 0) s_spill.n = 0;
    __syncthreads();
    const uint32_t W = gridDim.x, w = blockIdx.x;
    const uint64_t nrounds = (n + kPartRound - 1) / kPartRound;
    long long delta = 0;
    // Rounds are dealt round-robin and walked from the END of the batch: what overflows into the direct
    // path (a hot key fills its ring at once) then meets the newest records first, and the pre-read of
    // the direct path filters the older ones.
    if (nrounds > w) {
        int64_t rd = (int64_t)(w + ((nrounds - 1 - w) / W) * W);
        // Two-stage prefetch: the columns run two rounds ahead of the hash, the key bytes (whose addresses
        // come from the columns) one round ahead — no load waits for another inside a round.  Two register
        // sets alternate (the loop is unrolled by two): copying "next" into "current" at the end of a round
        // would make the wave wait for the prefetch at the very place it was issued.
        PartCols cols_a, cols_b;
        PartKeys keys_a, keys_b;
        load_cols(c, n, (uint64_t)rd * kPartRound, cols_a);
        load_cols(c, n, rd >= (int64_t)W ? (uint64_t)(rd - W) * kPartRound : n, cols_b);
        load_keys(c, cols_a, keys_a);
        // blocks completed in the previous round, written at the START of the next one (after its hash): the
        // stores then have a whole round to be acknowledged before this wave next waits on its memory counter
        uint32_t pend_b[kPartRecs], pend_from[kPartRecs], pend_to[kPartRecs];
#pragma unroll
        for (int j = 0; j < kPartRecs; j++) pend_b[j] = pend_from[j] = pend_to[j] = 0u;
        // one round: hash (r, keys), request the other set's keys and this set's columns of two rounds on
        auto round = [&](PartCols &r, PartKeys &keys, PartCols &r_next, PartKeys &keys_next, int64_t rd) {
            const uint64_t base = (uint64_t)rd * kPartRound;
            uint32_t h[kPartRecs], alive[kPartRecs];
            bool keyed[kPartRecs];
#pragma unroll
            for (int j = 0; j < kPartRecs; j++) {
                keyed[j] = r.kl[j] >= 0;         // key None: ignored (metric.rs:302)
                alive[j] = r.vl[j] >= 0 ? 1u : 0u;
#ifdef KTA_DBG_NOHASH   /* ablation build of tools/ubench_alive.hip only */
                h[j] = (keys.k16[j].x ^ keys.k16[j].y ^ keys.k16[j].z ^ keys.k16[j].w) * 0x9E3779B1u;
#else
                h[j] = r.kl[j] > 0 ? fnv32_prefetched(keys.k16[j], c.key_bytes + r.ko[j], (uint32_t)r.kl[j]) : kFnvInit;
#endif
            }
            KTA_PHASE(0, 0);   // waiting for the round's loads + hashing
            load_keys(c, r_next, keys_next);                      // their columns were requested a round ago
            load_cols(c, n, rd >= 2 * (int64_t)W ? (uint64_t)(rd - 2 * W) * kPartRound : n, r);   // r is spent: hashed
#ifdef KTA_DBG_NOLDS   /* ablation build of tools/ubench_alive.hip only: stream + hash, nothing else */
            delta += (long long)(h[0] ^ h[1] ^ h[2] ^ h[3]) + alive[0];
#ifdef KTA_DBG_BARRIERS
            __syncthreads();
            __syncthreads();
#endif
            return;
#endif
            flush_blocks<BLOG2>(s_ring, pairs, W, w, cap, pend_b, pend_from, pend_to);
            __syncthreads();   // the ring entries of the flushed blocks are free again
            // Arrivals.  Straight-line, so that the four LDS atomics (and then the four reads) of a thread
            // are in flight together.  rank = arrival order in the bucket's round; the ring holds the
            // positions [fill & ~7, (fill & ~7) + 16) of the segment, the segment holds cap pairs.
            uint32_t bk[kPartRecs], rank[kPartRecs], fill[kPartRecs], room[kPartRecs];
#pragma unroll
            for (int j = 0; j < kPartRecs; j++) bk[j] = h[j] >> (32 - BLOG2);
#pragma unroll
            for (int j = 0; j < kPartRecs; j++) rank[j] = keyed[j] ? atomicAdd(&s_cnt[bk[j]], 1u) : ~0u;
#pragma unroll
            for (int j = 0; j < kPartRecs; j++) fill[j] = s_fill[bk[j]];
            bool any_direct = false;
#pragma unroll
            for (int j = 0; j < kPartRecs; j++) {
                room[j] = (fill[j] & ~7u) + kRing - fill[j];
                if (cap - fill[j] < room[j]) room[j] = cap - fill[j];
                if (rank[j] < room[j]) {
                    const uint64_t i = base + (uint64_t)j * kPartThreads + threadIdx.x;
                    s_ring[ring_at(bk[j], fill[j] + rank[j])] =
                        ((unsigned long long)h[j] << 32) | ((unsigned long long)(i + 1) << 1) | alive[j];
                } else if (keyed[j]) {
                    any_direct = true;
                }
            }
            if (any_direct) {   // no room in the ring or the segment (rare): parked for the direct path
#pragma unroll
                for (int j = 0; j < kPartRecs; j++) {
                    if (!keyed[j] || rank[j] < room[j]) continue;
                    const uint64_t i = base + (uint64_t)j * kPartThreads + threadIdx.x;
                    const unsigned long long v = ((unsigned long long)(base_seq + i + 1) << 1) | alive[j];
                    if (!spill_push(s_spill, h[j], v)) delta += direct_update(table, h[j], v);
                }
            }
            KTA_PHASE(0, 4);   // arrivals: LDS atomics, ring writes, parking
            __syncthreads();
            // The first arrival of every bucket closes the bucket's round: accept what fitted, write the
            // completed 64-byte blocks to the segment (aligned: cap is a multiple of 8).
            uint32_t arrivals[kPartRecs];
#pragma unroll
            for (int j = 0; j < kPartRecs; j++) arrivals[j] = rank[j] == 0u ? s_cnt[bk[j]] : 0u;
#pragma unroll
            for (int j = 0; j < kPartRecs; j++) {
                if (rank[j] != 0u) continue;
                const uint32_t newf = fill[j] + (arrivals[j] < room[j] ? arrivals[j] : room[j]);
                s_fill[bk[j]] = newf;
                s_cnt[bk[j]] = 0;
                pend_b[j] = bk[j];
                pend_from[j] = fill[j] & ~7u;
                pend_to[j] = newf;
            }
            const bool drain = s_spill.n >= kSpill / 2;     // uniform: nobody pushes between two arrival phases
            if (drain) {
                __syncthreads();
                delta += spill_drain(s_spill, table);
                __syncthreads();
                if (threadIdx.x == 0) s_spill.n = 0;
                __syncthreads();
            }
            KTA_PHASE(0, 6);   // closing the buckets
        };
        for (; rd >= 0; rd -= 2 * (int64_t)W) {
            round(cols_a, keys_a, cols_b, keys_b, rd);
            if (rd < (int64_t)W) break;
            round(cols_b, keys_b, cols_a, keys_a, rd - W);
        }
        __syncthreads();
        flush_blocks<BLOG2>(s_ring, pairs, W, w, cap, pend_b, pend_from, pend_to);
    }
    __syncthreads();
    delta += spill_drain(s_spill, table);
    // the last, partial block of every segment, and the segment fills for pass 2
    for (uint32_t b = threadIdx.x; b < B; b += kPartThreads) {
        const uint32_t f = s_fill[b];
        unsigned long long *seg = pairs + ((uint64_t)b * W + w) * cap;
        for (uint32_t p = f & ~7u; p < f; p++) seg[p] = s_ring[ring_at(b, p)];
        counts[(uint64_t)b * W + w] = f;
    }
    add_running(delta, running, s_w);
    KTA_PHASE(0, 7);
}

// ------------------------------------------------------------------------------------------------------
// pass 2: per-bucket merge in LDS, then the owner's read / compare / write on its table region
// ------------------------------------------------------------------------------------------------------
constexpr int kApplyThreads = 1024;
constexpr int kApplyWaves = kApplyThreads / 64;
constexpr int kApplyUnroll = 4;                    // 16-byte loads a lane has in flight
constexpr uint32_t kMaxSegWGs = 1024;              // partition workgroups (segments per bucket) at most

// The LDS table of a workgroup: T entries in 8-way sets, as two parallel u32 arrays — tag[e] = (h << PBITS) | 1
// (the hash without the PBITS top bits all of the workgroup's pairs share; never zero, zero = free) and val[e] = the largest
// (local sequence + 1) << 1 | alive seen for that slot.  The set is chosen from the slot's table LINE
// (h >> 3: the 8 slots of one 64-byte line of the table).  A merge reads the set's 8 tags (32 bytes),
// claims a free entry with a 32-bit CAS when the slot is new, and does one 32-bit max on the value.  No
// probing beyond the set: a wave costs what its unluckiest lane costs, so every lane does the same thing.
// Returns 1 merged into an existing entry, 2 claimed a new one, 0 the set is full (the caller parks the
// record for the direct path).
// The pairs of one unit (N per lane) are merged in four phases — all set reads, then all decisions, all
// claims, all value updates — so that a lane's N LDS round trips overlap instead of queueing up behind
// each other.  ok[i]: 1 merged into an existing entry, 2 claimed a new one, 0 not merged (the set is full,
// or a record of another slot took the chosen entry in the same instant: the caller parks the record for the
// direct path).  Pairs with valid[i] false are skipped.
template <int PBITS, int TLOG2, int N>
__device__ __forceinline__ void set_merge_unit(uint32_t *s_tag, uint32_t *s_val, const uint32_t (&h)[N],
                                               const uint32_t (&lo)[N], const bool (&valid)[N], int (&ok)[N])
{
    constexpr uint32_t kSetBits = TLOG2 - 3;
    uint32_t sbase[N], tagv[N];
    uint4 t0[N], t1[N];
#pragma unroll
    for (int i = 0; i < N; i++) {
        tagv[i] = (h[i] << PBITS) | 1u;
        sbase[i] = (((h[i] >> 3) ^ (h[i] >> (3 + kSetBits))) & ((1u << kSetBits) - 1u)) << 3;
        t0[i] = *reinterpret_cast<const uint4 *>(s_tag + sbase[i]);
        t1[i] = *reinterpret_cast<const uint4 *>(s_tag + sbase[i] + 4);
    }
    uint32_t idx[N];
    bool claim[N];
#pragma unroll
    for (int i = 0; i < N; i++) {
        const uint32_t t[8] = {t0[i].x, t0[i].y, t0[i].z, t0[i].w, t1[i].x, t1[i].y, t1[i].z, t1[i].w};
        uint32_t m = 8u, e = 8u;       // the entry holding this slot, else the first free entry, else none (8)
#pragma unroll
        for (int j = 7; j >= 0; j--) {
            e = t[j] == 0u ? (uint32_t)j : e;
            m = t[j] == tagv[i] ? (uint32_t)j : m;
        }
        claim[i] = valid[i] && m == 8u && e < 8u;
        ok[i] = valid[i] && m < 8u ? 1 : 0;
        idx[i] = m < 8u ? m : e;
    }
    uint32_t old[N];
#pragma unroll
    for (int i = 0; i < N; i++) old[i] = claim[i] ? atomicCAS(&s_tag[sbase[i] + idx[i]], 0u, tagv[i]) : 1u;
#pragma unroll
    for (int i = 0; i < N; i++) {
        // the entry is this slot's if the CAS took it or another record of the same slot just did
        if (claim[i]) ok[i] = old[i] == 0u ? 2 : (old[i] == tagv[i] ? 1 : 0);
        if (ok[i]) atomicMax(&s_val[sbase[i] + idx[i]], lo[i]);
    }
}

// The survivors of the LDS table, one per slot, go to the table region this workgroup owns: it is the
// region's only writer during this kernel (its own direct-path atomics are complete: barrier), so read /
// compare / write needs no RMW atomic.  Loads and stores are agent-scope so that they see, and are seen
// by, the atomics of the direct path and of other kernels.  Leaves the LDS table empty.
// One 8-byte update of a 64-byte block is a memory-side read-modify-write at 28 G/s whichever instruction
// asks for it (tools/ubench_scatter.hip), and that rate is what this sweep runs at (8.8 k updates per bucket
// in 87 us).  Two ways around it were built, measured on the config-3 shape and dropped: (1) appending the
// survivors to a list that a third kernel applies with atomicMax on a second stream, under the next batch's
// partition kernel — the concurrent atomics slowed that kernel by more than they saved (1.61 vs 1.43 ms per
// batch); (2) whole-line updates — the sets are chosen by table line, so the first survivor of a line in
// its set can read the 64-byte line, apply the line's survivors and write it back whole (no partial
// write) — 140 us per bucket instead of 87: random 64-byte read + write pairs through L2 cost more than the
// partial writes they replace.
template <int PBITS, int TLOG2>
__device__ __forceinline__ long long sweep_table(uint32_t *s_tag, uint32_t *s_val, uint32_t prefix,
                                                 unsigned long long *__restrict__ table, uint64_t seq2)
{
    constexpr uint32_t T = 1u << TLOG2;
    constexpr int kSweep = T / kApplyThreads;
    static_assert(kSweep % 4 == 0, "sweep unroll");
    long long delta = 0;
    for (int e0 = 0; e0 < kSweep; e0 += 4) {
        uint32_t tg[4], lo[4], slot[4];
        unsigned long long old[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t e = (uint32_t)(e0 + u) * kApplyThreads + threadIdx.x;
            tg[u] = s_tag[e];
            lo[u] = s_val[e];
            s_tag[e] = 0u;
            s_val[e] = 0u;
            slot[u] = (prefix << (32 - PBITS)) | (tg[u] >> PBITS);
        }
#pragma unroll
        for (int u = 0; u < 4; u++)
            old[u] = tg[u] ? __hip_atomic_load(&table[slot[u]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : ~0ull;
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const unsigned long long v = seq2 + lo[u];
            if (tg[u] && v > old[u]) {
                __hip_atomic_store(&table[slot[u]], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                delta += (long long)(v & 1ull) - (long long)(old[u] & 1ull);
            }
        }
    }
    return delta;
}

template <int BLOG2, int TLOG2>
__global__ __launch_bounds__(kApplyThreads) void kta_alive_apply(const unsigned long long *__restrict__ pairs,
                                                                 const uint32_t *__restrict__ counts, uint32_t cap,
                                                                 uint32_t W, uint64_t base_seq,
                                                                 unsigned long long *__restrict__ table,
                                                                 long long *__restrict__ running,
                                                                 unsigned long long *__restrict__ stats)
{
    constexpr uint32_t T = 1u << TLOG2;
    KTA_PHASE_BEGIN;
    // A bucket with more distinct slots than the table takes (a batch of mostly unique keys) is applied in
    // instalments: once this many entries are claimed the survivors so far are swept out and the table
    // starts empty again — exact, because the region's entries carry their sequence numbers.
    constexpr uint32_t kFlushAt = T / 2 + T / 16 + T / 32;
    // aligned: a set's 8 tags are read as two 16-byte pieces, and the dynamic LDS starts wherever the static
    // LDS of the kernel ends — a misaligned ds_read_b128 is split by the hardware
    extern __shared__ __attribute__((aligned(128))) uint32_t s_tag[];   // T tags, T values, then the segment fills
    uint32_t *s_val = s_tag + T;
    uint32_t *s_cnt = s_val + T;
    __shared__ uint32_t s_occ, s_pairs, s_claims;
    __shared__ long long s_w[kApplyWaves];
    __shared__ SpillList s_spill;
    constexpr int PBITS = BLOG2;       // the hash bits all pairs of the bucket share
    const uint32_t b = blockIdx.x, prefix = b;
    for (uint32_t e = threadIdx.x; e < 2 * T; e += kApplyThreads) s_tag[e] = 0u;
    if (threadIdx.x == 0) {
        s_occ = 0;
        s_pairs = 0;
        s_claims = 0;
        s_spill.n = 0;
    }
    __syncthreads();
    for (uint32_t w = threadIdx.x; w < W; w += kApplyThreads) {
        const uint32_t cw = counts[(uint64_t)b * W + w];
        s_cnt[w] = cw;
        atomicAdd(&s_pairs, cw);
    }
    __syncthreads();
    KTA_PHASE(1, 0);
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    long long delta = 0;
    const uint64_t seq2 = base_seq << 1;
    // A wave reads its segments (every kApplyWaves-th of the bucket) in units of up to kApplyUnroll 16-byte
    // loads per lane = 512 pairs, masked by the segment's fill (cap is a multiple of 128).  The loads of
    // unit u + 1 are issued before unit u is merged, so a wave always has a unit in flight.
    const unsigned long long *region = pairs + (uint64_t)b * W * cap;
    const uint32_t loads = cap >> 7;
    const uint32_t chunks = (loads + kApplyUnroll - 1) / kApplyUnroll;
    const uint32_t units = ((W + kApplyWaves - 1) / kApplyWaves) * chunks;      // the same for every wave
    ulonglong2 p[kApplyUnroll], pn[kApplyUnroll];
    uint32_t nv[kApplyUnroll], nvn[kApplyUnroll];    // valid pairs of each load: 0, 1 or 2
    auto issue = [&](uint32_t u, ulonglong2 (&q)[kApplyUnroll], uint32_t (&qv)[kApplyUnroll]) {
        const uint32_t w = (u / chunks) * kApplyWaves + wave, r0 = (u % chunks) * kApplyUnroll;
        const uint32_t cnt = u < units && w < W ? s_cnt[w] : 0u;
        const unsigned long long *seg = region + (uint64_t)(u < units && w < W ? w : 0u) * cap;
#pragma unroll
        for (int x = 0; x < kApplyUnroll; x++) {
            const uint32_t k = ((r0 + (uint32_t)x) << 7) + 2u * lane;
            qv[x] = (r0 + (uint32_t)x) < loads && k < cnt ? (cnt - k >= 2u ? 2u : 1u) : 0u;
            // unconditional (clamped address, masked by qv): a predicated load is a branch, and the wait for this
            // unit's data would then be vmcnt(0) — it would wait for the NEXT unit's loads as well
            q[x] = *reinterpret_cast<const ulonglong2 *>(seg + (qv[x] ? k : 2u * lane));
        }
    };
    issue(0, p, nv);
    uint32_t claimed = 0;
    // merge the unit held in (q, qv)
    auto merge = [&](const ulonglong2 (&q)[kApplyUnroll], const uint32_t (&qv)[kApplyUnroll]) {
        uint32_t failed = 0;                                  // bit i: pair i of the unit was not merged
#pragma unroll
        for (int x = 0; x < 2 * kApplyUnroll; x++) {          // one pair after the other (batching them costs registers, gains nothing)
            const uint32_t hh[1] = {(uint32_t)((x & 1 ? q[x >> 1].y : q[x >> 1].x) >> 32)};
            const uint32_t ll[1] = {(uint32_t)(x & 1 ? q[x >> 1].y : q[x >> 1].x)};
            const bool vv[1] = {qv[x >> 1] > (uint32_t)(x & 1)};
            int ok[1];
            set_merge_unit<PBITS, TLOG2, 1>(s_tag, s_val, hh, ll, vv, ok);
            claimed += ok[0] == 2 ? 1u : 0u;
            failed |= (vv[0] && ok[0] == 0 ? 1u : 0u) << x;
        }
        if (failed) {   // once per unit: a set was full, or lost a claim to another slot's record — park for the direct path
#pragma unroll
            for (int x = 0; x < kApplyUnroll; x++) {
                const uint32_t h0 = (uint32_t)(q[x].x >> 32), l0 = (uint32_t)q[x].x;
                const uint32_t h1 = (uint32_t)(q[x].y >> 32), l1 = (uint32_t)q[x].y;
                if ((failed >> (2 * x)) & 1u)
                    if (!spill_push(s_spill, h0, seq2 + l0)) delta += direct_update(table, h0, seq2 + l0);
                if ((failed >> (2 * x)) & 2u)
                    if (!spill_push(s_spill, h1, seq2 + l1)) delta += direct_update(table, h1, seq2 + l1);
            }
        }
    };
    // every fourth unit: does the table need sweeping out, or the parked records draining?
    auto checkpoint = [&](uint32_t u) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) claimed += __shfl_xor(claimed, off);
        if (lane == 0 && claimed) atomicAdd(&s_occ, claimed);
        claimed = 0;
        __syncthreads();
        const bool flush = s_occ >= kFlushAt && u + 1 < units;      // uniform: read between two barriers
        const bool drain = s_spill.n >= kSpill / 2 || flush;        // (the direct path must not run beside a sweep)
        if (drain) {
            delta += spill_drain(s_spill, table);
            __syncthreads();
            if (threadIdx.x == 0) s_spill.n = 0;
            if (flush) {
                delta += sweep_table<PBITS, TLOG2>(s_tag, s_val, prefix, table, seq2);
                if (threadIdx.x == 0) {
                    s_claims += s_occ;
                    s_occ = 0;
                }
            }
            __syncthreads();
        }
    };
    // Two units per trip, the buffers alternating: copying the prefetched registers into the current ones at the
    // end of an iteration would make the compiler wait for the prefetch right where it was issued.
    // The prefetch is issued unconditionally (a unit past the end loads a clamped address and merges nothing):
    // under a branch the compiler could not count it and would wait with vmcnt(0).
    for (uint32_t u = 0; u < units; u += 2) {
        issue(u + 1, pn, nvn);
        merge(p, nv);
        if ((u & 3u) == 3u || u + 1 >= units) checkpoint(u);
        issue(u + 2, p, nv);
        merge(pn, nvn);
        if (((u + 1) & 3u) == 3u || u + 2 == units) checkpoint(u + 1);
    }
    delta += spill_drain(s_spill, table);
    __syncthreads();
    KTA_PHASE(1, 1);
    delta += sweep_table<PBITS, TLOG2>(s_tag, s_val, prefix, table, seq2);
    add_running(delta, running, s_w);
    // what the host's choice of kernel for the NEXT batch feeds on: pairs read, entries claimed (one per
    // distinct slot and instalment) — their ratio says how much of the batch died in LDS
    if (stats && threadIdx.x == 0) {
        atomicAdd(&stats[0], (unsigned long long)s_pairs);
        atomicAdd(&stats[1], (unsigned long long)(s_claims + s_occ));
    }
    KTA_PHASE(1, 2);
}

template <int BLOG2, int TLOG2>
hipError_t launch_pair(const AliveColumns &c, uint64_t n, uint64_t base_seq, uint64_t *table, int64_t *running,
                       const AlivePartitionPlan &pl, uint64_t *pairs, uint32_t *counts, uint64_t *stats, hipStream_t s)
{
    unsigned long long *t = reinterpret_cast<unsigned long long *>(table);
    unsigned long long *pp = reinterpret_cast<unsigned long long *>(pairs);
    long long *run = reinterpret_cast<long long *>(running);
    const size_t lds1 = ((size_t)8 * kRing + 8) << BLOG2;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&kta_alive_partition<BLOG2>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((kta_alive_partition<BLOG2>), dim3(pl.segment_wgs), dim3(kPartThreads), lds1, s, c, n, base_seq, pp,
                       counts, pl.cap, t, run);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    const size_t lds2 = ((size_t)8 << TLOG2) + (size_t)pl.segment_wgs * 4;
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(&kta_alive_apply<BLOG2, TLOG2>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((kta_alive_apply<BLOG2, TLOG2>), dim3(1u << BLOG2), dim3(kApplyThreads), lds2, s, pp, counts, pl.cap,
                       pl.segment_wgs, base_seq, t, run, reinterpret_cast<unsigned long long *>(stats));
    return hipGetLastError();
}

} // namespace

AlivePartitionPlan plan_alive_partition(uint64_t n, int bucket_log2, int req_wgs, int cu_count)
{
    AlivePartitionPlan pl;
    pl.bucket_log2 = bucket_log2 == 9 ? 9u : 10u;
    pl.max_records = kAlivePartitionMax;
    if (n > pl.max_records) n = pl.max_records;
    // one partition workgroup per CU (its rings take most of a CU's LDS); every workgroup sees n / W records
    uint64_t wgs = req_wgs > 0 ? (uint64_t)req_wgs : (uint64_t)(cu_count > 0 ? cu_count : 256);
    if (wgs > kMaxSegWGs) wgs = kMaxSegWGs;
    const uint64_t rounds = (n + kPartRound - 1) / kPartRound;
    if (wgs > rounds) wgs = rounds ? rounds : 1;
    pl.segment_wgs = (uint32_t)wgs;
    // a segment receives n / (W * B) pairs on average; 1/8 + 48 of slack (8 sigma at 2^26 records) before it
    // overflows into the direct path, rounded up to what a wave reads with one load instruction (128 pairs)
    const uint64_t mean = n / (wgs << pl.bucket_log2) + 1;
    pl.cap = (uint32_t)((mean + mean / 8 + 48 + 127) & ~127ull);
    pl.pair_words = ((uint64_t)pl.segment_wgs << pl.bucket_log2) * pl.cap;
    pl.count_words = (uint64_t)pl.segment_wgs << pl.bucket_log2;
    return pl;
}

hipError_t launch_alive_partitioned(const AliveColumns &c, uint64_t n, uint64_t base_seq, uint64_t *table,
                                    int64_t *running, const AlivePartitionPlan &pl, uint64_t *pairs, uint32_t *counts,
                                    uint64_t *stats, hipStream_t s)
{
    switch (pl.bucket_log2) {
    case 9: return launch_pair<9, 14>(c, n, base_seq, table, running, pl, pairs, counts, stats, s);
    default: return launch_pair<10, 14>(c, n, base_seq, table, running, pl, pairs, counts, stats, s);
    }
}

} // namespace kta
