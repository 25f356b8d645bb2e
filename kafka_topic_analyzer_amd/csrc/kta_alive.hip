// kta_alive.hip — the alive-key pass (gfx950): LogCompactionInMemoryMetrics::handle_message
// (/root/reference/src/metric.rs:288-305) over one batch, as two kernels that keep the random accesses on chip.
//
// What the reference leaves behind is a bit per 32-bit hash slot: the alive flag of the LAST record that hashed
// there, in consumption order (BitSet::insert / remove, metric.rs:273-280).  One memory-side atomic (or even
// one L2-missing read) per record caps a direct implementation at 20-50 G records/s, so:
//
//   pass 1  kta_alive_partition32 (bit set state) / kta_alive_partition (table state)   streams the batch once, hashes
//           every key (fnv32.rs:92-101) and appends a pair — what pass 2 needs to know of the record — to the segment
//           [bucket = top 10 hash bits][workgroup].  Workgroups take contiguous ranges of the batch, so segment w of a
//           bucket holds an older range than segment w + 1.  Pairs reach memory through a ring of two 64-byte blocks
//           per bucket in LDS and leave it in whole aligned blocks (a partial block write costs a memory-side
//           read-modify-write, 28 G/s whatever its size: tools/ubench_scatter.hip).  No workgroup barrier in the loop:
//           twelve PRODUCER waves stream, hash and insert (one returning LDS atomic, one read, one write per record),
//           four CONSUMER waves sweep the rings' counters and write completed blocks out.
//             bit set state: a pair is 4 bytes — slot in the bucket, window, alive — and the ORDER of two pairs of one
//               slot is implicit in (workgroup, window, position in the segment); a guard keeps two records of one hash
//               out of the same wave instruction (exactly, in one round).  With at most 256 partitions the same kernel
//               also does the metrics handler's work (FUSE: both handlers of kafka.rs:107-109 in one pass).
//             table state: a pair is 8 bytes — hash, batch-local index, alive: a survivor's global sequence number comes
//               from its index; a batch's seq column is checked for order on the way.
//   pass 2  kta_alive_apply       one workgroup owns one bucket, i.e. one contiguous region of the slot
//           space.  It merges the bucket's pairs in an LDS table (8-way sets of 16-bit tags chosen by the
//           slot's ADDRESS, one 16-byte read per lookup; the newest pair per slot survives), then applies
//           the survivors:
//             bit set state (in-order batches, the single-GPU default): the region of the reference's own
//               512 MiB bit set is streamed through LDS, 2 KiB per wave at a time — survivors set / clear their bit
//               with LDS atomics, the piece goes back in whole lines.  No partial writes, no 32 GiB table.
//             table state (global sequence numbers: sharded runs): table[slot] = max(table[slot],
//               ((seq + 1) << 1) | alive) by the region's only writer.
//
// Exactness never depends on sizes or luck.  A pair that finds no room in its segment goes to a pool; a
// record that finds its set full goes to a small LDS side table that is applied with the sets.  In table
// state both end in the direct path (pre-read + atomicMax, commutative).  In bitmap state a bucket that cannot
// be finished on chip (pool pairs, list overflow) is left untouched and reported to kta_alive_fallback, which
// resolves it exactly in direct-indexed passes over the bucket's pairs.  A bucket with more distinct slots
// than the LDS table takes is applied in instalments, in segment order (older ranges first).
#include "kta_kernels.h"

#include <limits.h>

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

// h ^ byte N of w in ONE instruction: gfx9's sub-dword addressing selects the byte inside the xor.  (Left to itself
// the compiler does that for byte 3 only; bytes 1 and 2 cost a shift and an and-xor each — a quarter of the chain.)
#define KTA_XOR_BYTE(N)                                                                                         \
    __device__ __forceinline__ uint32_t xor_byte##N(uint32_t h, uint32_t w)                                      \
    {                                                                                                           \
        uint32_t r;                                                                                             \
        asm("v_xor_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_" #N    \
            : "=v"(r)                                                                                           \
            : "v"(h), "v"(w));                                                                                  \
        return r;                                                                                               \
    }
KTA_XOR_BYTE(0)
KTA_XOR_BYTE(1)
KTA_XOR_BYTE(2)
KTA_XOR_BYTE(3)
#undef KTA_XOR_BYTE

__device__ __forceinline__ uint32_t fnv_word(uint32_t h, uint32_t w)
{
    h = xor_byte0(h, w) * kFnvMul;
    h = xor_byte1(h, w) * kFnvMul;
    h = xor_byte2(h, w) * kFnvMul;
    return xor_byte3(h, w) * kFnvMul;
}

__device__ __forceinline__ uint32_t fnv_16(uint32_t h, const uint4 &v)
{
    return fnv_word(fnv_word(fnv_word(fnv_word(h, v.x), v.y), v.z), v.w);
}

// Four 16-byte keys at once, their chains interleaved (one chain is 32 dependent instructions).
__device__ __forceinline__ void fnv_16x4(uint32_t (&h)[4], const uint4 (&k)[4])
{
    const uint32_t w[4][4] = {{k[0].x, k[0].y, k[0].z, k[0].w}, {k[1].x, k[1].y, k[1].z, k[1].w},
                              {k[2].x, k[2].y, k[2].z, k[2].w}, {k[3].x, k[3].y, k[3].z, k[3].w}};
#pragma unroll
    for (int j = 0; j < 4; j++) h[j] = kFnvInit;
#pragma unroll
    for (int d = 0; d < 4; d++) {
#pragma unroll
        for (int j = 0; j < 4; j++) h[j] = xor_byte0(h[j], w[j][d]) * kFnvMul;
#pragma unroll
        for (int j = 0; j < 4; j++) h[j] = xor_byte1(h[j], w[j][d]) * kFnvMul;
#pragma unroll
        for (int j = 0; j < 4; j++) h[j] = xor_byte2(h[j], w[j][d]) * kFnvMul;
#pragma unroll
        for (int j = 0; j < 4; j++) h[j] = xor_byte3(h[j], w[j][d]) * kFnvMul;
    }
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

// FNV of a key whose first 16 bytes are in registers
__device__ __forceinline__ uint32_t fnv32_prefetched(const uint4 &k16, const uint8_t *key, uint32_t len)
{
    if (len == 16u) return fnv_16(kFnvInit, k16);
    if (len > 16u) return fnv32_more(fnv_16(kFnvInit, k16), key + 16, len - 16u);
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

// The direct path of the table state: what kta_alive_update_filtered does for one record.
__device__ __forceinline__ long long direct_update(unsigned long long *table, uint32_t h, unsigned long long v, const WrittenList &wl)
{
    const unsigned long long seen = __hip_atomic_load(&table[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (seen >= v) return 0;
    const unsigned long long old = atomicMax(&table[h], v);
    note_new_slot(wl, old == 0ull, h);
    return v > old ? (long long)(v & 1ull) - (long long)(old & 1ull) : 0;
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

// LDS operations of one wave are performed in program order; what has to be prevented is the COMPILER moving
// an LDS access across the atomic that publishes (or acquires) it.  A release / acquire atomic would do that
// too, but it also waits for the wave's outstanding global loads (vmcnt(0)) — the prefetches.
#define KTA_LDS_ORDER() asm volatile("" ::: "memory")

// Workgroup barrier for LDS state only.  __syncthreads() is a fence over all memory: the compiler waits for the
// wave's outstanding global loads (vmcnt(0)) — the prefetches that are meant to stay in flight across it.
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

__device__ __forceinline__ uint32_t lds_add(uint32_t *p, uint32_t v)
{
    return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}


// ------------------------------------------------------------------------------------------------------
// pass 1: hash + partition with software write combining: producer waves insert, consumer waves write out
// ------------------------------------------------------------------------------------------------------
constexpr int kPartThreads = 1024;                 // one workgroup per CU: the rings take most of the LDS
constexpr int kPartWaves = kPartThreads / 64;
constexpr int kConsumers = 4;                      // waves that move completed blocks from the rings to memory (one per SIMD)
constexpr int kProducers = kPartWaves - kConsumers; // waves that stream, hash and insert
constexpr uint32_t kTile = 256;                    // records of one wave step: instruction j of it takes the records 64 j + lane
static_assert(kConsumers >= 1 && kConsumers <= 8 && kProducers >= 1, "waves of the partition workgroup");

typedef uint32_t v4u __attribute__((ext_vector_type(4)));
typedef unsigned long long v2ull __attribute__((ext_vector_type(2)));

// pool control words (device memory, zeroed before every launch pair)
enum : uint32_t { POOL_CURSOR = 0, POOL_FAILED = 1, POOL_DENSE = 2, POOL_RANGES = 3, POOL_WORDS = 4 };

// ------------------------------------------------------------------------------------------------------
// pass 1, bit set state: the same partition with 4-byte pairs whose order is implicit
// ------------------------------------------------------------------------------------------------------
// What pass 1 costs is the scattered 64-byte stores of the pairs, not the LDS protocol and not the hash (measured:
// stream + hash 1.26 ms, + protocol 1.42 ms, + stores 2.34 ms at 2^28 records with 8-byte pairs; 1.76 ms with half
// the bytes).  An 8-byte pair is half index.  The bit set state needs no index — only, per slot, WHICH pair is the
// newest — so its pairs carry none:
//
//   pair32 = slot-in-bucket (22 bits) << 10 | window (1 ... 255: never zero) << 1 | alive          (bit 9 is zero)
//
// and the order of two pairs of one slot is the order of (workgroup w, window, position in the segment):
//   * workgroup w takes a contiguous range of the batch (older than w + 1's), cut into windows of consecutive tiles
//     (window c is older than c + 1); a window is walked by ONE producer wave, in order;
//   * the positions of ONE wave's pairs in a ring are handed out in program order (LDS operations of a wave are
//     performed in order) — tile after tile, and inside a tile instruction j (records 64 j + lane) before j + 1;
//   * inside one instruction the order of two lanes' atomics on one counter is the hardware's business, so no two
//     records of one hash are ever inserted by the same instruction: every record first writes its lane number to a
//     small per-wave guard table (bytes) at an index taken from its hash and reads it back (LDS operations of a wave
//     are performed in order: the read sees the instruction's last writer, and instruction j + 1 writes after j has
//     read).  A record that reads its own lane is alone at its index, or it is the last writer there; a record that
//     reads another lane shares the index with that lane — by chance (different hashes: 64 records over 1024 indices,
//     about two such pairs per instruction) or because the hashes are equal.  For every record that lost its index
//     the wave looks at the lanes with that record's very hash (one readlane + one compare per group, scalar
//     bookkeeping): the NEWEST of them (the highest lane: lane order is record order) stays, the others are
//     superseded and DROPPED — exactly what the reference's insert / remove sequence leaves (metric.rs:289-304).
//     Records with equal hashes always share their index, so every group with two or more members has a loser and is
//     looked at: after the guard no two records of one instruction have the same hash.  One round, exact.
// Hot keys — which used to fill their bucket's ring and stall the workgroup — mostly die in the guard.
// The stream is read once: non-temporal loads (measured against plain ones: the same time, within the noise).
#define KTA_P32_LOAD(p) __builtin_nontemporal_load(p)
// ... at a 32-bit byte offset from a base that lives in scalar registers
template <typename T>
__device__ __forceinline__ T ld_nt(const T *base, uint32_t byte_off)
{
    return __builtin_nontemporal_load(reinterpret_cast<const T *>(reinterpret_cast<const char *>(base) + byte_off));
}
typedef uint32_t v4u_any __attribute__((ext_vector_type(4), aligned(1)));   // 16 key bytes at any address (unaligned access mode)
constexpr uint32_t kBlk32 = 16;                    // pairs of a block: what leaves the ring in one piece (64 bytes)
constexpr uint32_t kRing32 = 2 * kBlk32;           // pairs per bucket ring: two blocks
constexpr uint32_t kBlkLanes = kBlk32 / 4;         // lanes that move a block (16 bytes each)
constexpr uint32_t kBlkPerTrip = 64 / kBlkLanes;   // blocks a consumer wave moves at a time
constexpr uint32_t kPoolChunk = 64;                // pool blocks a consumer wave takes with one device atomic (pool_tag_note)
// (the block stores as non-temporal or write-through stores measured the same as plain ones: ± 0.03 ms of 1.9)
constexpr uint32_t kGuard = 1024;                  // guard bytes (the lane that wrote last) per producer wave
constexpr uint32_t kPair32Shift = 10;

__device__ __forceinline__ uint32_t ring32_at(uint32_t b, uint32_t p)   // (rows rotated by whole 16-byte pieces)
{
    return b * kRing32 + ((p + 4u * (b & 7u)) & (kRing32 - 1));
}

// a pool pair of the bit set state (segment overflow; they are ordered by their place in the pool, see the fallback)
__device__ __forceinline__ unsigned long long pool_pair32(uint32_t bucket, uint32_t p32, uint32_t w, uint32_t RBITS)
{
    const uint32_t h = (bucket << RBITS) | (p32 >> kPair32Shift);
    return ((unsigned long long)h << 32) | 0x80000000u | (w << 9) | (p32 & 0x1FFu);   // w, window, alive
}

#define KTA_LDS(T, off) (reinterpret_cast<__attribute__((address_space(3))) T *>(static_cast<uintptr_t>(off)))
__device__ __forceinline__ uint32_t lds_offset(const void *p) { return (uint32_t)reinterpret_cast<uintptr_t>(p); }   // (the low half of an LDS address is its offset)
// The hot-key filter of a producer's step (see there), out of line.  guard_off: the wave's guard table (LDS byte offset); h0..h3:
// the hashes of the lane's records of the instructions 0..3 of the tile; keyed: bit j = record j takes part.  Returns the
// records that stay (bits 0..3) and, from bit 8, how many records of the wave were dropped (the same on every lane).
__device__ __noinline__ uint32_t hot_filter(uint32_t guard_off, uint32_t lane, uint32_t h0, uint32_t h1, uint32_t h2, uint32_t h3, uint32_t keyed)
{
    const uint32_t h[4] = {h0, h1, h2, h3};
#pragma unroll
    for (int j = 0; j < 4; j++) {
        if ((keyed >> j) & 1u) *KTA_LDS(uint8_t, guard_off + (h[j] & (kGuard - 1))) = (uint8_t)(lane | ((uint32_t)j << 6));
        KTA_LDS_ORDER();
    }
    uint32_t last[3];
#pragma unroll
    for (int j = 0; j < 3; j++) last[j] = *KTA_LDS(const volatile uint8_t, guard_off + (h[j] & (kGuard - 1)));
    KTA_LDS_ORDER();
    uint32_t dropped = 0;
#pragma unroll
    for (int j = 0; j < 3; j++) {
        const uint32_t jl = last[j] >> 6;
        const int at = (int)((last[j] & 63u) << 2);
        uint32_t hh = 0;
#pragma unroll
        for (int j2 = j + 1; j2 < 4; j2++) {
            const uint32_t c = (uint32_t)__builtin_amdgcn_ds_bpermute(at, (int)h[j2]);
            hh = jl == (uint32_t)j2 ? c : hh;
        }
        const bool gone = ((keyed >> j) & 1u) && jl > (uint32_t)j && hh == h[j];
        dropped += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(gone));
        keyed &= gone ? ~(1u << j) : ~0u;
    }
    return keyed | (dropped << 8);
}

// ---- both handlers in one pass over the batch (kafka.rs:107-109 calls every handler for every message) ----
// With FUSE the producers also read partition and ts_ms (12 B more per record: 40 instead of 28 + 20 for two kernels) and do
// what kta_metrics_scan does (MessageMetrics::handle_message, metric.rs:207-252), in the scan's own terms: per partition
// three LDS words — A += 1 | tombstone << 21 | key None << 42, K += key length, V += value length, replicated by lane as far
// as 12 KiB go (fuse_replicas_log2) — and the global extrema in registers.  The workgroup writes one row of the scan's partial
// workspace, which kta_fold_partials folds as it folds the scan's.  A workgroup takes less than 2^21 records (the
// host checks), so the 21-bit counts cannot overflow.
struct FuseArgs {
    const int32_t *partition;
    const int64_t *ts_ms;
    uint32_t P;                  // <= kFuseMaxP
    uint32_t rep_log2;           // replicas of a partition's sums in LDS: (P << rep_log2) <= kFuseSlots (fuse_replicas)
    uint64_t *partials;          // rows of row_len words, one per workgroup
    uint32_t row_len;
};
constexpr uint32_t kFuseMaxP = 256;
constexpr uint32_t kFuseSlots = 512;    // partition x replica slots of three 8-byte sums: 12 KiB
// A wave instruction's 64 records that share a partition serialise on its three sums (64 partitions: four or five records
// deep on average, and rounds 4-5 paid for it: the fused form cost 0.54 ms of a 2.3 ms pass on config 3's 64 partitions) — so
// the sums are replicated by lane, like the scan's (kta_kernels.hip): slot = partition << rep_log2 | lane & (replicas - 1).
inline uint32_t fuse_replicas_log2(uint32_t P)
{
    uint32_t r = 0;
    while (r < 4u && ((P ? P : 1u) << (r + 1u)) <= kFuseSlots) r++;
    return r;
}
constexpr uint32_t kFuseCntBits = 21;   // (the scan's packing: kta_kernels.hip)

// FUSE, after the workgroup's last barrier: its row of the scan's partial workspace (kta_fold_partials) from the sums in
// s_acc and the waves' extrema in s_red ([kPartWaves][5]: min / max timestamp, min / max size, bad-partition records).
__device__ __forceinline__ void fuse_write_row(const FuseArgs &fz, uint32_t w, const unsigned long long *s_acc, const long long *s_red)
{
    uint64_t *row = fz.partials + (uint64_t)w * fz.row_len;
    constexpr unsigned long long kMask = (1ull << kFuseCntBits) - 1ull;
    for (uint32_t p = threadIdx.x; p < fz.P; p += kPartThreads) {
        uint64_t t[5] = {0, 0, 0, 0, 0};
        for (uint32_t r = 0; r < (1u << fz.rep_log2); r++) {
            const unsigned long long *sl = s_acc + 3u * ((p << fz.rep_log2) | r);
            const unsigned long long a = sl[0];
            t[0] += a & kMask, t[1] += (a >> kFuseCntBits) & kMask, t[2] += (a >> (2 * kFuseCntBits)) & kMask;
            t[3] += sl[1], t[4] += sl[2];
        }
        uint64_t *o = row + (uint64_t)p * kScanCols;
        o[0] = t[0], o[1] = t[1], o[2] = t[2], o[3] = t[3], o[4] = t[4];
    }
    if (threadIdx.x == 0) {
        long long tmin = LLONG_MAX, tmax = LLONG_MIN, smin = 0xFFFFFFFFll, smax = 0, bad = 0;
        for (uint32_t v = 0; v < (uint32_t)kPartWaves; v++) {
            const long long *o = s_red + v * 5u;
            tmin = o[0] < tmin ? o[0] : tmin;
            tmax = o[1] > tmax ? o[1] : tmax;
            smin = o[2] < smin ? o[2] : smin;
            smax = o[3] > smax ? o[3] : smax;
            bad += o[4];
        }
        uint64_t *g = row + (uint64_t)fz.P * kScanCols;
        g[SG_TMIN] = (uint64_t)tmin;
        g[SG_TMAX] = (uint64_t)tmax;
        g[SG_SMIN] = smin == 0xFFFFFFFFll ? (uint64_t)LLONG_MAX : (uint64_t)smin;     // (no record with a payload: as the scan says it)
        g[SG_SMAX] = (uint64_t)smax;
        g[SG_BAD] = (uint64_t)bad;
        g[SG_NREC] = 0;
        g[6] = 0;
        g[7] = 0;
    }
}

template <int BLOG2, bool FUSE>
__global__ __launch_bounds__(kPartThreads) void kta_alive_partition32(AliveColumns c, uint64_t n, uint32_t tiles_per_wg,
                                                                      uint32_t *__restrict__ pairs, uint32_t *__restrict__ counts,
                                                                      uint32_t cap, unsigned long long *__restrict__ pool,
                                                                      uint64_t pool_pairs, unsigned long long *__restrict__ pool_ctl,
                                                                      uint32_t *__restrict__ pool_hist, FuseArgs fz)
{
    constexpr uint32_t B = 1u << BLOG2;
    constexpr uint32_t RBITS = 32 - BLOG2;
    uint32_t *pool_tags = reinterpret_cast<uint32_t *>(pool + pool_pairs);        // per pool block of 16 pairs: bucket + 1, or 0 (pool_tag_note)
    static_assert(RBITS == 32 - kPair32Shift, "a pair32 holds the hash bits below the bucket");
    static_assert(B % (64u * kConsumers) == 0, "a consumer's buckets are whole lanes-of-64 chunks");
    extern __shared__ __attribute__((aligned(128))) uint32_t s_ring32[];           // B x kRing32 pairs
    uint32_t *s_pos = s_ring32 + (size_t)B * kRing32;                              // positions handed out
    unsigned short *s_out = reinterpret_cast<unsigned short *>(s_pos + B);         // pairs written out, modulo 2^16 (its consumer's)
    uint8_t *s_guard = reinterpret_cast<uint8_t *>(s_out + B);                     // kProducers x kGuard
    uint32_t *s_list = reinterpret_cast<uint32_t *>(s_guard + (size_t)kProducers * kGuard);   // kConsumers x 64
    uint32_t *s_misc = s_list + kConsumers * 64;                                   // [0] producers that are done (20 words)
    unsigned long long *s_acc = reinterpret_cast<unsigned long long *>(s_misc + 20);      // FUSE: A, K, V per partition
    long long *s_red = reinterpret_cast<long long *>(s_acc + 3 * kFuseSlots);             // FUSE: [kPartWaves][5] extrema of the waves
    {
        uint4 *z = reinterpret_cast<uint4 *>(s_ring32);
        for (uint32_t e = threadIdx.x; e < B * kRing32 / 4u; e += kPartThreads) z[e] = make_uint4(0u, 0u, 0u, 0u);
        for (uint32_t b = threadIdx.x; b < B; b += kPartThreads) {
            s_pos[b] = 0u;
            s_out[b] = 0;
        }
        if (threadIdx.x < 2) s_misc[threadIdx.x] = 0u;
        if (FUSE)
            for (uint32_t e = threadIdx.x; e < 3 * kFuseSlots; e += kPartThreads) s_acc[e] = 0ull;
    }
    // FUSE: the lane's share of the global extrema (metric.rs:56-72) and of the records outside [0, P)
    long long f_tmin = LLONG_MAX, f_tmax = LLONG_MIN;
    uint32_t f_smin = 0xFFFFFFFFu, f_smax = 0u, f_bad = 0u;   // (0xFFFFFFFF is no size: both lengths are below 2^31)
    __syncthreads();
    const uint32_t W = gridDim.x, w = blockIdx.x;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint64_t ntiles = (n + kTile - 1) / kTile;
    const uint64_t first = (uint64_t)w * tiles_per_wg;
    const uint64_t end = first + tiles_per_wg < ntiles ? first + tiles_per_wg : ntiles;

    if (wave < (uint32_t)kProducers) {
        // ---------------------------------------------- producer ----------------------------------------------
        const uint32_t v = wave;
        uint8_t *guard = s_guard + v * kGuard;
        // instruction j of a step handles the records 64 j + lane of the tile: 4-byte loads, 256 bytes per
        // instruction and column.  Unconditional (index clamped into the batch, result masked), two steps ahead; the
        // key bytes — whose addresses come from the columns — one step ahead; two register sets alternate.
        struct Cols {
            int32_t kl[4], vl[4];
            uint32_t ko[4];
            uint32_t win;                                  // the tile's window (wave-uniform); 0: no tile
            int32_t pt[FUSE ? 4 : 1];                      // FUSE: partition (-1: no record), timestamp
            long long ts[FUSE ? 4 : 1];
        };
        auto load_cols32 = [&](uint64_t tile, bool ok, Cols &r) __attribute__((always_inline)) {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const uint64_t i = tile * kTile + 64u * j + lane;
                const bool in = ok && i < n;
                const uint64_t ic = in ? i : n - 1;
                r.kl[j] = KTA_P32_LOAD(c.key_len + ic);
                r.vl[j] = KTA_P32_LOAD(c.val_len + ic);
                r.ko[j] = KTA_P32_LOAD(c.key_off + ic);
                r.kl[j] = in ? r.kl[j] : -1;               // key None: ignored (metric.rs:302)
                if (FUSE) {
                    r.pt[j] = KTA_P32_LOAD(fz.partition + ic);
                    r.ts[j] = KTA_P32_LOAD(fz.ts_ms + ic);
                    r.vl[j] = in ? r.vl[j] : 0;
                    r.pt[j] = in ? r.pt[j] : -2;           // no record here (a record's bad id stays what it is)
                }
            }
        };
        auto load_keys32 = [&](const Cols &r, uint4 (&k)[4]) __attribute__((always_inline)) {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const v4u_any kk = KTA_P32_LOAD(reinterpret_cast<const v4u_any *>(c.key_bytes + (r.kl[j] > 0 ? r.ko[j] : 0u)));
                k[j] = make_uint4(kk.x, kk.y, kk.z, kk.w);
            }
        };
        // The workgroup's range is cut into (at most 255) windows of consecutive tiles, and the waves take them as they
        // come (a counter in LDS; from a start that differs from workgroup to workgroup, and around: the ranges lie a
        // power of two apart, and in step all workgroups would ask the same few memory channels).  A window is
        // walked by ONE wave, in order, and its number — its place in the range, not the order of the taking — is
        // what the pairs carry.
        const uint64_t span = end > first ? end - first : 0u;
        const uint32_t wtiles = (uint32_t)((span + 254u) / 255u);                 // tiles per window
        const uint32_t nwin = wtiles ? (uint32_t)((span + wtiles - 1u) / wtiles) : 0u;
        const uint32_t woff = nwin ? (w * 37u) % nwin : 0u;
        uint64_t cur_tile = 0, cur_stop = 0;                                     // the walk's cursor (wave-uniform)
        uint32_t cur_win = 0;
        auto next_tile = [&](uint64_t &tile, uint32_t &win) __attribute__((always_inline)) {   // win = 0: the range is used up
            if (cur_tile + 1 < cur_stop) {
                cur_tile++;
            } else {
                uint32_t g = 0;
                if (lane == 0) g = lds_add(&s_misc[1], 1u);
                g = __builtin_amdgcn_readfirstlane(g);
                if (g < nwin) {
                    const uint32_t cw = g + woff < nwin ? g + woff : g + woff - nwin;
                    cur_tile = first + (uint64_t)cw * wtiles;
                    cur_stop = cur_tile + wtiles < end ? cur_tile + wtiles : end;
                    cur_win = cw + 1u;
                } else {
                    cur_win = 0u;
                    cur_stop = 0u;
                    cur_tile = 0u;
                }
            }
            tile = cur_tile;
            win = cur_win;
        };
        {
            Cols cols_a, cols_b;
            uint4 keys_a[4], keys_b[4];
            uint64_t tl;
            next_tile(tl, cols_a.win);
            load_cols32(tl, cols_a.win != 0u, cols_a);
            next_tile(tl, cols_b.win);
            load_cols32(tl, cols_b.win != 0u, cols_b);
            load_keys32(cols_a, keys_a);
            bool hot = false;                                        // (wave-uniform) the last step dropped records: hot keys about
            auto step = [&](Cols &r, uint4 (&keys)[4], Cols &r_next, uint4 (&keys_next)[4]) __attribute__((always_inline)) {
                uint32_t h[4];
                if (__all(r.kl[0] == 16 && r.kl[1] == 16 && r.kl[2] == 16 && r.kl[3] == 16)) {
                    fnv_16x4(h, keys);
                } else {
#pragma unroll
                    for (int j = 0; j < 4; j++)
                        h[j] = r.kl[j] > 0 ? fnv32_prefetched(keys[j], c.key_bytes + r.ko[j], (uint32_t)r.kl[j]) : kFnvInit;
                }
                if (FUSE) {
                    // MessageMetrics::handle_message (metric.rs:207-252) for the tile's records, keyed or not
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const bool there = r.pt[j] != -2;
                        const bool ok = there && (uint32_t)r.pt[j] < fz.P;                 // (unsigned: negative ids are out as well)
                        const uint32_t tomb = (uint32_t)r.vl[j] >> 31, knull = (uint32_t)r.kl[j] >> 31;   // payload None / key None (metric.rs:227-244)
                        const uint32_t ks = knull ? 0u : (uint32_t)r.kl[j], vs = tomb ? 0u : (uint32_t)r.vl[j];
                        const long long t = r.ts[j] == -1ll ? 0ll : r.ts[j];               // to_millis() None -> unwrap_or(0) (metric.rs:209)
                        f_bad += there && !ok ? 1u : 0u;
                        if (ok) {
                            f_tmin = t < f_tmin ? t : f_tmin;
                            f_tmax = t > f_tmax ? t : f_tmax;
                            if (!tomb) {                                                   // metric.rs:249-251
                                f_smin = min(f_smin, ks + vs);
                                f_smax = max(f_smax, ks + vs);
                            }
                            unsigned long long *a = s_acc + 3u * (((uint32_t)r.pt[j] << fz.rep_log2) | (lane & ((1u << fz.rep_log2) - 1u)));
                            atomicAdd(a, 1ull | ((unsigned long long)tomb << kFuseCntBits) | ((unsigned long long)knull << (2 * kFuseCntBits)));
                            atomicAdd(a + 1, (unsigned long long)ks);
                            atomicAdd(a + 2, (unsigned long long)vs);
                        }
                    }
                }
                uint32_t pr[4];
                bool keyed[4];
                const uint32_t vv = r.win;                           // the window: what the order sees of the tile's place
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    keyed[j] = r.kl[j] >= 0;
                    pr[j] = (h[j] << kPair32Shift) | (vv << 1) | (r.vl[j] >= 0 ? 1u : 0u);
                }
                load_keys32(r_next, keys_next);                      // their columns were requested a step ago
                {
                    uint64_t tn;
                    next_tile(tn, r.win);
                    load_cols32(tn, r.win != 0u, r);                 // r is spent: hashed
                }
                // ---- hot keys: a record whose hash comes again in a LATER instruction of this tile is superseded by that
                // record — same wave, same window, a later position — and need not be inserted at all.  Only while the guard
                // below has been dropping records (a compacted topic's instructions hold no two records of one key): the
                // records write (instruction, lane) into the guard table in program order, read the last writer at their
                // index back and fetch ITS hash (ds_bpermute); an equal hash from a later instruction drops the record.  What
                // shares an index by chance is kept: the filter only ever drops what is certainly superseded.  With 40 hot
                // keys a tile of 256 records leaves 40 pairs instead of about 130.
                uint32_t dropped = 0;                                // (wave-uniform) records this step did not insert
                if (hot) {                                           // (out of line: inlined, it cost the ordinary path 3 % of pass 1)
                    const uint32_t km = (keyed[0] ? 1u : 0u) | (keyed[1] ? 2u : 0u) | (keyed[2] ? 4u : 0u) | (keyed[3] ? 8u : 0u);
                    const uint32_t res = hot_filter(lds_offset(guard), lane, h[0], h[1], h[2], h[3], km);
#pragma unroll
                    for (int j = 0; j < 4; j++) keyed[j] = (res >> j) & 1u;
                    dropped = (uint32_t)__builtin_amdgcn_readfirstlane((int)(res >> 8));
                }
                // ---- the guard: among the records of one instruction, one record per hash (the newest) ----
                uint32_t seen[4];
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    uint8_t *g = guard + (h[j] & (kGuard - 1));      // (the low bits: any function of the hash will do)
                    if (keyed[j]) *g = (uint8_t)lane;
                    KTA_LDS_ORDER();
                    seen[j] = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    KTA_LDS_ORDER();
                }
                bool ins[4];
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    unsigned long long todo = __builtin_amdgcn_ballot_w64(keyed[j] && seen[j] != lane);   // lost their index to another lane
                    unsigned long long drop = 0;
                    while (todo) {                                   // (wave-uniform) one trip per group of equal hashes
                        const uint32_t hl = (uint32_t)__builtin_amdgcn_readlane((int)h[j], (int)__builtin_ctzll(todo));
                        const unsigned long long grp = __builtin_amdgcn_ballot_w64(keyed[j] && h[j] == hl);
                        drop |= grp & ~(0x8000000000000000ull >> __builtin_clzll(grp));   // all but the highest lane
                        todo &= ~grp;
                    }
                    ins[j] = keyed[j] && !((drop >> lane) & 1ull);
                    dropped += (uint32_t)__popcll(drop);
                }
                hot = dropped >= 8u;                                 // (of 256: a tile of distinct keys drops none)
                // ---- positions, then the pairs once their ring entries are free: straight-line, so that a lane's four
                // LDS atomics and its four reads are in flight together ----
                uint32_t bk[4], p[4], out[4];
#pragma unroll
                for (int j = 0; j < 4; j++) bk[j] = h[j] >> RBITS;
#pragma unroll
                for (int j = 0; j < 4; j++) p[j] = ins[j] ? lds_add(&s_pos[bk[j]], 1u) : 0u;
#pragma unroll
                for (int j = 0; j < 4; j++) out[j] = __hip_atomic_load(&s_out[bk[j]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                KTA_LDS_ORDER();
                uint32_t pending = 0;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    if (ins[j]) {
                        // (`out` may have been read before the position was handed out: it only grows, so an old
                        // reading errs on the side of waiting; differences are taken modulo 2^16)
                        if (((p[j] - out[j]) & 0xFFFFu) < kRing32) s_ring32[ring32_at(bk[j], p[j])] = pr[j];
                        else pending |= 1u << j;
                    }
                }
                while (__any(pending != 0u)) {                        // rare: 32 arrivals of one bucket since its last block left
                    __builtin_amdgcn_s_sleep(2);
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        if (!((pending >> j) & 1u)) continue;
                        const uint32_t o = __hip_atomic_load(&s_out[bk[j]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        if (((p[j] - o) & 0xFFFFu) >= kRing32) continue;
                        KTA_LDS_ORDER();
                        s_ring32[ring32_at(bk[j], p[j])] = pr[j];
                        pending &= ~(1u << j);
                    }
                }
            };
            while (cols_a.win != 0u) {
                step(cols_a, keys_a, cols_b, keys_b);
                if (cols_b.win == 0u) break;
                step(cols_b, keys_b, cols_a, keys_a);
            }
        }
        KTA_LDS_ORDER();
        if (lane == 0) lds_add(&s_misc[0], 1u);            // (LDS operations of a wave are performed in order: its pairs are in)
    } else {
        // ---------------------------------------------- consumer ----------------------------------------------
        constexpr uint32_t kChunks = B / (64u * kConsumers);
        const uint32_t cw = wave - (uint32_t)kProducers;
        uint32_t *list = s_list + cw * 64u;
        // pool_tag_note.  Blocks of a full segment go to the pool, with what a pair32 leaves implicit spelled out.  The pool is
        // handed out in CHUNKS of kPoolChunk blocks per consumer wave (one device atomic per chunk: with one per block, a
        // batch of 40 hot keys spent 12 ms of pass 1 on 1.8 M atomics on ONE address), and every block of 16 pairs carries a
        // tag — its bucket + 1 — so that kta_alive_fallback, which resolves a bucket, reads the tags (4 bytes per block) and
        // only its own blocks (it read the whole pool, for every sub-range: 25 ms for those 40 keys).  Every block below the
        // cursor gets a tag: what a wave leaves of its last chunk is tagged 0.  A bucket belongs to ONE consumer wave, whose
        // pool positions ascend in time: among a segment's pool pairs the pool index still grows with the position.
        uint32_t pc_next = 0, pc_end = 0;                                         // this wave's chunk, in blocks (wave-uniform)
        for (;;) {
            const uint32_t done = __hip_atomic_load(&s_misc[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // BEFORE the sweep
            KTA_LDS_ORDER();
            bool any_ready = false;                                               // wave-uniform
            for (uint32_t ch = 0; ch < kChunks; ch++) {
                const uint32_t b = (cw * kChunks + ch) * 64u + lane;
                const uint32_t pos = __hip_atomic_load(&s_pos[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                const uint32_t have = (pos - s_out[b]) & 0xFFFFu;                  // handed out and not yet written out
                const bool ready = have >= kBlk32;                                    // every position of the oldest block is handed out
                const unsigned long long m = __ballot(ready);
                if (m == 0ull) continue;
                any_ready = true;
                const uint32_t nready = (uint32_t)__popcll(m);
                if (ready) list[__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] = b | (((pos - have) / kBlk32) << BLOG2);
                KTA_LDS_ORDER();
                for (uint32_t g0 = 0; g0 < nready; g0 += kBlkPerTrip) {             // sixteen blocks at a time, four lanes per block
                    const uint32_t e = g0 + lane / kBlkLanes, piece = lane % kBlkLanes;
                    const bool on = e < nready;
                    const uint32_t ent = list[on ? e : 0u];
                    const uint32_t bb = ent & (B - 1), k = ent >> BLOG2;           // bucket, block number
                    uint4 *src = reinterpret_cast<uint4 *>(s_ring32 + ring32_at(bb, k * kBlk32 + piece * 4u));
                    const uint4 d = *src;
                    KTA_LDS_ORDER();
                    const unsigned long long vm = __ballot(on && d.x != 0u && d.y != 0u && d.z != 0u && d.w != 0u);
                    const bool go = ((uint32_t)(vm >> (lane & ~(kBlkLanes - 1u))) & ((1u << kBlkLanes) - 1u)) == (1u << kBlkLanes) - 1u;   // all its pairs have arrived
                    // the blocks of this trip whose segment is full: their places in the wave's chunk of the pool (pool_tag_note)
                    const bool to_pool = go && (k + 1u) * kBlk32 > cap;
                    const unsigned long long pm = __builtin_amdgcn_ballot_w64(to_pool && piece == 0u);
                    uint32_t pool_blk = 0;                                         // this block's place (its four lanes agree)
                    if (pm != 0ull) {                                              // (wave-uniform)
                        const uint32_t cnt = (uint32_t)__popcll(pm), room = pc_end - pc_next;
                        const uint32_t rank = (uint32_t)__popcll(pm & ((1ull << (lane & ~(kBlkLanes - 1u))) - 1ull));
                        if (cnt > room) {                                          // what is left of the chunk, then a new one
                            unsigned long long base = 0;
                            if (lane == 0u) base = atomicAdd(&pool_ctl[POOL_CURSOR], (unsigned long long)kPoolChunk * kBlk32);
                            const uint32_t nb = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(base / kBlk32));
                            pool_blk = rank < room ? pc_next + rank : nb + (rank - room);
                            pc_next = nb + (cnt - room);
                            pc_end = nb + kPoolChunk;
                        } else {
                            pool_blk = pc_next + rank;
                            pc_next += cnt;
                        }
                    }
                    if (go) {
                        if (!to_pool) {
                            const uint64_t at = ((uint64_t)bb * W + w) * cap + (uint64_t)k * kBlk32 + piece * 4u;
                            *reinterpret_cast<v4u *>(pairs + at) = (v4u){d.x, d.y, d.z, d.w};
                        } else {
                            unsigned long long *dst = pool + (uint64_t)pool_blk * kBlk32 + piece * 4u;
                            *reinterpret_cast<v2ull *>(dst) = (v2ull){pool_pair32(bb, d.x, w, RBITS), pool_pair32(bb, d.y, w, RBITS)};
                            *reinterpret_cast<v2ull *>(dst + 2) = (v2ull){pool_pair32(bb, d.z, w, RBITS), pool_pair32(bb, d.w, w, RBITS)};
                            if (piece == 0u) {
                                pool_tags[pool_blk] = bb + 1u;
                                pool_hist[bb] = 1u;                                // (a flag: pass 2 leaves the bucket to the fallback kernel)
                            }
                        }
                        *src = make_uint4(0u, 0u, 0u, 0u);
                        KTA_LDS_ORDER();
                        if (piece == 0u) s_out[bb] = (unsigned short)((k + 1u) * kBlk32);   // after the zeroes (program order); its only writer
                    }
                }
                KTA_LDS_ORDER();
            }
            if (!any_ready) {
                if (done == (uint32_t)kProducers) break;                          // nothing left that is complete
                __builtin_amdgcn_s_sleep(8);
            }
        }
        for (uint32_t blk = pc_next + lane; blk < pc_end; blk += 64u) pool_tags[blk] = 0u;   // what is left of the last chunk holds nothing
    }
    if (FUSE) {                                        // the waves' extrema (the consumers' are the neutral elements)
        long long tmin = f_tmin, tmax = f_tmax, smin = (long long)f_smin, smax = (long long)f_smax, bad = (long long)f_bad;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const long long a = __shfl_xor(tmin, off), b2 = __shfl_xor(tmax, off), c2 = __shfl_xor(smin, off), d2 = __shfl_xor(smax, off);
            bad += __shfl_xor(bad, off);
            tmin = a < tmin ? a : tmin;
            tmax = b2 > tmax ? b2 : tmax;
            smin = c2 < smin ? c2 : smin;
            smax = d2 > smax ? d2 : smax;
        }
        if (lane == 0) {
            long long *o = s_red + wave * 5u;
            o[0] = tmin, o[1] = tmax, o[2] = smin, o[3] = smax, o[4] = bad;
        }
    }
    __syncthreads();
    if (FUSE) fuse_write_row(fz, w, s_acc, s_red);
    // the last, partial block of every segment, and the segment fills for pass 2
    for (uint32_t b = threadIdx.x; b < B; b += kPartThreads) {
        const uint32_t f = s_pos[b], k = f / kBlk32, rem = f % kBlk32;
        if (rem) {
            if ((k + 1u) * kBlk32 <= cap) {
                // the WHOLE block, its unwritten entries zero (a zero pair is no pair, and pass 2 reads a segment as far as
                // its fill): four 16-byte stores of one line — entry by entry these were 2 M four-byte writes per batch,
                // each a read-modify-write at the memory's side (round 6's WRITE_SIZE: 6.5 % over the pairs' bytes)
                uint4 *dst = reinterpret_cast<uint4 *>(pairs + ((uint64_t)b * W + w) * cap + (uint64_t)k * kBlk32);
#pragma unroll
                for (uint32_t q = 0; q < kBlk32 / 4u; q++) dst[q] = *reinterpret_cast<const uint4 *>(s_ring32 + ring32_at(b, k * kBlk32 + 4u * q));
            } else {
                // a whole pool block, padded (a zero pair is no pair); behind everything the consumers wrote (pool_tag_note)
                const unsigned long long at = atomicAdd(&pool_ctl[POOL_CURSOR], (unsigned long long)kBlk32);
                unsigned long long *dst = pool + at;
                for (uint32_t q = 0; q < kBlk32; q++) dst[q] = q < rem ? pool_pair32(b, s_ring32[ring32_at(b, k * kBlk32 + q)], w, RBITS) : 0ull;
                pool_tags[at / kBlk32] = b + 1u;
                pool_hist[b] = 1u;
            }
        }
        counts[(uint64_t)b * W + w] = f < cap ? f : cap;
    }
}

// ------------------------------------------------------------------------------------------------------
// pass 1, table state: 6-byte pairs with the order spelled out
// ------------------------------------------------------------------------------------------------------
// The table keeps ((global sequence number + 1) << 1) | alive per slot, so pass 2 needs a survivor's place in the batch
// (its sequence number is base_seq + index, or seq[index]).  Rounds 1-5 spelled the whole batch-local index out: 8-byte
// pairs, 2 GB written and read again per 2^28 records — and the kernel read its columns 16 bytes per lane, four
// consecutive records each, with a register budget (119 of 128) that left no room for the metrics handler.  Now:
//
//   * the stream, the window walk and the FUSE arm are kta_alive_partition32's (4-byte column loads, lane j of an
//     instruction = one record; both handlers of kafka.rs:107-109 in the one pass for a sharded rank as well);
//   * pair48 = slot in the bucket (22 bits), alive, and the record's index INSIDE THE WORKGROUP'S RANGE (22 bits: a
//     workgroup takes at most 2^22 records, the plan sees to it) — the workgroup is the pair's segment, so the batch-local
//     index is segment x range + index.  The order of two pairs of one slot is their index: no windows, no guard;
//   * a pair is three 16-bit words, each with bit 15 set:
//         word 0 = index bits 0..14        word 1 = slot bits 0..14        word 2 = slot bits 15..21 | alive << 7 | index bits 15..21 << 8
//     Bit 15 says "arrived": a ring word is zero from the moment its block leaves until its next tenant writes it, and the
//     three words of a pair are three LDS writes;
//   * a segment is a DENSE stream of 6-byte pairs.  A bucket's ring is 128 bytes = two 64-byte blocks = 21 1/3 pairs: the
//     words of position p are the ring's words (3 p + i) mod 64, so a pair may straddle two blocks (and the ring's end).
//     A block leaves — one aligned 64-byte store, as ever — when every position that reaches into it is handed out and
//     all its 32 words have arrived; `out` counts the BYTES written out, and position p may be written once
//     6 p + 6 - out <= 128 (every ring word it takes has been zeroed: the consumer zeroes a block before it advances `out`);
//   * a position at or behind the segment's capacity never enters the ring: the producer writes the pair, in the pool's
//     8-byte form with the batch-local index spelled out, straight to the pool (kta_alive_pool_direct takes it from there:
//     pre-read + atomicMax, commutative).  The pool is handed out to the producer waves in chunks (one device atomic per
//     chunk; what a wave leaves of its last chunk is zeroed: a zero pair is no pair).
// SEQ: the batch carries a seq column, which has to ascend inside the batch for pass 2 (it orders a batch's records by their
// index): the producers read it next to the other columns and raise *order_flag, which pass 2 and what runs in its stead read.
constexpr uint32_t kRingW48 = 64;                  // 16-bit words of a bucket's ring: two 64-byte blocks
constexpr uint32_t kBlkW48 = 32;                   // words of a block
constexpr uint32_t kIdxBits48 = 22;                // index inside the workgroup's range
constexpr uint32_t kPoolChunk48 = 2048;            // pool pairs a producer wave takes with one device atomic

__device__ __forceinline__ uint32_t ring48_at(uint32_t b, uint32_t word)   // (rows rotated by whole 16-byte pieces)
{
    return b * kRingW48 + ((word + 8u * (b & 7u)) & (kRingW48 - 1));
}

template <int BLOG2, bool SEQ, bool FUSE>
__global__ __launch_bounds__(kPartThreads) void kta_alive_partition48(AliveColumns c, uint64_t n, uint32_t tiles_per_wg,
                                                                      unsigned short *__restrict__ pairs, uint32_t *__restrict__ counts,
                                                                      uint32_t cap, unsigned long long *__restrict__ pool,
                                                                      unsigned long long *__restrict__ pool_ctl,
                                                                      uint32_t *__restrict__ order_flag, FuseArgs fz)
{
    constexpr uint32_t B = 1u << BLOG2;
    constexpr uint32_t RBITS = 32 - BLOG2;
    static_assert(RBITS == 22, "a pair48 holds 22 hash bits below the bucket");
    static_assert(B % (64u * kConsumers) == 0, "a consumer's buckets are whole lanes-of-64 chunks");
    extern __shared__ __attribute__((aligned(128))) unsigned short s_ring48[];    // B x kRingW48 words
    uint2 *s_ctl = reinterpret_cast<uint2 *>(s_ring48 + (size_t)B * kRingW48);     // per bucket: x = positions handed out, y = bytes written out
    uint32_t *s_list = reinterpret_cast<uint32_t *>(s_ctl + B);                    // kConsumers x 64: the chunk's ready blocks
    uint32_t *s_misc = s_list + kConsumers * 64;                                   // [0] producers that are done, [1] next window (4 words)
    unsigned long long *s_acc = reinterpret_cast<unsigned long long *>(s_misc + 4);       // FUSE: A, K, V per partition
    long long *s_red = reinterpret_cast<long long *>(s_acc + 3 * kFuseSlots);             // FUSE: [kPartWaves][5] extrema of the waves
    {
        uint4 *z = reinterpret_cast<uint4 *>(s_ring48);
        for (uint32_t e = threadIdx.x; e < B * kRingW48 / 8u; e += kPartThreads) z[e] = make_uint4(0u, 0u, 0u, 0u);
        for (uint32_t b = threadIdx.x; b < B; b += kPartThreads) s_ctl[b] = make_uint2(0u, 0u);
        if (threadIdx.x < 2) s_misc[threadIdx.x] = 0u;
        if (FUSE)
            for (uint32_t e = threadIdx.x; e < 3 * kFuseSlots; e += kPartThreads) s_acc[e] = 0ull;
    }
    // FUSE: the lane's share of the global extrema (metric.rs:56-72) and of the records outside [0, P)
    long long f_tmin = LLONG_MAX, f_tmax = LLONG_MIN;
    uint32_t f_smin = 0xFFFFFFFFu, f_smax = 0u, f_bad = 0u;   // (0xFFFFFFFF is no size: both lengths are below 2^31)
    __syncthreads();
    const uint32_t W = gridDim.x, w = blockIdx.x;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint64_t ntiles = (n + kTile - 1) / kTile;
    const uint64_t first = (uint64_t)w * tiles_per_wg;
    const uint64_t end = first + tiles_per_wg < ntiles ? first + tiles_per_wg : ntiles;
    unsigned short *seg_base = pairs + (uint64_t)w * cap * 3u;                     // + bucket * W * cap * 3: the segment's words

    if (wave < (uint32_t)kProducers) {
        // ---------------------------------------------- producer ----------------------------------------------
        // The batch has at most 2^28 records (kAlivePartitionMax), so every column offset fits 32 bits: the loads take the
        // column's base from scalar registers and a 32-bit byte offset per lane (a 64-bit index per lane costs two
        // registers and two additions per load).
        struct Cols {                                      // two sets: requested two steps ahead
            int32_t kl[4], vl[4];
            uint32_t ko[4];
            uint32_t on;                                   // a tile (wave-uniform); 0: none
            uint32_t at;                                   // the batch-local index of the tile's first record
        };
        const uint32_t nn = (uint32_t)n, first_rec = (uint32_t)(first * kTile);
        auto load_cols48 = [&](uint64_t tile, bool ok, Cols &r) __attribute__((always_inline)) {
            r.at = ok ? (uint32_t)(tile * kTile) : 0u;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const uint32_t i = r.at + 64u * (uint32_t)j + lane;
                const bool in = ok && i < nn;
                const uint32_t ic = (in ? i : nn - 1u) * 4u;
                r.kl[j] = ld_nt(c.key_len, ic);
                r.vl[j] = ld_nt(c.val_len, ic);
                r.ko[j] = ld_nt(c.key_off, ic);
                r.kl[j] = in ? r.kl[j] : -1;               // key None: ignored (metric.rs:302)
                if (FUSE) r.vl[j] = in ? r.vl[j] : 0;
            }
        };
        auto load_keys48 = [&](const Cols &r, uint4 (&k)[4]) __attribute__((always_inline)) {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const v4u_any kk = KTA_P32_LOAD(reinterpret_cast<const v4u_any *>(c.key_bytes + (r.kl[j] > 0 ? r.ko[j] : 0u)));
                k[j] = make_uint4(kk.x, kk.y, kk.z, kk.w);
            }
        };
        // What only the head of a tile's step reads — FUSE: partition (-2: no record) and timestamp — lives in ONE register
        // set, requested with the tile's key bytes (a step ahead) and spent before the next tile's are requested.
        int32_t x_pt[FUSE ? 4 : 1];
        long long x_ts[FUSE ? 4 : 1];
        auto load_extra48 = [&](const Cols &r) __attribute__((always_inline)) {
            if (FUSE) {
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const uint32_t i = r.at + 64u * (uint32_t)j + lane;
                    const bool in = r.on != 0u && i < nn;
                    const uint32_t ic = in ? i : nn - 1u;
                    x_pt[j] = ld_nt(fz.partition, ic * 4u);
                    x_ts[j] = ld_nt(fz.ts_ms, ic * 8u);
                    x_pt[j] = in ? x_pt[j] : -2;           // no record here (a record's bad id stays what it is)
                }
            }
        };
        // the walk: kta_alive_partition32's — windows of consecutive tiles taken from a counter, from a start that differs from
        // workgroup to workgroup (here a window is only the unit of work: the pairs carry their index)
        const uint64_t span = end > first ? end - first : 0u;
        const uint32_t wtiles = (uint32_t)((span + 254u) / 255u);                 // tiles per window
        const uint32_t nwin = wtiles ? (uint32_t)((span + wtiles - 1u) / wtiles) : 0u;
        const uint32_t woff = nwin ? (w * 37u) % nwin : 0u;
        uint64_t cur_tile = 0, cur_stop = 0;                                     // the walk's cursor (wave-uniform)
        uint32_t cur_on = 0;
        auto next_tile = [&](uint64_t &tile, uint32_t &on) __attribute__((always_inline)) {   // on = 0: the range is used up
            if (cur_tile + 1 < cur_stop) {
                cur_tile++;
            } else {
                uint32_t g = 0;
                if (lane == 0) g = lds_add(&s_misc[1], 1u);
                g = __builtin_amdgcn_readfirstlane(g);
                if (g < nwin) {
                    const uint32_t cw = g + woff < nwin ? g + woff : g + woff - nwin;
                    cur_tile = first + (uint64_t)cw * wtiles;
                    cur_stop = cur_tile + wtiles < end ? cur_tile + wtiles : end;
                    cur_on = 1u;
                } else {
                    cur_on = 0u;
                    cur_stop = 0u;
                    cur_tile = 0u;
                }
            }
            tile = cur_tile;
            on = cur_on;
        };
        uint32_t pc_next = 0, pc_end = 0;                  // this wave's chunk of the pool, in pairs (wave-uniform)
        {
            Cols cols_a, cols_b;
            uint4 keys_a[4], keys_b[4];
            uint64_t tl;
            next_tile(tl, cols_a.on);
            load_cols48(tl, cols_a.on != 0u, cols_a);
            next_tile(tl, cols_b.on);
            load_cols48(tl, cols_b.on != 0u, cols_b);
            load_keys48(cols_a, keys_a);
            load_extra48(cols_a);
            auto step = [&](Cols &r, uint4 (&keys)[4], Cols &r_next, uint4 (&keys_next)[4]) __attribute__((always_inline)) {
                uint32_t h[4];
                if (__all(r.kl[0] == 16 && r.kl[1] == 16 && r.kl[2] == 16 && r.kl[3] == 16)) {
                    fnv_16x4(h, keys);
                } else {
#pragma unroll
                    for (int j = 0; j < 4; j++)
                        h[j] = r.kl[j] > 0 ? fnv32_prefetched(keys[j], c.key_bytes + r.ko[j], (uint32_t)r.kl[j]) : kFnvInit;
                }
                if (FUSE) {
                    // MessageMetrics::handle_message (metric.rs:207-252) for the tile's records, keyed or not
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const bool there = x_pt[j] != -2;
                        const bool ok = there && (uint32_t)x_pt[j] < fz.P;                 // (unsigned: negative ids are out as well)
                        const uint32_t tomb = (uint32_t)r.vl[j] >> 31, knull = (uint32_t)r.kl[j] >> 31;   // payload None / key None (metric.rs:227-244)
                        const uint32_t ks = knull ? 0u : (uint32_t)r.kl[j], vs = tomb ? 0u : (uint32_t)r.vl[j];
                        const long long t = x_ts[j] == -1ll ? 0ll : x_ts[j];               // to_millis() None -> unwrap_or(0) (metric.rs:209)
                        f_bad += there && !ok ? 1u : 0u;
                        if (ok) {
                            f_tmin = t < f_tmin ? t : f_tmin;
                            f_tmax = t > f_tmax ? t : f_tmax;
                            if (!tomb) {                                                   // metric.rs:249-251
                                f_smin = min(f_smin, ks + vs);
                                f_smax = max(f_smax, ks + vs);
                            }
                            unsigned long long *a = s_acc + 3u * (((uint32_t)x_pt[j] << fz.rep_log2) | (lane & ((1u << fz.rep_log2) - 1u)));
                            atomicAdd(a, 1ull | ((unsigned long long)tomb << kFuseCntBits) | ((unsigned long long)knull << (2 * kFuseCntBits)));
                            atomicAdd(a + 1, (unsigned long long)ks);
                            atomicAdd(a + 2, (unsigned long long)vs);
                        }
                    }
                }
                // what the inserts need of the tile, in few registers (the columns' are about to be requested again)
                const uint32_t at = r.at - first_rec + lane;         // the index, inside the workgroup's range, of the lane's first record
                uint32_t alive_m = 0, ins_m = 0;                     // bit j: record j has a payload / a key
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    alive_m |= (r.vl[j] >= 0 ? 1u : 0u) << j;
                    ins_m |= (r.kl[j] >= 0 ? 1u : 0u) << j;
                }
                load_keys48(r_next, keys_next);                      // their columns were requested a step ago
                load_extra48(r_next);
                {
                    uint64_t tn;
                    next_tile(tn, r.on);
                    load_cols48(tn, r.on != 0u, r);                  // r is spent: hashed
                }
                // ---- positions, then the pairs once their ring words are free: straight-line, so that a lane's four
                // LDS atomics and its four reads are in flight together ----
                uint32_t bk[4], p[4], out[4];
                bool ins[4];
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    bk[j] = h[j] >> RBITS;
                    ins[j] = (ins_m >> j) & 1u;
                }
#pragma unroll
                for (int j = 0; j < 4; j++) p[j] = ins[j] ? lds_add(&s_ctl[bk[j]].x, 1u) : 0u;
#pragma unroll
                for (int j = 0; j < 4; j++) out[j] = __hip_atomic_load(&s_ctl[bk[j]].y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                KTA_LDS_ORDER();
                auto put = [&](int j) __attribute__((always_inline)) {   // the pair's three words
                    const uint32_t q = 3u * p[j], idx = at + 64u * (uint32_t)j, slot = h[j] & ((1u << RBITS) - 1u);
                    s_ring48[ring48_at(bk[j], q)] = (unsigned short)(0x8000u | (idx & 0x7FFFu));
                    s_ring48[ring48_at(bk[j], q + 1u)] = (unsigned short)(0x8000u | (slot & 0x7FFFu));
                    s_ring48[ring48_at(bk[j], q + 2u)] = (unsigned short)(0x8000u | (slot >> 15) | (((alive_m >> j) & 1u) << 7) | ((idx >> 15) << 8));
                };
                uint32_t pending = 0, over = 0;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    if (ins[j]) {
                        // (`out` may have been read before the position was handed out: it only grows, so an old
                        // reading errs on the side of waiting)
                        if (p[j] >= cap) over |= 1u << j;
                        else if (6u * p[j] + 6u - out[j] <= 2u * kRingW48) put(j);
                        else pending |= 1u << j;
                    }
                }
                while (__any(pending != 0u)) {                        // rare: 21 arrivals of one bucket since its last block left
                    __builtin_amdgcn_s_sleep(2);
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        if (!((pending >> j) & 1u)) continue;
                        const uint32_t o = __hip_atomic_load(&s_ctl[bk[j]].y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        if (6u * p[j] + 6u - o > 2u * kRingW48) continue;
                        KTA_LDS_ORDER();
                        put(j);
                        pending &= ~(1u << j);
                    }
                }
                // ---- rare: pairs whose segment is full go to the pool, spelled out (hot keys, or a bucket 8 sigma over its share)
                if (__any(over != 0u)) {
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const bool o = (over >> j) & 1u;
                        const unsigned long long m = __builtin_amdgcn_ballot_w64(o);
                        if (m == 0ull) continue;                     // (wave-uniform)
                        const uint32_t cnt = (uint32_t)__popcll(m), room = pc_end - pc_next;
                        const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                        uint32_t at_pool;
                        if (cnt > room) {                            // what is left of the chunk, then a new one
                            unsigned long long nb = 0;
                            if (lane == 0u) nb = atomicAdd(&pool_ctl[POOL_CURSOR], (unsigned long long)kPoolChunk48);
                            const uint32_t nb0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)nb);
                            at_pool = rank < room ? pc_next + rank : nb0 + (rank - room);
                            pc_next = nb0 + (cnt - room);
                            pc_end = nb0 + kPoolChunk48;
                        } else {
                            at_pool = pc_next + rank;
                            pc_next += cnt;
                        }
                        // the pool's form of a pair: hash << 32 | (batch-local index + 1) << 1 | alive
                        if (o) pool[at_pool] = ((unsigned long long)h[j] << 32) | (((unsigned long long)first_rec + at + 64u * (uint32_t)j + 1ull) << 1) | ((alive_m >> j) & 1u);
                    }
                }
            };
            while (cols_a.on != 0u) {
                step(cols_a, keys_a, cols_b, keys_b);
                if (cols_b.on == 0u) break;
                step(cols_b, keys_b, cols_a, keys_a);
            }
        }
        for (uint32_t k = pc_next + lane; k < pc_end; k += 64u) pool[k] = 0ull;   // what is left of the last chunk holds nothing
        KTA_LDS_ORDER();
        if (lane == 0) lds_add(&s_misc[0], 1u);            // (LDS operations of a wave are performed in order: its pairs are in)
    } else {
        // ---------------------------------------------- consumer ----------------------------------------------
        constexpr uint32_t kChunks = B / (64u * kConsumers);
        const uint32_t cw = wave - (uint32_t)kProducers;
        uint32_t *list = s_list + cw * 64u;
        // SEQ: the consumer waves, which have registers and time to spare (a block of a bucket completes every 19 us, a sweep
        // takes a fraction of one), also read the workgroup's range of the seq column — a quarter each, kSeqRows rows of 64
        // records per sweep, requested before the sweep and looked at before the next one: every pair of neighbours of the
        // column has to ascend.  Lane l of row t holds record 64 t + l (one instruction per 512 bytes: every line is asked
        // for once — non-temporal loads of overlapping pieces fetched their lines again and again); its successor is the
        // next lane's, the next row's first, or — the last row's last — the record behind the rows (which may be the next
        // quarter's, or the next workgroup's, first).  Every index is clamped into the column, what is clamped is masked, all
        // loads are unconditional.  (In the producers the column cost ten registers that the fused form does not have.)
        const uint32_t nn = (uint32_t)n;                   // (at most 2^28 records: kAlivePartitionMax)
        constexpr uint32_t kSeqRows = 24;                  // (12 KiB on the way per consumer wave; 8, 24 and 40 rows measured the same)
        const uint32_t q_tiles = (uint32_t)((end > first ? end - first : 0u) + kConsumers - 1u) / kConsumers;
        const uint32_t sq_lo = ((uint32_t)first + cw * q_tiles) * kTile;                       // the quarter's first record
        const uint32_t sq_stop_t = (uint32_t)first + (cw + 1u) * q_tiles < (uint32_t)end ? (uint32_t)first + (cw + 1u) * q_tiles : (uint32_t)end;
        const uint32_t sq_stop = sq_stop_t * kTile < nn ? sq_stop_t * kTile : nn;              // (the quarter's end, inside the batch)
        // The quarter in chunks of kSeqRows rows, each complete in itself (it reads its own successor), from a chunk that
        // differs from wave to wave, and around: the quarters lie a power of two apart, and in step all 1024 consumer waves of
        // the chip would ask the same few memory channels at every moment (measured: 2.8 instead of 2.3 ms for the pass).
        const uint32_t sq_chunks = sq_lo < sq_stop ? (sq_stop - sq_lo + 64u * kSeqRows - 1u) / (64u * kSeqRows) : 0u;
        const uint32_t sq_rot = sq_chunks ? ((w * (uint32_t)kConsumers + cw) * 37u) % sq_chunks : 0u;
        uint32_t sq_k = 0;                                 // chunks requested so far
        unsigned long long sq[SEQ ? kSeqRows : 1], sq_edge = 0;
        uint32_t sq_base = 0;                              // the rows in the registers begin at this record ...
        bool sq_have = false;                              // ... if there are any (wave-uniform)
        bool disorder = false;
        for (;;) {
            const uint32_t done = __hip_atomic_load(&s_misc[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // BEFORE the sweep
            KTA_LDS_ORDER();
            if (SEQ) {
                if (sq_have) {
#pragma unroll
                    for (uint32_t t = 0; t < kSeqRows; t++) {
                        const unsigned long long mine = sq[t];
                        unsigned long long next = __shfl_down(mine, 1);
                        const unsigned long long row0 = t + 1 < kSeqRows
                            ? ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(sq[t + 1 < kSeqRows ? t + 1 : t] >> 32), 0) << 32) |
                                  (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)sq[t + 1 < kSeqRows ? t + 1 : t], 0)
                            : sq_edge;
                        next = lane == 63u ? row0 : next;
                        const uint32_t i = sq_base + 64u * t + lane;                           // this record; i + 1: its successor
                        disorder |= i < sq_stop && i + 1u < nn && mine >= next;
                    }
                }
                sq_have = sq_k < sq_chunks;
                {
                    const uint32_t ck = sq_k + sq_rot < sq_chunks ? sq_k + sq_rot : sq_k + sq_rot - sq_chunks;
                    sq_base = sq_have ? sq_lo + ck * (64u * kSeqRows) : 0u;
                }
#pragma unroll
                for (uint32_t t = 0; t < kSeqRows; t++) {
                    const uint32_t i = sq_base + 64u * t + lane;
                    sq[t] = ld_nt(reinterpret_cast<const unsigned long long *>(c.seq), (i < nn ? i : nn - 1u) * 8u);
                }
                {
                    const uint32_t i = sq_base + 64u * kSeqRows;
                    sq_edge = ld_nt(reinterpret_cast<const unsigned long long *>(c.seq), (i < nn ? i : nn - 1u) * 8u);
                }
                sq_k += sq_have ? 1u : 0u;
            }
            bool any_ready = false;                                               // wave-uniform
            for (uint32_t ch = 0; ch < kChunks; ch++) {
                const uint32_t b = (cw * kChunks + ch) * 64u + lane;
                const unsigned long long ctl = __hip_atomic_load(reinterpret_cast<unsigned long long *>(&s_ctl[b]), __ATOMIC_RELAXED,
                                                                 __HIP_MEMORY_SCOPE_WORKGROUP);
                const uint32_t pos = (uint32_t)ctl, out = (uint32_t)(ctl >> 32);
                const uint32_t lim = pos < cap ? pos : cap;                        // (what lies behind the capacity went to the pool)
                const bool ready = 6u * lim - out >= 2u * kBlkW48;                 // every position that reaches into block out / 64 is handed out
                const unsigned long long m = __ballot(ready);
                if (m == 0ull) continue;
                any_ready = true;
                const uint32_t nready = (uint32_t)__popcll(m);
                if (ready) list[__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] = b | ((out >> 6) << BLOG2);
                KTA_LDS_ORDER();
                for (uint32_t g0 = 0; g0 < nready; g0 += 16u) {                    // sixteen blocks at a time, four lanes per block
                    const uint32_t e = g0 + (lane >> 2), piece = lane & 3u;
                    const bool on = e < nready;
                    const uint32_t ent = list[on ? e : 0u];
                    const uint32_t bb = ent & (B - 1), k = ent >> BLOG2;           // bucket, block number
                    uint4 *src = reinterpret_cast<uint4 *>(s_ring48 + ring48_at(bb, (k & 1u) * kBlkW48 + piece * 8u));
                    const uint4 d = *src;
                    KTA_LDS_ORDER();
                    const unsigned long long vm = __ballot(on && ((d.x & d.y & d.z & d.w) & 0x80008000u) == 0x80008000u);
                    const bool go = ((uint32_t)(vm >> (lane & ~3u)) & 15u) == 15u;  // all its 32 words have arrived
                    if (go) {
                        unsigned short *dst = seg_base + (uint64_t)bb * W * cap * 3u + (uint64_t)k * kBlkW48 + piece * 8u;
                        *reinterpret_cast<v4u *>(dst) = (v4u){d.x, d.y, d.z, d.w};
                        *src = make_uint4(0u, 0u, 0u, 0u);
                        KTA_LDS_ORDER();
                        if (piece == 0u) s_ctl[bb].y = (k + 1u) * 64u;            // after the zeroes (program order); its only writer
                    }
                }
                KTA_LDS_ORDER();
            }
            if (!any_ready) {
                if (done == (uint32_t)kProducers && (!SEQ || !sq_have)) break;   // nothing left that is complete (SEQ: or unread)
                if (done != (uint32_t)kProducers) __builtin_amdgcn_s_sleep(8);
            }
        }
        if (SEQ && __any(disorder) && lane == 0) atomicOr(order_flag, 1u);
    }
    if (FUSE) {                                        // the waves' extrema (the consumers' are the neutral elements)
        long long tmin = f_tmin, tmax = f_tmax, smin = (long long)f_smin, smax = (long long)f_smax, bad = (long long)f_bad;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const long long a = __shfl_xor(tmin, off), b2 = __shfl_xor(tmax, off), c2 = __shfl_xor(smin, off), d2 = __shfl_xor(smax, off);
            bad += __shfl_xor(bad, off);
            tmin = a < tmin ? a : tmin;
            tmax = b2 > tmax ? b2 : tmax;
            smin = c2 < smin ? c2 : smin;
            smax = d2 > smax ? d2 : smax;
        }
        if (lane == 0) {
            long long *o = s_red + wave * 5u;
            o[0] = tmin, o[1] = tmax, o[2] = smin, o[3] = smax, o[4] = bad;
        }
    }
    __syncthreads();
    if (FUSE) fuse_write_row(fz, w, s_acc, s_red);
    // the last, partial block of every segment, and the segment fills for pass 2
    for (uint32_t b = threadIdx.x; b < B; b += kPartThreads) {
        const uint32_t pos = s_ctl[b].x, out = s_ctl[b].y;
        const uint32_t lim = pos < cap ? pos : cap;
        // (less than a block of words is still in the ring: the WHOLE block goes out, its unwritten words zero — pass 2 reads a
        // segment as far as its fill —, four 16-byte stores of one line instead of up to 31 two-byte ones)
        if (6u * lim != out) {
            uint4 *dst = reinterpret_cast<uint4 *>(seg_base + (uint64_t)b * W * cap * 3u + out / 2u);
#pragma unroll
            for (uint32_t q = 0; q < 4u; q++) dst[q] = *reinterpret_cast<const uint4 *>(s_ring48 + ring48_at(b, out / 2u + 8u * q));
        }
        counts[(uint64_t)b * W + w] = lim;
    }
}

// ------------------------------------------------------------------------------------------------------
// pass 2: per-bucket merge in LDS, then the survivors go to the bucket's region
// ------------------------------------------------------------------------------------------------------
constexpr int kApplyThreads = 1024;
constexpr int kApplyWaves = kApplyThreads / 64;
constexpr int kApplyUnroll = 2;                    // 16-byte loads of a unit: 256 pairs, 4 per lane
constexpr uint32_t kMaxSegWGs = 1024;              // partition workgroups (segments per bucket) at most
constexpr uint32_t kSetLog2 = 11;                  // 2048 sets of 8 entries = 16384 entries
constexpr uint32_t kSets = 1u << kSetLog2;
constexpr uint32_t kEntries = kSets * 8u;
constexpr uint32_t kOvfLog2 = 11;
constexpr uint32_t kOvf = 1u << kOvfLog2;          // side table for the slots that found their set full
constexpr uint32_t kOvfSubLog2 = kOvfLog2 - 4;      // one sub-table per slice of 128 sets
constexpr uint32_t kOvfSub = 1u << kOvfSubLog2;
constexpr uint32_t kOvfProbes = 32;                // linear probes before the side table counts as full
constexpr uint32_t kSliceSets = 128;               // sets of one bitmap slice
constexpr uint32_t kNoFail = 0xFFFFFFFFu;
constexpr uint32_t kMissQueue = 320;               // pairs a wave queues for the long way of the merge
constexpr int kApplyDepth = 4;                     // units of a wave's walk: one being merged, the others in flight (2, 3, 4: the same time)

// One lookup: which of the set's eight 16-bit tags equals `tag` (8 = none).  t = the set's 16 bytes.  Entry i < 4 is
// the low half of dword i, entry i >= 4 the high half of dword i - 4 (the order a search finds them in: of two
// matching entries the lower number wins — tags are unique in a set, the order matters for tag 0 = "first free
// entry", which all racers for one slot must agree on); the entry's value is s_val[8 set + i].
// x = t ^ tag:tag has a zero half where a tag matches; v_pk_min_u16(x, 1:1) turns every half into "differs" (1 / 0) —
// two operations per dword; the compiler, asked for a packed minimum, produces eight 16-bit compares + selects +
// permutes, hence the asm.  Then the eight flags are gathered into one byte and the lowest clear bit is the entry.
__device__ __forceinline__ uint32_t pk_differs(uint32_t x)
{
    uint32_t r;
    asm("v_pk_min_u16 %0, %1, %2" : "=v"(r) : "v"(x), "s"(0x00010001u));
    return r;
}
__device__ __forceinline__ uint32_t find_tag(const uint4 &t, uint32_t tag)
{
    const uint32_t rep = tag * 0x10001u;
    const uint32_t m0 = pk_differs(t.x ^ rep), m1 = pk_differs(t.y ^ rep), m2 = pk_differs(t.z ^ rep), m3 = pk_differs(t.w ^ rep);
    const uint32_t comb = m0 | (m1 << 1) | (m2 << 2) | (m3 << 3);   // bits 0..3: low halves of dwords 0..3 differ; 16..19: high halves
    return (uint32_t)__builtin_ctz(~(comb | (comb >> 12)));         // (bit 8 of the argument is always set: 8 = none)
}
// where entry i of a set lies: the dword of the set's 16 tag bytes, and the shift of its half
__device__ __forceinline__ uint32_t entry_dword(uint32_t i) { return i & 3u; }
__device__ __forceinline__ uint32_t entry_shift(uint32_t i) { return (i >> 2) * 16u; }
// the entry number of the tag at half-word position ph (0..7) of a set
__device__ __forceinline__ uint32_t entry_of_half(uint32_t ph) { return ((ph & 1u) << 2) | (ph >> 1); }

// The long way of the merge — a pair that found no entry for its slot — out of line: it is taken by whole waves of
// queued pairs, from several places, and inlined everywhere it made the kernel three times its size (and what the
// scalar registers could not hold went through v_writelane in the hot loop).  A function has no idea of the kernel's
// LDS layout, so the arrays come as LDS byte offsets and are addressed through address-space pointers (a generic
// pointer would make every access a flat_ instruction, which counts on both memory counters).
//
// A new slot takes the set's first free entry with a compare-and-swap on the dword that holds it.  All racers for
// one slot pick the same entry (first free), so exactly one wins and the others find its tag when they look again:
// a slot never holds two entries.  A slot whose set is full goes to the side table — open addressing over 64-bit
// entries (slot + 1) << 32 | value, claimed with a compare-and-swap, the value kept with a 64-bit max (same slot,
// same upper half); one sub-table of kOvf / 16 entries per slice of 128 sets (the wave that sweeps a slice of the
// bit set reads its sub-table into registers).
// Returns 0: merged into an entry that was there; 1: merged into an entry it claimed; 2: merged into the side table;
// 3: the side table had no room either (the caller gives the attempt up, or takes the direct path).
__device__ __noinline__ uint32_t merge_new_slot(uint32_t lds_tag, uint32_t lds_val, uint32_t lds_ovf, uint32_t lds_ovf_n, uint32_t lds_any_ovf,
                                               uint32_t tagbits, uint32_t h, uint32_t lo)
{
    const uint32_t set = h >> tagbits, tag = (h & ((1u << tagbits) - 1u)) + 1u;
    uint32_t e, ret = 0;
    for (;;) {
        KTA_LDS_ORDER();
        const v4u raw = *KTA_LDS(const v4u, lds_tag + set * 16u);
        const uint4 tt = make_uint4(raw.x, raw.y, raw.z, raw.w);
        KTA_LDS_ORDER();
        e = find_tag(tt, tag);
        if (e < 8u) break;
        const uint32_t f = find_tag(tt, 0u);
        if (f == 8u) break;                           // the set is full
        const uint32_t d = entry_dword(f);
        uint32_t old = d == 0u ? tt.x : (d == 1u ? tt.y : (d == 2u ? tt.z : tt.w));
        if (__hip_atomic_compare_exchange_strong(KTA_LDS(uint32_t, lds_tag + set * 16u + d * 4u), &old, old | (tag << entry_shift(f)),
                                                 __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) {
            e = f;
            ret = 1;
            break;
        }
    }
    if (e < 8u) {
        __hip_atomic_fetch_max(KTA_LDS(uint32_t, lds_val + (set * 8u + e) * 4u), lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        return ret;
    }
    const unsigned long long pr = ((unsigned long long)(h + 1u) << 32) | lo;
    const uint32_t sub = (set / kSliceSets) * kOvfSub;
    uint32_t pos = sub + ((h * 0x9E3779B1u) >> (32 - kOvfSubLog2));
    for (uint32_t probe = 0; probe < kOvfProbes; probe++) {
        auto *slot = KTA_LDS(unsigned long long, lds_ovf + pos * 8u);
        unsigned long long cur = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (cur == 0ull) {
            if (__hip_atomic_compare_exchange_strong(slot, &cur, pr, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) {
                __hip_atomic_fetch_add(KTA_LDS(uint32_t, lds_ovf_n), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                *KTA_LDS(uint32_t, lds_any_ovf) = 1u;
                return 2;
            }                                             // (cur now holds what got there first)
        }
        if ((uint32_t)(cur >> 32) == h + 1u) {
            __hip_atomic_fetch_max(slot, pr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            return 2;
        }
        pos = sub + ((pos + 1u) & (kOvfSub - 1u));
    }
    return 3;
}

struct ApplyShared {
    uint32_t pairs, claims, instalments, ovf_total;
    // Counters of the current instalment in three rotating slots: between checkpoint t - 1 and t the waves add to
    // slot t % 3; after the barrier of checkpoint t everybody reads it, and thread 0 clears slot (t + 2) % 3 — the
    // one of interval t - 1, which nobody reads any more (all are past that) and nobody adds to before the barrier
    // of checkpoint t + 1 is behind them.  One barrier per checkpoint.
    uint32_t occ[3], ovf_n[3], fail[3];
    uint32_t any_ovf;        // the side table holds something (this instalment)
    uint32_t next_seg;       // fast attempt: the next segment to hand out
    uint32_t dense;          // POOL_DENSE as thread 0 found it: ONE reading for the whole workgroup
    long long w[kApplyWaves];
};

// BITMAP: the persistent state is the reference's bit set (u32 words, bit h & 31 of word h >> 5); else the
// u64 last-writer table.
// RANGES (bit set state only): the instantiation that takes the buckets whose distinct slots did not fit the table in the
// plain attempt — listed by kta_alive_apply<BLOG2, true, false> behind fail_from's 2 B words, POOL_RANGES of them.  Such a
// bucket is applied in 2^rho PASSES over ALL its pairs, pass r taking the slots of the r-th 2^rho-th of the bucket's slot
// range (the top rho bits of the slot in the bucket) and sweeping only that part of the bucket's bit set: every pair of a slot
// is in one pass, so a pass is a plain attempt — segments in any order, no checkpoints — on a table whose sets are chosen by
// the next 11 bits down.  rho starts at 1 (or at what an earlier bucket of the batch needed: POOL_DENSE) and doubles the
// parts when a pass does not fit — what earlier passes applied is the final state of their slots, applying it again changes
// nothing — up to 16 parts; what defeats that too (a few sets taking everything) goes to kta_alive_fallback.  Rounds 4-5 sent every
// overflowing bucket into careful mode (instalments in segment order — still the table state's way, which has the direct
// path as its last resort): a batch of 20 M distinct keys (19.5 k slots per bucket against the table's 16 k) was applied in 26 instalments
// per bucket, each a sweep of the bucket's 512 KiB — 7.6 ms instead of 2.6 —, config 5's law on one GPU (97 k slots per
// bucket) in 32.
constexpr uint32_t kMaxRho = 4;
template <int BLOG2, bool BITMAP, bool RANGES>
__global__ __launch_bounds__(kApplyThreads) void kta_alive_apply(const unsigned long long *__restrict__ pairs,
                                                                 const uint32_t *__restrict__ counts, uint32_t cap,
                                                                 uint32_t W, uint32_t range, uint64_t base_seq,
                                                                 const uint64_t *__restrict__ seq_col,
                                                                 unsigned long long *__restrict__ table,
                                                                 uint32_t *__restrict__ bitmap,
                                                                 long long *__restrict__ running,
                                                                 unsigned long long *__restrict__ stats,
                                                                 const uint32_t *__restrict__ pool_hist,
                                                                 uint32_t *__restrict__ fail_from,
                                                                 unsigned long long *__restrict__ pool_ctl,
                                                                 const uint32_t *__restrict__ skip_flag, WrittenList wl)
{
    constexpr uint32_t RBITS = 32 - BLOG2;             // hash bits below the bucket
    constexpr uint32_t TAGBITS = RBITS - kSetLog2;     // slots of one set = 2^TAGBITS, a tag = slot in set + 1
    static_assert(TAGBITS <= 15, "tags are 16 bit");
    static_assert(BITMAP || !RANGES, "slot-range passes are the bit set state's");
    constexpr bool kHasCareful = !BITMAP;              // (bit set state: what does not fit goes to the RANGES instantiation, and from there to kta_alive_fallback)
    constexpr uint32_t kSliceWords = (kSliceSets << TAGBITS) / 32;   // u32 words of one bitmap slice
    static_assert(!BITMAP || kSliceWords == 2 * 4 * kApplyThreads, "a thread moves two 16-byte pieces of a slice");
    // a new instalment once this many entries are claimed (see checkpoint): with 8-way sets the lists of
    // records that found their set full stay short up to about 0.75 load
    constexpr uint32_t kFlushAt = kEntries / 2 + kEntries / 4;
    KTA_PHASE_BEGIN;
    extern __shared__ __attribute__((aligned(128))) uint32_t s_val[];        // kEntries values
    unsigned short *s_tag = reinterpret_cast<unsigned short *>(s_val + kEntries);   // kEntries tags
    unsigned long long *s_ovf = reinterpret_cast<unsigned long long *>(s_tag + kEntries);   // kOvf pairs
    uint32_t *s_cnt = reinterpret_cast<uint32_t *>(s_ovf + kOvf);            // W segment fills
    uint32_t *s_slice = s_cnt + ((W + 31u) & ~31u);                          // BITMAP: one slice of the region
    __shared__ ApplyShared sh;
    uint32_t b = blockIdx.x;
    if (skip_flag && *skip_flag) return;               // the batch was handed to another path (unordered seq column)
    if (RANGES) {                                      // the b-th bucket of the list
        if (b >= (uint32_t)pool_ctl[POOL_RANGES]) return;
        b = fail_from[(2u << BLOG2) + b];
    }
    if (BITMAP && !RANGES) {
        // a bucket with pool pairs is not attempted: its pairs are not all in its segments
        if (pool_hist[b] != 0u) {
            if (threadIdx.x == 0) {
                fail_from[b] = 0u;
                fail_from[(1u << BLOG2) + (uint32_t)atomicAdd(&pool_ctl[POOL_FAILED], 1ull)] = b;   // (the list of such buckets: kta_alive_fallback)
            }
            return;
        }
        if (threadIdx.x == 0) fail_from[b] = kNoFail;
    }
    for (uint32_t e = threadIdx.x; e < kEntries + kEntries / 2 + 2 * kOvf; e += kApplyThreads) s_val[e] = 0u;   // values, tags, side table
    if (threadIdx.x == 0) {
        sh.pairs = 0;
        sh.claims = 0;
        sh.instalments = 0;
        sh.ovf_total = 0;
        sh.occ[0] = sh.occ[1] = sh.occ[2] = sh.ovf_n[0] = sh.ovf_n[1] = sh.ovf_n[2] = sh.fail[0] = sh.fail[1] = sh.fail[2] = 0;
        sh.any_ovf = 0;
        sh.next_seg = 0;
        // Another bucket's workgroup may set the word at any moment.  Waves that read it for themselves could
        // disagree, and a workgroup whose waves run in different modes does not meet at the same barriers.
        // (bit set state: the word is the rho an earlier bucket needed, 0 = none did)
        sh.dense = BITMAP ? (uint32_t)pool_ctl[POOL_DENSE] : (pool_ctl[POOL_DENSE] != 0ull ? 1u : 0u);
    }
    __syncthreads();
    if (BITMAP && !RANGES && sh.dense != 0u) {         // an earlier bucket of this batch did not fit: no attempt, straight to the list
        if (threadIdx.x == 0) fail_from[(2u << BLOG2) + (uint32_t)atomicAdd(&pool_ctl[POOL_RANGES], 1ull)] = b;
        return;
    }
    for (uint32_t w = threadIdx.x; w < W; w += kApplyThreads) {
        const uint32_t cw = counts[(uint64_t)b * W + w];
        s_cnt[w] = cw;
        if (cw) atomicAdd(&sh.pairs, cw);
    }
    __syncthreads();
    if (sh.pairs == 0u) return;                        // nothing hashed into this bucket: the region is not touched
    KTA_PHASE(1, 0);
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    long long delta = 0;
    // RANGES: the bucket's slot range is taken in 2^rho parts, the current pass takes part ridx (wave-uniform; rho = 0 in
    // careful mode and in the other instantiations, where everything below folds to the constants it was written with)
    uint32_t rho = 0, ridx = 0;

    // global value of a survivor (table state): ((sequence + 1) << 1) | alive
    auto global_val = [&](uint32_t lo) __attribute__((always_inline)) -> unsigned long long {
        const uint64_t idx = (uint64_t)(lo >> 1) - 1u;
        const uint64_t s = seq_col ? seq_col[idx] : base_seq + idx;
        return ((unsigned long long)(s + 1) << 1) | (lo & 1u);
    };

    // ---- end of an instalment: resolve duplicates and the full-set list, apply, leave the table empty ----
    // (every lambda of this kernel is forced inline: out of line, its captures live in scratch memory and the LDS
    // pointers among them lose their address space — flat instead of ds instructions)
    auto end_instalment = [&]() __attribute__((always_inline)) {
        lds_barrier();                                  // every wave's merges are done
        // The slots that found their set full sit in the side table, one entry per slot (never also in a set: tags
        // do not leave a set during an instalment, so a slot that was refused once is refused every time).
        const uint32_t novf = sh.any_ovf ? kOvf : 0u;
        if (BITMAP) {
            // The bucket's region of the bit set goes through LDS, and no workgroup barrier is needed for it: wave v
            // owns the slots of the sets [128 v, 128 v + 128) and walks them in 16 pieces of 8 sets = 2 KiB of bit
            // set = 32 bytes per lane; the piece's 64 table entries are one per lane.  The next piece is requested
            // while this one is worked on.
            // (RANGES: a set holds 2^(TAGBITS - rho) slots, so the wave's 128 sets are 1 / 2^rho of its share of the bucket's
            // region, a piece of 2 KiB of bit set is 8 << rho sets = 64 << rho entries, 2^rho per lane, and the wave has
            // 16 >> rho pieces)
            const uint32_t tbits = TAGBITS - rho;
            const uint32_t psets = 8u << rho, np = (kSliceSets / 8u) >> rho;       // sets of a piece; pieces of this wave
            uint32_t *mine = bitmap + ((size_t)b << (RBITS - 5)) + (size_t)ridx * ((1u << (RBITS - 5)) >> rho) +
                             (size_t)wave * (kSliceWords >> rho);
            uint32_t *buf = s_slice + wave * (kSliceWords / kSliceSets * 8u);       // 512 words of this wave
            constexpr uint32_t kPieces = kSliceSets / 8u;                          // 16: at most; also "no piece"
            constexpr uint32_t kPieceWords = kSliceWords / kPieces;                // 512
            // Which of the wave's pieces hold entries at all (a small batch leaves most of the region alone).  The
            // pieces are walked from a start that differs from wave to wave and bucket to bucket: in step, 4096
            // waves would otherwise ask for addresses that differ by multiples of 32 KiB — a handful of the
            // memory's channels — at every moment.  Bit i of `occupied` = piece (i + start) % 16.
            const uint32_t start = (wave + b * 5u) & (np - 1u);
            uint32_t occupied = 0;
            for (uint32_t i = 0; i < np; i++) {
                bool any_tag = false;
                for (uint32_t k = 0; k < (1u << rho); k++)
                    any_tag |= s_tag[(wave * kSliceSets + ((i + start) & (np - 1u)) * psets) * 8u + k * 64u + lane] != 0;
                occupied |= (__any(any_tag) ? 1u : 0u) << i;
            }
            const uint4 *src = reinterpret_cast<const uint4 *>(mine);
            // The occupied pieces, one request ahead.  Loads and stores share one in-order counter (vmcnt): the next
            // piece is requested BEFORE this piece's stores, so the wait for it lets exactly those stores pend.
            auto next_piece = [&](uint32_t after) __attribute__((always_inline)) -> uint32_t {   // kPieces: none
                const uint32_t rest = after + 1u < 32u ? occupied & ~((2u << after) - 1u) : 0u;
                return rest ? (uint32_t)__builtin_ctz(rest) : kPieces;
            };
            auto request = [&](uint32_t i, uint4 &q0, uint4 &q1) __attribute__((always_inline)) {   // unconditional: countable
                const uint32_t mc = ((i < kPieces ? i : 0u) + start) & (np - 1u);
                q0 = src[(size_t)mc * (kPieceWords / 4u) + lane];
                q1 = src[(size_t)mc * (kPieceWords / 4u) + lane + 64u];
            };
            // the side table's entries of this wave's slots: two per lane (their sets are full, so their pieces are
            // among the occupied ones)
            static_assert(kOvfSub == 128, "two side entries per lane");
            const unsigned long long side0 = novf ? s_ovf[wave * kOvfSub + lane] : 0ull;
            const unsigned long long side1 = novf ? s_ovf[wave * kOvfSub + 64u + lane] : 0ull;
            auto apply_side = [&](unsigned long long pr, uint32_t m) __attribute__((always_inline)) {
                if (pr == 0ull) return;
                const uint32_t h = (uint32_t)(pr >> 32) - 1u, lo = (uint32_t)pr;
                const uint32_t set_local = (h >> tbits) & (kSliceSets - 1u);
                if ((set_local >> (3u + rho)) != m) return;
                const uint32_t bit = ((set_local & (psets - 1u)) << tbits) | (h & ((1u << tbits) - 1u));
                const uint32_t mk = 1u << (bit & 31u);
                const uint32_t old = lo & 1u ? atomicOr(&buf[bit >> 5], mk) : atomicAnd(&buf[bit >> 5], ~mk);
                delta += (long long)(lo & 1u) - (long long)((old & mk) != 0u);
            };
            auto piece = [&](uint32_t i, const uint4 &q0, const uint4 &q1) __attribute__((always_inline)) {
                const uint32_t m = (i + start) & (np - 1u);
                uint4 *sl = reinterpret_cast<uint4 *>(buf);
                sl[lane] = q0;
                sl[lane + 64u] = q1;
                KTA_LDS_ORDER();
                for (uint32_t k = 0; k < (1u << rho); k++) {
                    const uint32_t e = (wave * kSliceSets + m * psets) * 8u + k * 64u + lane;   // the lane's tag: half-word lane & 7 of the piece's set 8 k + (lane >> 3)
                    const uint32_t ev = (e & ~7u) | entry_of_half(lane & 7u);             // ... and its value
                    const uint32_t tag = s_tag[e], lo = s_val[ev];
                    if (tag) {
                        const uint32_t bit = ((8u * k + (lane >> 3)) << tbits) | (tag - 1u);
                        const uint32_t mk = 1u << (bit & 31u);
                        const uint32_t old = lo & 1u ? atomicOr(&buf[bit >> 5], mk) : atomicAnd(&buf[bit >> 5], ~mk);
                        delta += (long long)(lo & 1u) - (long long)((old & mk) != 0u);
                        s_tag[e] = 0;
                        s_val[ev] = 0u;
                    }
                }
                apply_side(side0, m);
                apply_side(side1, m);
                KTA_LDS_ORDER();
                uint4 *dst = reinterpret_cast<uint4 *>(mine + (size_t)m * kPieceWords);
                dst[lane] = sl[lane];
                dst[lane + 64u] = sl[lane + 64u];
                KTA_LDS_ORDER();
            };
            // Two pieces are in flight while one is worked on (one wave holds 4 KiB of requests: 16 waves x 256 CUs =
            // 16 MiB on the way, what the memory's latency asks for at its full rate; with one piece the sweep ran at less
            // than half of it).  Two register sets take turns — copying "next" into "current" would wait for the request
            // at the very place it was issued — and every request is unconditional: countable.
            uint32_t m0 = occupied ? (uint32_t)__builtin_ctz(occupied) : kPieces;
            if (m0 < kPieces) {
                uint32_t m1 = next_piece(m0);
                uint4 x0, x1, y0, y1;
                request(m0, x0, x1);
                request(m1, y0, y1);
#pragma unroll 1
                for (;;) {
                    {
                        const uint32_t m2 = next_piece(m1 < kPieces ? m1 : kPieces - 1u);
                        const uint4 t0 = x0, t1 = x1;
                        request(m2, x0, x1);
                        piece(m0, t0, t1);
                        m0 = m2;
                    }
                    if (m1 >= kPieces) break;
                    {
                        const uint32_t m3 = next_piece(m0 < kPieces ? m0 : kPieces - 1u);
                        const uint4 t0 = y0, t1 = y1;
                        request(m3, y0, y1);
                        piece(m1, t0, t1);
                        m1 = m3;
                    }
                    if (m0 >= kPieces) break;
                }
            }
            lds_barrier();                              // all waves have read the side table
        } else {
            // (3) the workgroup is its region's only writer during this kernel (its own direct-path atomics
            // are complete: barrier), so read / compare / write needs no RMW atomic.  Loads and stores are
            // agent-scope so that they see, and are seen by, the atomics of the direct path and other kernels.
            // Four entries per thread at a time (eight spill), and every load of them UNCONDITIONAL (an entry without a tag reads its set's
            // first slot and record 0's sequence number, and changes nothing): under `if (tag)` each load was a branch of its own
            // with a full wait behind it — 32 memory round trips one after the other per instalment, 123 of pass 2's 280 us per
            // bucket (round 6's phase counters).  The new slots of a round are appended to the written list with ONE device
            // atomic per wave.
            constexpr int U = 4;
            for (uint32_t e0 = 0; e0 < kEntries; e0 += U * kApplyThreads) {
                uint32_t tg[U], lo[U], slot[U];
                unsigned long long sv[U], old[U];
#pragma unroll
                for (int u = 0; u < U; u++) {
                    const uint32_t e = e0 + (uint32_t)u * kApplyThreads + threadIdx.x;      // a tag by its half-word position ...
                    const uint32_t ev = (e & ~7u) | entry_of_half(e & 7u);                    // ... and its value
                    tg[u] = s_tag[e];
                    lo[u] = s_val[ev];
                    s_tag[e] = 0;
                    s_val[ev] = 0u;
                    slot[u] = (b << RBITS) | ((e >> 3) << TAGBITS) | (tg[u] ? tg[u] - 1u : 0u);
                }
#pragma unroll
                for (int u = 0; u < U; u++) {
                    const uint64_t idx = tg[u] ? (uint64_t)(lo[u] >> 1) - 1u : 0u;
                    sv[u] = seq_col ? seq_col[idx] : base_seq + idx;
                    old[u] = __hip_atomic_load(&table[slot[u]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                bool fresh[U];
#pragma unroll
                for (int u = 0; u < U; u++) {
                    const unsigned long long v = ((unsigned long long)(sv[u] + 1) << 1) | (lo[u] & 1u);
                    const bool upd = tg[u] && v > old[u];
                    fresh[u] = upd && old[u] == 0ull;
                    if (upd) {
                        __hip_atomic_store(&table[slot[u]], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        delta += (long long)(v & 1ull) - (long long)(old[u] & 1ull);
                    }
                }
                if (wl.slots) {                          // (uniform control flow from here: every lane takes part in the ballots)
                    unsigned long long m[U];
                    uint32_t total = 0;
#pragma unroll
                    for (int u = 0; u < U; u++) {
                        m[u] = __builtin_amdgcn_ballot_w64(fresh[u]);
                        total += (uint32_t)__popcll(m[u]);
                    }
                    if (total) {                         // (wave-uniform)
                        unsigned long long base = 0;
                        if (lane == 0u) base = atomicAdd(wl.n, (unsigned long long)total);
                        base = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(base >> 32)) << 32) |
                               (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)base);
#pragma unroll
                        for (int u = 0; u < U; u++) {
                            const unsigned long long at = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(m[u] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m[u], 0u));
                            if (fresh[u] && at < wl.cap) wl.slots[at] = slot[u];
                            base += (uint32_t)__popcll(m[u]);
                        }
                    }
                }
            }
            lds_barrier();
            for (uint32_t k = threadIdx.x; k < novf; k += kApplyThreads) {
                const unsigned long long pr = s_ovf[k];
                if (pr) delta += direct_update(table, (b << RBITS) | ((uint32_t)(pr >> 32) - 1u), global_val((uint32_t)pr), wl);
            }
        }
        for (uint32_t k = threadIdx.x; k < novf; k += kApplyThreads) s_ovf[k] = 0ull;
        if (threadIdx.x == 0) {
            sh.claims += sh.occ[0] + sh.occ[1] + sh.occ[2];
            sh.instalments++;
            sh.ovf_total += sh.ovf_n[0] + sh.ovf_n[1] + sh.ovf_n[2];
            sh.occ[0] = sh.occ[1] = sh.occ[2] = sh.ovf_n[0] = sh.ovf_n[1] = sh.ovf_n[2] = 0;
            sh.any_ovf = 0;
        }
        lds_barrier();
    };

    // A wave reads its segments in units of 256 pairs (cap is a multiple of 256), four per lane, masked by the segment's
    // fill: bit set state — ONE 16-byte load per lane, its four 4-byte pairs; table state — the lane's 24 bytes of the
    // dense stream of 6-byte pairs, a 16-byte and an 8-byte load (kta_alive_partition48).  kApplyDepth - 1 (table state:
    // one) units are in flight while one is merged.
    const uint8_t *region_pairs48 = reinterpret_cast<const uint8_t *>(pairs) + (uint64_t)b * W * cap * 6u;
    const uint32_t *region_pairs32 = reinterpret_cast<const uint32_t *>(pairs) + (uint64_t)b * W * cap;
    const uint32_t loads = cap >> 8;
    const uint32_t chunks = loads;
    // Careful mode (table state only since round 6: the bit set state's overflowing buckets are applied in slot-range
    // passes, see RANGES) merges one GROUP of sixteen segments — one per wave, each older than the next group's — between
    // two checkpoints; what a group's slots overflow of table and side table takes the direct path.  (Rounds 4-5 sized the
    // bit set state's groups from the segment fills, pick_group_size: a group had to fit whatever its keys were.)
    const uint32_t groups16 = (W + kApplyWaves - 1) / kApplyWaves, units16 = groups16 * chunks;
#define KTA_GS ((uint32_t)kApplyWaves)                      /* segments of a group */
#define KTA_UNITS units16                                   /* units of a wave's walk in careful mode (the same for every wave) */
    // bit set state: what pass 2 maximises per slot is (segment w, window, position in the segment) << 1 | alive; the
    // position takes ksh = ceil(log2(cap)) bits (cap >= 256, so the window's field starts above bit 8)
    const uint32_t ksh = 32u - (uint32_t)__builtin_clz(cap - 1u);
    struct Unit {
        ulonglong2 q[kApplyUnroll];  // bit set state: q[0] = four pair32; table state: q[0], q[1].x = four pair48
        uint32_t nv[kApplyUnroll];   // [0] = the lane's valid pairs, 0..4; [1] = bit set state: the value of the lane's first pair
                                     // without its window and alive bit; table state: the batch-local index of the segment's range
    };
    // Which segment a unit reads: in careful mode wave v takes the segments v, v + 16, ... (every group of 16 is
    // older than the next); in the fast attempt the order does not matter and the waves take whatever segment is
    // next (they finish together instead of waiting for whoever had the slow records).
    uint32_t seg = 0, seg_unit = 0, seg_units = 0;       // wave-uniform
    bool dynamic = false;
    auto issue = [&](uint32_t u, Unit &un) __attribute__((always_inline)) {
        uint32_t r0;
        // (wave-uniform by construction; said again, because loop-carried values of this kernel's loops end up in
        // vector registers and every use of them in vector instructions)
        seg_unit = __builtin_amdgcn_readfirstlane(seg_unit);
        seg_units = __builtin_amdgcn_readfirstlane(seg_units);
        u = __builtin_amdgcn_readfirstlane(u);
        if (dynamic) {                                   // a handed-out segment is read in as many units as it has pairs for
            if (seg_unit >= seg_units) {
                uint32_t g = 0;
                if (lane == 0) g = lds_add(&sh.next_seg, 1u);
                seg = __builtin_amdgcn_readfirstlane(g);
                const uint32_t c = __builtin_amdgcn_readfirstlane(s_cnt[seg < W ? seg : 0u]);
                seg_units = (c + 255u) / 256u;                                   // (256 pairs a unit either way)
                if (seg_units == 0u) seg_units = 1u;
                seg_unit = 0;
            }
            r0 = seg_unit;
            seg_unit++;
        } else {
            r0 = u % chunks;
            seg = (u / chunks) * KTA_GS + __builtin_amdgcn_readfirstlane(wave);
        }
        seg = __builtin_amdgcn_readfirstlane(seg);
        const bool on = seg < W && (dynamic || u < KTA_UNITS);
        const uint32_t cnt = on ? (uint32_t)__builtin_amdgcn_readfirstlane(s_cnt[on ? seg : 0u]) : 0u;
        if (BITMAP) {
            const uint32_t *sp = region_pairs32 + (uint64_t)(on ? seg : 0u) * cap;
            // (a segment's 64-byte blocks read as scattered halves of 128-byte lines — even blocks, then odd ones — cost pass 2
            // 0.02 ms of 0.68: gathered blocks would be affordable)
            const uint32_t k = (r0 << 8) + 4u * lane;
            un.nv[0] = r0 < loads && k < cnt ? (cnt - k >= 4u ? 4u : cnt - k) : 0u;
            un.nv[1] = ((on ? seg : 0u) << (ksh + 9u)) | (k << 1);
            // unconditional (clamped address, masked by nv): a predicated load is a branch, and the wait for this
            // unit's data would then be vmcnt(0) — it would wait for the units behind it as well
            const v2ull four = __builtin_nontemporal_load(reinterpret_cast<const v2ull *>(sp + (un.nv[0] ? k : 4u * lane)));   // (read once)
            un.q[0] = make_ulonglong2(four.x, four.y);
            return;
        }
        {
            const uint8_t *sp = region_pairs48 + (uint64_t)(on ? seg : 0u) * cap * 6u;
            const uint32_t k = (r0 << 8) + 4u * lane;
            un.nv[0] = r0 < loads && k < cnt ? (cnt - k >= 4u ? 4u : cnt - k) : 0u;
            un.nv[1] = (on ? seg : 0u) * range;
            const uint8_t *at = sp + (uint64_t)(un.nv[0] ? k : 4u * lane) * 6u;     // unconditional, as above (8-byte aligned: 24 bytes per lane)
            typedef unsigned long long ull_a8 __attribute__((ext_vector_type(2), aligned(8)));
            // (ordinary loads: the two instructions ask for the same lines, and a non-temporal load's line does not wait in the
            // cache for the second one)
            const ull_a8 lo = *reinterpret_cast<const ull_a8 *>(at);
            un.q[0] = make_ulonglong2(lo.x, lo.y);
            un.q[1] = make_ulonglong2(*reinterpret_cast<const unsigned long long *>(at + 16), 0ull);
        }
    };
    uint32_t claimed = 0;
    // table state: an earlier bucket of this batch overflowed its table: check as we go.  Bit set state: the plain
    // instantiation never is careful (what does not fit goes to the RANGES one, which starts with slot-range passes)
    bool careful = !BITMAP && __builtin_amdgcn_readfirstlane(sh.dense) != 0u;
    if (RANGES) {
        rho = (uint32_t)__builtin_amdgcn_readfirstlane(sh.dense);
        rho = rho < 1u ? 1u : (rho > kMaxRho ? kMaxRho : rho);
    }
    // The merge of one pair that found no entry for its slot: merge_new_slot, out of line.
    const uint32_t o_tag = lds_offset(s_tag), o_val = lds_offset(s_val), o_ovf = lds_offset(s_ovf), o_any = lds_offset(&sh.any_ovf);
    auto merge_new = [&](uint32_t h, uint32_t lo, uint32_t par) __attribute__((always_inline)) {
        const uint32_t how = merge_new_slot(o_tag, o_val, o_ovf, lds_offset(&sh.ovf_n[par]), o_any, TAGBITS - rho, h, lo);
        claimed += how == 1u ? 1u : 0u;
        if (how == 3u) {
            // Full as well: the attempt is given up (fast attempt; bit set state) or the record takes the direct path
            // (table state, careful mode).
            if (BITMAP || !careful) sh.fail[par] = 1u;
            else delta += direct_update(table, (b << RBITS) | h, global_val(lo), wl);
        }
    };
    // Pairs whose slot has an entry (most of a compacted topic's) are merged where they stand: one lookup, one
    // max.  The others are queued, per wave, and handled 64 at a time: a lane that needs the long way would
    // otherwise make its whole wave walk it — and with 64 lanes there is always one.  The queues live in the LDS
    // the sweep uses later (kMissQueue pairs per wave).
    unsigned long long *missq = reinterpret_cast<unsigned long long *>(s_slice) + wave * kMissQueue;
    uint32_t mq = 0;                                     // wave-uniform
    auto drain = [&](uint32_t par) __attribute__((always_inline)) {   // everything that is queued
        KTA_LDS_ORDER();
        for (uint32_t q0 = 0; q0 < mq; q0 += 64u) {
            const bool on = q0 + lane < mq;
            const unsigned long long pr = missq[on ? q0 + lane : 0u];
            if (on) merge_new((uint32_t)(pr >> 32), (uint32_t)pr, par);
        }
        KTA_LDS_ORDER();
        mq = 0;
    };
    auto drain_full = [&](uint32_t par) __attribute__((always_inline)) {   // whole waves of 64 only, from the queue's end (the order does not matter)
        KTA_LDS_ORDER();
        while (mq >= 64u) {
            mq -= 64u;
            const unsigned long long pr = missq[mq + lane];
            merge_new((uint32_t)(pr >> 32), (uint32_t)pr, par);
        }
        KTA_LDS_ORDER();
    };
    // Merge the unit `un`: four pairs per lane, their four lookups in flight together (one form for both states since round 6:
    // the table state's used to collect misses over several units and make a miss that found the queue full wait).
    // RANGES: one pair per lane — (slot inside this pass's part) << 32 | value —, what merge does with four
    unsigned long long *todoq = missq + 192;             // (the wave's own pairs waiting for a full wave of them: < 128; its misses: < 128)
    uint32_t tq = 0;                                     // wave-uniform
    auto merge_one = [&](unsigned long long pr, bool on, uint32_t par) __attribute__((always_inline)) {
        const uint32_t tb = TAGBITS - rho;
        const uint32_t x = on ? (uint32_t)(pr >> 32) : 0u, v = (uint32_t)pr;
        const uint4 t = *reinterpret_cast<const uint4 *>(s_tag + (x >> tb) * 8u);
        const uint32_t e = find_tag(t, (x & ((1u << tb) - 1u)) + 1u);
        if (on && e < 8u) atomicMax(&s_val[(x >> tb) * 8u + e], v);
        const bool miss = on && e >= 8u;
        const unsigned long long m = __builtin_amdgcn_ballot_w64(miss);
        mq = __builtin_amdgcn_readfirstlane(mq);
        if (miss) missq[mq + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] = ((unsigned long long)x << 32) | v;
        mq += (uint32_t)__popcll(m);
        drain_full(par);
    };
    auto merge = [&](const Unit &un, uint32_t par) __attribute__((always_inline)) {
        static_assert(kApplyUnroll == 2 && kMissQueue >= 5 * 64, "a unit's misses fit the queue behind what drain_full leaves");
        mq = __builtin_amdgcn_readfirstlane(mq);         // (wave-uniform by construction: see issue)
        // the lane's four pairs as (slot in the bucket, value to maximise)
        uint32_t hr[4], val[4];
        if (BITMAP) {
            const uint32_t w32[4] = {(uint32_t)un.q[0].x, (uint32_t)(un.q[0].x >> 32), (uint32_t)un.q[0].y, (uint32_t)(un.q[0].y >> 32)};
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const uint32_t w = w32[i];
                // value = (segment, window, position) << 1 | alive: the window and the alive bit are the pair's low bits
                val[i] = ((w & 0x1FEu) << ksh) | (un.nv[1] + 2u * (uint32_t)i);
                val[i] = (val[i] & ~1u) | (w & 1u);
                hr[i] = w >> kPair32Shift;
            }
        } else {
            // pair48: word 0 = index bits 0..14, word 1 = slot bits 0..14, word 2 = slot bits 15..21 | alive << 7 | index bits
            // 15..21 << 8, bit 15 of every word set (kta_alive_partition48); value = (batch-local index + 1) << 1 | alive
            const uint32_t d6[6] = {(uint32_t)un.q[0].x, (uint32_t)(un.q[0].x >> 32), (uint32_t)un.q[0].y, (uint32_t)(un.q[0].y >> 32),
                                    (uint32_t)un.q[1].x, (uint32_t)(un.q[1].x >> 32)};
            const uint32_t wa[4] = {d6[0], d6[1] >> 16, d6[3], d6[4] >> 16}, wb[4] = {d6[0] >> 16, d6[2], d6[3] >> 16, d6[5]},
                           wc[4] = {d6[1], d6[2] >> 16, d6[4], d6[5] >> 16};                  // (their low 16 bits)
#pragma unroll
            for (int i = 0; i < 4; i++) {
                hr[i] = (wb[i] & 0x7FFFu) | ((wc[i] & 0x7Fu) << 15);
                const uint32_t idx = (wa[i] & 0x7FFFu) | ((wc[i] & 0x7F00u) << 7);
                val[i] = (un.nv[1] + idx + 1u) << 1 | ((wc[i] >> 7) & 1u);
            }
        }
        // RANGES: this pass takes the slots whose top rho bits are ridx — one pair in 2^rho; of those, the next 11 bits choose
        // the set.  A unit's own pairs are compacted into a second queue of the wave first and merged 64 at a time, a pair per
        // lane (merge_one): the foreign pairs cost the unpacking and a ballot, not the lookup (with eight parts a pass paid
        // the whole merge for seven eighths of its pairs: 5.3 ms of pass 2 on config 5's law).
        const uint32_t tbits = TAGBITS - rho;
        if (RANGES) {
            tq = __builtin_amdgcn_readfirstlane(tq);
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const bool own = (uint32_t)i < un.nv[0] && (hr[i] >> (RBITS - rho)) == ridx;
                const unsigned long long m = __builtin_amdgcn_ballot_w64(own);
                if (own)
                    todoq[tq + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] =
                        ((unsigned long long)(hr[i] & ((1u << (RBITS - rho)) - 1u)) << 32) | val[i];
                tq += (uint32_t)__popcll(m);
                if (tq >= 64u) {                           // (wave-uniform)
                    tq -= 64u;
                    KTA_LDS_ORDER();
                    merge_one(todoq[tq + lane], true, par);
                }
            }
            return;
        }
        uint4 t[4];
#pragma unroll
        for (int i = 0; i < 4; i++) t[i] = *reinterpret_cast<const uint4 *>(s_tag + (hr[i] >> tbits) * 8u);
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint32_t tag = (hr[i] & ((1u << tbits) - 1u)) + 1u;
            const uint32_t e = find_tag(t[i], tag);
            const bool valid = (uint32_t)i < un.nv[0];
            if (valid && e < 8u) atomicMax(&s_val[(hr[i] >> tbits) * 8u + e], val[i]);
            const bool miss = valid && e >= 8u;
            const unsigned long long m = __builtin_amdgcn_ballot_w64(miss);   // (__ballot goes through an int: a select and a compare)
            if (miss)
                missq[mq + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] =
                    ((unsigned long long)hr[i] << 32) | val[i];
            mq += (uint32_t)__popcll(m);
        }
        drain_full(par);
    };
    // The driver.  First without checkpoints (no barrier until the end: a compacted topic's bucket fits the table);
    // a bucket that overflows table AND side table that way starts over in careful mode, and tells the buckets
    // still to come (POOL_DENSE).  Careful mode: after every unit that ends a group of segments (every wave has
    // finished whole segments: everything merged so far is older than everything that follows — what the bit set
    // state needs) the workgroup decides whether the table is emptied before it goes on.  Every thread keeps the
    // instalment's totals in registers, fed from the counter slot of the interval just ended (ApplyShared).
    // (Whatever the workgroup decides on comes out of LDS through readfirstlane: a value loaded per lane is divergent
    // to the compiler, and so is every loop counter of a loop that such a value leaves.)
    uint32_t inst_start = 0;                            // first segment of the current instalment
    uint32_t tot_occ = 0, tot_ovf = 0, last_occ = 0, par = 0;
    uint32_t u = 0;                                     // the unit that is merged next
    constexpr int D = BITMAP ? kApplyDepth : 3;        // (table state: six registers a unit instead of four — the registers for three)
    Unit ring[D];
    for (;;) {                                          // one pass per instalment (one more per restart)
        // D units per trip, the buffers taking turns: copying prefetched registers into "current" ones
        // would make the compiler wait for the prefetch right where it was issued.  The prefetch is issued
        // unconditionally (a unit past the end loads a clamped address and merges nothing): under a branch the
        // compiler could not count it and would wait with vmcnt(0).
        dynamic = !careful;
        if (RANGES && dynamic) {                         // a new pass over all the bucket's segments: hand them out again
            lds_barrier();
            if (threadIdx.x == 0) sh.next_seg = 0;
            lds_barrier();
            seg = 0, seg_unit = 0, seg_units = 0, u = 0;
        }
#pragma unroll
        for (int s = 0; s + 1 < D; s++) issue(u + (uint32_t)s, ring[s]);
        bool flush = false, failed = false, stop = false;
        // (fast attempt: `seg` is the segment of the unit issued LAST; segments are handed out in ascending order, so
        // once it is past the end every unit still in flight is empty or about to be merged by the trip)
        while (!stop && (dynamic ? seg < W : u < KTA_UNITS)) {
            if (!careful && __builtin_amdgcn_readfirstlane(sh.fail[0])) break;   // (fast attempt) some wave ran out of room: stop early
#pragma unroll
            for (int s = 0; s < D; s++) {
                if (stop) break;
                issue(u + (uint32_t)D - 1u, ring[(s + D - 1) % D]);
                merge(ring[s], par);
                u++;
                KTA_PHASE(1, 3);
                if (kHasCareful && careful && u % chunks == 0u) {
                    drain(par);                           // the queued misses are in before anybody looks at the table as a whole
                    if (u < KTA_UNITS) {
#pragma unroll
                        for (int off = 32; off > 0; off >>= 1) claimed += __shfl_xor(claimed, off);
                        if (lane == 0 && claimed) atomicAdd(&sh.occ[par], claimed);
                        claimed = 0;
                        lds_barrier();
                        tot_occ += __builtin_amdgcn_readfirstlane(sh.occ[par]);
                        tot_ovf += __builtin_amdgcn_readfirstlane(sh.ovf_n[par]);
                        failed = __builtin_amdgcn_readfirstlane(sh.fail[par]) != 0u;
                        if (threadIdx.x == 0) sh.occ[(par + 2u) % 3u] = sh.ovf_n[(par + 2u) % 3u] = 0;
                        par = (par + 1u) % 3u;
                        // empty the table when the next interval, growing like the last one, would take it past kFlushAt
                        // (measured and dropped in round 5: no forecast — merge until a group finds the table full, apply,
                        // merge that group again: the brim-full table's long lookups cost more than the instalments saved,
                        // 13.1 instead of 8.1 ms at 2^28 records of 100 M keys)
                        const uint32_t grew = tot_occ - last_occ;
                        last_occ = tot_occ;
                        flush = tot_occ + grew + grew / 4 > kFlushAt || tot_ovf > kOvf / 2;
                        KTA_PHASE(1, 4);
                        if (failed || flush) stop = true;   // (the units in flight are issued again by the next pass, from u)
                    }
                }
            }
        }
        // the trip's remaining units (fast attempt: in flight and not empty) are merged before the verdict
        if (dynamic && !stop) {
#pragma unroll
            for (int s = 0; s + 1 < D; s++) merge(ring[s], par);
        }
        if (RANGES) {                                    // what is left of the wave's own pairs (less than a wave of them)
            tq = __builtin_amdgcn_readfirstlane(tq);
            KTA_LDS_ORDER();
            merge_one(todoq[lane < tq ? lane : 0u], lane < tq, par);
            tq = 0;
        }
        drain(par);
        if (dynamic && !flush) u = KTA_UNITS;              // (segments were handed out: every wave counted its own units)
        if (!failed && !flush) {                         // the last units are in: did everything fit?
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) claimed += __shfl_xor(claimed, off);
            if (lane == 0 && claimed) atomicAdd(&sh.occ[par], claimed);
            claimed = 0;
            lds_barrier();
            failed = __builtin_amdgcn_readfirstlane(sh.fail[0] | sh.fail[1] | sh.fail[2]) != 0u;
        }
        if (failed) {
            if (kHasCareful && careful) {
                if (BITMAP) {                            // handed to kta_alive_fallback from this instalment on
                    if (threadIdx.x == 0) {
                        fail_from[b] = inst_start;
                        fail_from[(1u << BLOG2) + (uint32_t)atomicAdd(&pool_ctl[POOL_FAILED], 1ull)] = b;
                    }
                    // (the instalments before this one ARE applied: what they changed belongs to the running count.  Round 5:
                    // this path returned without it, and a bucket that gave up after its first instalment left sum_all_alive
                    // short of the bit set — found by holding bench.py's c5 leg against the oracle)
                    add_running(delta, running, sh.w);
                    return;
                }
                // (table state never fails in careful mode: its last resort is the direct path)
            } else if (BITMAP && !RANGES) {
                // The plain attempt did not fit — nothing has been applied, nothing counted: the bucket goes on the list of
                // the RANGES instantiation, which runs next, and the buckets of this batch still to come skip their attempt.
                if (threadIdx.x == 0) {
                    fail_from[(2u << BLOG2) + (uint32_t)atomicAdd(&pool_ctl[POOL_RANGES], 1ull)] = b;
                    atomicMax(&pool_ctl[POOL_DENSE], 1ull);
                }
                return;
            } else if (RANGES && rho >= kMaxRho) {
                // Sixteen parts do not hold it either (a few sets take everything): the bucket goes to kta_alive_fallback, from
                // its first segment on.  What the passes so far applied stays applied — it is the final state of its slots, which
                // the fallback kernel, resolving everything again, finds as it should be — and counted.
                if (threadIdx.x == 0) {
                    fail_from[b] = 0u;
                    fail_from[(1u << BLOG2) + (uint32_t)atomicAdd(&pool_ctl[POOL_FAILED], 1ull)] = b;
                }
                add_running(delta, running, sh.w);
                return;
            } else {                                     // the pass did not fit: start over — in more parts (RANGES), or carefully (table state)
                lds_barrier();                           // everybody has seen the verdict
                for (uint32_t e = threadIdx.x; e < kEntries + kEntries / 2 + 2 * kOvf; e += kApplyThreads) s_val[e] = 0u;
                if (threadIdx.x == 0) {
                    sh.occ[0] = sh.occ[1] = sh.occ[2] = sh.ovf_n[0] = sh.ovf_n[1] = sh.ovf_n[2] = sh.fail[0] = sh.fail[1] = sh.fail[2] = 0;
                    sh.any_ovf = 0;
                    if (RANGES) atomicMax(&pool_ctl[POOL_DENSE], (unsigned long long)(rho + 1u));
                    else pool_ctl[POOL_DENSE] = 1ull;
                }
                mq = 0;
                tq = 0;
                claimed = 0;
                u = 0;
                if (RANGES) {
                    // (the parts applied so far hold the FINAL state of their slots: their slots' pairs were all in their
                    // passes.  Taking them again in smaller parts finds every bit as it should be and counts nothing.)
                    rho++;
                    ridx = 0;
                    lds_barrier();
                    continue;
                }
                careful = true;
                lds_barrier();
                continue;
            }
        }
        KTA_PHASE(1, 1);
        end_instalment();                                // (the one call site: the sweep is long)
        if (RANGES && !careful) {                        // the next part of the slot range, or done
            if (++ridx < (1u << rho)) continue;
            break;
        }
        if (u >= KTA_UNITS) break;
        inst_start = (u / chunks) * KTA_GS;
        tot_occ = tot_ovf = last_occ = 0;
    }
    add_running(delta, running, sh.w);
    // what the host's choice of kernel for the NEXT batch feeds on: pairs read, entries claimed (one per
    // distinct slot and instalment) — their ratio says how much of the batch died in LDS
    if (stats && threadIdx.x == 0) {
        atomicAdd(&stats[0], (unsigned long long)sh.pairs);
        atomicAdd(&stats[1], (unsigned long long)sh.claims);
#ifdef KTA_ALIVE_PHASES
        atomicAdd(&stats[2], (unsigned long long)sh.instalments);
        atomicAdd(&stats[3], (unsigned long long)sh.ovf_total);
#endif
    }
    KTA_PHASE(1, 2);
}

// Table state: the pool pairs (segments that were full) take the direct path.
__global__ __launch_bounds__(kWG) void kta_alive_pool_direct(const unsigned long long *__restrict__ pool,
                                                             const unsigned long long *__restrict__ pool_ctl,
                                                             uint64_t base_seq, const uint64_t *__restrict__ seq_col,
                                                             unsigned long long *__restrict__ table,
                                                             long long *__restrict__ running,
                                                             const uint32_t *__restrict__ skip_flag, WrittenList wl)
{
    __shared__ long long s_w[kWG / 64];
    const unsigned long long n = pool_ctl[POOL_CURSOR];
    if (n == 0ull || (skip_flag && *skip_flag)) return;
    long long delta = 0;
    for (unsigned long long k = (unsigned long long)blockIdx.x * kWG + threadIdx.x; k < n; k += (unsigned long long)gridDim.x * kWG) {
        const unsigned long long pr = pool[k];
        if (pr == 0ull) continue;
        const uint64_t idx = (uint64_t)((uint32_t)pr >> 1) - 1u;
        const uint64_t s = seq_col ? seq_col[idx] : base_seq + idx;
        delta += direct_update(table, (uint32_t)(pr >> 32), ((unsigned long long)(s + 1) << 1) | (pr & 1ull), wl);
    }
    add_running(delta, running, s_w);
}

// Bit set state: the buckets pass 2 gave up (fail_from[b] != kNoFail), exactly, whatever they hold: the
// bucket's region in sub-ranges of 2^14 slots, each resolved in a direct-indexed LDS array (the newest pair per
// slot) over ALL pairs of the bucket from segment fail_from[b] on plus the bucket's pool pairs.  Slow (every
// pass re-reads the bucket's pairs) and only ever needed by batches that defeat the sizes of pass 1 / 2.
// Order of two pairs of a slot: (segment w, producer v, in the pool?, position) — a segment's blocks go to the pool
// once it is full, in order, by the one consumer wave that owns the bucket: a pool pair is newer than every pair
// of its producer in the segment, and among a segment's pool pairs the pool index grows with the position.
template <int BLOG2>
__global__ __launch_bounds__(kApplyThreads) void kta_alive_fallback(const uint32_t *__restrict__ pairs,
                                                                    const uint32_t *__restrict__ counts, uint32_t cap,
                                                                    uint32_t W, const unsigned long long *__restrict__ pool,
                                                                    uint64_t pool_pairs, const unsigned long long *__restrict__ pool_ctl,
                                                                    const uint32_t *__restrict__ fail_from,
                                                                    const uint32_t *__restrict__ pool_hist,
                                                                    uint32_t *__restrict__ bitmap,
                                                                    long long *__restrict__ running,
                                                                    unsigned long long *__restrict__ failed_total)
{
    constexpr uint32_t RBITS = 32 - BLOG2;
    constexpr uint32_t kSub = 1u << 14;                  // slots per pass
    constexpr uint32_t kRanges = (1u << RBITS) / kSub;   // 256 sub-ranges of a bucket
    extern __shared__ __attribute__((aligned(128))) unsigned long long s_max[];   // kSub values: order << 1 | alive
    __shared__ long long s_w[kApplyWaves];
    __shared__ uint8_t s_has[kRanges];                   // the sub-ranges that hold a pair at all
    // One workgroup per bucket is launched; the buckets to resolve are few (pass 2 lists them behind fail_from's B words), so
    // they share the workgroups: each gets B / (their number) of them, 32 at most.  A bucket with pool pairs is a hot-key
    // bucket — its pairs sit in one or a few sub-ranges — and ONE of its workgroups resolves it; a bucket without, one that
    // careful mode gave up, holds pairs in all 256 sub-ranges, a pass over all its pairs for each: its workgroups share the
    // sub-ranges (twelve such buckets took 20 ms with one workgroup each while 244 CUs idled).
    const uint32_t nfail = (uint32_t)pool_ctl[POOL_FAILED];
    if (nfail == 0u) return;
    if (failed_total && blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(failed_total, (unsigned long long)nfail);   // kta_alive_pass_info
    constexpr uint32_t B = 1u << BLOG2;
    uint32_t share = B / (nfail < B ? nfail : B);
    share = share > 32u ? 32u : share;
    const uint32_t which = blockIdx.x / share, part = blockIdx.x % share;
    if (which >= nfail) return;
    const uint32_t b = fail_from[B + which], from = fail_from[b];
    if (from == kNoFail) return;                         // (never: the list holds the buckets that were given up)
    const uint32_t nparts = pool_hist[b] != 0u ? 1u : share;
    if (part >= nparts) return;
    // the pool in blocks of 16 pairs, each tagged with its bucket + 1 (pass 1: pool_tag_note): this bucket's blocks are found by
    // their tags, four of which a thread has in flight
    const uint32_t nblk = (uint32_t)(pool_ctl[POOL_CURSOR] / kBlk32);
    const uint32_t *tags = reinterpret_cast<const uint32_t *>(pool + pool_pairs);
    auto for_my_pool_pairs = [&](auto &&f) __attribute__((always_inline)) {      // f(pair, index in the pool)
        // a wave looks at 4 x 64 tags at a time (the four loads in flight together) and reads the blocks that are this bucket's
        // four at a time, sixteen lanes per block: one coalesced 128-byte read each
        const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
        for (uint32_t b0 = wave * 256u; b0 < nblk; b0 += 256u * kApplyWaves) {
            uint32_t t[4];
#pragma unroll
            for (uint32_t x = 0; x < 4u; x++) {
                const uint32_t blk = b0 + 64u * x + lane;
                t[x] = tags[blk < nblk ? blk : b0];
            }
#pragma unroll
            for (uint32_t x = 0; x < 4u; x++) {
                unsigned long long m = __builtin_amdgcn_ballot_w64(b0 + 64u * x + lane < nblk && t[x] == b + 1u);
                while (m != 0ull) {                      // (wave-uniform) the next four of them
                    uint32_t mine = 64u;                 // the block of this lane's group of sixteen: none
#pragma unroll
                    for (uint32_t g = 0; g < 4u; g++) {
                        if (m == 0ull) break;
                        const uint32_t bit = (uint32_t)__builtin_ctzll(m);
                        m &= m - 1ull;
                        if ((lane >> 4) == g) mine = bit;
                    }
                    if (mine < 64u) {
                        const unsigned long long k = (unsigned long long)(b0 + 64u * x + mine) * kBlk32 + (lane & 15u);
                        const unsigned long long pr = pool[k];
                        if (pr != 0ull) f(pr, k);
                    }
                }
            }
        }
    };
    uint32_t *region = bitmap + ((size_t)b << (RBITS - 5));
    long long delta = 0;
    // What sends a bucket here is mostly a handful of hot keys (their pairs overflow the segments into the pool): its
    // pairs then sit in very few of the 256 sub-ranges, and a pass reads ALL pairs of the bucket (its segments and its pool
    // blocks).  The first pass resolves the sub-range of the bucket's first pair — a hot key's, if there is one — and notes on
    // the way which other sub-ranges hold anything (every writer writes the same 1: no atomic); one more pass for each of them.
    __shared__ uint32_t s_first;
    for (uint32_t r = threadIdx.x; r < kRanges; r += kApplyThreads) s_has[r] = 0;
    for (uint32_t e = threadIdx.x; e < kSub; e += kApplyThreads) s_max[e] = 0ull;
    if (threadIdx.x == 0) {
        uint32_t first = 0;
        for (uint32_t w = from; w < W; w++)
            if (counts[(uint64_t)b * W + w] != 0u) {
                first = (pairs[((uint64_t)b * W + w) * cap] >> kPair32Shift) / kSub;
                break;
            }
        s_first = first;
    }
    __syncthreads();
    const uint32_t r_first = s_first;
    auto walk = [&](uint32_t r, bool note) __attribute__((always_inline)) {      // the newest pair of every slot of sub-range r
        for (uint32_t w = from; w < W; w++) {
            const uint32_t cnt = counts[(uint64_t)b * W + w];
            const uint32_t *seg = pairs + ((uint64_t)b * W + w) * cap;
            for (uint32_t k = threadIdx.x; k < cnt; k += kApplyThreads) {
                const uint32_t p32 = seg[k];
                const uint32_t h = p32 >> kPair32Shift;
                if (note) s_has[h / kSub] = 1;
                if (h / kSub != r) continue;
                const unsigned long long order = ((unsigned long long)(w * 256u + ((p32 >> 1) & 255u)) << 30) | k;
                atomicMax(&s_max[h % kSub], (order << 1) | (p32 & 1u));
            }
        }
        for_my_pool_pairs([&](unsigned long long pr, unsigned long long k) {
            const uint32_t hh = (uint32_t)(pr >> 32), lo = (uint32_t)pr;
            const uint32_t h = hh & ((1u << RBITS) - 1u);
            const uint32_t w = (lo >> 9) & 1023u;
            if (w < from) return;                        // (an instalment that pass 2 finished held no pool pairs: never true)
            if (note) s_has[h / kSub] = 1;
            if (h / kSub != r) return;
            const unsigned long long order = ((unsigned long long)(w * 256u + ((lo >> 1) & 255u)) << 30) | (1ull << 29) | k;
            atomicMax(&s_max[h % kSub], (order << 1) | (lo & 1u));
        });
    };
    auto apply_range = [&](uint32_t r) __attribute__((always_inline)) {          // ... into the bit set; leaves s_max empty
        for (uint32_t wd = threadIdx.x; wd < kSub / 32; wd += kApplyThreads) {
            uint32_t set_m = 0, clr_m = 0;
            for (uint32_t bit = 0; bit < 32; bit++) {
                const unsigned long long v = s_max[wd * 32 + bit];
                if (v) {
                    (v & 1ull ? set_m : clr_m) |= 1u << bit;
                    s_max[wd * 32 + bit] = 0ull;
                }
            }
            if (set_m | clr_m) {
                uint32_t *word = region + (size_t)r * (kSub / 32) + wd;
                const uint32_t old = *word, neu = (old | set_m) & ~clr_m;
                *word = neu;
                delta += (long long)__popc(neu) - (long long)__popc(old);
            }
        }
    };
    walk(r_first, true);
    __syncthreads();
    if (r_first % nparts == part) {
        apply_range(r_first);
    } else {                                             // (another workgroup's: only the notes were wanted)
        for (uint32_t e = threadIdx.x; e < kSub; e += kApplyThreads) s_max[e] = 0ull;
    }
    __syncthreads();
    for (uint32_t r = 0; r < kRanges; r++) {
        if (r == r_first || !s_has[r] || r % nparts != part) continue;   // (the same for every thread: s_has is not written any more)
        walk(r, false);
        __syncthreads();
        apply_range(r);
        __syncthreads();
    }
    add_running(delta, running, s_w);
}

// bit set state: table of bits -> count (sum_all_alive, metric.rs:282-284)
__global__ __launch_bounds__(kWG) void kta_bitmap_count(const uint4 *__restrict__ words, uint64_t n16, unsigned long long *out)
{
    __shared__ unsigned long long s_w[kWG / 64];
    unsigned long long cnt = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * kWG + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * kWG) {
        const uint4 v = words[i];
        cnt += __popc(v.x) + __popc(v.y) + __popc(v.z) + __popc(v.w);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) cnt += __shfl_xor(cnt, off);
    if ((threadIdx.x & 63u) == 0u) s_w[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long t = 0;
        for (int w = 0; w < kWG / 64; w++) t += s_w[w];
        if (t) atomicAdd(out, t);
    }
}

// tools/ubench_alive.hip times the kernels of a pair one by one: events between the launches (compiled out of the library)
#ifdef KTA_UBENCH_EVENTS
hipEvent_t g_ub_ev[4];
#define KTA_UB_MARK(i) (void)hipEventRecord(g_ub_ev[i], s)
#else
#define KTA_UB_MARK(i)
#endif

template <int BLOG2>
hipError_t launch_pair(const AliveColumns &c, uint64_t n, uint64_t base_seq, const AliveState &st,
                       const AlivePartitionPlan &pl, const AliveWorkspace &ws, uint64_t *stats, hipStream_t s, const AliveFuse *fuse)
{
    constexpr uint32_t B = 1u << BLOG2;
    unsigned long long *pp = reinterpret_cast<unsigned long long *>(ws.pairs);
    unsigned long long *pool = reinterpret_cast<unsigned long long *>(ws.pool);
    unsigned long long *ctl = reinterpret_cast<unsigned long long *>(ws.pool_ctl);
    long long *run = reinterpret_cast<long long *>(st.running);
    // pool control words, pool histogram and (seq column) the order flag: [ctl 2 x u64][hist B x u32][flag u32]
    hipError_t e = hipMemsetAsync(ws.pool_ctl, 0, POOL_WORDS * 8 + (size_t)B * 4 + 4, s);
    if (e != hipSuccess) return e;
    uint32_t *hist = reinterpret_cast<uint32_t *>(ctl + POOL_WORDS);
    uint32_t *flag = hist + B;
    const bool bitmap = st.bitmap != nullptr;
    if (bitmap != pl.pair32) return hipErrorInvalidValue;      // the plan sized the workspace for the other pair format
    // (the area a wave's slice of the bit set goes through doubles as the waves' miss queues: the larger of the two)
    constexpr size_t kSliceBytes = (size_t)(kSliceSets << (32 - BLOG2 - kSetLog2)) / 8, kQueueBytes = (size_t)kApplyWaves * kMissQueue * 8;
    const size_t lds2 = (size_t)kEntries * 6 + (size_t)kOvf * 8 + (size_t)((pl.segment_wgs + 31u) & ~31u) * 4 +
                        (kSliceBytes > kQueueBytes ? kSliceBytes : kQueueBytes);
    if (bitmap) {
        // 4-byte pairs; 4-byte column loads: any alignment of the columns will do
        const size_t lds0 = (size_t)B * kRing32 * 4 + (size_t)B * 6 + (size_t)kProducers * kGuard + (size_t)kConsumers * 64 * 4 + 16 + 64;
        if (fuse) {
            // both handlers in this pass: the scan's sums in LDS behind the rings (see FuseArgs)
            if (fuse->P > kFuseMaxP || (uint64_t)pl.tiles_per_wg * kTile >= (1ull << kFuseCntBits)) return hipErrorInvalidValue;
            const size_t lds1 = lds0 + (size_t)3 * kFuseSlots * 8 + (size_t)kPartWaves * 5 * 8;
            e = hipFuncSetAttribute(reinterpret_cast<const void *>(&kta_alive_partition32<BLOG2, true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1);
            if (e != hipSuccess) return e;
            KTA_UB_MARK(0);
            hipLaunchKernelGGL((kta_alive_partition32<BLOG2, true>), dim3(pl.segment_wgs), dim3(kPartThreads), lds1, s, c, n, pl.tiles_per_wg,
                               reinterpret_cast<uint32_t *>(pp), ws.counts, pl.cap, pool, pl.pool_pairs, ctl, hist,
                               FuseArgs{fuse->partition, fuse->ts_ms, fuse->P, fuse_replicas_log2(fuse->P), fuse->partials, fuse->row_len});
        } else {
            e = hipFuncSetAttribute(reinterpret_cast<const void *>(&kta_alive_partition32<BLOG2, false>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds0);
            if (e != hipSuccess) return e;
            KTA_UB_MARK(0);
            hipLaunchKernelGGL((kta_alive_partition32<BLOG2, false>), dim3(pl.segment_wgs), dim3(kPartThreads), lds0, s, c, n, pl.tiles_per_wg,
                               reinterpret_cast<uint32_t *>(pp), ws.counts, pl.cap, pool, pl.pool_pairs, ctl, hist, FuseArgs{nullptr, nullptr, 0, 0, nullptr, 0});
        }
        e = hipGetLastError();
        if (e != hipSuccess) return e;
        KTA_UB_MARK(1);
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(&kta_alive_apply<BLOG2, true, false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((kta_alive_apply<BLOG2, true, false>), dim3(B), dim3(kApplyThreads), lds2, s, pp, ws.counts, pl.cap,
                           pl.segment_wgs, 0u, base_seq, (const uint64_t *)nullptr, (unsigned long long *)nullptr, st.bitmap, run,
                           reinterpret_cast<unsigned long long *>(stats), hist, ws.fail_from, ctl, (const uint32_t *)nullptr,
                           WrittenList{nullptr, nullptr, 0});
        e = hipGetLastError();
        if (e != hipSuccess) return e;
        // the buckets that did not fit the table in one piece, in slot-range passes (returns at once when there are none)
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(&kta_alive_apply<BLOG2, true, true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((kta_alive_apply<BLOG2, true, true>), dim3(B), dim3(kApplyThreads), lds2, s, pp, ws.counts, pl.cap,
                           pl.segment_wgs, 0u, base_seq, (const uint64_t *)nullptr, (unsigned long long *)nullptr, st.bitmap, run,
                           reinterpret_cast<unsigned long long *>(stats), hist, ws.fail_from, ctl, (const uint32_t *)nullptr,
                           WrittenList{nullptr, nullptr, 0});
        e = hipGetLastError();
        if (e != hipSuccess) return e;
        KTA_UB_MARK(2);
        const size_t lds3 = (size_t)8 << 14;
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(&kta_alive_fallback<BLOG2>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((kta_alive_fallback<BLOG2>), dim3(B), dim3(kApplyThreads), lds3, s, reinterpret_cast<const uint32_t *>(pp),
                           ws.counts, pl.cap, pl.segment_wgs, pool, pl.pool_pairs, ctl, ws.fail_from, hist, st.bitmap, run,
                           reinterpret_cast<unsigned long long *>(ws.failed_total));
        KTA_UB_MARK(3);
        return hipGetLastError();
    }
    // table state: 6-byte pairs (a survivor's sequence number comes from its batch-local index: segment x range + index)
    if ((uint64_t)pl.tiles_per_wg * kTile > (1ull << kIdxBits48)) return hipErrorInvalidValue;   // (the plan keeps a workgroup's range inside a pair's index)
    const uint32_t *skip = c.seq ? flag : nullptr;       // (raised by pass 1 when the batch's seq column does not ascend)
    const size_t lds0 = (size_t)B * kRingW48 * 2 + (size_t)B * 8 + (size_t)kConsumers * 64 * 4 + 16;
    const size_t lds1 = lds0 + (fuse ? (size_t)3 * kFuseSlots * 8 + (size_t)kPartWaves * 5 * 8 : 0);
    FuseArgs fz{nullptr, nullptr, 0, 0, nullptr, 0};
    if (fuse) {
        if (fuse->P > kFuseMaxP || (uint64_t)pl.tiles_per_wg * kTile >= (1ull << kFuseCntBits)) return hipErrorInvalidValue;
        fz = FuseArgs{fuse->partition, fuse->ts_ms, fuse->P, fuse_replicas_log2(fuse->P), fuse->partials, fuse->row_len};
    }
    unsigned short *pp16 = reinterpret_cast<unsigned short *>(pp);
#define KTA_LAUNCH_P48(SEQ, FUSE)                                                                                                    \
    do {                                                                                                                             \
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(&kta_alive_partition48<BLOG2, SEQ, FUSE>),                            \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1);                                              \
        if (e != hipSuccess) return e;                                                                                               \
        hipLaunchKernelGGL((kta_alive_partition48<BLOG2, SEQ, FUSE>), dim3(pl.segment_wgs), dim3(kPartThreads), lds1, s, c, n,       \
                           pl.tiles_per_wg, pp16, ws.counts, pl.cap, pool, ctl, flag, fz);                                           \
    } while (0)
    KTA_UB_MARK(0);
    if (c.seq) {
        if (fuse) KTA_LAUNCH_P48(true, true);
        else KTA_LAUNCH_P48(true, false);
    } else {
        if (fuse) KTA_LAUNCH_P48(false, true);
        else KTA_LAUNCH_P48(false, false);
    }
#undef KTA_LAUNCH_P48
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    KTA_UB_MARK(1);
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(&kta_alive_apply<BLOG2, false, false>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
    if (e != hipSuccess) return e;
    unsigned long long *t = reinterpret_cast<unsigned long long *>(st.table);
    hipLaunchKernelGGL((kta_alive_apply<BLOG2, false, false>), dim3(B), dim3(kApplyThreads), lds2, s, pp, ws.counts, pl.cap,
                       pl.segment_wgs, pl.tiles_per_wg * kTile, base_seq, c.seq, t, (uint32_t *)nullptr, run,
                       reinterpret_cast<unsigned long long *>(stats), hist, (uint32_t *)nullptr, ctl, skip, st.written);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    KTA_UB_MARK(2);
    hipLaunchKernelGGL(kta_alive_pool_direct, dim3(256), dim3(kWG), 0, s, pool, ctl, base_seq, c.seq, t, run, skip, st.written);
    KTA_UB_MARK(3);
    return hipGetLastError();
}

} // namespace

const uint32_t *alive_order_flag(const AliveWorkspace &ws, int bucket_log2)
{
    return reinterpret_cast<const uint32_t *>(reinterpret_cast<const unsigned long long *>(ws.pool_ctl) + POOL_WORDS) +
           (1u << bucket_log2);
}

AlivePartitionPlan plan_alive_partition(uint64_t n, int req_wgs, int cu_count, bool pair32)
{
    AlivePartitionPlan pl;
    pl.pair32 = pair32;
    pl.bucket_log2 = 10u;
    pl.max_records = kAlivePartitionMax;
    if (n > pl.max_records) n = pl.max_records;
    if (n == 0) n = 1;
    // one partition workgroup per CU (its rings take most of a CU's LDS); workgroup w takes the tiles
    // [w * tiles_per_wg, (w + 1) * tiles_per_wg): a contiguous, older-to-newer range of the batch
    uint64_t wgs = req_wgs > 0 ? (uint64_t)req_wgs : (uint64_t)(cu_count > 0 ? cu_count : 256);
    if (wgs > kMaxSegWGs) wgs = kMaxSegWGs;
    const uint64_t ntiles = (n + kTile - 1) / kTile;
    const uint64_t want = (ntiles + kProducers - 1) / kProducers;     // at least one tile per producer wave
    if (wgs > want) wgs = want;
    pl.tiles_per_wg = (uint32_t)((ntiles + wgs - 1) / wgs);
    // table state: a pair48 holds its record's index inside the workgroup's range in kIdxBits48 bits (more workgroups if need be:
    // 2^28 records in 64 ranges at the least)
    if (!pair32 && (uint64_t)pl.tiles_per_wg * kTile > (1ull << kIdxBits48)) pl.tiles_per_wg = (uint32_t)((1ull << kIdxBits48) / kTile);
    pl.segment_wgs = (uint32_t)((ntiles + pl.tiles_per_wg - 1) / pl.tiles_per_wg);
    // a segment receives n / (W * B) pairs on average; 1/8 + 48 of slack (8 sigma at 2^26 records) before it
    // overflows into the pool, rounded up to what a wave reads with one load instruction (128 pairs)
    const uint64_t mean = n / ((uint64_t)pl.segment_wgs << pl.bucket_log2) + 1;
    // (and to what a wave of pass 2 reads with one load instruction of four pairs per lane: 256)
    pl.cap = (uint32_t)((mean + mean / 8 + 48 + 255) & ~255ull);
    pl.pair_words = ((uint64_t)pl.segment_wgs << pl.bucket_log2) * pl.cap * (pair32 ? 4 : 6) / 8;
    pl.count_words = (uint64_t)pl.segment_wgs << pl.bucket_log2;
    // the pool: at most one pair per record; bit set state: + a padded block per segment tail + what the consumer waves leave of
    // their last chunks (pool_tag_note), and behind the pairs one 4-byte tag per block of 16; table state: + what the producer
    // waves leave of theirs
    pl.pool_pairs = ((n + 15) & ~15ull) + (pair32 ? ((uint64_t)16 * pl.segment_wgs << pl.bucket_log2) + (uint64_t)kPoolChunk * kBlk32 * kConsumers * pl.segment_wgs
                                                  : (uint64_t)kPoolChunk48 * kProducers * pl.segment_wgs);
    pl.pool_words = pl.pool_pairs + (pair32 ? pl.pool_pairs / 32 + 1 : 0);
    pl.ctl_bytes = POOL_WORDS * 8 + ((size_t)4 << pl.bucket_log2) + 4;
    return pl;
}

hipError_t launch_alive_partitioned(const AliveColumns &c, uint64_t n, uint64_t base_seq, const AliveState &st,
                                    const AlivePartitionPlan &pl, const AliveWorkspace &ws, uint64_t *stats, hipStream_t s,
                                    const AliveFuse *fuse)
{
    return launch_pair<10>(c, n, base_seq, st, pl, ws, stats, s, fuse);
}

bool alive_fuse_possible(const AlivePartitionPlan &pl, uint32_t P)
{
    return P <= kFuseMaxP && (uint64_t)pl.tiles_per_wg * kTile < (1ull << kFuseCntBits);
}

hipError_t launch_bitmap_count(const uint32_t *bitmap, uint64_t *out, hipStream_t s)
{
    hipLaunchKernelGGL(kta_bitmap_count, dim3(2048), dim3(kWG), 0, s, reinterpret_cast<const uint4 *>(bitmap),
                       (uint64_t)(kAliveSlots / 128), reinterpret_cast<unsigned long long *>(out));
    return hipGetLastError();
}

} // namespace kta
