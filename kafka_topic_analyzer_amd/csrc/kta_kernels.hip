// kta_kernels.hip — hand-written gfx950 (MI355X / CDNA4) kernels for the per-record
// metric-accumulation hot path of kafka-topic-analyzer.  Integer/byte work, HBM-bound:
// no MFMA.  wave = 64 lanes, 256-thread workgroups, 16 B/lane coalesced loads,
// LDS-staged per-workgroup partial counters, deterministic integer reductions.
//
// Reference semantics (paths under /root/reference):
//   kta_metrics_scan   src/metric.rs:207-252  MessageMetrics::handle_message
//   kta_alive_update   src/metric.rs:289-304  LogCompactionInMemoryMetrics::handle_message
//                      src/metric.rs:256-260  fnv1a        src/fnv32.rs:92-101  FnvHasher::write
//   kta_alive_count    src/metric.rs:282-284  sum_all_alive
#include "kta_kernels.h"

#include <limits.h>

namespace kta {

// ---------------------------------------------------------------------------------------
// K1  metrics scan
// ---------------------------------------------------------------------------------------
//
// Input: four columns, 20 B/record (partition i32, key_len i32, val_len i32, ts_ms i64).
// Each lane owns 4 consecutive records per tile (one int4 from each i32 column, two
// longlong2 from the timestamp column: 5 x 16 B loads, fully coalesced: a wave reads
// 3 x 1 KiB + 2 KiB contiguous).  A tile is 256 lanes x 4 records; tiles are dealt
// round-robin to workgroups, and the next tile's loads are issued before the current
// tile is accumulated (register double buffer) to keep HBM requests in flight.
//
// Accumulation: per-partition partial counters live in LDS, replicated 2^rep_log2 times
// (replica = lane & (R-1)) so records of one partition that sit in the same wave do not
// serialise on one LDS address (a Kafka consumer delivers per-partition runs, and small
// P is common).  Three non-returning 64-bit LDS atomics per record — or per QUAD when the
// lane's four consecutive records share a partition (inside a per-partition run they do):
//     A += 1 | tombstone << 21 | key_null << 42      (three 21-bit counts in one word)
//     K += key_len (Some)      V += val_len (Some)
// LDS partials are flushed to this workgroup's row of the partial workspace every 2^18
// records (so the 21-bit fields cannot overflow) and at the end; a second tiny kernel
// folds the rows.  No floating point anywhere; integer adds commute, so results are
// independent of scheduling.
//
// Globals: min/max of ts_ms (-1 => 0 first, metric.rs:209) — the division by 1000
// (metric.rs:210) is monotone, so it is applied once to the extrema on the host — and
// min/max of key+value size over non-tombstones (metric.rs:249-251), both carried in
// registers and reduced with wave shuffles at the end.

constexpr uint32_t kCntBits = 21;
constexpr uint64_t kCntMask = (1ull << kCntBits) - 1;
constexpr uint32_t kFlushTiles = 256;          // 256 tiles x 1024 records = 2^18 records

struct Quad {
    int4 p, k, v;
    longlong2 t0, t1;
};

typedef int v4i __attribute__((ext_vector_type(4)));
typedef long long v2l __attribute__((ext_vector_type(2)));

// NT: non-temporal loads (the columns are streamed exactly once; keep them out of L2/MALL)
template <bool NT>
__device__ __forceinline__ void load_quad(Quad &q, const ScanColumns &c, uint64_t qi)
{
    if (NT) {
        const v4i p = __builtin_nontemporal_load(reinterpret_cast<const v4i *>(c.partition) + qi);
        const v4i k = __builtin_nontemporal_load(reinterpret_cast<const v4i *>(c.key_len) + qi);
        const v4i v = __builtin_nontemporal_load(reinterpret_cast<const v4i *>(c.val_len) + qi);
        const v2l t0 = __builtin_nontemporal_load(reinterpret_cast<const v2l *>(c.ts_ms) + 2 * qi);
        const v2l t1 = __builtin_nontemporal_load(reinterpret_cast<const v2l *>(c.ts_ms) + 2 * qi + 1);
        q.p = make_int4(p.x, p.y, p.z, p.w);
        q.k = make_int4(k.x, k.y, k.z, k.w);
        q.v = make_int4(v.x, v.y, v.z, v.w);
        q.t0 = make_longlong2(t0.x, t0.y);
        q.t1 = make_longlong2(t1.x, t1.y);
    } else {
        q.p = reinterpret_cast<const int4 *>(c.partition)[qi];
        q.k = reinterpret_cast<const int4 *>(c.key_len)[qi];
        q.v = reinterpret_cast<const int4 *>(c.val_len)[qi];
        q.t0 = reinterpret_cast<const longlong2 *>(c.ts_ms)[2 * qi];
        q.t1 = reinterpret_cast<const longlong2 *>(c.ts_ms)[2 * qi + 1];
    }
}

struct LaneState {
    long long tmin, tmax;
    uint32_t smin, smax;
    uint32_t bad;
};

constexpr uint32_t kHistBuckets = 34; // [0] None, [1] length 0, [2+k] 2^k <= length < 2^(k+1)
constexpr uint32_t kHistReps = 16;    // lane-replicated LDS counters per bucket

__device__ __forceinline__ uint32_t size_bucket(uint32_t is_null, uint32_t len)
{
    return is_null ? 0u : (len == 0u ? 1u : 2u + (31u - (uint32_t)__clz((int)len)));
}

// What one record contributes, derived in registers.
struct Rec {
    uint32_t part, tomb, knull, ks, vs;
    long long t;
    bool ok;
};

__device__ __forceinline__ Rec make_rec(uint32_t part, int32_t kl, int32_t vl, long long ts, uint32_t P, bool valid)
{
    Rec r;
    r.part = part;
    r.ok = valid && (part < P);           // unsigned compare also rejects negative ids
    r.tomb = (uint32_t)vl >> 31;          // payload None  (metric.rs:241-244)
    r.knull = (uint32_t)kl >> 31;         // key None      (metric.rs:227-230)
    r.ks = r.knull ? 0u : (uint32_t)kl;
    r.vs = r.tomb ? 0u : (uint32_t)vl;
    r.t = (ts == -1ll) ? 0ll : ts;        // to_millis() None -> unwrap_or(0)  (metric.rs:209)
    return r;
}

// global extrema in registers (metric.rs:56-72) + the bad-partition count
__device__ __forceinline__ void lane_extrema(const Rec &r, bool valid, LaneState &st)
{
    st.bad += (valid && !r.ok) ? 1u : 0u;
    if (r.ok) {
        st.tmin = r.t < st.tmin ? r.t : st.tmin;
        st.tmax = r.t > st.tmax ? r.t : st.tmax;
        if (!r.tomb) {                     // metric.rs:249-251
            const uint32_t sz = r.ks + r.vs; // < 2^32: both < 2^31
            st.smin = min(st.smin, sz);
            st.smax = max(st.smax, sz);
        }
    }
}

struct ScanLds {
    unsigned long long *A, *K, *V;     // counters
    long long *X;                      // ANALYTICS: [4][slots] signed-max arrays [~ts, ts, ~size, size]
    uint32_t *H;                       // ANALYTICS: [2][34][16] histogram counters
    uint32_t slots;
};

// LDS side of ONE record.
template <int VARIANT, bool ANALYTICS>
__device__ __forceinline__ void lds_record(const Rec &r, const ScanLds &L, uint32_t rep_log2, uint32_t rep)
{
    if (VARIANT == 9) return; // diagnostic: loads + register work only
    const uint32_t slot = (r.part << rep_log2) | rep;
    const unsigned long long a = 1ull | ((unsigned long long)r.tomb << kCntBits) |
                                 ((unsigned long long)r.knull << (2 * kCntBits));
    if (r.ok) {
        atomicAdd(&L.A[slot], a);
        atomicAdd(&L.K[slot], (unsigned long long)r.ks);
        atomicAdd(&L.V[slot], (unsigned long long)r.vs);
    }
    if (ANALYTICS && r.ok) {
        atomicMax(&L.X[slot], ~r.t);
        atomicMax(&L.X[L.slots + slot], r.t);
        if (!r.tomb) {
            const long long sz = (long long)(r.ks + r.vs);
            atomicMax(&L.X[2 * L.slots + slot], ~sz);
            atomicMax(&L.X[3 * L.slots + slot], sz);
        }
    }
}

template <bool ANALYTICS>
__device__ __forceinline__ void lds_histograms(const Rec &r, const ScanLds &L)
{
    if (ANALYTICS && r.ok) {
        const uint32_t hr = threadIdx.x & (kHistReps - 1u);
        atomicAdd(&L.H[size_bucket(r.knull, r.ks) * kHistReps + hr], 1u);
        atomicAdd(&L.H[(kHistBuckets + size_bucket(r.tomb, r.vs)) * kHistReps + hr], 1u);
    }
}

// The lane's 4 consecutive records of a tile.  A Kafka consumer delivers per-partition runs, so the
// four usually share a partition: then their contributions are combined in registers and cost one
// set of LDS atomics instead of four (4x fewer same-address conflicts inside a run).
template <int VARIANT, bool ANALYTICS>
__device__ __forceinline__ void accumulate_quad(const Quad &q, bool valid, uint32_t P, uint32_t rep_log2,
                                                uint32_t rep, const ScanLds &L, LaneState &st)
{
    Rec r[4] = {make_rec((uint32_t)q.p.x, q.k.x, q.v.x, q.t0.x, P, valid),
                make_rec((uint32_t)q.p.y, q.k.y, q.v.y, q.t0.y, P, valid),
                make_rec((uint32_t)q.p.z, q.k.z, q.v.z, q.t1.x, P, valid),
                make_rec((uint32_t)q.p.w, q.k.w, q.v.w, q.t1.y, P, valid)};
#pragma unroll
    for (int j = 0; j < 4; j++) lane_extrema(r[j], valid, st);
    if (VARIANT == 9) {
        st.smax = max(st.smax, r[0].part + r[1].part + r[2].part + r[3].part); // keep the loads alive
        return;
    }
#pragma unroll
    for (int j = 0; j < 4; j++) lds_histograms<ANALYTICS>(r[j], L);
    const bool uniform = r[0].ok && r[0].part == r[1].part && r[0].part == r[2].part && r[0].part == r[3].part;
    if (uniform) {
        const uint32_t slot = (r[0].part << rep_log2) | rep;
        const uint32_t tombs = r[0].tomb + r[1].tomb + r[2].tomb + r[3].tomb;
        const uint32_t knulls = r[0].knull + r[1].knull + r[2].knull + r[3].knull;
        atomicAdd(&L.A[slot], 4ull | ((unsigned long long)tombs << kCntBits) |
                                  ((unsigned long long)knulls << (2 * kCntBits)));
        atomicAdd(&L.K[slot], (unsigned long long)r[0].ks + r[1].ks + r[2].ks + r[3].ks);
        atomicAdd(&L.V[slot], (unsigned long long)r[0].vs + r[1].vs + r[2].vs + r[3].vs);
        if (ANALYTICS) {
            long long nt = LLONG_MIN, tx = LLONG_MIN, ns = LLONG_MIN, sx = LLONG_MIN;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                nt = ~r[j].t > nt ? ~r[j].t : nt;
                tx = r[j].t > tx ? r[j].t : tx;
                if (!r[j].tomb) {
                    const long long sz = (long long)(r[j].ks + r[j].vs);
                    ns = ~sz > ns ? ~sz : ns;
                    sx = sz > sx ? sz : sx;
                }
            }
            atomicMax(&L.X[slot], nt);
            atomicMax(&L.X[L.slots + slot], tx);
            if (tombs < 4u) {
                atomicMax(&L.X[2 * L.slots + slot], ns);
                atomicMax(&L.X[3 * L.slots + slot], sx);
            }
        }
        return;
    }
    // interleaved partitions: per-record atomics, spread over the replicas
#pragma unroll
    for (int j = 0; j < 4; j++) lds_record<VARIANT, ANALYTICS>(r[j], L, rep_log2, rep);
}

// ANALYTICS (opt-in, no reference counterpart — the additive outputs named by the project brief):
// log2 histograms of key and value sizes and per-partition timestamp / message-size extrema, kept
// in additional LDS arrays (extrema as signed-max arrays of [~ts, ts, ~size, size], histograms as
// u32 counters replicated 16x by lane) and flushed with the counters.
template <int VARIANT, bool NT, bool ANALYTICS>
__global__ __launch_bounds__(kWG) void kta_metrics_scan(ScanColumns c, uint64_t n, uint32_t P,
                                                        uint32_t rep_log2,
                                                        uint64_t *__restrict__ partials,
                                                        uint32_t row_len)
{
    extern __shared__ unsigned long long lds[];
    __shared__ long long s_red[kWG / 64][6];

    const uint32_t tid = threadIdx.x;
    const uint32_t slots = P << rep_log2;
    const uint32_t n_arrays = 3u;
    unsigned long long *sA = lds;
    unsigned long long *sK = lds + slots;
    unsigned long long *sV = lds + 2 * slots;
    long long *sX = reinterpret_cast<long long *>(lds + n_arrays * slots);          // ANALYTICS: [4][slots]
    uint32_t *sH = reinterpret_cast<uint32_t *>(lds + (n_arrays + 4) * slots);      // ANALYTICS: [2][34][16]
    const ScanLds L{sA, sK, sV, sX, sH, slots};

    for (uint32_t i = tid; i < n_arrays * slots; i += kWG) lds[i] = 0ull;
    if (ANALYTICS) {
        for (uint32_t i = tid; i < 4 * slots; i += kWG) sX[i] = LLONG_MIN;
        for (uint32_t i = tid; i < 2 * kHistBuckets * kHistReps; i += kWG) sH[i] = 0u;
    }
    __syncthreads();

    const uint32_t rep = tid & ((1u << rep_log2) - 1u);
    LaneState st;
    st.tmin = LLONG_MAX;
    st.tmax = LLONG_MIN;
    st.smin = 0xFFFFFFFFu; // never a real size (max real size is 2^32-2)
    st.smax = 0u;
    st.bad = 0u;

    uint64_t *row = partials + (uint64_t)blockIdx.x * row_len;
    bool first_flush = true;

    auto flush = [&]() {
        __syncthreads();
        for (uint32_t p = tid; p < P; p += kWG) {
            unsigned long long a = 0, k = 0, v = 0;
            for (uint32_t r = 0; r < (1u << rep_log2); r++) {
                const uint32_t s = (p << rep_log2) | r;
                a += sA[s];
                k += sK[s];
                v += sV[s];
                sA[s] = 0ull;
                sK[s] = 0ull;
                sV[s] = 0ull;
            }
            uint64_t *o = row + (uint64_t)p * kScanCols;
            const uint64_t c0 = a & kCntMask, c1 = (a >> kCntBits) & kCntMask,
                           c2 = (a >> (2 * kCntBits)) & kCntMask;
            if (first_flush) {
                o[0] = c0; o[1] = c1; o[2] = c2; o[3] = k; o[4] = v;
            } else {
                o[0] += c0; o[1] += c1; o[2] += c2; o[3] += k; o[4] += v;
            }
            if (ANALYTICS) {
                long long *x = reinterpret_cast<long long *>(row + (uint64_t)P * kScanCols + kScanGlobals) + 4ull * p;
                for (uint32_t j = 0; j < 4; j++) {
                    long long m = LLONG_MIN;
                    for (uint32_t r = 0; r < (1u << rep_log2); r++) {
                        const uint32_t s = j * slots + ((p << rep_log2) | r);
                        m = sX[s] > m ? sX[s] : m;
                        sX[s] = LLONG_MIN;
                    }
                    x[j] = (first_flush || m > x[j]) ? m : x[j];
                }
            }
        }
        if (ANALYTICS) {
            uint64_t *hrow = row + (uint64_t)P * kScanCols + kScanGlobals + 4ull * P;
            for (uint32_t bkt = tid; bkt < 2 * kHistBuckets; bkt += kWG) {
                uint64_t t = 0;
                for (uint32_t r = 0; r < kHistReps; r++) {
                    t += sH[bkt * kHistReps + r];
                    sH[bkt * kHistReps + r] = 0u;
                }
                hrow[bkt] = first_flush ? t : hrow[bkt] + t;
            }
        }
        first_flush = false;
        __syncthreads();
    };

    const uint64_t nquads = n >> 2;
    const uint64_t ntiles = (nquads + kWG - 1) / kWG;
    uint64_t tile = blockIdx.x;
    uint32_t since_flush = 0;

    Quad cur, nxt;
    uint64_t qi = tile * kWG + tid;
    bool cur_valid = (tile < ntiles) && (qi < nquads);
    if (cur_valid) load_quad<NT>(cur, c, qi);

    while (tile < ntiles) { // uniform per workgroup
        const uint64_t ntile = tile + gridDim.x;
        const uint64_t nqi = ntile * kWG + tid;
        const bool nxt_valid = (ntile < ntiles) && (nqi < nquads);
        if (nxt_valid) load_quad<NT>(nxt, c, nqi);

        accumulate_quad<VARIANT, ANALYTICS>(cur, cur_valid, P, rep_log2, rep, L, st);

        if (++since_flush == kFlushTiles) {
            flush();
            since_flush = 0;
        }
        cur = nxt;
        cur_valid = nxt_valid;
        tile = ntile;
    }

    // tail: the last n % 4 records, scalar, by the first lanes of workgroup 0
    if (blockIdx.x == 0) {
        const uint64_t i = (nquads << 2) + tid;
        const bool v = (tid < 3) && (i < n);
        const uint64_t ii = v ? i : 0;
        if (tid < 64) {
            uint32_t p = 0; int32_t kl = 0, vl = 0; long long ts = 0;
            if (v) { p = (uint32_t)c.partition[ii]; kl = c.key_len[ii]; vl = c.val_len[ii]; ts = c.ts_ms[ii]; }
            const Rec r = make_rec(p, kl, vl, ts, P, v);
            lane_extrema(r, v, st);
            lds_histograms<ANALYTICS>(r, L);
            lds_record<VARIANT, ANALYTICS>(r, L, rep_log2, rep);
        }
    }

    flush();

    // per-workgroup extrema: wave shuffle tree, then 4 waves through LDS
    long long tmin = st.tmin, tmax = st.tmax;
    long long smin = (long long)st.smin, smax = (long long)st.smax, bad = (long long)st.bad;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const long long a = __shfl_xor(tmin, off), b = __shfl_xor(tmax, off);
        const long long cmin = __shfl_xor(smin, off), cmax = __shfl_xor(smax, off);
        const long long d = __shfl_xor(bad, off);
        tmin = a < tmin ? a : tmin;
        tmax = b > tmax ? b : tmax;
        smin = cmin < smin ? cmin : smin;
        smax = cmax > smax ? cmax : smax;
        bad += d;
    }
    const uint32_t wave = tid >> 6;
    if ((tid & 63u) == 0u) {
        s_red[wave][0] = tmin; s_red[wave][1] = tmax; s_red[wave][2] = smin;
        s_red[wave][3] = smax; s_red[wave][4] = bad;
    }
    __syncthreads();
    if (tid == 0) {
        for (uint32_t w = 1; w < kWG / 64; w++) {
            tmin = s_red[w][0] < tmin ? s_red[w][0] : tmin;
            tmax = s_red[w][1] > tmax ? s_red[w][1] : tmax;
            smin = s_red[w][2] < smin ? s_red[w][2] : smin;
            smax = s_red[w][3] > smax ? s_red[w][3] : smax;
            bad += s_red[w][4];
        }
        uint64_t *g = row + (uint64_t)P * kScanCols;
        g[SG_TMIN] = (uint64_t)tmin;
        g[SG_TMAX] = (uint64_t)tmax;
        g[SG_SMIN] = (smin == 0xFFFFFFFFll) ? (uint64_t)LLONG_MAX : (uint64_t)smin;
        g[SG_SMAX] = (uint64_t)smax;
        g[SG_BAD] = (uint64_t)bad;
        g[SG_NREC] = 0;
        g[6] = 0;
        g[7] = 0;
    }
}

// ---------------------------------------------------------------------------------------
// K5  fold partial rows into the persistent counter vector
// ---------------------------------------------------------------------------------------
// Layout of vec: u64[P*7] (reference field order) followed by KTA_NGLOBALS globals
// (include/kta_hip.h).  grid = (column blocks, row slices); each thread sums one column of
// the partial workspace over its row slice (consecutive threads read consecutive words), then
// one device-scope integer atomic per (thread, output).  Integer atomics commute => exact.
__global__ __launch_bounds__(kWG) void kta_fold_partials(const uint64_t *__restrict__ partials,
                                                         uint32_t rows, uint32_t P,
                                                         uint64_t *__restrict__ vec, uint32_t row_len,
                                                         uint64_t *__restrict__ avec)
{
    const uint32_t base_len = P * kScanCols + kScanGlobals;
    const uint32_t col = blockIdx.x * kWG + threadIdx.x;
    if (col >= row_len) return;
    const uint32_t r0 = (uint32_t)(((uint64_t)rows * blockIdx.y) / gridDim.y);
    const uint32_t r1 = (uint32_t)(((uint64_t)rows * (blockIdx.y + 1)) / gridDim.y);
    if (r0 == r1) return;
    if (col >= base_len) {
        // analytics tail of the row: [P][4] signed-max extrema, then 2 x 34 histogram sums.
        // avec layout: [2 x 34 histogram][P][4 extrema]
        const uint32_t a = col - base_len;
        if (a < 4u * P) {
            long long m = LLONG_MIN;
            for (uint32_t r = r0; r < r1; r++) {
                const long long x = (long long)partials[(uint64_t)r * row_len + col];
                m = x > m ? x : m;
            }
            atomicMax(reinterpret_cast<long long *>(avec) + 2 * kHistBuckets + a, m);
        } else {
            unsigned long long t = 0;
            for (uint32_t r = r0; r < r1; r++) t += partials[(uint64_t)r * row_len + col];
            if (t) atomicAdd(reinterpret_cast<unsigned long long *>(avec) + (a - 4u * P), t);
        }
        return;
    }
    unsigned long long *v = reinterpret_cast<unsigned long long *>(vec);
    unsigned long long *g = v + (uint64_t)P * 7;
    if (col < P * kScanCols) {
        const uint32_t p = col / kScanCols, f = col % kScanCols;
        unsigned long long s = 0;
        for (uint32_t r = r0; r < r1; r++) s += partials[(uint64_t)r * row_len + col];
        if (s == 0) return;
        unsigned long long *o = v + (uint64_t)p * 7;
        switch (f) {
        case 0: // record count: total += s, alive += s, key_non_null += s, records += s
            atomicAdd(&o[0], s);
            atomicAdd(&o[2], s);
            atomicAdd(&o[4], s);
            atomicAdd(&g[2], s);
            break;
        case 1: // tombstones: tombstones += s, alive -= s   (alive = total - tombstones)
            atomicAdd(&o[1], s);
            atomicAdd(&o[2], 0ull - s);
            break;
        case 2: // key None: key_null += s, key_non_null -= s
            atomicAdd(&o[3], s);
            atomicAdd(&o[4], 0ull - s);
            break;
        case 3: atomicAdd(&o[5], s); break; // key_size_sum
        default: atomicAdd(&o[6], s); break; // value_size_sum
        }
    } else {
        const uint32_t gi = col - P * kScanCols;
        long long *gs = reinterpret_cast<long long *>(g);
        if (gi == SG_TMIN || gi == SG_SMIN) { // minima are kept bit-complemented: all four are MAX
            long long m = LLONG_MAX;
            for (uint32_t r = r0; r < r1; r++) {
                const long long x = (long long)partials[(uint64_t)r * row_len + col];
                m = x < m ? x : m;
            }
            atomicMax(&gs[gi == SG_TMIN ? 4 : 6], ~m);
        } else if (gi == SG_TMAX || gi == SG_SMAX) {
            long long m = LLONG_MIN;
            for (uint32_t r = r0; r < r1; r++) {
                const long long x = (long long)partials[(uint64_t)r * row_len + col];
                m = x > m ? x : m;
            }
            atomicMax(&gs[gi == SG_TMAX ? 5 : 7], m);
        } else if (gi == SG_BAD) {
            unsigned long long s = 0;
            for (uint32_t r = r0; r < r1; r++) s += partials[(uint64_t)r * row_len + col];
            if (s) atomicAdd(&g[0], s);
        }
    }
}

__global__ void kta_init_vector(uint64_t *vec, uint32_t P, uint64_t *avec)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t nc = P * 7;
    if (i < nc) vec[i] = 0;
    if (avec) {
        if (i < 2 * kHistBuckets) avec[i] = 0;
        if (i < 4 * P) avec[2 * kHistBuckets + i] = (uint64_t)LLONG_MIN;
    }
    if (i == 0) {
        uint64_t *g = vec + nc;
        g[0] = 0; g[1] = 0; g[2] = 0; g[3] = 0;  // SUM globals
        g[4] = (uint64_t)~LLONG_MAX;             // ~min ts: nothing seen
        g[5] = (uint64_t)LLONG_MIN;              // max ts
        g[6] = (uint64_t)~LLONG_MAX;             // ~smallest: u64::MAX in the reference (metric.rs:42)
        g[7] = 0;                                // largest (metric.rs:41)
    }
}

// ---------------------------------------------------------------------------------------
// K2 + K3  FNV + last-writer-wins alive table
// ---------------------------------------------------------------------------------------
// fnv32.rs:92-101: h = 0x811c9dc5; for each byte: h ^= byte; h *= 0x811c9dc5 (wrapping).
// The multiplier is the offset basis, not the FNV prime.

constexpr uint32_t kFnvInit = 0x811c9dc5u;
constexpr uint32_t kFnvMul = 0x811c9dc5u;

__device__ __forceinline__ uint32_t fnv_step(uint32_t h, uint32_t byte) { return (h ^ byte) * kFnvMul; }

// Hash `len` bytes starting at byte pointer k (any alignment) with aligned 4-byte loads.
// Reads only whole dwords that overlap the key, so it may touch up to 3 bytes before and
// after the key inside the same 4-byte words (the key blob is allocated padded to 16 B).
__device__ __forceinline__ uint32_t fnv32_global(const uint8_t *k, uint32_t len)
{
    uint32_t h = kFnvInit;
    if (len == 0) return h;
    const uintptr_t a = reinterpret_cast<uintptr_t>(k);
    const uint32_t *w = reinterpret_cast<const uint32_t *>(a & ~(uintptr_t)3);
    uint32_t skip = (uint32_t)(a & 3u);
    uint32_t word = *w++;
    word >>= 8u * skip;
    uint32_t avail = 4u - skip;
    while (true) {
        const uint32_t take = len < avail ? len : avail;
        if (take == 4u) {
            h = fnv_step(h, word & 0xFFu);
            h = fnv_step(h, (word >> 8) & 0xFFu);
            h = fnv_step(h, (word >> 16) & 0xFFu);
            h = fnv_step(h, word >> 24);
        } else {
            for (uint32_t j = 0; j < take; j++) {
                h = fnv_step(h, word & 0xFFu);
                word >>= 8;
            }
        }
        len -= take;
        if (len == 0) break;
        word = *w++;
        avail = 4u;
    }
    return h;
}

// table[h] = max(table[h], ((seq+1) << 1) | alive): the entry with the largest sequence
// number is the last writer in consumption order, which is exactly what sequential
// BitSet insert/remove leaves behind (metric.rs:273-280, 291-303).  0 == never written.
__global__ __launch_bounds__(kWG) void kta_alive_update(AliveColumns c, uint64_t n, uint64_t base_seq,
                                                        unsigned long long *__restrict__ table, WrittenList wl)
{
    const uint64_t stride = (uint64_t)gridDim.x * kWG;
    for (uint64_t i = (uint64_t)blockIdx.x * kWG + threadIdx.x; i < n; i += stride) {
        const int32_t kl = c.key_len[i];
        if (kl < 0) continue; // key None: ignored (metric.rs:302)
        const uint32_t h = fnv32_global(c.key_bytes + c.key_off[i], (uint32_t)kl);
        const uint64_t s = c.seq ? c.seq[i] : base_seq + i;
        const unsigned long long v = ((unsigned long long)(s + 1) << 1) | (c.val_len[i] >= 0 ? 1ull : 0ull);
        note_new_slot(wl, atomicMax(&table[h], v) == 0ull, h);
    }
}

// Variant with a RUNNING alive count.  atomicMax returns the previous entry; when this record
// replaces it (v > old) the number of alive slots changes by flag(v) - flag(old).  Updates to one
// address are serialised by the memory-side atomic unit, so over any interleaving the deltas
// telescope to flag(final) - flag(initial): the running sum is exactly sum_all_alive()
// (metric.rs:282-284) without scanning the 32 GiB table.
__global__ __launch_bounds__(kWG) void kta_alive_update_counting(AliveColumns c, uint64_t n, uint64_t base_seq,
                                                                 unsigned long long *__restrict__ table,
                                                                 long long *__restrict__ running, WrittenList wl)
{
    __shared__ long long s_w[kWG / 64];
    const uint64_t stride = (uint64_t)gridDim.x * kWG;
    long long delta = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * kWG + threadIdx.x; i < n; i += stride) {
        const int32_t kl = c.key_len[i];
        if (kl < 0) continue;
        const uint32_t h = fnv32_global(c.key_bytes + c.key_off[i], (uint32_t)kl);
        const uint64_t s = c.seq ? c.seq[i] : base_seq + i;
        const unsigned long long v = ((unsigned long long)(s + 1) << 1) | (c.val_len[i] >= 0 ? 1ull : 0ull);
        const unsigned long long old = atomicMax(&table[h], v);
        note_new_slot(wl, old == 0ull, h);
        if (v > old) delta += (long long)(v & 1ull) - (long long)(old & 1ull);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) delta += __shfl_xor(delta, off);
    if ((threadIdx.x & 63u) == 0u) s_w[threadIdx.x >> 6] = delta;
    __syncthreads();
    if (threadIdx.x == 0) {
        long long t = 0;
        for (int w = 0; w < kWG / 64; w++) t += s_w[w];
        if (t) atomicAdd(reinterpret_cast<unsigned long long *>(running), (unsigned long long)t);
    }
}

// Variant 2: the counting kernel with a pre-read and the batch walked from its END.  A record only needs
// the table if no later record owns its slot yet; walking the batch backwards lets the LAST record of a
// key arrive first, so the earlier ones (a compacted topic repeats its keys) find a larger entry with a
// plain load and skip the memory-side atomic, which is the scarce resource (section 3.3 of DESIGN.md).
// Exactness does not depend on the order or on what the load sees: entries only grow, so `seen >= v`
// proves that a later record has already been applied, and a stale smaller value merely costs the atomic.
__global__ __launch_bounds__(kWG) void kta_alive_update_filtered(AliveColumns c, uint64_t n, uint64_t base_seq,
                                                                 unsigned long long *__restrict__ table,
                                                                 long long *__restrict__ running,
                                                                 const uint32_t *__restrict__ only_if, WrittenList wl)
{
    __shared__ long long s_w[kWG / 64];
    if (only_if && *only_if == 0u) return;   // stands in for the partitioned pass only when that gave the batch up
    const uint64_t stride = (uint64_t)gridDim.x * kWG;
    long long delta = 0;
    for (uint64_t j = (uint64_t)blockIdx.x * kWG + threadIdx.x; j < n; j += stride) {
        const uint64_t i = n - 1 - j;                      // highest sequence numbers first
        const int32_t kl = c.key_len[i];
        if (kl < 0) continue;
        const uint32_t h = fnv32_global(c.key_bytes + c.key_off[i], (uint32_t)kl);
        const uint64_t s = c.seq ? c.seq[i] : base_seq + i;
        const unsigned long long v = ((unsigned long long)(s + 1) << 1) | (c.val_len[i] >= 0 ? 1ull : 0ull);
        const unsigned long long seen = __hip_atomic_load(&table[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (seen >= v) continue;
        const unsigned long long old = atomicMax(&table[h], v);
        note_new_slot(wl, old == 0ull, h);
        if (v > old) delta += (long long)(v & 1ull) - (long long)(old & 1ull);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) delta += __shfl_xor(delta, off);
    if ((threadIdx.x & 63u) == 0u) s_w[threadIdx.x >> 6] = delta;
    __syncthreads();
    if (threadIdx.x == 0) {
        long long t = 0;
        for (int w = 0; w < kWG / 64; w++) t += s_w[w];
        if (t) atomicAdd(reinterpret_cast<unsigned long long *>(running), (unsigned long long)t);
    }
}

// Ablation kernels (alive_variant 8 / 9): hash only (h -> scratch) and table update only
// (h <- scratch).  Used to attribute time; the pair is also a valid two-phase implementation.
__global__ __launch_bounds__(kWG) void kta_alive_hash_only(AliveColumns c, uint64_t n, uint32_t *__restrict__ hout)
{
    const uint64_t stride = (uint64_t)gridDim.x * kWG;
    for (uint64_t i = (uint64_t)blockIdx.x * kWG + threadIdx.x; i < n; i += stride) {
        const int32_t kl = c.key_len[i];
        hout[i] = kl < 0 ? 0u : fnv32_global(c.key_bytes + c.key_off[i], (uint32_t)kl);
    }
}

__global__ __launch_bounds__(kWG) void kta_alive_apply_only(AliveColumns c, uint64_t n, uint64_t base_seq,
                                                            const uint32_t *__restrict__ hin,
                                                            unsigned long long *__restrict__ table, WrittenList wl)
{
    const uint64_t stride = (uint64_t)gridDim.x * kWG;
    for (uint64_t i = (uint64_t)blockIdx.x * kWG + threadIdx.x; i < n; i += stride) {
        if (c.key_len[i] < 0) continue;
        const uint64_t s = c.seq ? c.seq[i] : base_seq + i;
        const unsigned long long v = ((unsigned long long)(s + 1) << 1) | (c.val_len[i] >= 0 ? 1ull : 0ull);
        note_new_slot(wl, atomicMax(&table[hin[i]], v) == 0ull, hin[i]);
    }
}

__global__ __launch_bounds__(kWG) void kta_fnv32(const uint8_t *key_bytes, const uint32_t *key_off,
                                                 const int32_t *key_len, uint64_t n, uint32_t *out)
{
    const uint64_t i = (uint64_t)blockIdx.x * kWG + threadIdx.x;
    if (i >= n) return;
    const int32_t kl = key_len[i];
    out[i] = kl < 0 ? 0u : fnv32_global(key_bytes + key_off[i], (uint32_t)kl);
}

// K4: popcount of the alive bits.  16 B/lane streaming read of the u64 table.
__global__ __launch_bounds__(kWG) void kta_alive_count(const ulonglong2 *__restrict__ table2,
                                                       uint64_t n_pairs, unsigned long long *out)
{
    __shared__ unsigned long long s_w[kWG / 64];
    const uint64_t stride = (uint64_t)gridDim.x * kWG;
    unsigned long long cnt = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * kWG + threadIdx.x; i < n_pairs; i += stride) {
        const ulonglong2 e = table2[i];
        cnt += (e.x & 1ull) + (e.y & 1ull);
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

// Alive bits of the slots [lo, hi) only (any bounds): what the owner of a hash range counts after the
// hash-range exchange of a multi-GPU run.  8 B/lane.
__global__ __launch_bounds__(kWG) void kta_alive_count_span(const uint64_t *__restrict__ table, uint64_t lo,
                                                            uint64_t hi, unsigned long long *out)
{
    __shared__ unsigned long long s_w[kWG / 64];
    const uint64_t stride = (uint64_t)gridDim.x * kWG;
    unsigned long long cnt = 0;
    for (uint64_t i = lo + (uint64_t)blockIdx.x * kWG + threadIdx.x; i < hi; i += stride) cnt += table[i] & 1ull;
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

// The same, but only when the written list is no longer complete (see launch_written_alive_count).
__global__ __launch_bounds__(kWG) void kta_alive_count_span_if_overflowed(WrittenList wl, const uint64_t *__restrict__ table, uint64_t lo,
                                                                          uint64_t hi, unsigned long long *out)
{
    __shared__ unsigned long long s_w[kWG / 64];
    if (*wl.n <= wl.cap) return;
    const uint64_t stride = (uint64_t)gridDim.x * kWG;
    unsigned long long cnt = 0;
    for (uint64_t i = lo + (uint64_t)blockIdx.x * kWG + threadIdx.x; i < hi; i += stride) cnt += table[i] & 1ull;
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

// Compact export of the entries ever written (value != 0): (slot u32, value u64) pairs.  A shard's
// table holds at most one entry per distinct key hash it has seen, so this is what partition-sharded
// GPUs exchange instead of the 32 GiB table.  Each workgroup sweeps a contiguous slab, stages hits in
// LDS and reserves output space with ONE device atomic per flush (not per hit).
constexpr uint32_t kExportStage = 2048;

__global__ __launch_bounds__(kWG) void kta_alive_export(const unsigned long long *__restrict__ table,
                                                        uint64_t n_slots, uint32_t slot_base,
                                                        uint32_t *__restrict__ out_slots,
                                                        unsigned long long *__restrict__ out_vals,
                                                        unsigned long long *__restrict__ counter, uint64_t cap)
{
    __shared__ uint32_t s_slot[kExportStage];
    __shared__ unsigned long long s_val[kExportStage];
    __shared__ uint32_t s_n;
    __shared__ unsigned long long s_base;
    const uint64_t per = (n_slots + gridDim.x - 1) / gridDim.x;
    const uint64_t lo = (uint64_t)blockIdx.x * per;
    const uint64_t hi = lo + per < n_slots ? lo + per : n_slots;
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    for (uint64_t base = lo; base < hi; base += kWG * 4) {     // uniform trip count per workgroup
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint64_t i = base + (uint64_t)u * kWG + threadIdx.x;
            const unsigned long long v = i < hi ? table[i] : 0ull;
            if (v) {
                const uint32_t at = atomicAdd(&s_n, 1u);         // LDS counter; stage holds 2048 >= 4 * 256
                s_slot[at] = slot_base + (uint32_t)i;
                s_val[at] = v;
            }
        }
        __syncthreads();
        const uint32_t n = s_n;
        if (n > kExportStage - kWG * 4 || base + kWG * 4 >= hi) {  // flush when the next round might overflow
            if (threadIdx.x == 0 && n) s_base = atomicAdd(counter, (unsigned long long)n);
            __syncthreads();
            for (uint32_t k = threadIdx.x; k < n; k += kWG)
                if (s_base + k < cap) {
                    out_slots[s_base + k] = s_slot[k];
                    out_vals[s_base + k] = s_val[k];
                }
            __syncthreads();
            if (threadIdx.x == 0) s_n = 0;
        }
        __syncthreads();
    }
}

// Merge foreign entries: table[slot] = max(table[slot], value), keeping the running alive count exact
// (same telescoping argument as kta_alive_update_counting).
__global__ __launch_bounds__(kWG) void kta_alive_import(const uint32_t *__restrict__ slots,
                                                        const unsigned long long *__restrict__ vals, uint64_t n,
                                                        unsigned long long *__restrict__ table,
                                                        long long *__restrict__ running, WrittenList wl)
{
    __shared__ long long s_w[kWG / 64];
    const uint64_t stride = (uint64_t)gridDim.x * kWG;
    long long delta = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * kWG + threadIdx.x; i < n; i += stride) {
        const unsigned long long v = vals[i];
        if (!v) continue;
        const unsigned long long old = atomicMax(&table[slots[i]], v);
        note_new_slot(wl, old == 0ull, slots[i]);
        if (v > old) delta += (long long)(v & 1ull) - (long long)(old & 1ull);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) delta += __shfl_xor(delta, off);
    if ((threadIdx.x & 63u) == 0u) s_w[threadIdx.x >> 6] = delta;
    __syncthreads();
    if (threadIdx.x == 0) {
        long long t = 0;
        for (int w = 0; w < kWG / 64; w++) t += s_w[w];
        if (t) atomicAdd(reinterpret_cast<unsigned long long *>(running), (unsigned long long)t);
    }
}

// number of entries ever written (value != 0)
__global__ __launch_bounds__(kWG) void kta_alive_count_written(const ulonglong2 *__restrict__ table2, uint64_t n_pairs,
                                                               unsigned long long *out)
{
    __shared__ unsigned long long s_w[kWG / 64];
    const uint64_t stride = (uint64_t)gridDim.x * kWG;
    unsigned long long cnt = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * kWG + threadIdx.x; i < n_pairs; i += stride) {
        const ulonglong2 e = table2[i];
        cnt += (e.x != 0ull) + (e.y != 0ull);
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

// entries ever written in the slots [lo, hi) (any bounds)
__global__ __launch_bounds__(kWG) void kta_alive_count_written_span(const uint64_t *__restrict__ table, uint64_t lo,
                                                                    uint64_t hi, unsigned long long *out)
{
    __shared__ unsigned long long s_w[kWG / 64];
    const uint64_t stride = (uint64_t)gridDim.x * kWG;
    unsigned long long cnt = 0;
    for (uint64_t i = lo + (uint64_t)blockIdx.x * kWG + threadIdx.x; i < hi; i += stride) cnt += table[i] != 0ull;
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

// table -> bitmap: one wave turns 64 consecutive entries into one u64 of bits via ballot.
__global__ __launch_bounds__(kWG) void kta_alive_bitmap(const unsigned long long *__restrict__ table,
                                                        uint64_t n_slots,
                                                        unsigned long long *__restrict__ bitmap64)
{
    const uint64_t stride = (uint64_t)gridDim.x * kWG;
    for (uint64_t i = (uint64_t)blockIdx.x * kWG + threadIdx.x; i < n_slots; i += stride) {
        const unsigned long long m = __ballot((table[i] & 1ull) != 0ull);
        if ((threadIdx.x & 63u) == 0u) bitmap64[i >> 6] = m;
    }
}

// ---------------------------------------------------------------------------------------
// launch wrappers
// ---------------------------------------------------------------------------------------

ScanPlan plan_scan(uint32_t P, uint64_t n, int cu_count, int req_workgroups, int req_variant, bool analytics)
{
    ScanPlan pl;
    // req_variant: low bits 0 = accumulate, 9 = loads only (diagnostic); +16 = non-temporal loads
    const int base = req_variant & 15;
    pl.analytics = analytics;
    pl.nontemporal = analytics ? true : (req_variant & 16) != 0;
    pl.variant = analytics ? 0u : (base == 9 ? 9u : 0u);
    const uint32_t arrays = 3u + (analytics ? 4u : 0u);
    const uint32_t hist_bytes = analytics ? 2u * kHistBuckets * kHistReps * 4u : 0u;
    // LDS budget per workgroup: 32 KiB (4 workgroups = 16 waves per CU can be resident); the
    // analytics kernel carries 7 arrays and gets 64 KiB (2 workgroups per CU) to keep the replication.
    const uint32_t budget_slots = ((analytics ? 64u : 32u) * 1024u - hist_bytes) / (8u * arrays);
    uint32_t rep_log2 = 0;
    while (rep_log2 < 6 && (P << (rep_log2 + 1)) <= budget_slots) rep_log2++;
    pl.rep_log2 = rep_log2;
    pl.lds_bytes = (P << rep_log2) * 8u * arrays + hist_bytes;
    pl.row_len = scan_row_len(P, analytics);
    const uint64_t ntiles = ((n >> 2) + kWG - 1) / kWG;
    // 3 workgroups (12 waves) per CU saturate HBM (measured: 2-3 per CU best, more is slower), and
    // every workgroup is resident at once, so the static round-robin tile deal stays balanced.
    // (a workgroup needing more than 40 KiB of LDS only fits twice per CU: use 2 per CU then)
    uint64_t wgs = req_workgroups > 0 ? (uint64_t)req_workgroups
                                      : (uint64_t)cu_count * (pl.lds_bytes > 40u * 1024u ? 2u : 3u);
    if (wgs > ntiles) wgs = ntiles;
    if (wgs < 1) wgs = 1;
    pl.workgroups = (uint32_t)wgs;
    return pl;
}

hipError_t launch_metrics_scan(const ScanPlan &pl, const ScanColumns &c, uint64_t n, uint32_t P,
                               uint64_t *partials, hipStream_t s)
{
    dim3 grid(pl.workgroups), block(kWG);
    // > 64 KiB of dynamic LDS (P > ~2700) must be opted into per kernel
#define KTA_SCAN(V, NT, AN)                                                                                     \
    do {                                                                                                        \
        if (pl.lds_bytes > 48u * 1024u) {                                                                       \
            hipError_t ea = hipFuncSetAttribute(reinterpret_cast<const void *>(&kta_metrics_scan<V, NT, AN>),   \
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl.lds_bytes);  \
            if (ea != hipSuccess) return ea;                                                                    \
        }                                                                                                       \
        hipLaunchKernelGGL((kta_metrics_scan<V, NT, AN>), grid, block, pl.lds_bytes, s, c, n, P, pl.rep_log2,   \
                           partials, pl.row_len);                                                               \
    } while (0)
    if (pl.analytics) {
        KTA_SCAN(0, true, true);
        return hipGetLastError();
    }
    switch (pl.variant) {
    case 9: if (pl.nontemporal) KTA_SCAN(9, true, false); else KTA_SCAN(9, false, false); break;
    default: if (pl.nontemporal) KTA_SCAN(0, true, false); else KTA_SCAN(0, false, false); break;
    }
#undef KTA_SCAN
    return hipGetLastError();
}

hipError_t launch_fold_partials(const uint64_t *partials, uint32_t rows, uint32_t P, uint64_t *vec,
                                uint32_t row_len, uint64_t *avec, hipStream_t s)
{
    uint32_t slices = rows < 32 ? rows : 32;
    dim3 grid((row_len + kWG - 1) / kWG, slices), block(kWG);
    hipLaunchKernelGGL(kta_fold_partials, grid, block, 0, s, partials, rows, P, vec, row_len, avec);
    return hipGetLastError();
}

hipError_t launch_init_vector(uint64_t *vec, uint32_t P, uint64_t *avec, hipStream_t s)
{
    const uint32_t n = P * 7 + 1;
    hipLaunchKernelGGL(kta_init_vector, dim3((n + 255) / 256), dim3(256), 0, s, vec, P, avec);
    return hipGetLastError();
}

hipError_t launch_alive_update(const AliveColumns &c, uint64_t n, uint64_t base_seq, uint64_t *table,
                               int workgroups, int variant, uint32_t *scratch, int64_t *running, hipStream_t s,
                               const uint32_t *only_if, const WrittenList &wl)
{
    uint64_t wgs = (n + kWG - 1) / kWG;
    const uint64_t cap = workgroups > 0 ? (uint64_t)workgroups : 256ull * 8ull;
    if (wgs > cap) wgs = cap;
    if (wgs < 1) wgs = 1;
    unsigned long long *t = reinterpret_cast<unsigned long long *>(table);
    if (variant == 8 && scratch) {
        hipLaunchKernelGGL(kta_alive_hash_only, dim3((uint32_t)wgs), dim3(kWG), 0, s, c, n, scratch);
    } else if (variant == 9 && scratch) {
        hipLaunchKernelGGL(kta_alive_apply_only, dim3((uint32_t)wgs), dim3(kWG), 0, s, c, n, base_seq, scratch, t, wl);
    } else if (variant == 2 && running) {
        hipLaunchKernelGGL(kta_alive_update_filtered, dim3((uint32_t)wgs), dim3(kWG), 0, s, c, n, base_seq, t,
                           reinterpret_cast<long long *>(running), only_if, wl);
    } else if (variant == 1 && running) {
        hipLaunchKernelGGL(kta_alive_update_counting, dim3((uint32_t)wgs), dim3(kWG), 0, s, c, n, base_seq, t,
                           reinterpret_cast<long long *>(running), wl);
    } else {
        hipLaunchKernelGGL(kta_alive_update, dim3((uint32_t)wgs), dim3(kWG), 0, s, c, n, base_seq, t, wl);
    }
    return hipGetLastError();
}

hipError_t launch_alive_count(const uint64_t *table, uint64_t n_slots, uint64_t *out, hipStream_t s)
{
    hipError_t e = hipMemsetAsync(out, 0, sizeof(uint64_t), s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kta_alive_count, dim3(256 * 8), dim3(kWG), 0, s,
                       reinterpret_cast<const ulonglong2 *>(table), n_slots / 2,
                       reinterpret_cast<unsigned long long *>(out));
    return hipGetLastError();
}

hipError_t launch_alive_count_span(const uint64_t *table, uint64_t lo, uint64_t hi, uint64_t *out, hipStream_t s)
{
    hipError_t e = hipMemsetAsync(out, 0, sizeof(uint64_t), s);
    if (e != hipSuccess || lo >= hi) return e;
    hipLaunchKernelGGL(kta_alive_count_span, dim3(256 * 8), dim3(kWG), 0, s, table, lo, hi,
                       reinterpret_cast<unsigned long long *>(out));
    return hipGetLastError();
}

hipError_t launch_alive_count_written_span(const uint64_t *table, uint64_t lo, uint64_t hi, uint64_t *out, hipStream_t s)
{
    hipError_t e = hipMemsetAsync(out, 0, sizeof(uint64_t), s);
    if (e != hipSuccess || lo >= hi) return e;
    hipLaunchKernelGGL(kta_alive_count_written_span, dim3(256 * 8), dim3(kWG), 0, s, table, lo, hi,
                       reinterpret_cast<unsigned long long *>(out));
    return hipGetLastError();
}

hipError_t launch_alive_count_written(const uint64_t *table, uint64_t n_slots, uint64_t *out, hipStream_t s)
{
    hipError_t e = hipMemsetAsync(out, 0, sizeof(uint64_t), s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kta_alive_count_written, dim3(256 * 8), dim3(kWG), 0, s,
                       reinterpret_cast<const ulonglong2 *>(table), n_slots / 2,
                       reinterpret_cast<unsigned long long *>(out));
    return hipGetLastError();
}

hipError_t launch_alive_export(const uint64_t *table, uint64_t n_slots, uint32_t *out_slots, uint64_t *out_vals,
                               uint64_t *counter, uint64_t cap, hipStream_t s)
{
    return launch_alive_export_span(table, 0, n_slots, out_slots, out_vals, counter, cap, s);
}

hipError_t launch_alive_export_span(const uint64_t *table, uint64_t lo, uint64_t hi, uint32_t *out_slots,
                                    uint64_t *out_vals, uint64_t *counter, uint64_t cap, hipStream_t s)
{
    hipError_t e = hipMemsetAsync(counter, 0, sizeof(uint64_t), s);
    if (e != hipSuccess || lo >= hi) return e;
    const uint64_t n_slots = hi - lo;
    hipLaunchKernelGGL(kta_alive_export, dim3(256 * 16), dim3(kWG), 0, s,
                       reinterpret_cast<const unsigned long long *>(table + lo), n_slots, (uint32_t)lo, out_slots,
                       reinterpret_cast<unsigned long long *>(out_vals),
                       reinterpret_cast<unsigned long long *>(counter), cap);
    return hipGetLastError();
}

hipError_t launch_alive_import(const uint32_t *slots, const uint64_t *vals, uint64_t n, uint64_t *table,
                               int64_t *running, const WrittenList &wl, hipStream_t s)
{
    if (n == 0) return hipSuccess;
    uint64_t wgs = (n + kWG - 1) / kWG;
    if (wgs > 256 * 8) wgs = 256 * 8;
    hipLaunchKernelGGL(kta_alive_import, dim3((uint32_t)wgs), dim3(kWG), 0, s, slots,
                       reinterpret_cast<const unsigned long long *>(vals), n,
                       reinterpret_cast<unsigned long long *>(table), reinterpret_cast<long long *>(running), wl);
    return hipGetLastError();
}

// ---- the exchange over the written list -----------------------------------------------------------------
constexpr int kMaxOwners = 64;

// The list's length is read on the device (the host does not wait for it): a list that overflowed its capacity says so
// in *overflow, and the counts then cover its first `cap` entries only — the host, which reads the flag together with the
// counts, falls back to the sweeps.
__global__ __launch_bounds__(kWG) void kta_written_count(WrittenList wl, uint32_t nranks, unsigned long long *counts, unsigned long long *overflow)
{
    __shared__ uint32_t s_c[kMaxOwners];
    if (threadIdx.x < kMaxOwners) s_c[threadIdx.x] = 0;
    __syncthreads();
    const uint64_t have = *wl.n, n = have < wl.cap ? have : wl.cap;
    if (blockIdx.x == 0 && threadIdx.x == 0 && have > wl.cap) *overflow = 1ull;
    for (uint64_t i = (uint64_t)blockIdx.x * kWG + threadIdx.x; i < n; i += (uint64_t)gridDim.x * kWG)
        atomicAdd(&s_c[(uint32_t)(((uint64_t)wl.slots[i] * nranks) >> 32)], 1u);
    __syncthreads();
    if (threadIdx.x < nranks && s_c[threadIdx.x]) atomicAdd(&counts[threadIdx.x], (unsigned long long)s_c[threadIdx.x]);
}

// Every workgroup takes a contiguous piece of the list, counts its entries per owner in LDS, reserves their places
// in the owners' lists with ONE device atomic per owner, and writes (slot, table[slot]).
__global__ __launch_bounds__(kWG) void kta_written_export(WrittenList wl, const unsigned long long *__restrict__ table,
                                                          uint32_t nranks, uint32_t skip_rank,
                                                          const unsigned long long *__restrict__ owner_at,
                                                          unsigned long long *__restrict__ cursors, uint32_t *__restrict__ out_slots,
                                                          unsigned long long *__restrict__ out_vals)
{
    __shared__ uint32_t s_c[kMaxOwners];
    __shared__ unsigned long long s_base[kMaxOwners];
    const uint64_t have = *wl.n, n = have < wl.cap ? have : wl.cap;     // (the length is read on the device, as in kta_written_count)
    const uint64_t per = (n + gridDim.x - 1) / gridDim.x;
    const uint64_t lo = (uint64_t)blockIdx.x * per, hi = lo + per < n ? lo + per : n;
    constexpr uint64_t kChunk = (uint64_t)kWG * 8;
    for (uint64_t c0 = lo; c0 < hi; c0 += kChunk) {                    // uniform trip count per workgroup
        if (threadIdx.x < kMaxOwners) s_c[threadIdx.x] = 0;
        __syncthreads();
        uint32_t slot[8], own[8], at[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const uint64_t i = c0 + (uint64_t)u * kWG + threadIdx.x;
            slot[u] = i < hi ? wl.slots[i] : 0u;
            own[u] = i < hi ? (uint32_t)(((uint64_t)slot[u] * nranks) >> 32) : skip_rank;
            at[u] = own[u] != skip_rank ? atomicAdd(&s_c[own[u]], 1u) : 0u;
        }
        __syncthreads();
        if (threadIdx.x < nranks && s_c[threadIdx.x]) s_base[threadIdx.x] = atomicAdd(&cursors[threadIdx.x], (unsigned long long)s_c[threadIdx.x]);
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if (own[u] == skip_rank) continue;
            const unsigned long long k = owner_at[own[u]] + s_base[own[u]] + at[u];
            out_slots[k] = slot[u];
            out_vals[k] = table[slot[u]];
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(kWG) void kta_written_alive_count(WrittenList wl, const unsigned long long *__restrict__ table,
                                                               uint64_t lo, uint64_t hi, unsigned long long *out)
{
    __shared__ unsigned long long s_w[kWG / 64];
    unsigned long long cnt = 0;
    const uint64_t len = *wl.n;                           // (may have grown by the import that ran just before)
    if (len > wl.cap) return;                             // the list is not complete: kta_alive_count_span_if_overflowed counts
    for (uint64_t i = (uint64_t)blockIdx.x * kWG + threadIdx.x; i < len; i += (uint64_t)gridDim.x * kWG) {
        const uint64_t slot = wl.slots[i];
        if (slot >= lo && slot < hi) cnt += table[slot] & 1ull;
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

hipError_t launch_written_count(const WrittenList &wl, int nranks, uint64_t *counts, uint64_t *overflow, hipStream_t s)
{
    // (one LDS counter per owner: kta_comm.hip takes the sweeps — launch_alive_count_written_span / _export_span — for more
    // than kMaxOwners ranks and never comes here with them)
    if (nranks > kMaxOwners) return hipErrorInvalidValue;
    // (the grid does not depend on the list's length, which only the device knows: 4 workgroups per CU walk it)
    hipLaunchKernelGGL(kta_written_count, dim3(1024), dim3(kWG), 0, s, wl, (uint32_t)nranks,
                       reinterpret_cast<unsigned long long *>(counts), reinterpret_cast<unsigned long long *>(overflow));
    return hipGetLastError();
}

hipError_t launch_written_export(const WrittenList &wl, const uint64_t *table, int nranks, int skip_rank,
                                 const uint64_t *owner_at, uint64_t *cursors, uint32_t *out_slots, uint64_t *out_vals, hipStream_t s)
{
    if (nranks > kMaxOwners) return hipErrorInvalidValue;
    hipLaunchKernelGGL(kta_written_export, dim3(1024), dim3(kWG), 0, s, wl, reinterpret_cast<const unsigned long long *>(table),
                       (uint32_t)nranks, (uint32_t)skip_rank, reinterpret_cast<const unsigned long long *>(owner_at),
                       reinterpret_cast<unsigned long long *>(cursors), out_slots, reinterpret_cast<unsigned long long *>(out_vals));
    return hipGetLastError();
}

// The owner's count of its hash range over the written list — or, when the list has overflowed by now (the import may
// have added to it; only the device knows), over the range itself: both kernels are launched, one of them returns at once.
hipError_t launch_written_alive_count(const WrittenList &wl, const uint64_t *table, uint64_t lo, uint64_t hi,
                                      uint64_t *out, hipStream_t s)
{
    hipError_t e = hipMemsetAsync(out, 0, sizeof(uint64_t), s);
    if (e != hipSuccess || lo >= hi) return e;
    hipLaunchKernelGGL(kta_written_alive_count, dim3(1024), dim3(kWG), 0, s, wl,
                       reinterpret_cast<const unsigned long long *>(table), lo, hi, reinterpret_cast<unsigned long long *>(out));
    hipLaunchKernelGGL(kta_alive_count_span_if_overflowed, dim3(256 * 8), dim3(kWG), 0, s, wl, table, lo, hi,
                       reinterpret_cast<unsigned long long *>(out));
    return hipGetLastError();
}

hipError_t launch_alive_bitmap(const uint64_t *table, uint64_t n_slots, uint32_t *bitmap, hipStream_t s)
{
    hipLaunchKernelGGL(kta_alive_bitmap, dim3(256 * 8), dim3(kWG), 0, s,
                       reinterpret_cast<const unsigned long long *>(table), n_slots,
                       reinterpret_cast<unsigned long long *>(bitmap));
    return hipGetLastError();
}

hipError_t launch_fnv32(const uint8_t *key_bytes, const uint32_t *key_off, const int32_t *key_len,
                        uint64_t n, uint32_t *out, hipStream_t s)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(kta_fnv32, dim3((uint32_t)((n + kWG - 1) / kWG)), dim3(kWG), 0, s, key_bytes,
                       key_off, key_len, n, out);
    return hipGetLastError();
}

} // namespace kta
