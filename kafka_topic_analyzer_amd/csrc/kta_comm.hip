// kta_comm.hip — the exchange step of a partition-sharded run, behind the C ABI (include/kta_hip.h):
// one context = one rank = one GPU, RCCL over xGMI.
//
// The reference is one process with one MessageMetrics and one BitSet (/root/reference/src/main.rs:77-82);
// its report reads them after the consume loop (src/main.rs:121-179).  Sharded by partition, every
// per-partition counter is complete on the rank that owns the partition (src/metric.rs:74-100) and the
// globals are sums and extrema, so the step that replaces "read the handler" is:
//
//   counters   ONE grouped RCCL launch on the compute stream: all-reduce SUM (u64) over
//              vec[0 : P*7 + 4] and all-reduce MAX (i64) over vec[P*7 + 4 : P*7 + 8] of the SNAPSHOT vector
//              (kta_finish_device), never of the live accumulator.
//   alive set  (-c) global and order dependent (src/metric.rs:262-264, 289-304): every rank's table holds
//              GLOBAL sequence numbers; rank r owns the slots [ceil(r 2^32 / R), ceil((r+1) 2^32 / R)).
//              Each rank exports the entries it ever wrote, one contiguous list per owner (<= 12 bytes per
//              distinct key hash), the lists travel to their owners in one grouped ncclSend / ncclRecv
//              (every link carries 1/R of the entries, all links in parallel), the owner merges them with
//              atomicMax and counts its range; the range counts are disjoint, so the SUM all-reduce above
//              turns them into the reference's sum_all_alive() (src/metric.rs:282-284).
//
// RCCL is bound at run time (dlopen, RTLD_LOCAL): a process that also imports PyTorch keeps PyTorch's own
// bundled RCCL apart from this one, and contexts that never exchange never load it.
#include "../../include/kta_hip.h"
#include "kta_kernels.h"

#include <dlfcn.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <string>
#include <vector>

hipStream_t kta_internal_stream(kta_ctx *ctx);
int kta_internal_device(kta_ctx *ctx);
void kta_internal_set_error(kta_ctx *ctx, const char *msg);
void **kta_internal_comm_slot(kta_ctx *ctx, void (*free_fn)(void *));
bool kta_internal_count_alive(kta_ctx *ctx);
bool kta_internal_alive_table(kta_ctx *ctx);
uint64_t *kta_internal_vec_out(kta_ctx *ctx);
uint32_t kta_internal_partitions(kta_ctx *ctx);
uint64_t *kta_internal_table(kta_ctx *ctx);
int64_t *kta_internal_running(kta_ctx *ctx);
bool kta_internal_written(kta_ctx *ctx, kta::WrittenList *out);

namespace {

// the slice of rccl.h this file uses (ABI of RCCL 2.x / NCCL 2.x)
typedef struct ncclComm *ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
enum { ncclSuccess = 0 };
enum { ncclInt64 = 4, ncclUint64 = 5, ncclUint32 = 3 };      // ncclDataType_t
enum { ncclSum = 0, ncclMax = 2 };                            // ncclRedOp_t

struct Rccl {
    void *lib = nullptr;
    int (*GetUniqueId)(ncclUniqueId *) = nullptr;
    int (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    int (*CommAbort)(ncclComm_t) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, ncclComm_t, hipStream_t) = nullptr;
    int (*Send)(const void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*Recv)(void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    std::string error;
};

// Bound once per process (std::call_once): contexts live on their own threads and may reach their first
// kta_comm_create together.  The handle is published only after every symbol is bound.
static void rccl_load(Rccl &r)
{
    const char *env = getenv("KTA_RCCL_LIBRARY");
    const bool only_env = getenv("KTA_RCCL_ONLY_ENV") != nullptr;  // tests: no default search paths
    const char *names[] = {env, only_env ? nullptr : "/opt/rocm/lib/librccl.so.1", only_env ? nullptr : "librccl.so.1",
                           only_env ? nullptr : "librccl.so"};
    void *lib = nullptr;
    std::string why;
    for (const char *n : names) {
        if (!n || !*n) continue;
        lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (lib) break;
        const char *e = dlerror();  // one call: dlerror() clears the message it returns
        if (why.empty() && e) why = e;
    }
    if (!lib) {
        r.error = std::string("RCCL not found (set KTA_RCCL_LIBRARY): ") + why;
        return;
    }
    bool ok = true;
    auto sym = [&](const char *name) {
        void *p = dlsym(lib, name);
        if (!p && ok) {
            ok = false;
            r.error = std::string("RCCL symbol missing: ") + name;
        }
        return p;
    };
    r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
    r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
    r.CommAbort = reinterpret_cast<decltype(r.CommAbort)>(dlsym(lib, "ncclCommAbort"));  // optional
    r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(sym("ncclAllReduce"));
    r.AllGather = reinterpret_cast<decltype(r.AllGather)>(sym("ncclAllGather"));
    r.Send = reinterpret_cast<decltype(r.Send)>(sym("ncclSend"));
    r.Recv = reinterpret_cast<decltype(r.Recv)>(sym("ncclRecv"));
    r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
    r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
    if (!ok) {
        dlclose(lib);
        return;
    }
    r.lib = lib;
}

Rccl *rccl()
{
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] { rccl_load(r); });
    return &r;
}

struct CommState {
    ncclComm_t comm = nullptr;
    int nranks = 1, rank = 0;
    uint64_t *d_counts = nullptr;      // [nranks] entries this rank sends to every owner, then [nranks x nranks] gathered, ..., the overflow flag
    uint64_t *h_pinned = nullptr;      // pinned: [nranks + nranks x nranks + 1] read back, then [nranks] offsets of the owners' lists
    uint64_t *d_scalar = nullptr;
    uint32_t *d_send_slots = nullptr, *d_recv_slots = nullptr;
    uint64_t *d_send_vals = nullptr, *d_recv_vals = nullptr;
    uint64_t send_cap = 0, recv_cap = 0;
    uint64_t last_sent = 0, last_received = 0;
    bool aborted = false;
};

void free_comm(void *p)
{
    CommState *st = static_cast<CommState *>(p);
    if (!st) return;
    if (st->comm && rccl()->CommDestroy) (void)rccl()->CommDestroy(st->comm);
    if (st->d_counts) (void)hipFree(st->d_counts);
    if (st->h_pinned) (void)hipHostFree(st->h_pinned);
    if (st->d_scalar) (void)hipFree(st->d_scalar);
    if (st->d_send_slots) (void)hipFree(st->d_send_slots);
    if (st->d_recv_slots) (void)hipFree(st->d_recv_slots);
    if (st->d_send_vals) (void)hipFree(st->d_send_vals);
    if (st->d_recv_vals) (void)hipFree(st->d_recv_vals);
    delete st;
}

int fail(kta_ctx *ctx, int code, const std::string &m)
{
    kta_internal_set_error(ctx, m.c_str());
    return code;
}

#define CH(ctx, call)                                                                                     \
    do {                                                                                                  \
        hipError_t e__ = (call);                                                                          \
        if (e__ != hipSuccess)                                                                            \
            return fail(ctx, e__ == hipErrorOutOfMemory ? KTA_ERR_NOMEM : KTA_ERR_HIP,                    \
                        std::string(#call) + ": " + hipGetErrorString(e__));                              \
    } while (0)

#define CN(ctx, call)                                                                                     \
    do {                                                                                                  \
        int r__ = (call);                                                                                 \
        if (r__ != ncclSuccess)                                                                           \
            return fail(ctx, KTA_ERR_COMM, std::string(#call) + ": " + rccl()->GetErrorString(r__));      \
    } while (0)

// first slot owned by rank r of R: owner(slot) == (slot * R) >> 32
uint64_t range_lo(int r, int R) { return (((uint64_t)r << 32) + (uint64_t)R - 1) / (uint64_t)R; }

int grow(kta_ctx *ctx, uint32_t **slots, uint64_t **vals, uint64_t *cap, uint64_t need)
{
    if (*cap >= need) return KTA_OK;
    if (*slots) (void)hipFree(*slots);
    if (*vals) (void)hipFree(*vals);
    *slots = nullptr;
    *vals = nullptr;
    *cap = 0;
    const uint64_t want = need + need / 8 + 1024;
    CH(ctx, hipMalloc((void **)slots, want * sizeof(uint32_t)));
    CH(ctx, hipMalloc((void **)vals, want * sizeof(uint64_t)));
    *cap = want;
    return KTA_OK;
}

// The alive-set half of the exchange.  Leaves this rank's table merged for its own hash range and the
// count of that range in the snapshot vector.
//
// What a rank sends is "the entries it ever wrote": the context keeps the list of slots it wrote for the first
// time (WrittenList), so counting per owner, exporting per owner and counting the own range are passes over that
// list (4 bytes per distinct key hash) with gathers from the table — not sweeps of the 32 GiB table, whose cost
// would not depend on how few keys the rank saw.  The sweeps remain as the fallback for a list that overflowed or
// a table somebody else wrote into (kta_alive_table_modified).  ONE host synchronisation remains in either form:
// the sizes of the sends and receives have to be known to the host that issues them.  The list's length never comes
// to the host (the kernels read it on the device; a list that overflowed raises a flag that travels with the counts),
// and what the host hands to the device afterwards (the offsets of the owners' lists) leaves from pinned memory.
int exchange_alive(kta_ctx *ctx, CommState *st)
{
    Rccl *R = rccl();
    hipStream_t s = kta_internal_stream(ctx);
    uint64_t *table = kta_internal_table(ctx);
    const int n = st->nranks;
    // what a rank contributes to the all-gather: its n counts and its overflow flag; what comes back: n such rows
    const size_t row = (size_t)n + 1, n_read = row + row * n;
    if (!st->d_counts) CH(ctx, hipMalloc((void **)&st->d_counts, (n_read + 2 * (size_t)n) * sizeof(uint64_t)));
    if (!st->h_pinned) CH(ctx, hipHostMalloc((void **)&st->h_pinned, (n_read + (size_t)n) * sizeof(uint64_t), hipHostMallocDefault));
    if (!st->d_scalar) CH(ctx, hipMalloc((void **)&st->d_scalar, sizeof(uint64_t)));
    kta::WrittenList wl;
    bool listed = kta_internal_written(ctx, &wl);
    if (n > 64) listed = false;                          // the list kernels keep one LDS counter per owner, 64 of them: sweep
    uint64_t *d_flag = st->d_counts + n, *d_owner_at = st->d_counts + n_read, *d_cursors = d_owner_at + n;
    const uint64_t *send = st->h_pinned, *matrix = st->h_pinned + row;          // matrix[r * row + o]: rank r's entries for owner o
    uint64_t *h_owner_at = st->h_pinned + n_read;
    // 1. how many entries does this rank hold for every owner
    for (int attempt = 0;; attempt++) {
        CH(ctx, hipMemsetAsync(st->d_counts, 0, row * sizeof(uint64_t), s));
        if (listed) {
            CH(ctx, kta::launch_written_count(wl, n, st->d_counts, d_flag, s));
        } else {
            for (int r = 0; r < n; r++)
                CH(ctx, kta::launch_alive_count_written_span(table, range_lo(r, n), range_lo(r + 1, n), st->d_counts + r, s));
        }
        CN(ctx, R->AllGather(st->d_counts, st->d_counts + row, row, ncclUint64, st->comm, s));
        CH(ctx, hipMemcpyAsync(st->h_pinned, st->d_counts, n_read * sizeof(uint64_t), hipMemcpyDeviceToHost, s));
        CH(ctx, hipStreamSynchronize(s));                // THE synchronisation of the exchange
        // A list that overflowed (only its device knew) counted its first entries only: that rank counts again, with
        // sweeps — and every rank gathers again, because all of them have to issue the same collectives: the flags
        // travel with the counts, so all ranks see the same ones and decide alike.
        bool any = false;
        for (int r = 0; r < n; r++) any |= matrix[(size_t)r * row + n] != 0;
        if (!any || attempt) break;
        if (send[n] != 0) listed = false;
    }
    // (the second gather cannot carry a flag of this rank: a list that overflowed is not listed the second time, and sweeps
    // raise none.  Should that ever change, a rank must not go on with counts that stop at the list's capacity: it would
    // leave alive entries behind without a word — it fails instead, and kta_exchange aborts the communicator for its peers)
    if (listed && send[n] != 0) {
        kta_internal_set_error(ctx, "exchange: the written list still reports an overflow after the recount");
        return KTA_ERR_INVALID;
    }
    uint64_t send_total = 0, recv_total = 0;
    std::vector<uint64_t> send_at(n), recv_at(n);
    for (int r = 0; r < n; r++) {
        send_at[r] = send_total;
        if (r != st->rank) send_total += send[r];
        recv_at[r] = recv_total;
        if (r != st->rank) recv_total += matrix[(size_t)r * row + st->rank];
    }
    int rc = grow(ctx, &st->d_send_slots, &st->d_send_vals, &st->send_cap, send_total);
    if (rc != KTA_OK) return rc;
    rc = grow(ctx, &st->d_recv_slots, &st->d_recv_vals, &st->recv_cap, recv_total);
    if (rc != KTA_OK) return rc;
    // 2. one contiguous list per owner (the rank's own range stays where it is)
    if (listed) {
        if (send_total) {
            for (int r = 0; r < n; r++) h_owner_at[r] = send_at[r];
            CH(ctx, hipMemcpyAsync(d_owner_at, h_owner_at, (size_t)n * sizeof(uint64_t), hipMemcpyHostToDevice, s));   // (pinned: no wait)
            CH(ctx, hipMemsetAsync(d_cursors, 0, (size_t)n * sizeof(uint64_t), s));
            CH(ctx, kta::launch_written_export(wl, table, n, st->rank, d_owner_at, d_cursors, st->d_send_slots, st->d_send_vals, s));
        }
    } else {
        for (int r = 0; r < n; r++) {
            if (r == st->rank || send[r] == 0) continue;
            CH(ctx, kta::launch_alive_export_span(table, range_lo(r, n), range_lo(r + 1, n), st->d_send_slots + send_at[r],
                                                  st->d_send_vals + send_at[r], st->d_scalar, send[r], s));
        }
    }
    // 3. every list to its owner: one grouped launch, all links at once
    CN(ctx, R->GroupStart());
    for (int r = 0; r < n; r++) {
        if (r == st->rank) continue;
        const uint64_t ns = send[r], nr = matrix[(size_t)r * row + st->rank];
        if (ns) {
            CN(ctx, R->Send(st->d_send_slots + send_at[r], ns, ncclUint32, r, st->comm, s));
            CN(ctx, R->Send(st->d_send_vals + send_at[r], ns, ncclUint64, r, st->comm, s));
        }
        if (nr) {
            CN(ctx, R->Recv(st->d_recv_slots + recv_at[r], nr, ncclUint32, r, st->comm, s));
            CN(ctx, R->Recv(st->d_recv_vals + recv_at[r], nr, ncclUint64, r, st->comm, s));
        }
    }
    CN(ctx, R->GroupEnd());
    // 4. the owner merges (last writer by global sequence number; new slots join its list) and counts its range
    if (recv_total)
        CH(ctx, kta::launch_alive_import(st->d_recv_slots, st->d_recv_vals, recv_total, table, kta_internal_running(ctx), wl, s));
    uint64_t *dst = kta_internal_vec_out(ctx) + (size_t)kta_internal_partitions(ctx) * KTA_NCOUNTERS + KTA_G_ALIVE_KEYS;
    if (listed) {
        // (the list may have grown by the import, past its capacity even: the kernels read its length on the device, and
        // the range itself is counted when the list is no longer complete)
        CH(ctx, kta::launch_written_alive_count(wl, table, range_lo(st->rank, n), range_lo(st->rank + 1, n), dst, s));
    } else {
        CH(ctx, kta::launch_alive_count_span(table, range_lo(st->rank, n), range_lo(st->rank + 1, n), dst, s));
    }
    st->last_sent = send_total;
    st->last_received = recv_total;
    return KTA_OK;
}

} // namespace

extern "C" {

int kta_comm_unique_id(uint8_t id[KTA_COMM_ID_BYTES])
{
    if (!id) return KTA_ERR_INVALID;
    Rccl *R = rccl();
    if (!R->lib) return KTA_ERR_COMM;
    ncclUniqueId u;
    if (R->GetUniqueId(&u) != ncclSuccess) return KTA_ERR_COMM;
    static_assert(sizeof(u) == KTA_COMM_ID_BYTES, "unique id size");
    memcpy(id, &u, sizeof u);
    return KTA_OK;
}

int kta_comm_create(kta_ctx *ctx, int nranks, int rank, const uint8_t id[KTA_COMM_ID_BYTES])
{
    if (!ctx || nranks < 1 || rank < 0 || rank >= nranks || (nranks > 1 && !id)) return KTA_ERR_INVALID;
    void **slot = kta_internal_comm_slot(ctx, free_comm);
    if (*slot) return fail(ctx, KTA_ERR_INVALID, "kta_comm_create: the context already has a communicator");
    if (nranks > 1 && kta_internal_count_alive(ctx) && !kta_internal_alive_table(ctx))
        return fail(ctx, KTA_ERR_INVALID, "kta_comm_create: a -c rank of a sharded run needs KTA_FLAG_ALIVE_TABLE (global sequence numbers)");
    CH(ctx, hipSetDevice(kta_internal_device(ctx)));
    CommState *st = new CommState();
    st->nranks = nranks;
    st->rank = rank;
    // KTA_COMM_FORCE_RCCL=1: a real one-rank communicator, so that a box with a single GPU still runs the
    // collectives of the exchange (tests, bench.py's forced mode)
    const char *force = getenv("KTA_COMM_FORCE_RCCL");
    if (nranks > 1 || (force && *force == '1')) {
        Rccl *R = rccl();
        if (!R->lib) {
            delete st;
            return fail(ctx, KTA_ERR_COMM, R->error);
        }
        ncclUniqueId u;
        if (id) memcpy(&u, id, sizeof u);
        else if (R->GetUniqueId(&u) != ncclSuccess) {
            delete st;
            return fail(ctx, KTA_ERR_COMM, "ncclGetUniqueId failed");
        }
        int r = R->CommInitRank(&st->comm, nranks, u, rank);
        if (r != ncclSuccess) {
            delete st;
            return fail(ctx, KTA_ERR_COMM, std::string("ncclCommInitRank: ") + R->GetErrorString(r));
        }
    }
    *slot = st;
    return KTA_OK;
}

int kta_comm_destroy(kta_ctx *ctx)
{
    if (!ctx) return KTA_ERR_INVALID;
    void **slot = kta_internal_comm_slot(ctx, free_comm);
    if (*slot) {
        (void)hipSetDevice(kta_internal_device(ctx));
        (void)hipStreamSynchronize(kta_internal_stream(ctx));
        free_comm(*slot);
        *slot = nullptr;
    }
    return KTA_OK;
}

// A rank that fails locally (a kernel launch, an allocation) before or between the collectives would leave
// its peers blocked inside theirs for ever: abort the communicator, which makes the peers' pending and
// later collectives fail with an error (ncclCommAbort; a library without it keeps the communicator).
static int exchange_failed(CommState *st, int rc)
{
    if (st->comm && rccl()->CommAbort) {
        (void)rccl()->CommAbort(st->comm);
        st->comm = nullptr;
        st->aborted = true;
    }
    return rc;
}

static int exchange_collectives(kta_ctx *ctx, CommState *st)
{
    CH(ctx, hipSetDevice(kta_internal_device(ctx)));
    if (kta_internal_count_alive(ctx) && kta_internal_alive_table(ctx)) {
        int rc = exchange_alive(ctx, st);
        if (rc != KTA_OK) return rc;
    }
    Rccl *R = rccl();
    hipStream_t s = kta_internal_stream(ctx);
    uint64_t *vec = kta_internal_vec_out(ctx);
    const size_t sum_words = (size_t)kta_internal_partitions(ctx) * KTA_NCOUNTERS + KTA_NSUM_GLOBALS;
    CN(ctx, R->GroupStart());
    CN(ctx, R->AllReduce(vec, vec, sum_words, ncclUint64, ncclSum, st->comm, s));
    CN(ctx, R->AllReduce(vec + sum_words, vec + sum_words, KTA_NGLOBALS - KTA_NSUM_GLOBALS, ncclInt64, ncclMax, st->comm, s));
    CN(ctx, R->GroupEnd());
    return KTA_OK;
}

int kta_exchange(kta_ctx *ctx)
{
    if (!ctx) return KTA_ERR_INVALID;
    void **slot = kta_internal_comm_slot(ctx, free_comm);
    CommState *st = static_cast<CommState *>(*slot);
    if (!st) return fail(ctx, KTA_ERR_INVALID, "kta_exchange without kta_comm_create");
    if (st->aborted) return fail(ctx, KTA_ERR_COMM, "kta_exchange: the communicator was aborted by an earlier failure");
    int rc = kta_finish_device(ctx);           // flush + snapshot (+ this rank's alive count)
    if (rc != KTA_OK) return exchange_failed(st, rc);
    if (!st->comm) return rc;                  // one rank without a communicator: the snapshot is the job's result
    rc = exchange_collectives(ctx, st);
    return rc == KTA_OK ? rc : exchange_failed(st, rc);
}

int kta_comm_allreduce_i64(kta_ctx *ctx, int64_t *host_values, size_t n, int op_max)
{
    if (!ctx || (n && !host_values)) return KTA_ERR_INVALID;
    void **slot = kta_internal_comm_slot(ctx, free_comm);
    CommState *st = static_cast<CommState *>(*slot);
    if (!st) return fail(ctx, KTA_ERR_INVALID, "kta_comm_allreduce_i64 without kta_comm_create");
    if (!st->comm || n == 0) return KTA_OK;
    CH(ctx, hipSetDevice(kta_internal_device(ctx)));
    hipStream_t s = kta_internal_stream(ctx);
    int64_t *d = nullptr;
    CH(ctx, hipMalloc((void **)&d, n * sizeof(int64_t)));
    hipError_t e = hipMemcpyAsync(d, host_values, n * sizeof(int64_t), hipMemcpyHostToDevice, s);
    int r = ncclSuccess;
    if (e == hipSuccess) r = rccl()->AllReduce(d, d, n, ncclInt64, op_max ? ncclMax : ncclSum, st->comm, s);
    if (e == hipSuccess && r == ncclSuccess) e = hipMemcpyAsync(host_values, d, n * sizeof(int64_t), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(d);
    if (r != ncclSuccess) return fail(ctx, KTA_ERR_COMM, std::string("ncclAllReduce: ") + rccl()->GetErrorString(r));
    if (e != hipSuccess) return fail(ctx, KTA_ERR_HIP, std::string("kta_comm_allreduce_i64: ") + hipGetErrorString(e));
    return KTA_OK;
}

int kta_comm_info(kta_ctx *ctx, int *nranks, int *rank, uint64_t *entries_sent, uint64_t *entries_received)
{
    if (!ctx) return KTA_ERR_INVALID;
    void **slot = kta_internal_comm_slot(ctx, free_comm);
    CommState *st = static_cast<CommState *>(*slot);
    if (!st) return fail(ctx, KTA_ERR_INVALID, "kta_comm_info without kta_comm_create");
    if (nranks) *nranks = st->nranks;
    if (rank) *rank = st->rank;
    if (entries_sent) *entries_sent = st->last_sent;
    if (entries_received) *entries_received = st->last_received;
    return KTA_OK;
}

} // extern "C"
