// Test double of RCCL (test infrastructure, not product code): the slice of the API csrc/kta_comm.hip binds,
// for several ranks living as THREADS of one process on one GPU — the build container reaches a single
// device and RCCL refuses two ranks on the same device.  Every call drains the caller's stream and then
// moves the bytes through host memory between rendezvous barriers, so the semantics (who receives what,
// in which order, reduced how) are those of the real collectives without any of their machinery.
//   hipcc -O1 -shared -fPIC tests/mock_rccl.cpp -o <tmp>/libmock_rccl.so     (KTA_RCCL_LIBRARY points at it)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include <condition_variable>
#include <deque>
#include <map>
#include <mutex>
#include <string>
#include <vector>

namespace {

struct Group {
    int nranks = 0;
    std::mutex m;
    std::condition_variable cv;
    int arrived = 0;
    uint64_t generation = 0;
    std::vector<std::vector<uint8_t>> slot;                          // one contribution per rank
    std::map<std::pair<int, int>, std::deque<std::vector<uint8_t>>> mail;   // (src, dst) -> messages in order
    void barrier()
    {
        std::unique_lock<std::mutex> l(m);
        const uint64_t g = generation;
        if (++arrived == nranks) {
            arrived = 0;
            generation++;
            cv.notify_all();
        } else {
            cv.wait(l, [&] { return generation != g; });
        }
    }
};

struct Comm {
    Group *g;
    int rank;
    bool grouped = false;
    struct Op { bool send; void *buf; size_t bytes; int peer; hipStream_t s; };
    std::vector<Op> pending;
};

std::mutex g_reg_m;
std::map<std::string, Group *> g_registry;
thread_local Comm *t_comm = nullptr;   // ncclGroupStart / End carry no communicator

size_t dtype_size(int dt) { return dt == 0 || dt == 1 ? 1 : (dt == 2 || dt == 3 ? 4 : 8); }

int flush_group(Comm *c)
{
    // sends first (into the mailboxes), rendezvous, then receives in posting order
    for (auto &op : c->pending) {
        if (!op.send) continue;
        if (hipStreamSynchronize(op.s) != hipSuccess) return 1;
        std::vector<uint8_t> bytes(op.bytes);
        if (hipMemcpy(bytes.data(), op.buf, op.bytes, hipMemcpyDeviceToHost) != hipSuccess) return 1;
        std::lock_guard<std::mutex> l(c->g->m);
        c->g->mail[{c->rank, op.peer}].push_back(std::move(bytes));
    }
    c->g->barrier();
    for (auto &op : c->pending) {
        if (op.send) continue;
        std::vector<uint8_t> bytes;
        {
            std::lock_guard<std::mutex> l(c->g->m);
            auto &q = c->g->mail[{op.peer, c->rank}];
            if (q.empty()) return 2;
            bytes = std::move(q.front());
            q.pop_front();
        }
        if (bytes.size() != op.bytes) return 3;
        if (hipMemcpy(op.buf, bytes.data(), op.bytes, hipMemcpyHostToDevice) != hipSuccess) return 1;
    }
    c->pending.clear();
    c->g->barrier();
    return 0;
}

} // namespace

extern "C" {

typedef struct { char internal[128]; } ncclUniqueId;

int ncclGetUniqueId(ncclUniqueId *id)
{
    static std::mutex m;
    static uint64_t next = 1;
    std::lock_guard<std::mutex> l(m);
    memset(id, 0, sizeof *id);
    const uint64_t v = next++;
    memcpy(id->internal, &v, sizeof v);
    memcpy(id->internal + 8, "mock-rccl", 9);
    return 0;
}

int ncclCommInitRank(void **comm, int nranks, ncclUniqueId id, int rank)
{
    Group *g;
    {
        std::lock_guard<std::mutex> l(g_reg_m);
        std::string key(id.internal, sizeof id.internal);
        auto it = g_registry.find(key);
        if (it == g_registry.end()) {
            g = new Group();
            g->nranks = nranks;
            g->slot.resize(nranks);
            g_registry[key] = g;
        } else {
            g = it->second;
        }
    }
    if (g->nranks != nranks || rank < 0 || rank >= nranks) return 4;
    Comm *c = new Comm();
    c->g = g;
    c->rank = rank;
    *comm = c;
    t_comm = c;
    g->barrier();   // ncclCommInitRank synchronises the ranks
    return 0;
}

int ncclCommDestroy(void *comm)
{
    delete static_cast<Comm *>(comm);
    return 0;
}

const char *ncclGetErrorString(int r) { return r == 0 ? "no error" : "mock rccl error"; }

int ncclGroupStart()
{
    if (t_comm) t_comm->grouped = true;
    return 0;
}

int ncclGroupEnd()
{
    Comm *c = t_comm;
    if (!c) return 0;
    c->grouped = false;
    return flush_group(c);      // always: a rank with nothing to send or receive still meets the others here
}

int ncclSend(const void *buf, size_t count, int dt, int peer, void *comm, hipStream_t s)
{
    Comm *c = static_cast<Comm *>(comm);
    t_comm = c;
    c->pending.push_back({true, const_cast<void *>(buf), count * dtype_size(dt), peer, s});
    return c->grouped ? 0 : flush_group(c);
}

int ncclRecv(void *buf, size_t count, int dt, int peer, void *comm, hipStream_t s)
{
    Comm *c = static_cast<Comm *>(comm);
    t_comm = c;
    c->pending.push_back({false, buf, count * dtype_size(dt), peer, s});
    return c->grouped ? 0 : flush_group(c);
}

// collectives run at once even inside a group: every rank issues them in the same order
int ncclAllGather(const void *send, void *recv, size_t count, int dt, void *comm, hipStream_t s)
{
    Comm *c = static_cast<Comm *>(comm);
    t_comm = c;
    const size_t bytes = count * dtype_size(dt);
    if (hipStreamSynchronize(s) != hipSuccess) return 1;
    c->g->slot[c->rank].resize(bytes);
    if (hipMemcpy(c->g->slot[c->rank].data(), send, bytes, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    c->g->barrier();
    for (int r = 0; r < c->g->nranks; r++)
        if (hipMemcpy(static_cast<uint8_t *>(recv) + (size_t)r * bytes, c->g->slot[r].data(), bytes, hipMemcpyHostToDevice) != hipSuccess)
            return 1;
    c->g->barrier();
    return 0;
}

int ncclAllReduce(const void *send, void *recv, size_t count, int dt, int op, void *comm, hipStream_t s)
{
    Comm *c = static_cast<Comm *>(comm);
    t_comm = c;
    if (dtype_size(dt) != 8 || (op != 0 && op != 2)) return 5;       // u64 / i64, sum / max: all kta_comm uses
    const size_t bytes = count * 8;
    if (hipStreamSynchronize(s) != hipSuccess) return 1;
    c->g->slot[c->rank].resize(bytes);
    if (hipMemcpy(c->g->slot[c->rank].data(), send, bytes, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    c->g->barrier();
    std::vector<uint64_t> acc(count);
    memcpy(acc.data(), c->g->slot[0].data(), bytes);
    for (int r = 1; r < c->g->nranks; r++) {
        const uint64_t *o = reinterpret_cast<const uint64_t *>(c->g->slot[r].data());
        for (size_t i = 0; i < count; i++) {
            if (op == 0) acc[i] += o[i];
            else if (dt == 4 ? (int64_t)o[i] > (int64_t)acc[i] : o[i] > acc[i]) acc[i] = o[i];
        }
    }
    if (hipMemcpy(recv, acc.data(), bytes, hipMemcpyHostToDevice) != hipSuccess) return 1;
    c->g->barrier();
    return 0;
}

} // extern "C"
