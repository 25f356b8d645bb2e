// Test double of RCCL (test infrastructure, not product code): the slice of the API csrc/kta_comm.hip binds, for
// several ranks on ONE GPU — the build container reaches a single device and RCCL refuses two ranks on the same
// device.  The ranks may be THREADS of one process (tests/test_gpu_dist.py: up to eight contexts in one process) or
// PROCESSES (`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`, the driver's launch line): they
// meet through POSIX shared memory named by the unique id — a control block with a process-shared mutex / condition
// variable for the rendezvous barriers, and the payloads as files under /dev/shm.  Every call drains the caller's
// stream and then moves the bytes through host memory between barriers, so the semantics (who receives what, in
// which order, reduced how) are those of the real collectives without any of their machinery.
//   hipcc -O1 -shared -fPIC tests/mock_rccl.cpp -o <tmp>/libmock_rccl.so -lrt -lpthread    (KTA_RCCL_LIBRARY points at it)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <fcntl.h>
#include <pthread.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <atomic>
#include <map>
#include <mutex>
#include <string>
#include <vector>

namespace {

struct Control {                       // lives in shared memory: /dev/shm/<name>
    std::atomic<uint32_t> ready;       // set by the creator once the rest is initialised
    uint32_t nranks;
    pthread_mutex_t m;
    pthread_cond_t cv;
    int arrived;
    uint64_t generation;
    int attached;                      // communicators alive: the last one unlinks the block
};

struct Comm {
    Control *ctl = nullptr;
    std::string name;                  // shared-memory object ("/mockrccl-...") and prefix of the payload files
    int rank = 0, nranks = 0;
    bool grouped = false;
    struct Op { bool send; void *buf; size_t bytes; int peer; hipStream_t s; };
    std::vector<Op> pending;
    std::map<std::pair<int, int>, uint64_t> seqno;   // (src, dst) -> messages so far: both ends count alike
    void barrier()
    {
        pthread_mutex_lock(&ctl->m);
        const uint64_t g = ctl->generation;
        if (++ctl->arrived == (int)ctl->nranks) {
            ctl->arrived = 0;
            ctl->generation++;
            pthread_cond_broadcast(&ctl->cv);
        } else {
            while (ctl->generation == g) pthread_cond_wait(&ctl->cv, &ctl->m);
        }
        pthread_mutex_unlock(&ctl->m);
    }
    std::string file(const char *kind, int a, int b, uint64_t k) const
    {
        char buf[256];
        snprintf(buf, sizeof buf, "/dev/shm%s.%s%d_%d_%llu", name.c_str(), kind, a, b, (unsigned long long)k);
        return buf;
    }
};

thread_local Comm *t_comm = nullptr;   // ncclGroupStart / End carry no communicator

size_t dtype_size(int dt) { return dt == 0 || dt == 1 ? 1 : (dt == 2 || dt == 3 ? 4 : 8); }

bool put_file(const std::string &path, const void *data, size_t bytes)
{
    const int fd = open(path.c_str(), O_CREAT | O_TRUNC | O_WRONLY, 0600);
    if (fd < 0) return false;
    const uint8_t *p = static_cast<const uint8_t *>(data);
    size_t done = 0;
    while (done < bytes) {
        const ssize_t w = write(fd, p + done, bytes - done);
        if (w <= 0) { close(fd); return false; }
        done += (size_t)w;
    }
    close(fd);
    return true;
}

bool get_file(const std::string &path, std::vector<uint8_t> &out)
{
    const int fd = open(path.c_str(), O_RDONLY);
    if (fd < 0) return false;
    struct stat st;
    if (fstat(fd, &st) != 0) { close(fd); return false; }
    out.resize((size_t)st.st_size);
    size_t done = 0;
    while (done < out.size()) {
        const ssize_t r = read(fd, out.data() + done, out.size() - done);
        if (r <= 0) { close(fd); return false; }
        done += (size_t)r;
    }
    close(fd);
    return true;
}

int flush_group(Comm *c)
{
    // sends first (into the mail files), rendezvous, then receives in posting order
    for (auto &op : c->pending) {
        if (!op.send) continue;
        if (hipStreamSynchronize(op.s) != hipSuccess) return 1;
        std::vector<uint8_t> bytes(op.bytes);
        if (op.bytes && hipMemcpy(bytes.data(), op.buf, op.bytes, hipMemcpyDeviceToHost) != hipSuccess) return 1;
        if (!put_file(c->file("m", c->rank, op.peer, c->seqno[{c->rank, op.peer}]++), bytes.data(), bytes.size())) return 6;
    }
    c->barrier();
    for (auto &op : c->pending) {
        if (op.send) continue;
        std::vector<uint8_t> bytes;
        const std::string path = c->file("m", op.peer, c->rank, c->seqno[{op.peer, c->rank}]++);
        if (!get_file(path, bytes)) return 2;
        unlink(path.c_str());
        if (bytes.size() != op.bytes) return 3;
        if (op.bytes && hipMemcpy(op.buf, bytes.data(), op.bytes, hipMemcpyHostToDevice) != hipSuccess) return 1;
    }
    c->pending.clear();
    c->barrier();
    return 0;
}

} // namespace

extern "C" {

typedef struct { char internal[128]; } ncclUniqueId;

int ncclGetUniqueId(ncclUniqueId *id)
{
    static std::atomic<uint64_t> next{1};
    struct timespec ts;
    clock_gettime(CLOCK_REALTIME, &ts);
    memset(id, 0, sizeof *id);
    snprintf(id->internal, sizeof id->internal, "/mockrccl-%d-%llu-%llu", (int)getpid(), (unsigned long long)next++,
             (unsigned long long)ts.tv_sec * 1000000000ull + (unsigned long long)ts.tv_nsec);
    return 0;
}

int ncclCommInitRank(void **comm, int nranks, ncclUniqueId id, int rank)
{
    id.internal[sizeof id.internal - 1] = 0;
    const std::string name(id.internal);
    if (name.compare(0, 10, "/mockrccl-") != 0 || rank < 0 || rank >= nranks) return 4;
    Control *ctl = nullptr;
    int fd = shm_open(name.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
    if (fd >= 0) {                                   // the first rank to arrive sets the block up
        if (ftruncate(fd, sizeof(Control)) != 0) return 6;
        ctl = static_cast<Control *>(mmap(nullptr, sizeof(Control), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0));
        close(fd);
        if (ctl == MAP_FAILED) return 6;
        pthread_mutexattr_t ma;
        pthread_condattr_t ca;
        pthread_mutexattr_init(&ma);
        pthread_mutexattr_setpshared(&ma, PTHREAD_PROCESS_SHARED);
        pthread_condattr_init(&ca);
        pthread_condattr_setpshared(&ca, PTHREAD_PROCESS_SHARED);
        pthread_mutex_init(&ctl->m, &ma);
        pthread_cond_init(&ctl->cv, &ca);
        ctl->nranks = (uint32_t)nranks;
        ctl->arrived = 0;
        ctl->generation = 0;
        ctl->attached = 0;
        ctl->ready.store(1, std::memory_order_release);
    } else {
        for (int tries = 0; tries < 600000; tries++) {          // (the creator may still be between shm_open and ftruncate)
            fd = shm_open(name.c_str(), O_RDWR, 0600);
            struct stat st;
            if (fd >= 0 && fstat(fd, &st) == 0 && (size_t)st.st_size >= sizeof(Control)) break;
            if (fd >= 0) close(fd);
            fd = -1;
            usleep(100);
        }
        if (fd < 0) return 6;
        ctl = static_cast<Control *>(mmap(nullptr, sizeof(Control), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0));
        close(fd);
        if (ctl == MAP_FAILED) return 6;
        while (ctl->ready.load(std::memory_order_acquire) == 0) usleep(100);
    }
    if (ctl->nranks != (uint32_t)nranks) return 4;
    Comm *c = new Comm();
    c->ctl = ctl;
    c->name = name;
    c->rank = rank;
    c->nranks = nranks;
    pthread_mutex_lock(&ctl->m);
    ctl->attached++;
    pthread_mutex_unlock(&ctl->m);
    *comm = c;
    t_comm = c;
    c->barrier();   // ncclCommInitRank synchronises the ranks
    return 0;
}

int ncclCommDestroy(void *comm)
{
    Comm *c = static_cast<Comm *>(comm);
    if (t_comm == c) t_comm = nullptr;
    pthread_mutex_lock(&c->ctl->m);
    const bool last = --c->ctl->attached == 0;
    pthread_mutex_unlock(&c->ctl->m);
    unlink(c->file("s", c->rank, 0, 0).c_str());
    if (last) shm_unlink(c->name.c_str());
    munmap(c->ctl, sizeof(Control));
    delete c;
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
static int contribute(Comm *c, const void *send, size_t bytes, hipStream_t s)
{
    if (hipStreamSynchronize(s) != hipSuccess) return 1;
    std::vector<uint8_t> mine(bytes);
    if (bytes && hipMemcpy(mine.data(), send, bytes, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    if (!put_file(c->file("s", c->rank, 0, 0), mine.data(), bytes)) return 6;
    c->barrier();
    return 0;
}

int ncclAllGather(const void *send, void *recv, size_t count, int dt, void *comm, hipStream_t s)
{
    Comm *c = static_cast<Comm *>(comm);
    t_comm = c;
    const size_t bytes = count * dtype_size(dt);
    if (int e = contribute(c, send, bytes, s)) return e;
    for (int r = 0; r < c->nranks; r++) {
        std::vector<uint8_t> theirs;
        if (!get_file(c->file("s", r, 0, 0), theirs) || theirs.size() != bytes) return 2;
        if (bytes && hipMemcpy(static_cast<uint8_t *>(recv) + (size_t)r * bytes, theirs.data(), bytes, hipMemcpyHostToDevice) != hipSuccess)
            return 1;
    }
    c->barrier();               // (nobody overwrites its slot before everybody has read it)
    return 0;
}

int ncclAllReduce(const void *send, void *recv, size_t count, int dt, int op, void *comm, hipStream_t s)
{
    Comm *c = static_cast<Comm *>(comm);
    t_comm = c;
    if (dtype_size(dt) != 8 || (op != 0 && op != 2)) return 5;       // u64 / i64, sum / max: all kta_comm uses
    const size_t bytes = count * 8;
    if (int e = contribute(c, send, bytes, s)) return e;
    std::vector<uint64_t> acc(count);
    for (int r = 0; r < c->nranks; r++) {
        std::vector<uint8_t> theirs;
        if (!get_file(c->file("s", r, 0, 0), theirs) || theirs.size() != bytes) return 2;
        const uint64_t *o = reinterpret_cast<const uint64_t *>(theirs.data());
        for (size_t i = 0; i < count; i++) {
            if (r == 0) acc[i] = o[i];
            else if (op == 0) acc[i] += o[i];
            else if (dt == 4 ? (int64_t)o[i] > (int64_t)acc[i] : o[i] > acc[i]) acc[i] = o[i];
        }
    }
    if (bytes && hipMemcpy(recv, acc.data(), bytes, hipMemcpyHostToDevice) != hipSuccess) return 1;
    c->barrier();
    return 0;
}

} // extern "C"
