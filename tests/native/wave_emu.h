// wave_emu.h — test infrastructure: runs the SOURCE of a one-wave HIP kernel on the CPU, so the `-m "not gpu"` suite can
// execute the device code itself (window loads, LDS hand-over between lanes, barriers, cross-lane operations), not a
// restatement of it.  Not part of the product: nothing under kafka_topic_analyzer_amd/ includes this file.
//
// The 64 lanes of a workgroup are fibers (ucontext) on one OS thread.  A lane runs until it reaches a meeting point —
// __syncthreads(), __any(), __shfl_xor(), KTA_READLANE() — and parks there; when every live lane is parked the scheduler checks that
// they all wait at the SAME call site (the source line: a barrier under divergent control flow is reported, not
// emulated), exchanges the values and releases them.  Between two meeting points the lanes run one after the other in
// an order the test chooses — ascending, descending or a fresh pseudo-random permutation per phase — so code that
// only works because "lane 0 went first" (a missing barrier) fails under at least one of them.  `__shared__` becomes
// function-local static storage: one workgroup runs at a time, and like LDS it keeps what the last workgroup left.
//
// What this does NOT check: the compiler's gfx950 code generation, the memory model between waves, inline assembly
// (an empty asm with a register constraint is a no-op here as there), timing.  The GPU tests remain the parity gate.
#pragma once

#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <ucontext.h>

#include <functional>
#include <vector>

#define __device__
#define __host__
#define __global__
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__ static
#define __launch_bounds__(...)
#define KTA_WAVES_PER_EU(least, most)

struct uint4 {
    uint32_t x, y, z, w;
};
static inline uint4 make_uint4(uint32_t x, uint32_t y, uint32_t z, uint32_t w) { return uint4{x, y, z, w}; }

namespace wave_emu {

constexpr uint32_t kLanes = 64;
constexpr size_t kStackBytes = 256 * 1024;

struct Dim3 {
    uint32_t x, y, z;
};

enum Meeting : uint32_t { NONE = 0, BARRIER = 1, ANY = 2, SHFL_XOR = 3, READLANE = 4, BALLOT = 5, SHFL_UP = 6 };

struct Lane {
    ucontext_t ctx;
    std::vector<char> stack;
    Dim3 thread_idx;
    bool done = false;
    Meeting waiting = NONE;
    int site = 0;          // source line of the meeting point
    uint64_t in = 0;       // what the lane brings to the meeting
    uint64_t out = 0;      // what it takes away
    uint32_t operand = 0;  // shfl_xor: the lane mask
};

struct State {
    Lane lanes[kLanes];
    ucontext_t scheduler;
    Lane *current = nullptr;
    Dim3 block_idx{0, 0, 0};
    const std::function<void()> *body = nullptr;
    char error[256] = {0};
};

inline State &state()
{
    static State s;
    return s;
}

inline void park(Meeting what, int site, uint64_t in, uint32_t operand)
{
    State &s = state();
    Lane *me = s.current;
    me->waiting = what;
    me->site = site;
    me->in = in;
    me->operand = operand;
    swapcontext(&me->ctx, &s.scheduler);
}

inline void barrier(int site) { park(BARRIER, site, 0, 0); }

inline bool any(bool pred, int site)
{
    park(ANY, site, pred ? 1u : 0u, 0);
    return state().current->out != 0;
}

template <class T> inline T shfl_xor(T v, int mask, int site)
{
    static_assert(sizeof(T) <= 8, "shfl_xor: at most 64 bits");
    uint64_t bits = 0;
    memcpy(&bits, &v, sizeof(T));
    park(SHFL_XOR, site, bits, (uint32_t)mask);
    T r;
    memcpy(&r, &state().current->out, sizeof(T));
    return r;
}

// the lanes (of those still running) whose predicate holds
inline uint64_t ballot(bool pred, int site)
{
    park(BALLOT, site, pred ? 1u : 0u, 0);
    return state().current->out;
}

// lane - off's value; a lane below `off` (or whose source has finished) keeps its own
inline uint32_t shfl_up(uint32_t v, uint32_t off, int site)
{
    park(SHFL_UP, site, v, off);
    return (uint32_t)state().current->out;
}

// v_readlane: every lane takes lane `src`'s value (src is the same for all of them)
inline uint32_t readlane(uint32_t v, uint32_t src, int site)
{
    park(READLANE, site, v, src);
    return (uint32_t)state().current->out;
}

inline void lane_entry()
{
    State &s = state();
    (*s.body)();
    s.current->done = true;
    s.current->waiting = NONE;
    swapcontext(&s.current->ctx, &s.scheduler);
}

// Run `grid` workgroups of 64 lanes, one after the other.  order: 0 ascending, 1 descending, 2 a new pseudo-random
// permutation of the lanes in every phase (from `seed`).  Returns nullptr, or a message (divergent meeting points).
inline const char *launch(uint32_t grid, int order, uint32_t seed, const std::function<void()> &body)
{
    State &s = state();
    s.body = &body;
    s.error[0] = 0;
    uint64_t rng = 0x9E3779B97F4A7C15ull ^ ((uint64_t)seed << 17) ^ seed;
    for (uint32_t blk = 0; blk < grid; blk++) {
        s.block_idx = Dim3{blk, 0, 0};
        for (uint32_t l = 0; l < kLanes; l++) {
            Lane &ln = s.lanes[l];
            if (ln.stack.size() != kStackBytes) ln.stack.resize(kStackBytes);
            ln.thread_idx = Dim3{l, 0, 0};
            ln.done = false;
            ln.waiting = NONE;
            getcontext(&ln.ctx);
            ln.ctx.uc_stack.ss_sp = ln.stack.data();
            ln.ctx.uc_stack.ss_size = ln.stack.size();
            ln.ctx.uc_link = &s.scheduler;
            makecontext(&ln.ctx, (void (*)())lane_entry, 0);
        }
        for (;;) {
            uint32_t perm[kLanes];
            for (uint32_t l = 0; l < kLanes; l++) perm[l] = order == 1 ? kLanes - 1 - l : l;
            if (order >= 2)
                for (uint32_t l = kLanes - 1; l > 0; l--) {
                    rng = rng * 6364136223846793005ull + 1442695040888963407ull;
                    const uint32_t o = (uint32_t)((rng >> 33) % (l + 1));
                    const uint32_t t = perm[l]; perm[l] = perm[o]; perm[o] = t;
                }
            uint32_t live = 0;
            for (uint32_t i = 0; i < kLanes; i++) {          // one phase: every live lane up to its next meeting point
                Lane &ln = s.lanes[perm[i]];
                if (ln.done) continue;
                ln.waiting = NONE;
                s.current = &ln;
                swapcontext(&s.scheduler, &ln.ctx);
                if (!ln.done) live++;
            }
            if (live == 0) break;
            Meeting what = NONE;
            int site = 0;
            for (uint32_t l = 0; l < kLanes; l++) {
                const Lane &ln = s.lanes[l];
                if (ln.done) continue;
                if (what == NONE) { what = ln.waiting; site = ln.site; }
                if (ln.waiting != what || ln.site != site) {
                    snprintf(s.error, sizeof s.error, "workgroup %u: lanes meet at different points (line %d and line %d)",
                             blk, site, ln.site);
                    return s.error;   // (the fibers are abandoned where they stand)
                }
            }
            if (what == ANY) {
                uint64_t r = 0;
                for (uint32_t l = 0; l < kLanes; l++) if (!s.lanes[l].done) r |= s.lanes[l].in;
                for (uint32_t l = 0; l < kLanes; l++) s.lanes[l].out = r;
            } else if (what == READLANE) {
                uint32_t src = 0;
                for (uint32_t l = 0; l < kLanes; l++) if (!s.lanes[l].done) { src = s.lanes[l].operand; break; }
                const Lane &from = s.lanes[src % kLanes];
                for (uint32_t l = 0; l < kLanes; l++) s.lanes[l].out = from.done ? 0 : from.in;
            } else if (what == BALLOT) {
                uint64_t r = 0;
                for (uint32_t l = 0; l < kLanes; l++) if (!s.lanes[l].done && s.lanes[l].in) r |= 1ull << l;
                for (uint32_t l = 0; l < kLanes; l++) s.lanes[l].out = r;
            } else if (what == SHFL_UP) {
                for (uint32_t l = 0; l < kLanes; l++) {
                    const uint32_t off = s.lanes[l].operand;
                    const bool own = l < off || s.lanes[l - off].done;
                    s.lanes[l].out = own ? s.lanes[l].in : s.lanes[l - off].in;
                }
            } else if (what == SHFL_XOR) {
                for (uint32_t l = 0; l < kLanes; l++) {
                    const Lane &from = s.lanes[(l ^ s.lanes[l].operand) % kLanes];
                    s.lanes[l].out = from.done ? 0 : from.in;
                }
            }
        }
    }
    return nullptr;
}

} // namespace wave_emu

#define threadIdx (wave_emu::state().current->thread_idx)
#define blockIdx (wave_emu::state().block_idx)
#define __syncthreads() wave_emu::barrier(__LINE__)
#define __any(p) wave_emu::any((p), __LINE__)
#define __shfl_xor(v, m) wave_emu::shfl_xor((v), (m), __LINE__)
#define KTA_READLANE(v, l) wave_emu::readlane((uint32_t)(v), (l), __LINE__)
#define KTA_BALLOT64(p) wave_emu::ballot((p), __LINE__)
#define KTA_SHFL_UP(v, off) wave_emu::shfl_up((uint32_t)(v), (off), __LINE__)
#define KTA_UNI(v) ((uint32_t)(v))

// one OS thread: the atomics are plain operations
static inline unsigned long long atomicAdd(unsigned long long *p, unsigned long long v)
{
    const unsigned long long old = *p;
    *p = old + v;
    return old;
}
static inline uint32_t atomicMin(uint32_t *p, uint32_t v)
{
    const uint32_t old = *p;
    if (v < old) *p = v;
    return old;
}
