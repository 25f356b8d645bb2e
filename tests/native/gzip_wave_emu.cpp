// gzip_wave_emu.cpp — test infrastructure: the wave-per-batch gzip tokenizer (csrc/kta_gzip_wave.h, the kernel's own source)
// compiled for the host over tests/native/wave_emu.h and run with its 64 lanes as fibers.  tests/test_gzip_wave_emu.py
// builds this into a shared object and holds what it produces — literals in place + tokens, then the tokens applied by
// kta::gz_apply_tokens — against zlib's output, and its refusals against the lane tokenizer's verdicts.
#include "wave_emu.h"

#include "../../include/kta_kafka.h"
#include "kta_gzip.h"

namespace {
#define KTA_GW_STATS 1          // regions, repetitions of the confirming loop, lanes that decoded again in them, lanes with a segment
uint32_t gw_stats[4];
#include "kta_gzip_wave.h"
}

#include <vector>

extern "C" {

// One member (member[0 .. n)) placed `shift` bytes (0..15) behind a 16-byte boundary of a fetch buffer whose bytes around
// it are `poison`; dst: cap + 16 bytes (the bytes behind cap must come back untouched: the caller checks); tok_cap tokens.
// order, seed: wave_emu::launch.  Returns the number of tokens (>= 0; the tokens applied: dst holds the member's output),
// -1 if the kernel left the member to the lane kernel, -2 if the emulator reports divergent meeting points, -3 if the
// tokens it wrote do not apply.
int64_t kta_emu_gzip_wave(const uint8_t *member, uint64_t n, uint32_t shift, uint8_t poison, uint8_t *dst, uint64_t cap, uint64_t tok_cap,
                          int order, uint32_t seed, char *err_out, uint64_t err_cap)
{
    const uint64_t lead = 64 + (shift & 15u), total = ((lead + n + 15) & ~15ull) + 64;
    uint8_t *buf = static_cast<uint8_t *>(aligned_alloc(16, total));
    if (!buf) return -4;
    memset(buf, poison, total);
    memcpy(buf + lead, member, n);
    std::vector<uint32_t> tok(tok_cap + 1, 0xDEADBEEFu);
    uint32_t got = 0;
    const char *err = wave_emu::launch(1, order, seed, [&] {
        __shared__ GwShared sh;
        const uint32_t r = gw_tokenize_member(sh, buf, lead, n, dst, cap, tok.data(), tok_cap, threadIdx.x);
        if (threadIdx.x == 0) got = r;
    });
    free(buf);
    if (err) {
        if (err_out && err_cap) snprintf(err_out, err_cap, "%s", err);
        return -2;
    }
    if (got == kGwNotDone) return -1;
    if (tok[tok_cap] != 0xDEADBEEFu || got > tok_cap) return -3;
    if (!kta::gz_apply_tokens(dst, cap, tok.data(), got)) return -3;
    return (int64_t)got;
}

void kta_emu_gzip_wave_stats(uint32_t *out, int reset)
{
    for (int i = 0; i < 4; i++) {
        out[i] = gw_stats[i];
        if (reset) gw_stats[i] = 0;
    }
}

} // extern "C"
