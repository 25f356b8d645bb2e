// zstd_huf_emu.cpp — test infrastructure: the wave decoder of zstd's Huffman-coded literals (csrc/kta_zstd_huf_wave.h, the
// kernel's own source) compiled for the host over tests/native/wave_emu.h and run with its 64 lanes as fibers, beside the
// format's own statement of the same thing (kta_zstd.h: zs_huf_stream, one stream after the other).
// tests/test_zstd_huf_emu.py hands it literals sections cut out of frames libzstd wrote.
#include "wave_emu.h"

#include "kta_zstd.h"

namespace {
#include "kta_zstd_huf_wave.h"
}

#include <vector>

extern "C" {

// sec[0 .. n): a literals section with its own Huffman tree (type 2), as it lies in a compressed block.  The section is placed
// `shift` bytes behind a 16-byte boundary of a buffer whose other bytes are `poison`.  want[0 .. cap): the host statement's
// literals; got[0 .. cap + 16): the wave's (the 16 bytes behind the literals must come back untouched).  order, seed:
// wave_emu::launch.  Returns the number of literals; -1: the host statement refuses the section; -2: the emulator reports
// divergent meeting points; -3: the wave refuses what the host statement accepts; -4: cap is too small.
int64_t kta_emu_zstd_huf(const uint8_t *sec, uint64_t n, uint32_t shift, uint8_t poison, uint8_t *want, uint8_t *got, uint64_t cap,
                         int order, uint32_t seed, int expect_refusal, char *err_out, uint64_t err_cap)
{
    const uint64_t lead = 4096 + (shift & 15u), total = ((lead + n + 15) & ~15ull) + 64;
    uint8_t *buf = static_cast<uint8_t *>(aligned_alloc(16, total));
    if (!buf) return -5;
    memset(buf, poison, total);
    memcpy(buf + lead, sec, n);
    kta::ZsMem src{buf + lead};
    uint32_t type, regen, comp, streams;
    const uint32_t hdr = kta::zs_lit_header(src, 0, n, &type, &regen, &comp, &streams);
    static kta::ZsWork w;
    int64_t rc = -1;
    do {
        if (!hdr || type != 2 || hdr + comp > n) break;
        if (regen > cap) { rc = -4; break; }
        uint32_t q = hdr, qn = comp;
        const uint32_t used = kta::zs_read_huffman(w, src, q, qn);
        if (!used) break;
        q += used;
        qn -= used;
        uint32_t at[4] = {q, 0, 0, 0}, len[4] = {qn, 0, 0, 0}, count[4] = {regen, 0, 0, 0};
        if (streams == 4) {                               // (kta_zstd.h: zs_block)
            if (qn < 6) break;
            const uint32_t s1 = src.byte(q) | (src.byte(q + 1) << 8);
            const uint32_t s2 = src.byte(q + 2) | (src.byte(q + 3) << 8);
            const uint32_t s3 = src.byte(q + 4) | (src.byte(q + 5) << 8);
            if (6 + s1 + s2 + s3 > qn) break;
            const uint32_t each = (regen + 3) / 4;
            if (3 * each > regen) break;
            at[0] = q + 6; at[1] = at[0] + s1; at[2] = at[1] + s2; at[3] = at[2] + s3;
            len[0] = s1; len[1] = s2; len[2] = s3; len[3] = qn - 6 - s1 - s2 - s3;
            count[0] = count[1] = count[2] = each;
            count[3] = regen - 3 * each;
        }
        kta::ZsOutMem o{nullptr, 0};
        const bool host_ok = o.huf_streams(w, src, streams, at, len, count, want);
        bool wave_ok = false;
        const char *err = wave_emu::launch(1, order, seed, [&] {
            __shared__ uint32_t s_win[kZhWinBytes / 4];
            const bool ok = zh_streams(s_win, buf, lead, streams, at, len, count, w.huf_table(), w.huf_log, got, threadIdx.x);
            if (threadIdx.x == 0) wave_ok = ok;
        });
        if (err) {
            if (err_out && err_cap) snprintf(err_out, err_cap, "%s", err);
            rc = -2;
            break;
        }
        if (!host_ok) { rc = wave_ok ? -3 : -1; break; }    // what the host statement refuses the wave must refuse too
        if (!wave_ok) { rc = -3; break; }
        rc = (int64_t)regen;
    } while (false);
    (void)expect_refusal;
    free(buf);
    return rc;
}

} // extern "C"
