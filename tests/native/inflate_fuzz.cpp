// inflate_fuzz.cpp — TEST HARNESS: the product's inflaters (csrc/kta_snappy.h, kta_lz4.h, kta_gzip.h, kta_zstd.h — the
// very functions the device kernels run, compiled here for the host) under AddressSanitizer on mutated
// streams.  Input and output live in exact-size heap blocks, so any read past the input or write past
// `cap` aborts.  A malformed batch on the GPU must be "reported, never mis-decoded" — and must never
// fault the device.
//   inflate_fuzz <codec: snappy|lz4|gzip|gzip2|zstd> <seed-file>... ; prints "<codec> ok=<n> refused=<n>"
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "kta_gzip.h"
#include "kta_lz4.h"
#include "kta_snappy.h"
#include "kta_zstd.h"

namespace {

uint64_t rng_state = 0x9E3779B97F4A7C15ull;
uint64_t rnd()
{
    rng_state ^= rng_state << 13;
    rng_state ^= rng_state >> 7;
    rng_state ^= rng_state << 17;
    return rng_state;
}

int64_t run(const std::string &codec, const std::vector<uint8_t> &in, uint64_t cap)
{
    uint8_t *src = (uint8_t *)malloc(in.size() ? in.size() : 1);   // exact size: over-reads are caught
    memcpy(src, in.data(), in.size());
    uint8_t *dst = (uint8_t *)malloc(cap ? cap : 1);               // exact size: over-writes are caught
    int64_t got;
    if (codec == "snappy") got = kta::snappy_inflate(src, in.size(), dst, cap);
    else if (codec == "lz4") got = kta::lz4_inflate(src, in.size(), dst, cap);
    else if (codec == "zstd") {
        uint64_t bound = 0, lit = 0;
        if (!kta::zstd_scan(src, in.size(), &bound, &lit)) got = -1;
        else {
            uint8_t *l = (uint8_t *)malloc(lit ? lit : 1);        // exactly what the scan asked for
            if (rnd() & 1) {                                      // the wave kernel's table layout: Huffman table over the sequence tables
                kta::ZsWorkSmall *w = (kta::ZsWorkSmall *)malloc(sizeof(kta::ZsWorkSmall));
                kta::ZsSpill *sp = (kta::ZsSpill *)malloc(sizeof(kta::ZsSpill));
                got = kta::zstd_inflate_small(src, in.size(), dst, cap, w, sp, l, lit);
                free(w);
                free(sp);
            } else {
                kta::ZsWork *w = (kta::ZsWork *)malloc(sizeof(kta::ZsWork));
                got = kta::zstd_inflate(src, in.size(), dst, cap, w, l, lit);
                free(w);
            }
            free(l);
        }
    } else if (codec == "gzip2") {                                // the two-stage form: tokens in an exact-size block too
        uint16_t work[kta::GZ2_WORK];
        const uint64_t tcap = kta::gz_token_bound(cap);
        uint32_t *tok = (uint32_t *)malloc(tcap * 4 ? tcap * 4 : 1);
        uint64_t n_tok = 0;
        if (rnd() & 1) {                                          // through the stream window, as on the device
            uint32_t window[kta::GZ_WIN / 4];
            kta::GzBitsWin bits;
            bits.win = window;
            bits.wstride = 1;
            got = kta::gzip_tokenize(bits, src, in.size(), dst, cap, tok, tcap, &n_tok, work, 1);
        } else {
            kta::GzBits bits;
            got = kta::gzip_tokenize(bits, src, in.size(), dst, cap, tok, tcap, &n_tok, work, 1);
        }
        if (got >= 0 && (n_tok > tcap || !kta::gz_apply_tokens(dst, (uint64_t)got, tok, n_tok))) { fprintf(stderr, "bad tokens accepted\n"); abort(); }
        free(tok);
    } else {
        uint16_t work[kta::GZ_WORK];
        got = kta::gzip_inflate(src, in.size(), dst, cap, work, 1);
    }
    if (got > (int64_t)cap) { fprintf(stderr, "produced %lld > cap %llu\n", (long long)got, (unsigned long long)cap); abort(); }
    free(src);
    free(dst);
    return got;
}

}  // namespace

int main(int argc, char **argv)
{
    if (argc < 3) return 64;
    const std::string codec = argv[1];
    uint64_t ok = 0, refused = 0;
    for (int a = 2; a < argc; a++) {
        FILE *f = fopen(argv[a], "rb");
        if (!f) return 65;
        std::vector<uint8_t> seed;
        uint8_t buf[65536];
        size_t r;
        while ((r = fread(buf, 1, sizeof buf, f)) > 0) seed.insert(seed.end(), buf, buf + r);
        fclose(f);
        // the file: u64 uncompressed length, then the stream
        uint64_t want = 0;
        memcpy(&want, seed.data(), 8);
        seed.erase(seed.begin(), seed.begin() + 8);
        if (run(codec, seed, want) != (int64_t)want) { fprintf(stderr, "seed %s does not inflate\n", argv[a]); return 66; }
        // a buffer one byte short must be refused (LZ4 excepted: it takes cap as a bound and reports the size)
        if (want && run(codec, seed, want - 1) >= 0 && codec != "lz4") { fprintf(stderr, "short buffer accepted\n"); return 67; }
        for (int it = 0; it < 1500; it++) {
            std::vector<uint8_t> m = seed;
            const int kind = (int)(rnd() % 5);
            if (m.empty()) break;
            if (kind == 0) m[rnd() % m.size()] ^= (uint8_t)(1u << (rnd() % 8));                  // bit flip
            else if (kind == 1) m[rnd() % m.size()] = (uint8_t)rnd();                          // byte set
            else if (kind == 2) m.resize(rnd() % m.size());                                      // truncation
            else if (kind == 3) for (int k = 0; k < 8; k++) m[rnd() % m.size()] = (uint8_t)rnd();   // several bytes
            else { const size_t at = rnd() % m.size(); m.insert(m.begin() + at, (uint8_t)rnd()); }   // insertion
            // capacities around the truth: exact, smaller, larger (LZ4 treats cap as a bound)
            const uint64_t caps[3] = {want, want / 2, want + 4096};
            const int64_t got = run(codec, m, caps[rnd() % 3]);
            if (got < 0) refused++;
            else ok++;
        }
    }
    printf("%s ok=%llu refused=%llu\n", codec.c_str(), (unsigned long long)ok, (unsigned long long)refused);
    return 0;
}
