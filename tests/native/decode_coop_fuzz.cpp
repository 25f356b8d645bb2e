// decode_coop_fuzz.cpp — TEST HARNESS: kafka_decode_coop (csrc/kta_decode_coop.h, the kernel's own
// source over tests/native/wave_emu.h) under AddressSanitizer on damaged record sets.  The blob lies in a heap block
// of exactly the bytes the device contract promises (16-byte aligned, 64 readable bytes behind blob_len), the output
// columns in blocks of exactly n_records entries: a read or write outside them aborts — on the GPU it would be a
// fault of the device.  What the kernel delivers is checked too: a sound batch exactly as before the damage, a
// reported batch as a delivered prefix followed by -1 to its end, nothing outside the batches' record ranges.
//   decode_coop_fuzz <rounds> <seed-file>...        prints "decode ok=<n> reported=<n>"
// Seed file (little endian, tests/test_decode_fuzz.py writes it from the host index): u64 n_batches | u64 n_records |
// u64 blob_len | n_batches x kta_kafka_batch_desc | blob.
#include "decode_coop_emu.cpp"

#include <string>
#include <vector>

namespace {

uint64_t rng_state = 0x9E3779B97F4A7C15ull;
uint64_t rnd()
{
    rng_state ^= rng_state << 13;
    rng_state ^= rng_state >> 7;
    rng_state ^= rng_state << 17;
    return rng_state;
}

struct Columns {
    int32_t *part, *klen, *vlen;
    int64_t *ts;
    uint32_t *koff;
    uint64_t n;
    explicit Columns(uint64_t n_) : n(n_)
    {
        part = (int32_t *)malloc(n * 4 + 1); klen = (int32_t *)malloc(n * 4 + 1); vlen = (int32_t *)malloc(n * 4 + 1);
        ts = (int64_t *)malloc(n * 8 + 1); koff = (uint32_t *)malloc(n * 4 + 1);
        for (uint64_t i = 0; i < n; i++) { part[i] = klen[i] = vlen[i] = -7; ts[i] = -7; koff[i] = 0xFFFFFFF7u; }
    }
    ~Columns() { free(part); free(klen); free(vlen); free(ts); free(koff); }
};

struct Geometry {
    uint32_t lanes, window, per_round;
    int prefetch;
};
const Geometry kGeometries[] = {{16, 3072, 16, 0}, {8, 1024, 16, 0}, {32, 8192, 32, 0}, {64, 8192, 256, 0}, {8, 256, 8, 0}, {4, 64, 4, 0}};

bool decode(const Geometry &g, const std::vector<uint8_t> &blob, const std::vector<kta_kafka_batch_desc> &descs, Columns &c,
            uint64_t *bad)
{
    uint64_t keyb = 0;
    const int rc = kta_emu_decode_coop(g.lanes, g.window, g.per_round, g.prefetch, (int)(rnd() % 3), (uint32_t)rnd(),
                                       (uint8_t)rnd(), blob.data(), blob.size(), descs.data(), descs.size(), c.part, c.klen,
                                       c.vlen, c.ts, c.koff, nullptr, 0, &keyb, bad);
    if (rc != 0) fprintf(stderr, "emulator: rc %d %s\n", rc, kta_emu_last_error());
    return rc == 0;
}

} // namespace

int main(int argc, char **argv)
{
    if (argc < 3) return 2;
    const int rounds = atoi(argv[1]);
    uint64_t ok = 0, reported = 0;
    for (int a = 2; a < argc; a++) {
        FILE *f = fopen(argv[a], "rb");
        if (!f) return 2;
        uint64_t head[3];
        if (fread(head, 8, 3, f) != 3) return 2;
        std::vector<kta_kafka_batch_desc> descs(head[0]);
        std::vector<uint8_t> blob(head[2]);
        if (fread(descs.data(), sizeof(kta_kafka_batch_desc), descs.size(), f) != descs.size()) return 2;
        if (fread(blob.data(), 1, blob.size(), f) != blob.size()) return 2;
        fclose(f);
        const uint64_t n = head[1];
        Columns clean(n);
        uint64_t bad = 0;
        if (!decode(kGeometries[0], blob, descs, clean, &bad) || bad != 0) {
            fprintf(stderr, "%s: the seed itself does not decode\n", argv[a]);
            return 1;
        }
        for (int r = 0; r < rounds; r++) {
            std::vector<uint8_t> hurt = blob;
            std::vector<kta_kafka_batch_desc> d = descs;
            std::vector<char> touched(d.size(), 0);
            const int hits = 1 + (int)(rnd() % 3);
            for (int h = 0; h < hits; h++) {
                const size_t b = rnd() % d.size();
                if (d[b].payload_end == d[b].payload_off) continue;
                touched[b] = 1;
                const uint64_t at = d[b].payload_off + rnd() % (d[b].payload_end - d[b].payload_off);
                switch (rnd() % 5) {
                case 0: hurt[at] = (uint8_t)rnd(); break;
                case 1: hurt[at] ^= (uint8_t)(1u << (rnd() % 8)); break;
                case 2: hurt[at] = 0xFF; break;                                   // a varint that goes on
                case 3:                                                           // a forged count, as far as the index lets one through
                    d[b].n_records = (int32_t)(1 + rnd() % ((d[b].payload_end - d[b].payload_off) / 7 + 1));
                    break;
                default:                                                          // a batch cut short (never beyond the blob)
                    d[b].payload_end = d[b].payload_off + rnd() % (d[b].payload_end - d[b].payload_off + 1);
                    break;
                }
            }
            // record ranges must stay disjoint and inside the columns: place the (possibly forged) counts anew
            uint64_t total = 0;
            for (auto &x : d) { x.record_base = total; total += (uint64_t)x.n_records; }
            Columns c(total);
            const Geometry &g = kGeometries[rnd() % (sizeof kGeometries / sizeof kGeometries[0])];
            if (!decode(g, hurt, d, c, &bad)) return 1;
            uint64_t seen_bad = 0;
            for (size_t b = 0; b < d.size(); b++) {
                const uint64_t lo = d[b].record_base, cnt = (uint64_t)d[b].n_records;
                uint64_t delivered = 0;
                while (delivered < cnt && c.part[lo + delivered] != -1) delivered++;
                for (uint64_t i = 0; i < cnt; i++) {
                    const bool live = i < delivered;
                    if (live ? c.part[lo + i] != d[b].partition
                             : !(c.part[lo + i] == -1 && c.klen[lo + i] == -1 && c.vlen[lo + i] == -1 && c.ts[lo + i] == -1 &&
                                 c.koff[lo + i] == 0)) {
                        fprintf(stderr, "%s round %d geometry <%u,%u,%u,%d>: batch %zu record %llu is neither delivered nor withheld\n",
                                argv[a], r, g.lanes, g.window, g.per_round, g.prefetch, b, (unsigned long long)i);
                        return 1;
                    }
                }
                seen_bad += delivered < cnt;
                if (!touched[b]) {                                               // an undamaged batch decodes as before
                    const uint64_t was = descs[b].record_base;
                    if (delivered != cnt || memcmp(c.klen + lo, clean.klen + was, cnt * 4) || memcmp(c.vlen + lo, clean.vlen + was, cnt * 4) ||
                        memcmp(c.ts + lo, clean.ts + was, cnt * 8) || memcmp(c.koff + lo, clean.koff + was, cnt * 4)) {
                        fprintf(stderr, "%s round %d: the undamaged batch %zu changed\n", argv[a], r, b);
                        return 1;
                    }
                }
            }
            if (seen_bad != bad) {
                fprintf(stderr, "%s round %d: %llu batches reported, %llu with withheld records\n", argv[a], r,
                        (unsigned long long)bad, (unsigned long long)seen_bad);
                return 1;
            }
            if (bad) reported++; else ok++;
        }
    }
    printf("decode ok=%llu reported=%llu\n", (unsigned long long)ok, (unsigned long long)reported);
    return 0;
}
