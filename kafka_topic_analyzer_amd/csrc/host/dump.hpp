// dump.hpp — on-disk topic dump ("KTADUMP1"): the decoded records of a topic in consumption order,
// as the struct-of-arrays batches the C ABI takes.  Stands in for the consume loop of
// src/kafka.rs:74-137 when no broker / librdkafka is available, and makes "the same topic" concrete
// for CPU-vs-GPU parity runs.  Little endian.
//
//   header : "KTADUMP1" | u32 version=1 | u32 n_partitions | u64 n_records | u64 n_batches
//            | i64 start_offset[n_partitions] | i64 end_offset[n_partitions]
//   batch  : u64 n | u64 n_key_bytes | i32 partition[n] | i32 key_len[n] | i32 val_len[n]
//            | i64 ts_ms[n] | u32 key_off[n] | u8 key_bytes[n_key_bytes]   (each array padded to 8 B)
#pragma once

#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <string>
#include <vector>

namespace kta {

struct DumpHeader {
    uint32_t n_partitions = 0;
    uint64_t n_records = 0, n_batches = 0;
    std::vector<int64_t> start_offsets, end_offsets;
};

struct DumpBatch {
    uint64_t n = 0, n_key_bytes = 0;
    std::vector<int32_t> partition, key_len, val_len;
    std::vector<int64_t> ts_ms;
    std::vector<uint32_t> key_off;
    std::vector<uint8_t> key_bytes;
};

inline size_t pad8(size_t b) { return (b + 7) & ~(size_t)7; }

class DumpReader {
public:
    explicit DumpReader(const std::string &path) : f_(fopen(path.c_str(), "rb")) {}
    ~DumpReader() { if (f_) fclose(f_); }
    bool ok() const { return f_ != nullptr; }
    bool read_header(DumpHeader *h)
    {
        char magic[8];
        uint32_t ver = 0;
        if (!rd(magic, 8) || memcmp(magic, "KTADUMP1", 8) != 0) return false;
        if (!rd(&ver, 4) || ver != 1 || !rd(&h->n_partitions, 4) || !rd(&h->n_records, 8) || !rd(&h->n_batches, 8))
            return false;
        if (h->n_partitions == 0 || h->n_partitions > (1u << 20)) return false;
        h->start_offsets.resize(h->n_partitions);
        h->end_offsets.resize(h->n_partitions);
        return rd(h->start_offsets.data(), 8 * (size_t)h->n_partitions) &&
               rd(h->end_offsets.data(), 8 * (size_t)h->n_partitions);
    }
    bool read_batch(DumpBatch *b)
    {
        if (!rd(&b->n, 8) || !rd(&b->n_key_bytes, 8)) return false;
        if (b->n > (1ull << 32) || b->n_key_bytes >= (1ull << 32)) return false;
        b->partition.resize(b->n); b->key_len.resize(b->n); b->val_len.resize(b->n);
        b->ts_ms.resize(b->n); b->key_off.resize(b->n); b->key_bytes.resize(b->n_key_bytes);
        if (!(rdp(b->partition.data(), 4 * b->n) && rdp(b->key_len.data(), 4 * b->n) &&
              rdp(b->val_len.data(), 4 * b->n) && rdp(b->ts_ms.data(), 8 * b->n) &&
              rdp(b->key_off.data(), 4 * b->n) && rdp(b->key_bytes.data(), b->n_key_bytes)))
            return false;
        // the file is untrusted: every key must lie inside the batch's key bytes (the caller copies
        // key_len bytes from key_off), lengths below -1 do not exist
        for (uint64_t i = 0; i < b->n; i++) {
            if (b->key_len[i] < -1 || b->val_len[i] < -1) return false;
            if (b->key_len[i] > 0 && (uint64_t)b->key_off[i] + (uint64_t)b->key_len[i] > b->n_key_bytes) return false;
        }
        return true;
    }

private:
    bool rd(void *p, size_t n) { return n == 0 || fread(p, 1, n, f_) == n; }
    bool rdp(void *p, size_t n)
    {
        if (!rd(p, n)) return false;
        char pad[8];
        return rd(pad, pad8(n) - n);
    }
    FILE *f_;
};

class DumpWriter {
public:
    explicit DumpWriter(const std::string &path) : f_(fopen(path.c_str(), "wb")) {}
    ~DumpWriter() { if (f_) fclose(f_); }
    bool ok() const { return f_ != nullptr; }
    bool write_header(const DumpHeader &h)
    {
        const uint32_t ver = 1;
        return wr("KTADUMP1", 8) && wr(&ver, 4) && wr(&h.n_partitions, 4) && wr(&h.n_records, 8) &&
               wr(&h.n_batches, 8) && wr(h.start_offsets.data(), 8 * (size_t)h.n_partitions) &&
               wr(h.end_offsets.data(), 8 * (size_t)h.n_partitions);
    }
    bool write_batch(uint64_t n, uint64_t n_key_bytes, const int32_t *part, const int32_t *klen, const int32_t *vlen,
                     const int64_t *ts, const uint32_t *koff, const uint8_t *kbytes)
    {
        return wr(&n, 8) && wr(&n_key_bytes, 8) && wrp(part, 4 * n) && wrp(klen, 4 * n) && wrp(vlen, 4 * n) &&
               wrp(ts, 8 * n) && wrp(koff, 4 * n) && wrp(kbytes, n_key_bytes);
    }

private:
    bool wr(const void *p, size_t n) { return n == 0 || fwrite(p, 1, n, f_) == n; }
    bool wrp(const void *p, size_t n)
    {
        static const char zeros[8] = {0};
        return wr(p, n) && wr(zeros, pad8(n) - n);
    }
    FILE *f_;
};

}  // namespace kta
