// kta_lz4.h — LZ4 inflate for compressed Kafka record batches (attributes codec 3: LZ4 *frame* format).
// Same code on the host (index: output bound; CPU tests) and on the device (one lane per batch; the
// wave-cooperative kernel in kta_kafka.hip follows the same grammar).
//
// Frame (lz4_Frame_format.md): magic 04 22 4D 18 | FLG | BD | [content size 8 B] | [dict id 4 B] | HC |
//   blocks: u32 LE size (bit 31 = stored uncompressed, 0 = end mark) | data | [block checksum 4 B] ...
//   | [content checksum 4 B].  FLG: bits 7-6 version = 01, bit 5 block independence, bit 4 block
//   checksums, bit 3 content size present, bit 2 content checksum, bit 0 dict id.  BD bits 6-4: block
//   maximum size 4..7 = 64 KiB, 256 KiB, 1 MiB, 4 MiB.  Checksums are not verified (the batch CRC-32C
//   covers the compressed bytes).  Blocks may be linked: matches reach into earlier blocks' output.
// Block (lz4_Block_format.md): sequences of token (literal length << 4 | match length - 4, 15 = more
//   bytes follow, each adding up to 255) | literals | offset u16 LE (1..65535) | [more match length];
//   the last sequence ends after its literals.  Matches may overlap their own output.
#pragma once

#include <stdint.h>

#if defined(__HIPCC__)
#define KTA_LZ4_HD __host__ __device__ inline
#else
#define KTA_LZ4_HD inline
#endif

namespace kta {

struct Lz4Frame {
    uint64_t first_block; // offset of the first block header
    uint64_t block_max;   // block maximum size
    uint64_t content;     // content size if the frame states it, else ~0
    bool block_checksum;
};

KTA_LZ4_HD bool lz4_frame_header(const uint8_t *p, uint64_t n, Lz4Frame *f)
{
    if (n < 7 || p[0] != 0x04 || p[1] != 0x22 || p[2] != 0x4D || p[3] != 0x18) return false;
    const uint32_t flg = p[4], bd = p[5];
    if ((flg >> 6) != 1u) return false;
    const uint32_t bs = (bd >> 4) & 7u;
    if (bs < 4) return false;
    f->block_max = 1ull << (8 + 2 * bs); // 4 -> 64 KiB ... 7 -> 4 MiB
    f->block_checksum = (flg & 0x10u) != 0;
    f->first_block = 6 + ((flg & 0x08u) ? 8 : 0) + ((flg & 0x01u) ? 4 : 0) + 1; // + HC
    f->content = ~0ull;
    if ((flg & 0x08u) && n >= 14) {
        f->content = 0;
        for (int i = 0; i < 8; i++) f->content |= (uint64_t)p[6 + i] << (8 * i);
    }
    return f->first_block <= n;
}

// Upper bound of the inflated size; -1 if the framing is malformed.  A compressed block expands at most
// 255x (every match-length extension byte adds 255 bytes), and never beyond the block maximum size.
KTA_LZ4_HD int64_t lz4_inflate_bound(const uint8_t *p, uint64_t n)
{
    Lz4Frame f;
    if (!lz4_frame_header(p, n, &f)) return -1;
    uint64_t pos = f.first_block, bound = 0;
    while (true) {
        if (pos + 4 > n) return -1;
        const uint32_t w = (uint32_t)p[pos] | ((uint32_t)p[pos + 1] << 8) | ((uint32_t)p[pos + 2] << 16) | ((uint32_t)p[pos + 3] << 24);
        pos += 4;
        if (w == 0) break;
        const uint64_t sz = w & 0x7FFFFFFFu;
        if (sz > f.block_max || pos + sz > n) return -1;
        const uint64_t grown = sz * 255u + 64u;
        bound += (w & 0x80000000u) ? sz : (grown < f.block_max ? grown : f.block_max);
        pos += sz + (f.block_checksum ? 4 : 0);
    }
    // a frame that states its content size (librdkafka's do) gets a slice of that size: one that inflates to more is refused
    // where it crosses it, as one that crosses the blocks' bound is
    return (int64_t)(f.content < bound ? f.content : bound);
}

// Inflate one block into dst[op .. cap); `dst` holds the earlier blocks' output (linked blocks).
// Returns the new output position or -1.
KTA_LZ4_HD int64_t lz4_inflate_block(const uint8_t *p, uint64_t n, uint8_t *dst, uint64_t op, uint64_t cap)
{
    uint64_t ip = 0;
    while (ip < n) {
        const uint32_t token = p[ip++];
        uint64_t lit = token >> 4;
        if (lit == 15) {
            uint32_t b;
            do {
                if (ip >= n) return -1;
                b = p[ip++];
                lit += b;
            } while (b == 255);
        }
        if (ip + lit > n || op + lit > cap) return -1;
        for (uint64_t k = 0; k < lit; k++) dst[op + k] = p[ip + k];
        ip += lit;
        op += lit;
        if (ip == n) break; // last sequence: literals only
        if (ip + 2 > n) return -1;
        const uint64_t off = (uint64_t)p[ip] | ((uint64_t)p[ip + 1] << 8);
        ip += 2;
        uint64_t ml = token & 15u;
        if (ml == 15) {
            uint32_t b;
            do {
                if (ip >= n) return -1;
                b = p[ip++];
                ml += b;
            } while (b == 255);
        }
        ml += 4;
        if (off == 0 || off > op || op + ml > cap) return -1;
        for (uint64_t k = 0; k < ml; k++) dst[op + k] = dst[op - off + k];
        op += ml;
    }
    return (int64_t)op;
}

// Inflate a whole frame; returns the bytes produced or -1.
KTA_LZ4_HD int64_t lz4_inflate(const uint8_t *p, uint64_t n, uint8_t *dst, uint64_t cap)
{
    Lz4Frame f;
    if (!lz4_frame_header(p, n, &f)) return -1;
    uint64_t pos = f.first_block, op = 0;
    while (true) {
        if (pos + 4 > n) return -1;
        const uint32_t w = (uint32_t)p[pos] | ((uint32_t)p[pos + 1] << 8) | ((uint32_t)p[pos + 2] << 16) | ((uint32_t)p[pos + 3] << 24);
        pos += 4;
        if (w == 0) break;
        const uint64_t sz = w & 0x7FFFFFFFu;
        if (sz > f.block_max || pos + sz > n) return -1;
        if (w & 0x80000000u) {
            if (op + sz > cap) return -1;
            for (uint64_t k = 0; k < sz; k++) dst[op + k] = p[pos + k];
            op += sz;
        } else {
            const int64_t r = lz4_inflate_block(p + pos, sz, dst, op, cap);
            if (r < 0) return -1;
            op = (uint64_t)r;
        }
        pos += sz + (f.block_checksum ? 4 : 0);
    }
    return (int64_t)op;
}

} // namespace kta
