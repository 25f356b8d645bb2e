"""A small LZ4 COMPRESSOR (block + frame format) and decompressor in pure Python (test infrastructure).
Formats: lz4_Block_format.md / lz4_Frame_format.md.  The header checksum uses the xxhash module."""
import struct

import xxhash


def _len_bytes(n):
    out = bytearray()
    while n >= 255:
        out.append(255)
        n -= 255
    out.append(n)
    return bytes(out)


def compress_block(data: bytes, history: bytes = b"") -> bytes:
    """Greedy matcher over `history + data` (linked blocks); only `data` is emitted."""
    buf = history + data
    base = len(history)
    n = len(buf)
    out = bytearray()
    table = {}
    for j in range(max(0, base - 65535), base - 3):   # seed the table with the history window
        table[buf[j:j + 4]] = j
    i, lit = base, base
    limit = n - 12                                      # parsing restrictions: last match starts >= 12 B before the end
    while i < limit:
        key = buf[i:i + 4]
        cand = table.get(key)
        table[key] = i
        if cand is not None and 0 < i - cand <= 65535:
            m = 4
            while i + m < n - 5 and buf[cand + m] == buf[i + m]:   # the last 5 bytes are always literals
                m += 1
            ll = i - lit
            tok = (min(ll, 15) << 4) | min(m - 4, 15)
            out.append(tok)
            if ll >= 15:
                out += _len_bytes(ll - 15)
            out += buf[lit:i]
            out += struct.pack("<H", i - cand)
            if m - 4 >= 15:
                out += _len_bytes(m - 4 - 15)
            i += m
            lit = i
        else:
            i += 1
    ll = n - lit
    out.append(min(ll, 15) << 4)
    if ll >= 15:
        out += _len_bytes(ll - 15)
    out += buf[lit:n]
    return bytes(out)


def compress_frame(data: bytes, block_size_id=4, linked=True, content_size=False, block_checksum=False,
                   content_checksum=False) -> bytes:
    bmax = 1 << (8 + 2 * block_size_id)
    flg = (1 << 6) | (0 if linked else 1 << 5) | (1 << 4 if block_checksum else 0) | (1 << 3 if content_size else 0) | \
          (1 << 2 if content_checksum else 0)
    desc = bytes([flg, block_size_id << 4]) + (struct.pack("<Q", len(data)) if content_size else b"")
    out = bytearray(b"\x04\x22\x4d\x18" + desc + bytes([(xxhash.xxh32(desc, seed=0).intdigest() >> 8) & 0xFF]))
    for i in range(0, len(data), bmax):
        chunk = data[i:i + bmax]
        c = compress_block(chunk, data[max(0, i - 65535):i] if linked else b"")
        if len(c) >= len(chunk):                        # incompressible: stored block
            out += struct.pack("<I", len(chunk) | 0x80000000) + chunk
            blk = chunk
        else:
            out += struct.pack("<I", len(c)) + c
            blk = c
        if block_checksum:
            out += struct.pack("<I", xxhash.xxh32(blk, seed=0).intdigest())
    out += struct.pack("<I", 0)
    if content_checksum:
        out += struct.pack("<I", xxhash.xxh32(data, seed=0).intdigest())
    return bytes(out)


def decompress_frame(src: bytes) -> bytes:
    assert src[:4] == b"\x04\x22\x4d\x18"
    flg = src[4]
    i = 6 + (8 if flg & 8 else 0) + (4 if flg & 1 else 0) + 1
    out = bytearray()
    while True:
        w = struct.unpack_from("<I", src, i)[0]; i += 4
        if w == 0:
            break
        sz = w & 0x7FFFFFFF
        blk = src[i:i + sz]; i += sz + (4 if flg & 0x10 else 0)
        if w & 0x80000000:
            out += blk
            continue
        q = 0
        while q < len(blk):
            tok = blk[q]; q += 1
            ll = tok >> 4
            if ll == 15:
                while True:
                    b = blk[q]; q += 1; ll += b
                    if b != 255:
                        break
            out += blk[q:q + ll]; q += ll
            if q == len(blk):
                break
            off = blk[q] | (blk[q + 1] << 8); q += 2
            ml = tok & 15
            if ml == 15:
                while True:
                    b = blk[q]; q += 1; ml += b
                    if b != 255:
                        break
            ml += 4
            assert 0 < off <= len(out)
            for _ in range(ml):
                out.append(out[-off])
    return bytes(out)
