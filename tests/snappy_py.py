"""A small Snappy COMPRESSOR and decompressor in pure Python (test infrastructure): produces valid
blocks with literals and all three copy forms, so the product's and the oracle's inflaters can be
fed real compressed data.  Format: google/snappy format_description.txt; stream framing: snappy-java."""
import struct


def _varint(n):
    out = bytearray()
    while True:
        b = n & 0x7F
        n >>= 7
        if n:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _emit_literal(out, data):
    n = len(data) - 1
    if n < 60:
        out.append(n << 2)
    else:
        nb = (n.bit_length() + 7) // 8
        out.append((59 + nb) << 2)
        out += n.to_bytes(nb, "little")
    out += data


def _emit_copy(out, offset, length):
    while length > 0:
        if 4 <= length <= 11 and offset < 2048:
            out.append(1 | ((length - 4) << 2) | ((offset >> 8) << 5))
            out.append(offset & 0xFF)
            return
        l = min(length, 64)
        if length - l in (1, 2, 3):  # keep the remainder encodable by any form
            l = length - 4 if length - 4 >= 1 and length - 4 <= 64 else l
        if offset < 65536:
            out.append(2 | ((l - 1) << 2))
            out += struct.pack("<H", offset)
        else:
            out.append(3 | ((l - 1) << 2))
            out += struct.pack("<I", offset)
        length -= l


def compress_block(data: bytes) -> bytes:
    out = bytearray(_varint(len(data)))
    table = {}
    i, lit = 0, 0
    n = len(data)
    while i + 4 <= n:
        key = data[i:i + 4]
        cand = table.get(key)
        table[key] = i
        if cand is not None and i - cand <= 0xFFFFFF:
            m = 4
            while i + m < n and data[cand + m] == data[i + m] and m < 4096:
                m += 1
            if lit < i:
                _emit_literal(out, data[lit:i])
            _emit_copy(out, i - cand, m)
            i += m
            lit = i
        else:
            i += 1
    if lit < n:
        _emit_literal(out, data[lit:n])
    return bytes(out)


def compress_xerial(data: bytes, block=32768) -> bytes:
    out = bytearray(b"\x82SNAPPY\x00" + struct.pack(">ii", 1, 1))
    for i in range(0, len(data), block):
        c = compress_block(data[i:i + block])
        out += struct.pack(">I", len(c)) + c
    return bytes(out)


def decompress_block(src: bytes) -> bytes:
    ulen, shift, i = 0, 0, 0
    while True:
        b = src[i]; i += 1
        ulen |= (b & 0x7F) << shift
        shift += 7
        if not b & 0x80:
            break
    out = bytearray()
    while i < len(src):
        tag = src[i]; i += 1
        t = tag & 3
        if t == 0:
            l = tag >> 2
            if l >= 60:
                nb = l - 59
                l = int.from_bytes(src[i:i + nb], "little"); i += nb
            l += 1
            out += src[i:i + l]; i += l
        else:
            if t == 1:
                l = 4 + ((tag >> 2) & 7); off = ((tag >> 5) << 8) | src[i]; i += 1
            elif t == 2:
                l = 1 + (tag >> 2); off = struct.unpack_from("<H", src, i)[0]; i += 2
            else:
                l = 1 + (tag >> 2); off = struct.unpack_from("<I", src, i)[0]; i += 4
            assert 0 < off <= len(out)
            for _ in range(l):
                out.append(out[-off])
    assert len(out) == ulen
    return bytes(out)
