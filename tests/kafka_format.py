"""An independent Kafka record-batch v2 ENCODER (test infrastructure): builds wire-format record sets
from plain record lists, so the expected decode of every blob is known by construction.
Format: Apache Kafka protocol guide, "Record Batch" (KIP-98); CRC-32C (Castagnoli) over the bytes
from `attributes` to the end of the batch."""
import struct

_CRC_TABLE = []
for _i in range(256):
    _c = _i
    for _ in range(8):
        _c = (_c >> 1) ^ 0x82F63B78 if _c & 1 else _c >> 1
    _CRC_TABLE.append(_c)


def crc32c(data: bytes) -> int:
    c = 0xFFFFFFFF
    for b in data:
        c = _CRC_TABLE[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def varint(v: int) -> bytes:
    """zig-zag base-128 (works for 32- and 64-bit values)."""
    z = (v << 1) ^ (v >> 63)
    z &= (1 << 64) - 1
    out = bytearray()
    while True:
        b = z & 0x7F
        z >>= 7
        if z:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def encode_record(offset_delta, ts_delta, key, value, headers=()):
    body = bytearray()
    body += b"\x00"  # record attributes
    body += varint(ts_delta)
    body += varint(offset_delta)
    if key is None:
        body += varint(-1)
    else:
        body += varint(len(key)) + key
    if value is None:
        body += varint(-1)
    else:
        body += varint(len(value)) + value
    body += varint(len(headers))
    for hk, hv in headers:
        body += varint(len(hk)) + hk
        body += varint(-1) if hv is None else varint(len(hv)) + hv
    return varint(len(body)) + bytes(body)


def encode_batch(base_offset, records, base_ts, attributes=0, max_ts=None, producer_id=-1, magic=2,
                 raw_records=None, count=None, compression=None, last_offset_delta=None):
    """records: [(ts_delta, key|None, value|None, headers)]; returns the batch bytes.
    raw_records/count let tests build corrupt or compressed-looking batches.
    compression: None, "snappy" (one bare block, as librdkafka writes), "snappy-xerial" (snappy-java
    stream framing, as the Java clients write) -> codec 2; "lz4" (LZ4 frame, linked 64 KiB blocks, as
    librdkafka's LZ4F defaults) or "lz4-indep" (independent blocks + block checksums + content size,
    as the Java client's KafkaLZ4BlockOutputStream can) -> codec 3; "snappy-lib" / "lz4-lib": the same
    two codecs from the real libraries (Google snappy, lz4 frame) through pyarrow; "gzip" (zlib level 6:
    dynamic Huffman blocks), "gzip-fixed" (Z_FIXED), "gzip-stored" (level 0), "gzip-named" (GzipFile with
    a file name in the header) -> codec 1; "zstd" (libzstd level 3, one-shot: content size in the frame
    header), "zstd-19", "zstd-stream" (streaming API: no content size) through pyarrow -> codec 4.
    The records section is compressed."""
    recs = b"".join(encode_record(i, r[0], r[1], r[2], r[3] if len(r) > 3 else ()) for i, r in enumerate(records)) \
        if raw_records is None else raw_records
    if compression in ("snappy", "snappy-xerial"):
        import snappy_py
        recs = snappy_py.compress_block(recs) if compression == "snappy" else snappy_py.compress_xerial(recs, 4096)
        attributes = (attributes & ~0x07) | 2
    elif compression in ("gzip", "gzip-fixed", "gzip-stored", "gzip-named"):
        import gzip as _gzip
        import io
        import zlib
        if compression == "gzip-named":          # FNAME + MTIME header fields, as GzipFile writes them
            f = io.BytesIO()
            with _gzip.GzipFile(filename="records.bin", mode="wb", fileobj=f, mtime=1600000000) as g:
                g.write(recs)
            recs = f.getvalue()
        else:
            level = 0 if compression == "gzip-stored" else 6
            strategy = zlib.Z_FIXED if compression == "gzip-fixed" else zlib.Z_DEFAULT_STRATEGY
            co = zlib.compressobj(level, zlib.DEFLATED, 15 + 16, 8, strategy)
            recs = co.compress(recs) + co.flush()
        attributes = (attributes & ~0x07) | 1
    elif compression in ("zstd", "zstd-stream", "zstd-19"):
        import pyarrow as pa
        if compression == "zstd-stream":         # streaming API: no Frame_Content_Size, a window descriptor instead
            sink = pa.BufferOutputStream()
            with pa.CompressedOutputStream(sink, "zstd") as o:
                o.write(recs)
            recs = sink.getvalue().to_pybytes()
        else:
            recs = pa.Codec("zstd", compression_level=19 if compression == "zstd-19" else 3).compress(recs, asbytes=True)
        attributes = (attributes & ~0x07) | 4
    elif compression in ("snappy-lib", "lz4-lib"):
        import pyarrow as pa
        recs = pa.compress(recs, codec=compression[:-4], asbytes=True)
        attributes = (attributes & ~0x07) | (2 if compression == "snappy-lib" else 3)
    elif compression in ("lz4", "lz4-indep"):
        import lz4_py
        recs = lz4_py.compress_frame(recs) if compression == "lz4" else \
            lz4_py.compress_frame(recs, linked=False, content_size=True, block_checksum=True, content_checksum=True)
        attributes = (attributes & ~0x07) | 3
    n = len(records) if count is None else count
    if max_ts is None:
        max_ts = max([base_ts + r[0] for r in records], default=base_ts)
    lod = max(n - 1, 0) if last_offset_delta is None else last_offset_delta   # a compacted batch keeps its original extent
    after_crc = struct.pack(">hiqqqhii", attributes, lod, base_ts, max_ts, producer_id, -1, -1, n) + recs
    crc = crc32c(after_crc)
    tail = struct.pack(">iBI", 0, magic, crc) + after_crc  # partitionLeaderEpoch, magic, crc
    return struct.pack(">qi", base_offset, len(tail)) + tail


def expected_columns(partition, batches):
    """batches: [(base_ts, attributes, max_ts, records)] -> the columns a consumer would deliver."""
    part, klen, vlen, ts, keys = [], [], [], [], []
    for base_ts, attributes, max_ts, records in batches:
        if attributes & 0x20 or (attributes & 0x07) > 4:   # control, or an unknown codec
            continue
        for r in records:
            part.append(partition)
            klen.append(-1 if r[1] is None else len(r[1]))
            vlen.append(-1 if r[2] is None else len(r[2]))
            ts.append(max_ts if attributes & 0x08 else base_ts + r[0])
            keys.append(r[1])
    return part, klen, vlen, ts, keys
