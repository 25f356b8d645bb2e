"""Record sets used by the Kafka-decode tests (CPU and GPU): built with the independent encoder in
kafka_format.py, so the expected columns are known by construction."""
import numpy as np

import kafka_format as K


def random_record_set(rng, n_batches, max_records=40, partition=3, key_space=50, with_noise=True, big=False,
                      snappy=False):
    """-> (blob bytes, expected (part, klen, vlen, ts, keys) lists, info dict)."""
    blob = bytearray()
    batches = []
    info = {"control": 0, "compressed": 0, "old_magic": 0, "batches": 0}
    offset = int(rng.integers(0, 1 << 40))
    for b in range(n_batches):
        base_ts = int(1_600_000_000_000 + rng.integers(-10**9, 10**9))
        n = int(rng.integers(1, max_records + 1))
        recs = []
        for _ in range(n):
            kid = int(rng.integers(0, key_space))
            r = rng.random()
            key = None if r < 0.1 else (b"" if r < 0.13 else (b"key-%d-" % kid) + bytes([kid % 251]) * (kid % 37))
            if big and rng.random() < 0.05:
                key = bytes(rng.integers(0, 256, size=int(rng.integers(100, 400)), dtype=np.uint8))
            if big and rng.random() < 0.01:  # a key larger than the decoder's 8 KiB LDS window
                key = bytes(rng.integers(0, 256, size=int(rng.integers(9000, 20000)), dtype=np.uint8))
            r = rng.random()
            vlen = int(rng.integers(1, 20000 if big else 600))
            val = None if r < 0.2 else (b"" if r < 0.23 else
                                        (bytes(vlen) if not snappy else (b"payload-%d " % kid) * (vlen // 12 + 1)))
            headers = [(b"h%d" % i, None if i % 3 == 0 else b"x" * i) for i in range(int(rng.integers(0, 4)))]
            recs.append((int(rng.integers(-5000, 500000)), key, val, headers))
        kind = rng.random() if with_noise else 1.0
        attrs = 0
        max_ts = None
        if kind < 0.06:
            attrs = 0x20 | 0x10  # control batch (transactional marker): skipped
            info["control"] += 1
        elif kind < 0.10:
            attrs = int(rng.choice([5, 6, 7]))  # an unknown codec: skipped
            info["compressed"] += 1
        elif kind < 0.2:
            attrs = 0x08  # LogAppendTime: every record carries maxTimestamp
            max_ts = base_ts + 777
        elif kind < 0.25:
            attrs = 0x10  # transactional data batch: decoded like any other
        if attrs & 0x27 == 0 or attrs & 0x20 == 0 and attrs & 0x07 == 0:
            info["batches"] += 1
        mt = max_ts if max_ts is not None else max(base_ts + r[0] for r in recs)
        comp = None
        if snappy and attrs & 0x27 == 0 and rng.random() < 0.6:  # gzip / Snappy / LZ4 batches, all framings
            comp = str(rng.choice(["snappy", "snappy-xerial", "lz4", "lz4-indep", "gzip", "gzip-fixed", "gzip-stored",
                                   "gzip-named", "zstd", "zstd-stream", "zstd-19"]))
            kind_key = comp.split("-")[0]
            info[kind_key] = info.get(kind_key, 0) + 1
        blob += K.encode_batch(offset, recs, base_ts, attributes=attrs, max_ts=mt, compression=comp)
        codec = 0 if not comp else {"gzip": 1, "snappy": 2, "lz4": 3, "zstd": 4}[comp.split("-")[0]]
        batches.append((base_ts, attrs | codec, mt, recs))
        offset += n
        if with_noise and rng.random() < 0.03:  # an old-format (magic 1) message set: skipped
            blob += K.encode_batch(offset, [(0, b"old", b"fmt")], base_ts, magic=1)
            info["old_magic"] += 1
    return bytes(blob), K.expected_columns(partition, batches), info


def assert_columns(cols, expected, key_check=True):
    part, klen, vlen, ts, keys = expected
    assert list(cols["partition"]) == part
    assert list(cols["key_len"]) == klen
    assert list(cols["val_len"]) == vlen
    assert list(cols["ts_ms"]) == ts
    if key_check and "key_off" in cols:
        kb = cols["key_bytes"].tobytes()
        for i, k in enumerate(keys):
            if k:
                o = int(cols["key_off"][i])
                assert kb[o:o + len(k)] == k, i
