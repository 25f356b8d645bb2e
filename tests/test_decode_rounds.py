"""The record decode kernel's rounds on the host (kta_kafka_decode_rounds_host): the chain over the length prefixes
and the parse of one record are the device's own code (csrc/kta_records.h), run here lane after lane in the kernel's
geometries with poison behind every window's valid bytes, against the encoder's expectations and the C oracle
(oracle/kta_kafka_oracle.c).  The GPU tests of tests/test_kafka_decode.py run the same cases through the kernel."""
import ctypes as C
import functools

import numpy as np
import pytest

from kafka_topic_analyzer_amd import _native as N
import kafka_format as K
from kafka_cases import assert_columns, random_record_set
from oracle_c import kafka_decode
from test_kafka_decode import _forged_count_blob, index_host

# (lanes per batch, window bytes, records per round): the dispatcher's geometries (kta_kafka.hip) and two small ones
# that put a window edge into almost every record
GEOMETRIES = [(16, 3072, 16), (32, 8192, 32), (64, 8192, 256), (8, 1024, 16), (8, 256, 8), (4, 64, 4)]


def rounds_host(blob, partition, geometry, with_keys=True):
    lib = N.load()
    rc, descs, st = index_host(blob, partition)
    assert rc == N.KTA_OK
    n = int(st.n_records)
    cols = {"partition": np.full(n, -7, np.int32), "key_len": np.full(n, -7, np.int32),
            "val_len": np.full(n, -7, np.int32), "ts_ms": np.full(n, -7, np.int64), "key_off": np.zeros(n, np.uint32)}
    kb, bad = C.c_uint64(), C.c_uint64()
    lanes, window, per_round = geometry
    rc = lib.kta_kafka_decode_rounds_host(blob, len(blob), descs, st.n_batches, lanes, window, per_round,
                                          cols["partition"].ctypes.data, cols["key_len"].ctypes.data,
                                          cols["val_len"].ctypes.data, cols["ts_ms"].ctypes.data,
                                          cols["key_off"].ctypes.data if with_keys else None, C.byref(kb), C.byref(bad))
    assert rc == N.KTA_OK
    if with_keys:
        cols["key_bytes"] = np.frombuffer(blob, dtype=np.uint8)   # zero-copy keys: offsets into the blob
    else:
        del cols["key_off"]
    cols["n_key_bytes"] = kb.value
    return cols, descs, st, bad.value


@functools.lru_cache(maxsize=None)
def _random_case(seed, max_records):
    """One record set per seed for all geometries (the Python encoder is most of a case's time)."""
    rng = np.random.default_rng(seed)
    blob, expected, _ = random_record_set(rng, 60 if max_records < 1000 else 6, max_records=max_records, big=(seed == 5))
    want, _ = kafka_decode(blob, 3)
    return blob, expected, want


@pytest.mark.parametrize("geometry", GEOMETRIES)
@pytest.mark.parametrize("seed,max_records", [(1, 40), (2, 40), (3, 700), (5, 300), (6, 3000)])
def test_rounds_match_encoder_and_oracle(seed, max_records, geometry):
    blob, expected, want = _random_case(seed, max_records)
    cols, _, _, bad = rounds_host(blob, 3, geometry)
    assert bad == 0
    assert_columns(cols, expected)
    for k in ("partition", "key_len", "val_len", "ts_ms"):
        assert np.array_equal(cols[k], want[k]), k
    assert cols["n_key_bytes"] == int(np.maximum(want["key_len"], 0).sum())


@pytest.mark.parametrize("geometry", GEOMETRIES)
def test_rounds_report_corrupt_and_forged_batches(geometry):
    good = K.encode_batch(0, [(0, b"a", b"b"), (1, b"c", None)], 1000)
    short = K.encode_batch(10, [(0, b"a", b"b"), (1, b"c", None)], 1000, count=3)
    cols, _, _, bad = rounds_host(good + short + good, 0, geometry)
    assert bad == 1 and list(cols["partition"]) == [0, 0, 0, 0, -1, 0, 0]
    blob, _ = _forged_count_blob()
    want, ost = kafka_decode(blob, 5)
    cols, _, _, bad = rounds_host(blob, 5, geometry)
    assert bad == 1 == ost.bad_batches
    for k in ("partition", "key_len", "val_len", "ts_ms"):
        assert np.array_equal(cols[k], want[k]), k


def padded(v, nbytes):
    """zig-zag varint of v in exactly nbytes bytes (a padded encoding when that is more than it needs)."""
    z = (v << 1) ^ (v >> 63)
    out = bytearray()
    for i in range(nbytes):
        out.append((z & 0x7F) | (0x80 if i + 1 < nbytes else 0))
        z >>= 7
    assert z == 0
    return bytes(out)


def record(ts_delta, key, value, nb=(0, 0, 0, 0, 0), offset_delta=0, tail=b"\x00"):
    """A record whose five varints (length, timestamp delta, offset delta, key length, value length) take nb[i] bytes
    each (0: as short as possible)."""
    def vi(v, n):
        return padded(v, n) if n else K.varint(v)
    body = b"\x00" + vi(ts_delta, nb[1]) + vi(offset_delta, nb[2])
    body += vi(-1 if key is None else len(key), nb[3]) + (key or b"")
    body += vi(-1 if value is None else len(value), nb[4]) + (value or b"") + tail
    return vi(len(body), nb[0]) + body


UNUSUAL = [
    # (records as (ts_delta, key, value, byte counts), what it exercises)
    [(5, b"k", b"v", (4, 0, 0, 0, 0)), (6, b"kk", None, (0, 0, 0, 0, 0))],          # a four-byte length
    [(5, b"k", b"v", (5, 0, 0, 0, 0)), (6, b"kk", None, (0, 0, 0, 0, 0))],          # five bytes: the chain's long way
    [(5, b"k", b"v", (10, 0, 0, 0, 0)), (6, b"kk", b"", (3, 0, 0, 0, 0))],          # ten
    [(1 << 40, b"k", b"v", (0, 0, 0, 0, 0)), (-(1 << 50), None, b"x" * 300, (0, 0, 0, 0, 0))],   # long timestamp deltas
    [(5, b"k" * 200, b"v", (0, 4, 4, 4, 4)), (7, b"", b"v" * 5000, (2, 3, 2, 2, 4))],            # padded header fields
    [(5, b"k" * 70, b"v", (0, 5, 0, 0, 0)), (7, b"q", b"v", (0, 0, 5, 0, 0)), (8, b"q", b"v", (0, 0, 0, 5, 0)),
     (9, b"q", b"v" * 9, (0, 0, 0, 0, 5))],                                         # one five-byte field each
    [(5, bytes(range(256)) * 40, b"v" * 3, (0, 0, 0, 0, 0)), (7, b"q", b"w" * 30000, (0, 0, 0, 0, 0)),
     (9, b"after", None, (0, 0, 0, 0, 0))],                                         # key and value larger than a window
]


@pytest.mark.parametrize("geometry", GEOMETRIES)
@pytest.mark.parametrize("case", range(len(UNUSUAL)))
def test_rounds_take_unusual_encodings_like_the_oracle(case, geometry):
    recs = UNUSUAL[case]
    raw = b"".join(record(r[0], r[1], r[2], r[3], offset_delta=i) for i, r in enumerate(recs))
    filler = [(i, b"key-%d" % i, b"x" * (37 * i % 400)) for i in range(40)]
    blob = K.encode_batch(0, filler, 1000) + K.encode_batch(40, recs, 10**12, raw_records=raw) + \
        K.encode_batch(50, filler, 2000)
    want, ost = kafka_decode(blob, 2)
    assert ost.bad_batches == 0 and len(want["partition"]) == 80 + len(recs)
    assert list(want["key_len"][40:40 + len(recs)]) == [-1 if r[1] is None else len(r[1]) for r in recs]
    assert list(want["ts_ms"][40:40 + len(recs)]) == [10**12 + r[0] for r in recs]
    cols, _, _, bad = rounds_host(blob, 2, geometry)
    assert bad == 0
    for k in ("partition", "key_len", "val_len", "ts_ms"):
        assert np.array_equal(cols[k], want[k]), k
    for i, r in enumerate(recs):
        if r[1]:
            o = int(cols["key_off"][40 + i])
            assert blob[o:o + len(r[1])] == r[1]


MALFORMED = {
    "negative length": lambda: K.varint(-3) + b"\x00" * 8,
    "length overruns the batch": lambda: K.varint(500) + b"\x00\x02\x00\x02k\x02v\x00",
    "key length -2": lambda: (lambda body: K.varint(len(body)) + body)(b"\x00\x02\x00" + K.varint(-2) + b"\x02v\x00"),
    "key overruns the record": lambda: (lambda body: K.varint(len(body)) + body)(b"\x00\x02\x00" + K.varint(40) + b"kkk\x02v\x00"),
    "value length -2": lambda: (lambda body: K.varint(len(body)) + body)(b"\x00\x02\x00\x02k" + K.varint(-2) + b"\x00"),
    "value overruns the record": lambda: (lambda body: K.varint(len(body)) + body)(b"\x00\x02\x00\x02k" + K.varint(90) + b"vv\x00"),
    "header cut by the record's end": lambda: K.varint(2) + b"\x00\x02" + b"\x00\x02k\x02v\x00",
    "three stray bytes": lambda: b"\x10\x00\x02",
    "unterminated length": lambda: b"\xff" * 12,
}


@pytest.mark.parametrize("geometry", GEOMETRIES)
@pytest.mark.parametrize("what", sorted(MALFORMED))
@pytest.mark.parametrize("before", [0, 3, 60])
def test_rounds_condemn_malformed_records_like_the_oracle(before, what, geometry):
    """A malformed record after `before` good ones: the batch is reported, the records the oracle still delivers may
    be withheld from the start of the round on, the neighbours are untouched."""
    good = [(i, b"key-%d" % i, b"x" * (53 * i % 300)) for i in range(before)]
    raw = b"".join(K.encode_record(i, *r) for i, r in enumerate(good)) + MALFORMED[what]()
    filler = [(i, b"f%d" % i, b"y" * i) for i in range(20)]
    blob = K.encode_batch(0, filler, 1000) + K.encode_batch(20, good + [(0, b"?", b"?")], 5000, raw_records=raw) + \
        K.encode_batch(100, filler, 2000)
    want, ost = kafka_decode(blob, 4)
    assert ost.bad_batches == 1 and list(want["partition"][20:20 + before + 1]) == [4] * before + [-1]
    cols, _, _, bad = rounds_host(blob, 4, geometry)
    assert bad == 1
    part = cols["partition"]
    assert (part[:20] == 4).all() and (part[-20:] == 4).all() and part[20 + before] == -1
    delivered = int((part[20:20 + before] == 4).sum())
    assert (part[20:20 + delivered] == 4).all() and (part[20 + delivered:20 + before + 1] == -1).all()
    for k in ("key_len", "val_len", "ts_ms"):
        assert np.array_equal(cols[k][:20 + delivered], want[k][:20 + delivered]), k
        assert (cols[k][20 + delivered:20 + before + 1] == -1).all(), k
        assert np.array_equal(cols[k][-20:], want[k][-20:]), k


@pytest.mark.parametrize("seed", range(6))
def test_rounds_agree_with_the_oracle_on_mutated_records(seed):
    """Random damage inside the records of plain batches: the same batches are reported as by the oracle, undamaged
    batches decode as before, and inside a reported batch whatever is delivered is what the oracle delivers."""
    rng = np.random.default_rng(100 + seed)
    blob, _, _ = random_record_set(rng, 8, max_records=120, with_noise=False, big=(seed % 2 == 1))
    _, descs, st, _ = rounds_host(blob, 1, GEOMETRIES[0])
    spans = [(descs[i].payload_off, descs[i].payload_end, descs[i].record_base, descs[i].n_records)
             for i in range(st.n_batches)]
    heads = []                                         # where the records begin: half of the damage goes to their headers
    for lo, hi, _, n in spans:
        at = lo
        for _ in range(n):
            heads.append(at)
            z, shift = 0, 0
            while True:
                b = blob[at]
                at += 1
                z |= (b & 0x7F) << shift
                shift += 7
                if not b & 0x80:
                    break
            at += z >> 1
        assert at == hi
    reported = 0
    for trial in range(120):
        hurt = bytearray(blob)
        for _ in range(int(rng.integers(1, 4))):
            lo, hi, _, _ = spans[int(rng.integers(0, len(spans)))]
            at = int(rng.integers(lo, hi))
            if rng.random() < 0.5:
                at = min(heads[int(rng.integers(0, len(heads)))] + int(rng.integers(0, 9)), len(blob) - 1)
            hurt[at] = int(rng.integers(0, 256)) if rng.random() < 0.5 else hurt[at] ^ (1 << int(rng.integers(0, 8)))
        hurt = bytes(hurt)
        want, ost = kafka_decode(hurt, 1)
        geometry = GEOMETRIES[trial % len(GEOMETRIES)]
        cols, _, _, bad = rounds_host(hurt, 1, geometry)
        assert bad == ost.bad_batches, (trial, geometry)
        reported += bad
        for lo, hi, base, n in spans:
            sl = slice(base, base + n)
            oracle_bad = (want["partition"][sl] == -1).any()
            assert (cols["partition"][sl] == -1).any() == oracle_bad, (trial, geometry)
            if not oracle_bad:
                for k in ("partition", "key_len", "val_len", "ts_ms"):
                    assert np.array_equal(cols[k][sl], want[k][sl]), (trial, geometry, k)
                continue
            mine, theirs = cols["partition"][sl], want["partition"][sl]
            delivered = int((mine != -1).sum())
            assert (mine[:delivered] == 1).all() and (mine[delivered:] == -1).all()
            assert delivered <= int((theirs != -1).sum())
            for k in ("key_len", "val_len", "ts_ms"):
                assert np.array_equal(cols[k][sl][:delivered], want[k][sl][:delivered]), (trial, geometry, k)
    assert reported > 20
