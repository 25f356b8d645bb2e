"""Kafka record-batch v2 decode.  CPU: the oracle decoder and the product's host
header index against record sets built by the independent encoder (expected output = encoder input)
and against a committed byte-level fixture.  GPU: the device decode and the end-to-end
`kta_kafka_consume` against the oracle."""
import ctypes as C
import sys
import functools
import json
import os

import numpy as np
import pytest

import kafka_topic_analyzer_amd as kta
from kafka_topic_analyzer_amd import _native as N
import kafka_format as K
from helpers import GOLDEN, NOW
from kafka_cases import assert_columns, random_record_set
from oracle_c import Oracle, kafka_decode


def index_host(blob, partition, cap=None):
    lib = N.load()
    st = N.KtaKafkaIndexStats()
    n = cap if cap is not None else 1 << 16
    descs = (N.KtaKafkaBatchDesc * n)()
    inflate_at = (len(blob) + 127) & ~63  # the inflate area of compressed batches follows the raw bytes
    rc = lib.kta_kafka_index_host(blob, len(blob), partition, 0, 0, inflate_at, descs, n, C.byref(st))
    return rc, descs, st


def test_encoder_primitives():
    assert K.crc32c(b"123456789") == 0xE3069283  # the CRC-32C check value
    assert [K.varint(v) for v in (0, -1, 1, 63, 64, 300)] == [b"\x00", b"\x01", b"\x02", b"\x7e", b"\x80\x01", b"\xd8\x04"]


def test_golden_fixture_bytes_and_decode():
    fx = json.load(open(os.path.join(GOLDEN, "kafka_v2_recordset.json")))
    blob = bytes.fromhex(fx["blob_hex"])
    cols, st = kafka_decode(blob, fx["partition"])
    e = fx["expect"]
    assert list(cols["partition"]) == e["partition"] and list(cols["key_len"]) == e["key_len"]
    assert list(cols["val_len"]) == e["val_len"] and list(cols["ts_ms"]) == e["ts_ms"]
    assert list(cols["offset"]) == e["offset"]
    assert cols["key_bytes"].tobytes().hex() == e["key_bytes_hex"]
    assert (st.control_batches, st.compressed_batches, st.old_magic_batches, st.trailing_bytes) == \
        (e["control_batches"], e["compressed_batches"], e["old_magic_batches"], e["trailing_bytes"])
    # every batch in the fixture carries a valid CRC-32C (the encoder is a faithful producer)
    pos = 0
    while pos + 12 <= len(blob) - e["trailing_bytes"]:
        total = 12 + int.from_bytes(blob[pos + 8:pos + 12], "big")
        if blob[pos + 16] == 2:
            assert int.from_bytes(blob[pos + 17:pos + 21], "big") == K.crc32c(blob[pos + 21:pos + total])
        pos += total


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5])
def test_oracle_decodes_what_the_encoder_built(seed):
    rng = np.random.default_rng(seed)
    blob, expected, info = random_record_set(rng, 40, big=(seed == 5))
    cols, st = kafka_decode(blob, 3)
    assert_columns(cols, expected)
    assert (st.control_batches, st.compressed_batches, st.old_magic_batches) == \
        (info["control"], info["compressed"], info["old_magic"])
    assert st.trailing_bytes == 0 and st.bad_batches == 0


def test_host_index_matches_oracle_and_handles_partial_tail():
    rng = np.random.default_rng(11)
    blob, expected, info = random_record_set(rng, 60)
    tail = K.encode_batch(1, [(0, b"k", b"v" * 100)], 1)[:40]  # a fetch response may end mid-batch
    rc, descs, st = index_host(blob + tail, 7)
    assert rc == N.KTA_OK
    cols, ost = kafka_decode(blob + tail, 7)
    assert st.n_records == len(cols["partition"]) == len(expected[0])
    assert st.n_batches == ost.batches
    assert (st.n_control_batches, st.n_compressed, st.n_old_magic) == (info["control"], info["compressed"], info["old_magic"])
    assert st.trailing_bytes == len(tail) == ost.trailing_bytes and st.bytes_consumed == len(blob)
    run = 0
    for i in range(st.n_batches):
        d = descs[i]
        assert d.record_base == run and d.partition == 7 and d.n_records > 0
        assert int.from_bytes(blob[d.byte_off + 57:d.byte_off + 61], "big") == d.n_records
        assert d.batch_bytes == 12 + int.from_bytes(blob[d.byte_off + 8:d.byte_off + 12], "big")
        run += d.n_records
    # too few descriptors: reports how many are needed
    rc, _, st2 = index_host(blob, 7, cap=3)
    assert rc == N.KTA_ERR_CAPACITY and st2.n_batches == st.n_batches
    # garbage / empty input
    rc, _, st3 = index_host(b"\x00" * 7, 0)
    assert rc == N.KTA_OK and st3.n_batches == 0 and st3.trailing_bytes == 7


def test_host_index_reports_the_log_extent_a_consumer_sees():
    """Watermarks (src/kafka.rs:60-72 asks the broker; segment:// asks the file): the log starts at the first
    batch's baseOffset and ends at max(baseOffset + lastOffsetDelta + 1) — a compacted batch keeps its
    original extent although records are gone, and control batches (never delivered) count."""
    compacted = K.encode_batch(100, [(0, b"a", b"1"), (5, b"b", None)], 1000, last_offset_delta=9)   # offsets 100..109, 2 left
    data = K.encode_batch(110, [(0, b"c", b"2")], 1000)
    control = K.encode_batch(111, [(0, b"\0\0\0\0", b"")], 1000, attributes=0x30)                    # commit marker at 111
    rc, descs, st = index_host(compacted + data + control, 0)
    assert rc == N.KTA_OK and st.n_batches == 2 and st.n_records == 3 and st.n_control_batches == 1
    assert (st.any_offsets, st.first_offset, st.next_offset) == (1, 100, 112)
    rc, descs, st = index_host(control, 0)
    assert st.n_batches == 0 and (st.any_offsets, st.first_offset, st.next_offset) == (1, 111, 112)
    rc, descs, st = index_host(b"", 0)
    assert st.any_offsets == 0


def test_oracle_marks_corrupt_batches():
    good = K.encode_batch(0, [(0, b"a", b"b"), (1, b"c", None)], 1000)
    # recordsCount says 3 but only 2 records are present
    bad = K.encode_batch(10, [(0, b"a", b"b"), (1, b"c", None)], 1000, count=3)
    cols, st = kafka_decode(good + bad, 0)
    assert st.bad_batches == 1 and list(cols["partition"]) == [0, 0, 0, 0, -1]


def _forged_count_blob():
    good = K.encode_batch(0, [(0, b"a", b"b"), (1, b"c", None)], 1000)
    forged = K.encode_batch(10, [(0, b"k%d" % i, b"v") for i in range(5)], 1000, count=2**31 - 1)
    return good + forged + good, len(forged) - 61


def test_forged_record_count_is_clamped_and_reported():
    """A header announcing more records than its payload can hold (a record takes >= 7 bytes) is corrupt:
    the count is clamped — no billions of output slots — and none of the batch is delivered.  Index and
    oracle agree on what comes out."""
    blob, payload = _forged_count_blob()
    rc, descs, st = index_host(blob, 5)
    assert rc == N.KTA_OK and st.n_batches == 3
    assert [descs[i].status for i in range(3)] == [0, 2, 0]                  # KTA_KB_BAD_FRAMING
    assert descs[1].n_records == payload // 7 + 1 and st.n_records == 4 + descs[1].n_records
    assert descs[2].record_base == 2 + descs[1].n_records
    cols, ost = kafka_decode(blob, 5)
    assert ost.bad_batches == 1 and len(cols["partition"]) == st.n_records
    assert list(cols["partition"][:2]) == [5, 5] and (cols["partition"][2:-2] == -1).all()
    assert list(cols["partition"][-2:]) == [5, 5]


def test_corrupt_compressed_batch_keeps_its_announced_records():
    """A compressed batch whose size probe fails (here: a Snappy preamble claiming 34 GB) has no known
    payload size, so the >= 7 bytes-per-record clamp must not cut its record count: every record the
    header announces comes out flagged, and the record bases of the later batches do not move.  (CPU
    twin of the assertion in test_device_decodes_snappy_batches.)"""
    rng = np.random.default_rng(33)
    blob, _, info = random_record_set(rng, 160, max_records=120, snappy=True)
    rc, descs, st = index_host(blob, 3)
    assert rc == N.KTA_OK
    want, _ost = kafka_decode(blob, 3)
    for flag in (4, 8, 16):                                                    # Snappy, LZ4, gzip
        victim = next(i for i in range(st.n_batches) if descs[i].flags & flag)
        p = descs[victim].byte_off
        broken = bytearray(blob)
        broken[p + 61:p + 66] = b"\xff\xff\xff\xff\x7f"
        rc2, descs2, st2 = index_host(bytes(broken), 3)
        assert rc2 == N.KTA_OK and st2.n_batches == st.n_batches and st2.n_records == st.n_records
        assert [descs2[i].n_records for i in range(st.n_batches)] == [descs[i].n_records for i in range(st.n_batches)]
        assert [descs2[i].record_base for i in range(st.n_batches)] == [descs[i].record_base for i in range(st.n_batches)]
        if flag != 8:      # an LZ4 frame is sized from its block maximum, its corruption is found on the device
            assert descs2[victim].status == 2 and descs2[victim].payload_end == descs2[victim].payload_off
        got, ost = kafka_decode(bytes(broken), 3)                              # the oracle states the same rule
        lo, n = descs[victim].record_base, descs[victim].n_records
        assert ost.bad_batches == 1 and len(got["partition"]) == st.n_records
        assert (got["partition"][lo:lo + n] == -1).all() and (got["partition"] == -1).sum() == n
        keep = np.ones(st.n_records, dtype=bool)
        keep[lo:lo + n] = False
        for k in ("partition", "key_len", "val_len", "ts_ms"):
            assert np.array_equal(got[k][keep], want[k][keep]), k


def test_host_index_invariants_on_mutated_blobs():
    """The header walk on corrupted record sets: never crashes, and whatever it describes stays inside the
    blob / the inflate area it sized — the device trusts these descriptors."""
    rng = np.random.default_rng(123)
    base, _, _ = random_record_set(rng, 40, max_records=30, snappy=True)
    checked = 0
    for it in range(400):
        blob = bytearray(base)
        for _ in range(int(rng.integers(1, 6))):
            kind = int(rng.integers(0, 3))
            at = int(rng.integers(0, len(blob)))
            if kind == 0:
                blob[at] ^= 1 << int(rng.integers(0, 8))
            elif kind == 1:
                blob[at] = int(rng.integers(0, 256))
            else:
                del blob[at:at + int(rng.integers(1, 40))]
        blob = bytes(blob)
        inflate_at = (len(blob) + 127) & ~63
        rc, descs, st = index_host(blob, 1)
        assert rc == N.KTA_OK
        assert st.bytes_consumed + st.trailing_bytes == len(blob)
        rec, hi = 0, inflate_at
        for i in range(st.n_batches):
            d = descs[i]
            assert d.byte_off + d.batch_bytes <= st.bytes_consumed and d.batch_bytes >= 61
            assert d.record_base == rec and 0 < d.n_records
            rec += d.n_records
            if d.flags & (4 | 8 | 16 | 32):
                assert inflate_at <= d.payload_off <= d.payload_end <= d.scratch_end
                assert d.payload_off >= hi - 0 and d.payload_off % 64 == 0     # slices do not overlap
                hi = (d.scratch_end + 63) & ~63
                clen = d.batch_bytes - 61                                      # bound from the header alone:
                k = 1032 if d.flags & 16 else (22 if d.flags & 4 else (255 if d.flags & 8 else None))
                payload = clen * k + 64 if k else max(d.payload_end - d.payload_off, clen * 1032 + 64)
            else:
                assert (d.payload_off, d.payload_end) == (d.byte_off + 61, d.byte_off + d.batch_bytes)
                payload = d.batch_bytes - 61
            assert d.n_records <= payload // 7 + 1                            # forged counts are clamped
            checked += 1
        assert rec == st.n_records and hi - inflate_at <= st.inflate_bytes
    assert checked > 5000


# --------------------------------------------------------------------------------------------- GPU
def _decode_on_device(h, blob, partition, with_keys):
    lib = N.load()
    rc, descs, st = index_host(blob, partition)
    assert rc == N.KTA_OK
    n = st.n_records
    d_blob = C.c_void_p()
    out = h.device_batch_alloc(max(n, 1), 16 if with_keys else 0)  # key_off only: keys stay in the blob
    buf_bytes = ((len(blob) + 127) & ~63) + st.inflate_bytes + 128   # raw bytes + inflate area + padding
    blob_dev = h.device_batch_alloc(buf_bytes // 4 + 1)  # a column serves as the raw byte buffer
    arr = np.frombuffer(blob + b"\0" * ((-len(blob)) % 4), dtype=np.uint8).copy()
    h._check(lib.kta_copy_to_device(h._ctx, blob_dev.partition, arr.ctypes.data, arr.nbytes))
    kb, bad = C.c_uint64(), C.c_uint64()
    h._check(lib.kta_kafka_decode_device(h._ctx, blob_dev.partition, len(blob), descs, st.n_batches, n, C.byref(out),
                                         C.byref(kb), C.byref(bad)))
    cols = h.download_batch(out, n, 0)
    if with_keys:  # zero-copy: key_off indexes the device buffer (raw blob, then the inflate area)
        whole = np.empty(buf_bytes, dtype=np.uint8)
        h._check(lib.kta_copy_to_host(h._ctx, whole.ctypes.data, blob_dev.partition, buf_bytes))
        cols["key_bytes"] = whole
    cols["n_key_bytes"] = kb.value
    h.device_batch_free(out)
    h.device_batch_free(blob_dev)
    return cols, st, bad.value


@functools.lru_cache(maxsize=None)
def _device_decode_case(seed, max_records):
    """One record set (and the oracle's columns) per seed for all variants: seed 5's 360 MB take the Python encoder
    half a minute."""
    rng = np.random.default_rng(seed)
    blob, expected, _ = random_record_set(rng, 300 if max_records < 1000 else 12, max_records=max_records,
                                          big=(seed == 5))
    want, _ = kafka_decode(blob, 3)
    return blob, expected, want


@pytest.mark.gpu
@pytest.mark.parametrize("variant", [0, 1, 2, 10, 11])
@pytest.mark.parametrize("seed,with_keys,max_records", [(1, True, 40), (2, False, 40), (3, True, 700), (5, True, 300),
                                                        (6, True, 3000)])
def test_device_decode_matches_encoder_and_oracle(seed, with_keys, max_records, variant):
    """Every decode kernel (1 / 4 / 8 batches per wave through LDS windows, lane-per-batch); batches from one
    record up to thousands (many LDS windows), values/keys larger than a window with seed 5."""
    blob, expected, want = _device_decode_case(seed, max_records)
    with kta.HipMetricHandler(8, now=NOW) as h:
        h._check(N.load().kta_kafka_set_variant(h._ctx, variant))
        cols, st, bad = _decode_on_device(h, blob, 3, with_keys)
    assert bad == 0
    assert_columns(cols, expected, key_check=with_keys)   # keys compared by content through key_off
    for k in ("partition", "key_len", "val_len", "ts_ms"):
        assert np.array_equal(cols[k], want[k]), k
    if with_keys:
        assert cols["n_key_bytes"] == len(want["key_bytes"]) == int(np.maximum(want["key_len"], 0).sum())


@pytest.mark.gpu
def test_device_decode_reports_corrupt_batches():
    good = K.encode_batch(0, [(0, b"a", b"b"), (1, b"c", None)], 1000)
    bad = K.encode_batch(10, [(0, b"a", b"b"), (1, b"c", None)], 1000, count=3)
    blob = good + bad + good
    want, ost = kafka_decode(blob, 0)
    for variant in (0, 1, 2, 10, 11):
        with kta.HipMetricHandler(2, now=NOW) as h:
            h._check(N.load().kta_kafka_set_variant(h._ctx, variant))
            cols, st, nbad = _decode_on_device(h, blob, 0, True)
            assert nbad == 1 == ost.bad_batches
            assert list(cols["partition"]) == list(want["partition"]) == [0, 0, 0, 0, -1, 0, 0]


# kta_kafka_set_variant -> (lanes per batch, window bytes, records per round) of kafka_decode_coop (csrc/kta_kafka.hip)
GEOMETRY_OF_VARIANT = {2: (64, 8192, 256), 10: (16, 3072, 16), 11: (32, 8192, 32)}


@pytest.mark.gpu
def test_device_rounds_equal_their_host_statement_bit_for_bit():
    """The kernel against kta_kafka_decode_rounds_host in the same geometry: unusual encodings (padded and long
    varints, keys and values larger than a window), every malformed record of tests/test_decode_rounds.py and randomly
    damaged record sets — identical columns, including WHICH records of a reported batch are withheld."""
    import test_decode_rounds as R
    blobs = []
    filler = [(i, b"key-%d" % i, b"x" * (37 * i % 400)) for i in range(40)]
    for recs in R.UNUSUAL:
        raw = b"".join(R.record(r[0], r[1], r[2], r[3], offset_delta=i) for i, r in enumerate(recs))
        blobs.append(K.encode_batch(0, filler, 1000) + K.encode_batch(40, recs, 10**12, raw_records=raw) +
                     K.encode_batch(50, filler, 2000))
    for what in sorted(R.MALFORMED):
        for before in (0, 3, 60):
            good = [(i, b"key-%d" % i, b"x" * (53 * i % 300)) for i in range(before)]
            raw = b"".join(K.encode_record(i, *r) for i, r in enumerate(good)) + R.MALFORMED[what]()
            blobs.append(K.encode_batch(0, filler, 1000) +
                         K.encode_batch(20, good + [(0, b"?", b"?")], 5000, raw_records=raw) +
                         K.encode_batch(100, filler, 2000))
    rng = np.random.default_rng(77)
    clean, _, _ = random_record_set(rng, 8, max_records=120, with_noise=False, big=True)
    _, descs, st = index_host(clean, 1)
    for _ in range(40):
        hurt = bytearray(clean)
        for _ in range(int(rng.integers(1, 4))):
            d = descs[int(rng.integers(0, st.n_batches))]
            at = int(rng.integers(d.payload_off, d.payload_end))
            hurt[at] = int(rng.integers(0, 256))
        blobs.append(bytes(hurt))
    reported = 0
    with kta.HipMetricHandler(8, now=NOW) as h:
        for variant, geometry in sorted(GEOMETRY_OF_VARIANT.items()):
            h._check(N.load().kta_kafka_set_variant(h._ctx, variant))
            for n, blob in enumerate(blobs):
                want, _, _, want_bad = R.rounds_host(blob, 1, geometry)
                cols, _, bad = _decode_on_device(h, blob, 1, True)
                assert bad == want_bad, (variant, n)
                reported += bad
                for k in ("partition", "key_len", "val_len", "ts_ms", "key_off"):
                    assert np.array_equal(cols[k], want[k]), (variant, n, k)
                assert cols["n_key_bytes"] == want["n_key_bytes"], (variant, n)
    assert reported > 25 * len(GEOMETRY_OF_VARIANT)      # (27 batches of the case list are reported, per geometry)


@pytest.mark.gpu
def test_device_does_not_deliver_a_batch_with_a_forged_record_count():
    blob, _ = _forged_count_blob()
    want, ost = kafka_decode(blob, 5)
    for variant in (0, 1):
        with kta.HipMetricHandler(8, now=NOW) as h:
            h._check(N.load().kta_kafka_set_variant(h._ctx, variant))
            cols, st, nbad = _decode_on_device(h, blob, 5, True)
            assert nbad == 1 == ost.bad_batches
            for k in ("partition", "key_len", "val_len", "ts_ms"):
                assert np.array_equal(cols[k], want[k]), k


@pytest.mark.gpu
@pytest.mark.parametrize("count_alive", [False, True])
def test_consume_raw_record_sets_end_to_end(count_alive):
    """Raw record sets of several partitions -> kta_kafka_consume -> the same metrics (and alive-key
    set) as the oracle handlers run over the oracle's decode, in the same consumption order."""
    lib = N.load()
    rng = np.random.default_rng(42)
    P = 4
    o = Oracle(NOW, count_alive)
    with kta.HipMetricHandler(P, count_alive_keys=count_alive, now=NOW) as h:
        for fetch in range(6):
            part = fetch % P
            blob, expected, _ = random_record_set(rng, 50, partition=part, key_space=80)
            st = N.KtaKafkaIndexStats()
            h._check(lib.kta_kafka_consume(h._ctx, blob, len(blob), part, C.byref(st)))
            cols, _ = kafka_decode(blob, part)
            assert st.n_records == len(cols["partition"])
            o.run_soa({k: v for k, v in cols.items() if k != "offset"})
        res, c = h.finish()
        assert np.array_equal(c, o.counters(P))
        mm = kta.MessageMetrics(res, c, NOW)
        assert mm.earliest_message() == o.earliest() and mm.latest_message() == o.latest()
        assert mm.smallest_message() == o.get("smallest_message") and mm.largest_message() == o.get("largest_message")
        assert mm.overall_size() == o.get("overall_size")
        if count_alive:
            assert res.alive_keys == o.alive_keys()
            assert np.array_equal(h.export_alive_bitmap(), o.alive_words())


# --------------------------------------------------------------------------------------------- CRC-32C
def _batches_of(blob):
    out, pos = [], 0
    while pos + 12 <= len(blob):
        total = 12 + int.from_bytes(blob[pos + 8:pos + 12], "big")
        if total < 61 or pos + total > len(blob):
            break
        out.append((pos, total))
        pos += total
    return out


def test_crc32c_host_check_value_and_agreement():
    import ctypes as C2
    from oracle_c import lib as olib
    L = olib()
    L.kto_crc32c.restype = C2.c_uint32
    L.kto_crc32c.argtypes = [C2.c_char_p, C2.c_uint64]
    lib = N.load()
    assert lib.kta_crc32c_host(b"123456789", 9) == 0xE3069283 == L.kto_crc32c(b"123456789", 9) == K.crc32c(b"123456789")
    rng = np.random.default_rng(8)
    for n in (0, 1, 3, 4, 63, 64, 65, 4095, 4096, 4097, 10000):
        data = rng.integers(0, 256, size=n, dtype=np.uint8).tobytes()
        assert lib.kta_crc32c_host(data, n) == L.kto_crc32c(data, n) == K.crc32c(data)


@pytest.mark.gpu
def test_device_crc32c_check_flags_exactly_the_corrupted_batches():
    """check.crcs on the device: batches of every size class (shorter than a chunk .. many 4 KiB windows,
    arbitrary start alignment); flipping one byte anywhere after the CRC field must flag that batch, and
    only that batch; its records are not delivered (partition -1)."""
    import ctypes as C2
    from oracle_c import lib as olib
    L = olib()
    L.kto_kafka_batch_crc_ok.restype = C2.c_int
    L.kto_kafka_batch_crc_ok.argtypes = [C2.c_char_p, C2.c_uint64]
    rng = np.random.default_rng(77)
    blob = bytearray()
    for n, vmax in ((1, 3), (1, 40), (2, 70), (5, 900), (40, 300), (300, 600), (3, 30000), (1, 5), (17, 4096)):
        recs = [(int(rng.integers(0, 1000)), bytes(rng.integers(0, 256, size=int(rng.integers(0, 30)), dtype=np.uint8)),
                 bytes(rng.integers(0, 256, size=int(rng.integers(0, vmax)), dtype=np.uint8))) for _ in range(n)]
        blob += K.encode_batch(len(blob), recs, 1_600_000_000_000)
    blob = bytes(blob)
    batches = _batches_of(blob)
    assert all(L.kto_kafka_batch_crc_ok(blob[p:p + t], t) for p, t in batches)
    lib = N.load()
    with kta.HipMetricHandler(4, now=NOW) as h:
        h._check(lib.kta_kafka_set_check_crcs(h._ctx, 1))
        cols, st, nbad = _decode_on_device(h, blob, 1, True)
        assert nbad == 0 and (cols["partition"] == 1).all()
        # corrupt: one byte in batches 1, 4 and the last one (header region after the CRC, first and last byte)
        bad = bytearray(blob)
        victims = {1: 21, 4: None, len(batches) - 1: -1}
        for bi, where in victims.items():
            p, t = batches[bi]
            off = p + 21 if where == 21 else (p + t - 1 if where == -1 else p + 61 + (t - 61) // 2)
            bad[off] ^= 0x40
        bad = bytes(bad)
        assert [bool(L.kto_kafka_batch_crc_ok(bad[p:p + t], t)) for p, t in batches] == \
            [i not in victims for i in range(len(batches))]
        cols2, st2, nbad2 = _decode_on_device(h, bad, 1, True)
        n_err = C.c_uint64()
        h._check(lib.kta_kafka_crc_errors(h._ctx, C.byref(n_err)))
        assert n_err.value == len(victims) and nbad2 == len(victims)
        counts = [int.from_bytes(blob[p + 57:p + 61], "big") for p, _ in batches]
        want = np.concatenate([np.full(c, -1 if i in victims else 1, np.int32) for i, c in enumerate(counts)])
        assert np.array_equal(cols2["partition"], want)
        # with check.crcs off (librdkafka's default) the corrupted value/header bytes go unnoticed or are
        # caught by framing only: no CRC errors are reported
        h._check(lib.kta_kafka_set_check_crcs(h._ctx, 0))
        _decode_on_device(h, bad, 1, True)
        h._check(lib.kta_kafka_crc_errors(h._ctx, C.byref(n_err)))
        assert n_err.value == len(victims)


# --------------------------------------------------------------------------------------------- Snappy
def test_snappy_inflate_host_matches_oracle_and_python():
    """The product's inflater (the same function the device runs, compiled for the host) against the
    oracle's independent inflater and the Python decompressor, on data from the Python compressor."""
    import ctypes as C2
    import snappy_py as S
    from oracle_c import lib as olib
    L = olib()
    L.kto_snappy_inflate.restype = C2.c_int64
    L.kto_snappy_inflate.argtypes = [C2.c_char_p, C2.c_uint64, C2.c_char_p, C2.c_uint64]
    lib = N.load()
    rng = np.random.default_rng(3)
    cases = [b"", b"a", b"abcd" * 1000, bytes(rng.integers(0, 256, size=5000, dtype=np.uint8)), b"\0" * 100000,
             bytes(rng.integers(0, 4, size=70000, dtype=np.uint8)),
             b"".join(b"key-%d value-%d;" % (i % 97, i) for i in range(3000))]
    for d in cases:
        for comp in (S.compress_block(d), S.compress_xerial(d, 4096), S.compress_xerial(d)):
            out1, out2 = C2.create_string_buffer(len(d) + 1), C2.create_string_buffer(len(d) + 1)
            assert lib.kta_snappy_inflate_host(comp, len(comp), out1, len(d)) == len(d)
            assert L.kto_snappy_inflate(comp, len(comp), out2, len(d)) == len(d)
            assert out1.raw[:len(d)] == d == out2.raw[:len(d)]
            if not comp.startswith(b"\x82SNAPPY"):
                assert S.decompress_block(comp) == d
    # malformed input is refused, not mis-decoded: truncated stream, zero / too-far copy offset, short output buffer
    good = S.compress_block(b"abcdefgh" * 50)
    out = C2.create_string_buffer(1024)
    assert lib.kta_snappy_inflate_host(good[:-3], len(good) - 3, out, 1024) == -1
    assert lib.kta_snappy_inflate_host(good, len(good), out, 100) == -1
    bad_off = bytes([8, 0 << 2]) + b"a" + bytes([1 | (0 << 2), 9])   # len 8; literal "a"; copy offset 9 > produced
    assert lib.kta_snappy_inflate_host(bad_off, len(bad_off), out, 1024) == -1
    assert L.kto_snappy_inflate(bad_off, len(bad_off), out, 1024) == -1


def test_host_index_sizes_compressed_batches():
    rng = np.random.default_rng(21)
    blob, expected, info = random_record_set(rng, 60, snappy=True)
    rc, descs, st = index_host(blob, 3)
    assert rc == N.KTA_OK and st.n_snappy == info["snappy"] > 0 and st.n_lz4 == info["lz4"] > 0
    assert st.n_gzip == info["gzip"] > 0 and st.n_zstd == info["zstd"] > 0
    cols, ost = kafka_decode(blob, 3)
    assert_columns(cols, expected)
    inflate_at = (len(blob) + 127) & ~63
    run = 0
    for i in range(st.n_batches):
        d = descs[i]
        if d.flags & (4 | 8 | 16 | 32):  # KTA_KB_SNAPPY / _LZ4 / _GZIP / _ZSTD: a 64-byte aligned slice of the inflate area
            assert d.payload_off == inflate_at + run and d.payload_off % 64 == 0
            if d.flags & 8:    # LZ4 frames do not carry their size: the slice is a bound (<= 255x per block)
                clen = d.batch_bytes - 61
                assert 0 < d.payload_end - d.payload_off <= 255 * clen + 64 * (clen // 4)
            run += (d.payload_end - d.payload_off + 63) & ~63
            if d.flags & 32:   # zstd: the decoder's tables and literals follow the slice
                assert d.scratch_end > ((d.payload_end + 63) & ~63) + 10000
                run += (d.scratch_end - ((d.payload_end + 63) & ~63) + 63) & ~63
            elif d.flags & 16:  # gzip: the tokens of the two-stage inflate (count + out/3 + out/255 + 1 words, + the closing
                out = d.payload_end - d.payload_off     # tokens of the wave tokenizer's segments: kta_gzip.h, gz_closing_tokens)
                clen = d.batch_bytes - 61
                assert d.scratch_end == ((d.payload_end + 63) & ~63) + 8 + 4 * (out // 3 + out // 255 + 1 + 64 * (2 + clen // 2048))
                run += (d.scratch_end - ((d.payload_end + 63) & ~63) + 63) & ~63
            else:
                assert d.scratch_end == d.payload_end
        else:
            assert (d.payload_off, d.payload_end) == (d.byte_off + 61, d.byte_off + d.batch_bytes)
    assert run == st.inflate_bytes


def test_lz4_inflate_host_matches_oracle_and_python():
    import ctypes as C2
    import lz4_py as Z
    from oracle_c import lib as olib
    L = olib()
    L.kto_lz4_inflate.restype = C2.c_int64
    L.kto_lz4_inflate.argtypes = [C2.c_char_p, C2.c_uint64, C2.c_char_p, C2.c_uint64]
    lib = N.load()
    rng = np.random.default_rng(4)
    cases = [b"", b"a", b"abcd" * 1000, bytes(rng.integers(0, 256, size=5000, dtype=np.uint8)), b"\0" * 200000,
             bytes(rng.integers(0, 4, size=150000, dtype=np.uint8)),
             b"".join(b"key-%d value-%d;" % (i % 97, i) for i in range(9000))]
    for d in cases:
        for kw in ({}, {"linked": False}, {"content_size": True, "block_checksum": True, "content_checksum": True},
                   {"block_size_id": 5}):
            comp = Z.compress_frame(d, **kw)
            assert Z.decompress_frame(comp) == d
            out1, out2 = C2.create_string_buffer(len(d) + 1), C2.create_string_buffer(len(d) + 1)
            assert lib.kta_lz4_inflate_host(comp, len(comp), out1, len(d)) == len(d)
            assert L.kto_lz4_inflate(comp, len(comp), out2, len(d)) == len(d)
            assert out1.raw[:len(d)] == d == out2.raw[:len(d)]
    good = Z.compress_frame(b"abcdefgh" * 500)
    out = C2.create_string_buffer(8192)
    assert lib.kta_lz4_inflate_host(good[:-6], len(good) - 6, out, 8192) == -1      # truncated: no end mark
    assert lib.kta_lz4_inflate_host(good, len(good), out, 100) == -1                # output buffer too small
    assert lib.kta_lz4_inflate_host(b"\x04\x22\x4d\x19" + good[4:], len(good), out, 8192) == -1   # bad magic
    bad_off = good[:7] + bytes([5, 0, 0, 0]) + bytes([0x10, ord("a"), 9, 0, 0]) + bytes(4)  # match offset 9 > 1 byte produced
    assert lib.kta_lz4_inflate_host(bad_off, len(bad_off), out, 8192) == -1
    assert L.kto_lz4_inflate(bad_off, len(bad_off), out, 8192) == -1


def test_gzip_inflate_host_against_zlib():
    """The product's DEFLATE decoder (the function the device runs, compiled for the host) on members
    written by zlib at every level and strategy, by GzipFile (FNAME header) and by pyarrow; the oracle
    inflates with zlib itself."""
    import ctypes as C2
    import gzip
    import io
    import zlib
    from oracle_c import lib as olib
    L = olib()
    L.kto_gzip_inflate.restype = C2.c_int64
    L.kto_gzip_inflate.argtypes = [C2.c_char_p, C2.c_uint64, C2.c_char_p, C2.c_uint64]
    lib = N.load()

    def check(comp, d):
        out1, out2, out3 = (C2.create_string_buffer(len(d) + 1) for _ in range(3))
        assert lib.kta_gzip_inflate_host(comp, len(comp), out1, len(d)) == len(d)          # two stages (tokens)
        assert lib.kta_gzip_inflate_lane_host(comp, len(comp), out3, len(d)) == len(d)     # single pass
        assert L.kto_gzip_inflate(comp, len(comp), out2, len(d)) == len(d)
        assert out1.raw[:len(d)] == d == out2.raw[:len(d)] and out3.raw[:len(d)] == d

    for d in _library_cases():
        for level in (0, 1, 6, 9):
            for strategy in (zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE):
                co = zlib.compressobj(level, zlib.DEFLATED, 15 + 16, 8, strategy)
                check(co.compress(d) + co.flush(), d)
        co = zlib.compressobj(6, zlib.DEFLATED, 15 + 16)      # several deflate blocks: sync flushes in the middle
        third = len(d) // 3
        check(co.compress(d[:third]) + co.flush(zlib.Z_SYNC_FLUSH) + co.compress(d[third:2 * third]) +
              co.flush(zlib.Z_FULL_FLUSH) + co.compress(d[2 * third:]) + co.flush(), d)
        f = io.BytesIO()
        with gzip.GzipFile(filename="some-name.bin", mode="wb", fileobj=f, mtime=12345) as g:
            g.write(d)
        check(f.getvalue(), d)
    # many short matches and few literals: the worst case for the token area of the two-stage form (a token per
    # match; the bound is out/3 + out/255 + 1) — three- and four-byte pieces of a small dictionary in random order
    rng2 = np.random.default_rng(3)
    pieces = [bytes(rng2.integers(0, 256, size=int(k), dtype=np.uint8)) for k in rng2.integers(3, 5, size=40)]
    short = b"".join(pieces[int(i)] for i in rng2.integers(0, len(pieces), size=60000))
    for level in (1, 6, 9):
        co = zlib.compressobj(level, zlib.DEFLATED, 15 + 16)
        check(co.compress(short) + co.flush(), short)
    pa = pytest.importorskip("pyarrow")
    for d in _library_cases():
        check(pa.compress(d, codec="gzip", asbytes=True), d)
    # malformed members are refused: truncation, wrong magic, a second member, a corrupted code, short output
    d = b"".join(b"key-%d value-%d;" % (i % 97, i) for i in range(3000))
    good = gzip.compress(d, mtime=0)
    out = C2.create_string_buffer(len(d) + 64)
    assert lib.kta_gzip_inflate_host(good, len(good), out, len(d)) == len(d)
    assert lib.kta_gzip_inflate_host(good[:-20], len(good) - 20, out, len(d)) == -1
    assert lib.kta_gzip_inflate_host(b"\x1f\x8c" + good[2:], len(good), out, len(d)) == -1
    assert lib.kta_gzip_inflate_host(good + good, 2 * len(good), out, len(d) + 64) == -1
    assert lib.kta_gzip_inflate_host(good, len(good), out, 100) == -1
    rng = np.random.default_rng(5)
    refused = 0
    for _ in range(200):                                       # flipped bits never crash and never write past `cap`
        bad = bytearray(good)
        bad[int(rng.integers(10, len(good) - 8))] ^= 1 << int(rng.integers(0, 8))
        guard, guard2 = C2.create_string_buffer(len(d) + 64), C2.create_string_buffer(len(d) + 64)
        got = lib.kta_gzip_inflate_host(bytes(bad), len(bad), guard, len(d))
        assert got in (-1, len(d)) and guard.raw[len(d):] == bytes(64)
        got2 = lib.kta_gzip_inflate_lane_host(bytes(bad), len(bad), guard2, len(d))
        assert got2 == got and (got < 0 or guard.raw == guard2.raw)      # both forms accept the same streams
        refused += got == -1
    assert refused > 50


def test_zstd_inflate_host_against_libzstd():
    """The product's Zstandard decoder (the function the device runs, compiled for the host) on frames
    written by libzstd (through pyarrow): one-shot frames at levels -5..22 (content size, single segment),
    streaming frames (window descriptor, no content size, flushed mid-way), multi-block inputs, raw and
    RLE blocks; refusals for truncated / corrupted frames."""
    pa = pytest.importorskip("pyarrow")
    import ctypes as C2
    lib = N.load()
    rng = np.random.default_rng(8)
    cases = _library_cases() + [bytes(rng.integers(0, 16, size=400000, dtype=np.uint8)),
                                b"".join(bytes([int(x)]) * int(y) for x, y in zip(rng.integers(0, 256, 3000), rng.integers(1, 300, 3000)))]

    spilled = {"treeless": 0, "repeat_after_huffman": 0}

    def check(comp, d):
        # both table layouts: a Huffman table of its own (host, lane kernel) and the one that lies over the sequence
        # tables and spills what outlives a block (the wave kernel's: kta_zstd.h, ZsWorkSmall)
        for fn in (lib.kta_zstd_inflate_host, lib.kta_zstd_inflate_host_small):
            out = C2.create_string_buffer(len(d) + 1)
            assert fn(comp, len(comp), out, len(d)) == len(d), (len(d), len(comp))
            assert out.raw[:len(d)] == d
        for k, v in _zstd_spill_paths(comp).items():
            spilled[k] += v

    for d in cases:
        for level in (-5, 1, 3, 9, 19, 22):
            check(pa.Codec("zstd", compression_level=level).compress(d, asbytes=True), d)
        sink = pa.BufferOutputStream()
        with pa.CompressedOutputStream(sink, "zstd") as o:
            o.write(d[:len(d) // 2])
            o.flush()
            o.write(d[len(d) // 2:])
        check(sink.getvalue().to_pybytes(), d)
    # the frames above take both ways through the spill: a block without a tree after one with, and a Huffman-coded
    # block whose sequences repeat the tables of the block before
    assert spilled["treeless"] > 20 and spilled["repeat_after_huffman"] > 20, spilled
    d = cases[6]
    good = pa.Codec("zstd", compression_level=3).compress(d, asbytes=True)
    check(good + good, d + d)                                  # concatenated frames
    out = C2.create_string_buffer(len(d) + 64)
    assert lib.kta_zstd_inflate_host(good[:-7], len(good) - 7, out, len(d)) == -1
    assert lib.kta_zstd_inflate_host(b"\x28\xb5\x2f\xfe" + good[4:], len(good), out, len(d)) == -1
    assert lib.kta_zstd_inflate_host(good, len(good), out, len(d) - 1) == -1
    refused = 0
    for _ in range(300):                                       # flipped bits never crash and never write past `cap`
        bad = bytearray(good)
        bad[int(rng.integers(4, len(good)))] ^= 1 << int(rng.integers(0, 8))
        guard = C2.create_string_buffer(len(d) + 64)
        got = lib.kta_zstd_inflate_host(bytes(bad), len(bad), guard, len(d))
        assert -1 <= got <= len(d) and guard.raw[len(d):] == bytes(64)
        guard2 = C2.create_string_buffer(len(d) + 64)
        assert lib.kta_zstd_inflate_host_small(bytes(bad), len(bad), guard2, len(d)) == got      # both layouts: the same verdict
        assert got < 0 or guard2.raw == guard.raw
        refused += got == -1
    assert refused > 100


def _zstd_spill_paths(comp):
    """Walks the block headers of zstd frames (RFC 8878 3.1.1) and counts the blocks that make a decoder whose Huffman
    table shares its place with the sequence tables go through its spill: literals coded with the tree of an earlier
    block, and Huffman-coded literals in a block whose sequences repeat a table of the block before."""
    n = {"treeless": 0, "repeat_after_huffman": 0}
    pos = 0
    while pos + 6 <= len(comp):
        fhd = comp[pos + 4]
        fcs, single, did = fhd >> 6, (fhd >> 5) & 1, fhd & 3
        pos += 5 + (0 if single else 1) + (0, 1, 2, 4)[did] + ((1, 2, 4, 8)[fcs] if (fcs or single) else 0)
        while True:
            h = int.from_bytes(comp[pos:pos + 3], "little")
            pos += 3
            btype, size = (h >> 1) & 3, h >> 3
            if btype == 2:
                b = comp[pos:pos + size]
                lt, sf = b[0] & 3, (b[0] >> 2) & 3
                if lt < 2:
                    regen = b[0] >> 3 if sf in (0, 2) else ((b[0] >> 4) | (b[1] << 4) if sf == 1 else (b[0] >> 4) | (b[1] << 4) | (b[2] << 12))
                    q = (1, 2, 1, 3)[sf] + (regen if lt == 0 else 1)
                else:
                    v = int.from_bytes(b[0:5], "little") >> 4
                    bits, hdr = ((10, 3), (10, 3), (14, 4), (18, 5))[sf]
                    q = hdr + ((v >> bits) & ((1 << bits) - 1))
                    n["treeless"] += lt == 3
                n_seq = b[q]
                q += 1 if n_seq < 128 else (2 if n_seq < 255 else 3)
                if n_seq and lt >= 2:
                    modes = b[q]
                    n["repeat_after_huffman"] += 3 in (modes >> 6, (modes >> 4) & 3, (modes >> 2) & 3)
            pos += 1 if btype == 1 else size
            if h & 1:
                break
        pos += 4 if fhd & 4 else 0
    return n


def _library_cases():
    rng = np.random.default_rng(17)
    text = b"".join(b"user-%05d|%s|balance=%d;" % (i % 513, b"x" * (i % 37), i * 7919 % 100003) for i in range(40000))
    return [b"", b"a", b"abcd" * 1000, bytes(rng.integers(0, 256, size=70000, dtype=np.uint8)), b"\0" * 300000,
            bytes(rng.integers(0, 4, size=200000, dtype=np.uint8)), text,
            text[:100000] + bytes(rng.integers(0, 256, size=90000, dtype=np.uint8)) + text[:100000]]


def test_inflaters_against_the_real_snappy_and_lz4_libraries():
    """Streams written by Google's snappy and by liblz4 (LZ4 frame), through pyarrow: a third-party pin for
    the product's inflaters (same functions the device runs) and for the oracle's."""
    pa = pytest.importorskip("pyarrow")
    import ctypes as C2
    from oracle_c import lib as olib
    L = olib()
    lib = N.load()
    for name, mine, theirs in (("snappy", lib.kta_snappy_inflate_host, L.kto_snappy_inflate),
                               ("lz4", lib.kta_lz4_inflate_host, L.kto_lz4_inflate)):
        theirs.restype = C2.c_int64
        theirs.argtypes = [C2.c_char_p, C2.c_uint64, C2.c_char_p, C2.c_uint64]
        for d in _library_cases():
            comp = pa.compress(d, codec=name, asbytes=True)
            out1, out2 = C2.create_string_buffer(len(d) + 1), C2.create_string_buffer(len(d) + 1)
            assert mine(comp, len(comp), out1, len(d)) == len(d), (name, len(d))
            assert theirs(comp, len(comp), out2, len(d)) == len(d), (name, len(d))
            assert out1.raw[:len(d)] == d == out2.raw[:len(d)]


@pytest.mark.gpu
def test_device_inflates_streams_of_the_real_libraries():
    pytest.importorskip("pyarrow")
    rng = np.random.default_rng(23)
    cases = _library_cases()
    recs = [(i, b"k%d" % i, cases[3 + i % 5][: 3000 + 9000 * i]) for i in range(12)] + [(12, None, cases[6][:150000])]
    blob = b"".join(K.encode_batch(100 * i, recs[i % 3:], 1_600_000_000_000 + i, compression=c)
                    for i, c in enumerate(["snappy-lib", "lz4-lib", None, "lz4-lib", "snappy-lib"]))
    want, _ = kafka_decode(blob, 2)
    lib = N.load()
    for variant in (0, 1):
        with kta.HipMetricHandler(4, now=NOW) as h:
            h._check(N.load().kta_kafka_set_variant(h._ctx, variant))
            cols, st, bad = _decode_on_device(h, blob, 2, True)
            assert bad == 0 and st.n_snappy == 2 and st.n_lz4 == 2
            for k in ("partition", "key_len", "val_len", "ts_ms"):
                assert np.array_equal(cols[k], want[k]), k
            kb = cols["key_bytes"].tobytes()
            o = int(cols["key_off"][0])                       # first record of the first (snappy) batch: key, value follow
            assert kb[o:o + 2] == b"k0" and recs[0][2] in kb[o:o + len(recs[0][2]) + 16]


@pytest.mark.gpu
@pytest.mark.parametrize("variant", [0, 1, 2, 10])
def test_device_decodes_snappy_batches(variant):
    rng = np.random.default_rng(33)
    blob, expected, info = random_record_set(rng, 160, max_records=120, snappy=True)
    assert info["snappy"] > 8 and info["lz4"] > 8 and info["gzip"] > 15 and info["zstd"] > 10
    want, _ = kafka_decode(blob, 3)
    lib = N.load()
    with kta.HipMetricHandler(8, now=NOW) as h:
        h._check(N.load().kta_kafka_set_variant(h._ctx, variant))
        h._check(lib.kta_kafka_set_check_crcs(h._ctx, 1))   # the CRC covers the compressed bytes
        cols, st, bad = _decode_on_device(h, blob, 3, True)
        assert bad == 0 and st.n_snappy == info["snappy"] and st.n_lz4 == info["lz4"] and st.n_gzip == info["gzip"]
        assert st.n_zstd == info["zstd"]
        assert_columns(cols, expected, key_check=True)
        for k in ("partition", "key_len", "val_len", "ts_ms"):
            assert np.array_equal(cols[k], want[k]), k
        # a corrupted compressed stream is reported, never mis-decoded (check.crcs off so that the
        # inflater itself has to notice)
        h._check(lib.kta_kafka_set_check_crcs(h._ctx, 0))
        first = _batches_of(blob)
        rc, descs, _st = index_host(blob, 3)
        victim = next(i for i in range(_st.n_batches) if descs[i].flags & 4)   # a Snappy batch
        p = descs[victim].byte_off
        broken = bytearray(blob)
        broken[p + 61] = 0xFF
        broken[p + 62] = 0xFF
        broken[p + 63] = 0xFF
        broken[p + 64] = 0xFF
        broken[p + 65] = 0x7F                       # preamble claims a 34 GB block
        cols2, st2, bad2 = _decode_on_device(h, bytes(broken), 3, True)
        assert bad2 >= 1 and (cols2["partition"] == -1).sum() == descs[victim].n_records
        # same for an LZ4 batch: a match offset pointing before the start of the output
        victim = next(i for i in range(_st.n_batches) if descs[i].flags & 8)
        p = descs[victim].byte_off + 61
        broken = bytearray(blob)
        first_block = p + 7 + (8 if blob[p + 4] & 8 else 0)
        broken[first_block + 4] = 0x0F                 # token: no literals, match length 15+4
        broken[first_block + 5] = 0x10                 # offset 16 with nothing produced yet
        broken[first_block + 6] = 0x00
        cols3, st3, bad3 = _decode_on_device(h, bytes(broken), 3, True)
        assert bad3 >= 1 and (cols3["partition"] == -1).sum() == descs[victim].n_records


@pytest.mark.gpu
def test_device_inflate_far_matches_and_long_literals():
    """Copies that reach further back than the inflater's 16 KiB LDS history ring (global read-back
    path), literals longer than its 4 KiB input window, and self-overlapping copies (offset < length)."""
    import snappy_py as S
    rng = np.random.default_rng(91)
    chunk = bytes(rng.integers(0, 256, size=20000, dtype=np.uint8))           # incompressible: one long literal
    far = chunk + bytes(rng.integers(0, 256, size=30000, dtype=np.uint8)) + chunk   # second copy: offsets ~50 KB
    recs = [(0, b"far", far), (1, b"rle", b"\x07" * 5000), (2, b"pat", b"abcdefg" * 900), (3, None, chunk[:70] * 40)]
    blob = b"".join(K.encode_batch(10 * i, recs, 1_600_000_000_000 + i, compression=c)
                    for i, c in enumerate(["snappy", "snappy-xerial", "lz4", "lz4-indep", None]))
    # the far copy must really be encoded as a long-offset copy
    comp = S.compress_block(b"".join(K.encode_record(i, r[0], r[1], r[2]) for i, r in enumerate(recs)))
    assert len(comp) < len(far) and S.decompress_block(comp)
    want, _ = kafka_decode(blob, 1)
    lib = N.load()
    for variant in (0, 1):
        with kta.HipMetricHandler(2, now=NOW) as h:
            h._check(N.load().kta_kafka_set_variant(h._ctx, variant))
            cols, st, bad = _decode_on_device(h, blob, 1, True)
            assert bad == 0 and st.n_snappy == 2 and st.n_lz4 == 2
            for k in ("partition", "key_len", "val_len", "ts_ms"):
                assert np.array_equal(cols[k], want[k]), k
            # the inflated bytes themselves: every key readable through key_off, and the value region of the
            # first record equals the original (checks the far copy byte for byte)
            kb = cols["key_bytes"].tobytes()
            for i, key in enumerate([b"far", b"rle", b"pat"]):
                o = int(cols["key_off"][i])
                assert kb[o:o + 3] == key
            o = int(cols["key_off"][0])
            assert far in kb[o:o + len(far) + 64]


@pytest.mark.gpu
def test_device_gzip_and_zstd_copy_paths():
    """The copy stage of the two-stage gzip inflate and of the zstd kernel, byte for byte: matches further back
    than the 16 KiB LDS ring minus a chunk (read back from the written-back output; DEFLATE reaches 32 KiB,
    zstd further), matches and literal runs across 4 KiB chunk borders, self-overlapping copies (distance 1,
    7 and 70 against lengths up to 258 and beyond), stored and fixed-code blocks, batches of several hundred KiB."""
    pytest.importorskip("pyarrow")
    rng = np.random.default_rng(92)
    chunk = bytes(rng.integers(0, 256, size=9000, dtype=np.uint8))
    far = chunk + bytes(rng.integers(0, 256, size=14000, dtype=np.uint8)) + chunk      # second copy: offsets ~23 KB
    wide = bytes(rng.integers(0, 256, size=70000, dtype=np.uint8))
    farther = wide + bytes(rng.integers(0, 256, size=50000, dtype=np.uint8)) + wide    # zstd: offsets ~120 KB
    text = b"".join(b"user-%05d|%s|balance=%d;" % (i % 513, b"x" * (i % 37), i * 7919 % 100003) for i in range(9000))
    # (a period that does not divide the LDS ring, repeated past the ring's length: one match longer than the ring)
    recs = [(0, b"far", far), (1, b"rle", b"\x07" * 5000), (2, b"pat", b"abcdefg" * 9000), (3, b"p70", chunk[:70] * 40),
            (4, b"txt", text), (5, b"fth", farther), (6, b"zer", b"\0" * 300000), (7, None, text[:1000])]
    codecs = ["gzip", "zstd", "gzip-fixed", "zstd-19", "gzip-stored", "zstd-stream", "gzip-named", None, "lz4", "snappy",
              "lz4-indep"]
    blob = b"".join(K.encode_batch(10 * i, recs, 1_600_000_000_000 + i, compression=c) for i, c in enumerate(codecs))
    want, _ = kafka_decode(blob, 1)
    # the zstd batches (the level 19 one) have blocks without a tree of their own and Huffman-coded blocks that repeat sequence tables: the
    # wave kernel's Huffman table lies over its sequence tables, and these are the blocks that go through its spill
    rc, descs, _ = index_host(blob, 1)
    assert rc == N.KTA_OK
    paths = [_zstd_spill_paths(blob[d.byte_off + 61:d.byte_off + d.batch_bytes]) for d in descs[:len(codecs)] if d.flags & 32]
    assert len(paths) == 3 and sum(p["treeless"] for p in paths) > 3 and sum(p["repeat_after_huffman"] for p in paths) > 3, paths
    for variant in (0, 1):
        with kta.HipMetricHandler(2, now=NOW) as h:
            h._check(N.load().kta_kafka_set_variant(h._ctx, variant))
            cols, st, bad = _decode_on_device(h, blob, 1, True)
            assert bad == 0 and st.n_gzip == 4 and st.n_zstd == 3 and st.n_lz4 == 2 and st.n_snappy == 1
            for k in ("partition", "key_len", "val_len", "ts_ms"):
                assert np.array_equal(cols[k], want[k]), k
            kb = cols["key_bytes"].tobytes()
            for b in range(len(codecs)):                      # every batch: every keyed record's key and value bytes
                for i, (_, key, value) in enumerate(recs[:7]):
                    o = int(cols["key_off"][b * len(recs) + i])
                    assert kb[o:o + 3] == key, (variant, codecs[b], key)
                    at = kb.find(value[:64], o + 3, o + 3 + 8 + 64)    # the value follows its varint length
                    assert at > 0 and kb[at:at + len(value)] == value, (variant, codecs[b], key)


@pytest.mark.gpu
@pytest.mark.parametrize("values", ["pattern", "text"])
def test_device_inflates_the_generators_batches_of_every_codec(values):
    """The synthetic encoder's batches (bench.py's source) in every codec, with its two value laws — the 24-byte periodic
    pattern of the bench (raw literals and ~ 100 long matches per batch) and words / numbers / punctuation (Huffman-coded
    literals in zstd, ~ 1 500 short matches per batch: the regime where the inflate kernels copy in rounds) —: the decoded
    columns equal the generator's, and every inflated payload is, byte for byte, the uncompressed batch's."""
    pytest.importorskip("pyarrow")
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    lib = N.load()
    spec, _ = kta.synth_preset("c4")
    nc, rpb = 9000, 60
    flag = 0x200 if values == "text" else 0

    def encode(enc):
        ln = C.c_uint64()
        lib.kta_kafka_encode_synth_host_ex(C.byref(spec), 0, nc, rpb, enc | flag, None, 0, C.byref(ln))
        buf = np.zeros(ln.value + 128, np.uint8)
        lib.kta_kafka_encode_synth_host_ex(C.byref(spec), 0, nc, rpb, enc | flag, buf.ctypes.data, ln.value, C.byref(ln))
        return buf[:ln.value].tobytes()

    plain = encode(0x100)
    payloads = []
    pos = 0
    while pos + 61 <= len(plain):
        total = 12 + int.from_bytes(plain[pos + 8:pos + 12], "big")
        payloads.append(plain[pos + 61:pos + total])
        pos += total
    ref = kta.synth_fill_host(spec, 0, nc)
    for codec in (1, 2, 3, 4):
        blob = encode(codec) if codec in (2, 3) else bench._recompress_batches(lib, plain, codec)
        assert len(blob) < len(plain)
        for variant in (0, 1):
            with kta.HipMetricHandler(2, now=NOW) as h:
                h._check(lib.kta_kafka_set_variant(h._ctx, variant))
                cols, st, bad = _decode_on_device(h, blob, 1, True)
                assert bad == 0 and st.n_records == nc
                for k in ("key_len", "val_len", "ts_ms"):
                    assert np.array_equal(cols[k], ref[k]), (codec, variant, k)
                rc, descs, _ = index_host(blob, 1)
                whole = cols["key_bytes"]
                for b, want in enumerate(payloads):         # a batch's slice of the inflate area begins at payload_off
                    at = descs[b].payload_off
                    assert whole[at:at + len(want)].tobytes() == want, (codec, variant, b)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [314, 2718, 1618])
def test_device_inflate_verdicts_on_mutated_gzip_and_zstd_streams(seed):
    """The wave / lane executions of the gzip and zstd decoders against the host execution of the same text, on
    damaged streams: a batch whose stream the host function refuses must come back flagged (all of its records
    partition -1), undamaged batches must decode as before, and nothing may fault or hang — for both kernel
    families (variant 0: two-stage gzip, wave zstd; variant 1: one lane per batch)."""
    pytest.importorskip("pyarrow")
    import ctypes as C2
    lib = N.load()
    rng = np.random.default_rng(seed)
    text = b"".join(b"user-%05d|%s|balance=%d;" % (i % 513, b"x" * (i % 37), i * 7919 % 100003) for i in range(2500))
    noise = bytes(rng.integers(0, 256, size=6000, dtype=np.uint8))
    recs = [(0, b"k0", text[:20000]), (1, b"k1", noise), (2, b"k2", text[20000:50000] + noise[:900]), (3, None, b"\0" * 9000)]
    codecs = ["gzip", "zstd", "gzip-fixed", "zstd-19", "zstd-stream", "gzip"] * 6
    batches = [K.encode_batch(10 * i, recs, 1_600_000_000_000 + i, compression=c) for i, c in enumerate(codecs)]
    blob = bytearray(b"".join(batches))
    rc, descs, st = index_host(bytes(blob), 1)
    assert rc == N.KTA_OK and st.n_batches == len(codecs)
    damaged = set(range(0, len(codecs), 2)) | {1}
    for b in sorted(damaged):                                   # one to three damaged bytes inside the compressed payload
        lo, hi = descs[b].byte_off + 61 + 12, descs[b].byte_off + descs[b].batch_bytes - 8
        for _ in range(int(rng.integers(1, 4))):
            blob[int(rng.integers(lo, hi))] ^= 1 << int(rng.integers(0, 8))
    blob = bytes(blob)
    rc, descs, st = index_host(blob, 1)                          # (sizes come from headers / trailers the damage spared)
    assert rc == N.KTA_OK and st.n_batches == len(codecs)
    refused = []
    for b in range(len(codecs)):
        d = descs[b]
        comp = blob[d.byte_off + 61:d.byte_off + d.batch_bytes]
        cap = d.payload_end - d.payload_off
        out = C2.create_string_buffer(cap + 1)
        if codecs[b].startswith("gzip"):
            refused.append(d.status != 0 or lib.kta_gzip_inflate_host(comp, len(comp), out, cap) != cap)
        else:
            refused.append(d.status != 0 or lib.kta_zstd_inflate_host(comp, len(comp), out, cap) < 0)
    assert any(refused) and not all(refused[b] for b in damaged)          # both outcomes are in the sample
    assert not any(refused[b] for b in range(len(codecs)) if b not in damaged)
    want, _ = kafka_decode(b"".join(batches), 1)
    for variant in (0, 1):
        with kta.HipMetricHandler(2, now=NOW) as h:
            h._check(lib.kta_kafka_set_variant(h._ctx, variant))
            cols, st2, bad = _decode_on_device(h, blob, 1, False)
            assert bad >= sum(refused)
            for b in range(len(codecs)):
                part = cols["partition"][b * len(recs):(b + 1) * len(recs)]
                if refused[b]:
                    assert (part == -1).all(), (variant, b, codecs[b])
                elif b not in damaged:
                    assert (part == 1).all(), (variant, b, codecs[b])
                    for k in ("key_len", "val_len", "ts_ms"):
                        assert np.array_equal(cols[k][b * len(recs):(b + 1) * len(recs)], want[k][b * len(recs):(b + 1) * len(recs)])


@pytest.mark.gpu
@pytest.mark.parametrize("inflate_limit", [0, 150_000])
def test_consume_snappy_record_sets_end_to_end(inflate_limit):
    """All codecs through the staging pipeline; with a small inflate limit the batches of a blob are
    processed in many groups that reuse the inflate area (keys are hashed in place there)."""
    lib = N.load()
    rng = np.random.default_rng(44)
    P = 3
    o = Oracle(NOW, True)
    with kta.HipMetricHandler(P, count_alive_keys=True, now=NOW) as h:
        h._check(lib.kta_kafka_set_inflate_limit(h._ctx, inflate_limit))
        for fetch in range(5):
            part = fetch % P
            blob, expected, info = random_record_set(rng, 40, partition=part, key_space=60, snappy=True)
            st = N.KtaKafkaIndexStats()
            h._check(lib.kta_kafka_consume(h._ctx, blob, len(blob), part, C.byref(st)))
            cols, _ = kafka_decode(blob, part)
            assert st.n_records == len(cols["partition"]) and st.n_snappy == info["snappy"] and st.n_lz4 == info["lz4"]
            assert st.n_gzip == info.get("gzip", 0) and st.n_zstd == info.get("zstd", 0)
            o.run_soa({k: v for k, v in cols.items() if k != "offset"})
        res, c = h.finish()
        assert np.array_equal(c, o.counters(P))
        assert res.alive_keys == o.alive_keys()
        assert np.array_equal(h.export_alive_bitmap(), o.alive_words())   # keys hashed in place in the inflate area
