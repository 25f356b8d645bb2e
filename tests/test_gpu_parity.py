"""GPU parity tests: the HIP path (through the C ABI of libkta_hip.so) against the CPU oracle on
the same inputs — bit-exact for every counter, extremum and for the alive-key set — plus the
golden scenarios and size-independent properties at BASELINE.json scale."""
import os
import sys

import numpy as np
import pytest

import kafka_topic_analyzer_amd as kta
from kafka_topic_analyzer_amd import _native as N
from helpers import NOW, cols_to_records, load_golden, random_cols, records_to_cols, scenario_records
from oracle_c import Oracle, fnv32

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hc():
    """One -c context in the default state — the reference's bit set, batches applied in submission order —
    shared by the module; reset between tests."""
    h = kta.HipMetricHandler(256, count_alive_keys=True, batch_capacity=1 << 16,
                             key_bytes_capacity=1 << 22, n_staging=3, now=NOW)
    yield h
    h.close()


@pytest.fixture(scope="module")
def ht():
    """One -c context in the table state (KTA_FLAG_ALIVE_TABLE: 32 GiB of sequence-numbered entries, what a rank
    of a sharded run keeps), shared by the module."""
    h = kta.HipMetricHandler(256, count_alive_keys=True, batch_capacity=1 << 16,
                             key_bytes_capacity=1 << 22, n_staging=3, now=NOW, alive_table=True)
    yield h
    h.close()


@pytest.fixture(params=["bitset", "table"])
def hs(request, hc, ht):
    """Both states of the alive set, for what has to hold in either."""
    return hc if request.param == "bitset" else ht


def _compare(h, o, P, check_bitmap=False):
    res, c = h.finish()
    assert np.array_equal(c[:P], o.counters(P)), "per-partition counters differ"
    assert not c[P:].any()
    mm = kta.MessageMetrics(res, c, h.now)
    assert mm.earliest_message() == o.earliest()
    assert mm.latest_message() == o.latest()
    assert mm.smallest_message() == o.get("smallest_message")
    assert mm.largest_message() == o.get("largest_message")
    assert mm.overall_count() == o.get("overall_count")
    assert mm.overall_size() == o.get("overall_size")
    for p in range(P):
        assert np.float32(mm.dirty_ratio(p)) == np.float32(o.get("dirty_ratio", p))
        for name in ("key_size_avg", "value_size_avg", "message_size_avg"):
            want = o.avg(name, p)
            if want is None:
                with pytest.raises(kta.DivideByZeroPanic):
                    getattr(mm, name)(p)
            else:
                assert getattr(mm, name)(p) == want
    if o.lc:
        assert res.alive_keys == o.alive_keys()
        if check_bitmap:
            assert np.array_equal(h.export_alive_bitmap(), o.alive_words()), "alive-key sets differ"


# ------------------------------------------------------------------------------------- FNV
def test_fnv_device_golden_and_random(hc):
    kats = [(bytes.fromhex(v["key_hex"]), v["hash"]) for v in load_golden("fnv32_kats.json")]
    got = hc.fnv32([k for k, _ in kats])
    assert [int(x) for x in got] == [w for _, w in kats]
    assert [int(x) for x in hc.fnv32([k for k, _ in kta.fnv_reference_kats()])] == \
        [w for _, w in kta.fnv_reference_kats()]
    rng = np.random.default_rng(11)
    keys = [rng.integers(0, 256, size=int(n), dtype=np.uint8).tobytes()
            for n in rng.integers(0, 300, size=4000)]  # every alignment / length class
    got = hc.fnv32(keys)
    assert [int(x) for x in got] == [fnv32(k) for k in keys]


# ------------------------------------------------------------------------------------- golden
@pytest.mark.parametrize("name", sorted(load_golden("scenarios.json")["scenarios"].keys()))
def test_golden_scenarios_per_message_entry(hc, name):
    """Feed the reference-test-style scenarios through kta_handle_message (the MetricHandler binding)."""
    g = load_golden("scenarios.json")
    sc, P = g["scenarios"][name], g["n_partitions"]
    hc.reset()
    for p, ts, k, v in scenario_records(sc):
        hc.handle_message(kta.Message(p, ts, k, v))
    res, c = hc.finish()
    e = sc["expect"]
    mm = kta.MessageMetrics(res, c, tuple(g["now"]))
    for p in range(P):
        assert [int(x) for x in c[p]] == e["partitions"][p]["counters"]
        for nm in ("key_size_avg", "value_size_avg", "message_size_avg"):
            if e["partitions"][p][nm] == "panic":
                with pytest.raises(kta.DivideByZeroPanic):
                    getattr(mm, nm)(p)
            else:
                assert getattr(mm, nm)(p) == e["partitions"][p][nm]
    assert list(mm.earliest_message()) == e["earliest"] and list(mm.latest_message()) == e["latest"]
    assert (mm.smallest_message(), mm.largest_message()) == (e["smallest"], e["largest"])
    assert (mm.overall_count(), mm.overall_size()) == (e["overall_count"], e["overall_size"])
    assert res.alive_keys == e["alive_keys"]
    if name in ("mixed_400", "hash_collision_later_wins_alive", "empty_key"):
        bm = hc.export_alive_bitmap()
        slots = np.nonzero(bm)[0]
        got = sorted(int(w) * 32 + b for w in slots for b in range(32) if (int(bm[w]) >> b) & 1)
        assert got == e["alive_slots"]


# ------------------------------------------------------------------------------------- random streams
@pytest.mark.parametrize("state", ["bitset", "table"])
def test_native_replay_through_the_per_message_entry(hc, ht, state):
    """kta_replay_messages = kta_handle_message for every record of host columns (the reference's loop, kafka.rs:107-109), as a
    native loop: counters, extrema, alive count and the alive set equal the oracle's; null and empty keys stay distinct; the
    library's own bookkeeping counts every message and the staging batches it submitted (a staging batch holds 2^16 records)."""
    hc = ht if state == "table" else hc
    rng = np.random.default_rng(31)
    cols = random_cols(rng, 150_001, 8, key_space=4000, tomb=0.35)
    o = Oracle(NOW, True)
    o.run_soa(cols)
    hc.reset()
    hc.replay_messages(cols)
    hc.flush()
    st = hc.handle_message_stats()
    assert st["messages"] == len(cols["partition"]) and st["batches"] == len(cols["partition"]) // (1 << 16)
    _compare(hc, o, 8, check_bitmap=True)


@pytest.mark.parametrize("P,n,runs,variant", [
    (1, 5000, False, 0), (1, 70001, True, 16), (3, 20000, False, 16), (8, 150000, False, 0),
    (8, 150000, True, 16), (64, 200003, False, 16), (256, 300000, False, 0), (256, 300000, True, 16),
])
def test_random_stream_through_staging_ring(hc, P, n, runs, variant):
    rng = np.random.default_rng(P * 1000 + n)
    cols = random_cols(rng, n, P, key_space=max(10, n // 7), runs=runs)
    o = Oracle(NOW, True)
    o.run_soa(cols)
    hc.reset()
    hc.set_tuning(scan_variant=variant)
    hc.submit_columns(**cols)  # 65536-record staging batches: several batches, ring wraps
    _compare(hc, o, P, check_bitmap=(P in (3, 256)))
    hc.set_tuning()


def test_many_partitions_and_no_alive_context():
    P = 1500
    rng = np.random.default_rng(5)
    cols = random_cols(rng, 120000, P, key_space=5000)
    o = Oracle(NOW, False)
    o.run_soa(cols)
    with kta.HipMetricHandler(P, batch_capacity=50000, now=NOW) as h:
        h.submit_columns(cols["partition"], cols["key_len"], cols["val_len"], cols["ts_ms"])
        _compare(h, o, P)


def test_sizes_up_to_i32_max(hc):
    """val_len up to 2^31-1: 64-bit sums (no 32-bit shortcut anywhere)."""
    rng = np.random.default_rng(99)
    cols = random_cols(rng, 100000, 8, key_space=100, big_sizes=True)
    cols["val_len"][:3] = [2**31 - 1, 2**31 - 1, 0]
    cols["key_len"][:3] = [-1, 1, -1]
    o = Oracle(NOW, True)
    o.run_soa(cols)
    for variant in (0, 16):
        hc.reset()
        hc.set_tuning(scan_variant=variant)
        hc.submit_columns(**cols)
        _compare(hc, o, 8)
    hc.set_tuning()
    assert o.get("largest_message") >= 2**31 - 1


@pytest.mark.parametrize("n", [0, 1, 2, 3, 4, 5, 7, 255, 256, 257, 1023, 1024, 1025, 4099])
def test_empty_and_ragged_batch_sizes(hc, n):
    rng = np.random.default_rng(n + 1)
    cols = random_cols(rng, n, 5, key_space=7) if n else records_to_cols([])
    o = Oracle(NOW, True)
    if n:
        o.run_soa(cols)
    hc.reset()
    b, _ = hc.upload_batch(cols, with_keys=True)
    hc.submit_device(b, n, 0)
    _compare(hc, o, 5)
    hc.device_batch_free(b)


def test_timestamp_outside_chronos_range_is_reported_where_the_reference_panics(hc):
    """metric.rs:210 / kafka.rs:104: NaiveDateTime::from_timestamp(ts / 1000, 0) panics outside chrono 0.4.19's
    years [-262144, 262143].  The scan counts such a record like any other; kta_finish says that the reference
    would not have got this far (the oracle stops at the record).  The bounds themselves are fine."""
    lo, hi = N.KTA_CHRONO_MIN_SEC, N.KTA_CHRONO_MAX_SEC
    ok = records_to_cols([(0, 1000, b"a", 1), (1, hi * 1000 + 999, b"b", 2), (2, lo * 1000 - 999, b"c", 3)])
    o = Oracle(NOW, True)
    o.run_soa(ok)
    assert not o.panicked()
    hc.reset()
    hc.submit_columns(**ok)
    _compare(hc, o, 5)
    for ts in ((hi + 1) * 1000, (lo - 1) * 1000, 2**63 - 1, -2**63):
        cols = records_to_cols([(0, 1000, b"a", 1), (1, ts, b"b", 2), (2, 3000, b"c", 3)])
        o = Oracle(NOW, True)
        o.run_soa(cols)
        assert o.panicked() and o.get("overall_count") == 1
        hc.reset()
        hc.submit_columns(**cols)
        with pytest.raises(kta.DateTimeRangePanic) as e:
            hc.finish()
        assert e.value.code == N.KTA_ERR_TIMESTAMP_RANGE and "out-of-range datetime" in str(e.value)
    hc.reset()


def test_bad_partition_is_reported_not_counted(hc):
    cols = records_to_cols([(0, 1000, b"a", 1), (256, 2000, b"b", 2), (-1, 3000, b"c", 3), (255, 4000, b"d", 4)])
    hc.reset()
    hc.submit_columns(**cols)
    with pytest.raises(kta.KtaError) as e:
        hc.finish()
    assert e.value.code == N.KTA_ERR_BAD_PARTITION
    res, c = hc.finish(allow_bad_partition=True)
    assert res.bad_partition_records == 2 and res.overall_count == 2
    assert int(c[0, 0]) == 1 and int(c[255, 0]) == 1 and res.max_ts_sec == 4


def test_reset_and_accumulation_across_batches(hc):
    rng = np.random.default_rng(3)
    a = random_cols(rng, 30000, 16, key_space=300)
    b = random_cols(rng, 41000, 16, key_space=300)
    o = Oracle(NOW, True)
    o.run_soa(a)
    o.run_soa(b)
    hc.reset()
    hc.submit_columns(**a)
    r1, _ = hc.finish()  # finish is non-destructive: more batches may follow
    assert r1.overall_count == 30000
    hc.submit_columns(**b)
    _compare(hc, o, 16, check_bitmap=True)
    hc.reset()
    res, c = hc.finish()
    assert res.any_records == 0 and res.alive_keys == 0 and not c.any()


# ------------------------------------------------------------------------------------- alive-key order
def test_last_writer_wins_is_by_sequence_not_by_submission_order(ht):
    """Table state: two device batches with explicit seq columns, submitted in the WRONG order, must still
    reproduce sequential BitSet semantics (what a partition-sharded multi-GPU merge relies on)."""
    hc = ht
    rng = np.random.default_rng(21)
    cols = random_cols(rng, 60000, 4, key_space=500, tomb=0.5)
    o = Oracle(NOW, True)
    o.run_soa(cols)
    n = len(cols["partition"])
    half = n // 2
    hc.reset()
    parts = []
    for lo, hi in ((half, n), (0, half)):  # second half first
        sub = {k: v[lo:hi] for k, v in cols.items() if k != "key_bytes"}
        base = int(cols["key_off"][lo])
        end = int(cols["key_off"][hi - 1]) + max(int(cols["key_len"][hi - 1]), 0)
        sub["key_bytes"] = cols["key_bytes"][base:end]
        sub["key_off"] = (cols["key_off"][lo:hi] - base).astype(np.uint32)
        sub["seq"] = np.arange(lo, hi, dtype=np.uint64)
        b, m = hc.upload_batch(sub, with_keys=True)
        hc.submit_device(b, m, 0)
        parts.append(b)
    _compare(hc, o, 4, check_bitmap=True)
    for b in parts:
        hc.device_batch_free(b)


def test_partitioned_pass_with_seq_columns_ascending_and_not(ht):
    """Table state, the partitioned pass forced (variant 13) on batches that carry a seq column of GLOBAL sequence
    numbers: the survivors' values come from the column (not base_seq + index).  Three batches submitted in the
    wrong order: two whose column ascends (partition + per-bucket merge) and one whose ROWS are shuffled — its
    column does not ascend, the device's order check hands it to the single-kernel update without a host round
    trip.  Last writer by sequence number: the oracle's set and count, whatever the order."""
    hc = ht
    rng = np.random.default_rng(77)
    cols = random_cols(rng, 300000, 8, key_space=6000, tomb=0.4)
    o = Oracle(NOW, True)
    o.run_soa(cols)
    n = len(cols["partition"])
    seq = 3 * np.arange(n, dtype=np.uint64) + 7            # global, ascending with the topic, not contiguous
    a, b_ = n // 3, 2 * n // 3 + 1
    last = np.arange(b_, n)
    rng.shuffle(last)
    hc.reset()
    hc.set_tuning(alive_variant=13)
    batches = []
    for idx in (last, np.arange(a, b_), np.arange(0, a)):
        sub = {k: cols[k][idx] for k in ("partition", "key_len", "val_len", "ts_ms", "key_off")}
        sub["key_bytes"] = cols["key_bytes"]                # offsets into the whole blob
        sub["seq"] = seq[idx]
        b, m = hc.upload_batch(sub, with_keys=True)
        hc.submit_device(b, m, 0)
        batches.append(b)
    _compare(hc, o, 8, check_bitmap=True)
    hc.set_tuning()
    for b in batches:
        hc.device_batch_free(b)


def test_bit_set_state_refuses_what_needs_sequence_numbers(hc):
    """The default state is the reference's bit set: no table to hand out, export or import."""
    for call in (hc.alive_table, hc.alive_export_entries, lambda: hc.alive_count_range(0, 10)):
        with pytest.raises(kta.KtaError) as e:
            call()
        assert e.value.code == N.KTA_ERR_INVALID and "KTA_FLAG_ALIVE_TABLE" in str(e.value)


def test_collision_across_partitions_and_batches(hs):
    hc = hs
    g = load_golden("scenarios.json")["scenarios"]
    recs = scenario_records(g["hash_collision_later_wins_dead"])
    (p0, t0, ka, _), (p1, t1, kb, _) = recs
    assert ka != kb and fnv32(ka) == fnv32(kb)
    for order, want in (((ka, 1), (kb, None)), 0), (((ka, None), (kb, 3)), 1), (((kb, 3), (ka, None)), 0):
        hc.reset()
        for i, (k, v) in enumerate(order):  # separate batches: flush after each message
            hc.handle_message(kta.Message(i, 1000 + i, k, v))
            hc.flush()
        res, _ = hc.finish()
        assert res.alive_keys == want


def test_running_alive_count_equals_table_scan(hs):
    """The running count == a recount of the state (popcount of the bit set / of the table's alive flags), also
    (table state) after interleaving counting and non-counting update kernels."""
    hc = hs
    rng = np.random.default_rng(77)
    cols = random_cols(rng, 200000, 16, key_space=3000, tomb=0.4)
    o = Oracle(NOW, True)
    o.run_soa(cols)
    hc.reset()
    hc.submit_columns(**cols)
    r_run, _ = hc.finish()               # running count
    hc.alive_table_modified()            # force a recount from the table
    r_scan, _ = hc.finish()
    assert r_run.alive_keys == r_scan.alive_keys == o.alive_keys()
    more = random_cols(rng, 50000, 16, key_space=3000, tomb=0.6)
    o.run_soa(more)
    hc.set_tuning(alive_variant=0)       # non-counting kernel invalidates the running count ...
    hc.submit_columns(**more)
    hc.set_tuning()
    r3, _ = hc.finish()                  # ... so this finish scans
    assert r3.alive_keys == o.alive_keys()


@pytest.mark.parametrize("state,variant", [("table", 0), ("table", 1), ("table", 2), ("table", 13), ("bitset", 3)])
def test_alive_kernels_under_heavy_slot_contention(hc, ht, state, variant):
    """Every alive-update kernel (table state: plain atomicMax; returning atomicMax + running count; the same
    walked backwards with a pre-read that skips superseded records; the partitioned pass forced onto small
    batches — and the bit set state, which always takes the partitioned pass) leaves exactly what sequential
    BitSet semantics leaves — 40 keys over 120k records is heavy same-slot contention: the hot keys' buckets
    overflow their segments into the pool, i.e. the direct path (table) / the fallback kernel (bit set)."""
    hc = ht if state == "table" else hc
    rng = np.random.default_rng(500 + variant)
    o = Oracle(NOW, True)
    hc.reset()
    hc.set_tuning(alive_variant=variant)
    for key_space, n in ((40, 120000), (3000, 200000), (10**6, 150000)):
        cols = random_cols(rng, n, 32, key_space=key_space, tomb=0.45)
        o.run_soa(cols)
        hc.submit_columns(**cols)
    res, c = hc.finish()
    hc.set_tuning()
    assert res.alive_keys == o.alive_keys()
    assert np.array_equal(c[:32], o.counters(32)) and not c[32:].any()
    assert np.array_equal(hc.export_alive_bitmap(), o.alive_words())


@pytest.mark.parametrize("state,variant,wgs", [("table", 3, 0), ("table", 13, 100), ("table", 13, 37),
                                               ("bitset", 3, 0), ("bitset", 3, 100), ("bitset", 3, 37)])
def test_partitioned_alive_pass_across_batches_vs_oracle(hc, ht, state, variant, wgs):
    """The partitioned pass (kta_alive_partition + kta_alive_apply) over several batches that revisit each
    other's keys — tombstones killing earlier batches' keys, re-insertions, empty and null keys, key
    lengths 0..40 at every alignment — against the oracle's BitSet: count, running count and all 2^32 bits.
    Table state: batches below the partitioning threshold take the single-kernel path on the same table, and so
    do (variant 3, the automatic choice) the batches after one whose keys were mostly unique; 13 partitions every
    batch.  Bit set state: every batch is partitioned."""
    hc = ht if state == "table" else hc
    rng = np.random.default_rng(900 + variant + wgs)
    o = Oracle(NOW, True)
    hc.reset()
    hc.set_tuning(alive_workgroups=wgs, alive_variant=variant)
    consumed = 0
    for key_space, n, tomb in ((3_000_000, 2_400_000, 0.3), (50, 2_200_000, 0.5), (3_000_000, 2_300_000, 0.6),
                               (200_000, 300_000, 0.2)):
        cols = random_cols(rng, n, 16, key_space=key_space, tomb=tomb)
        o.run_soa(cols)
        b, nb = hc.upload_batch(cols, with_keys=True)
        hc.submit_device(b, nb, consumed, which=2)   # base_seq = records consumed before this batch
        consumed += nb
        hc.device_batch_free(b)
    res, _ = hc.finish()
    hc.set_tuning()
    assert res.alive_keys == o.alive_keys()
    assert np.array_equal(hc.export_alive_bitmap(), o.alive_words())
    hc.alive_table_modified()
    assert hc.finish()[0].alive_keys == o.alive_keys()       # the running count equals a recount of the table


@pytest.mark.parametrize("state", ["bitset", "table"])
@pytest.mark.parametrize("preset,log2n", [("c3", 25), ("c5", 25)])
def test_partitioned_alive_pass_at_scale_vs_oracle(hc, ht, state, preset, log2n):
    """2^25 records of the config-3 topic (10 M distinct keys: everything merges in LDS) and of the config-5
    key law (100 M distinct: the buckets hold more distinct slots than the LDS table, so the overflow of
    the table into the direct path carries most of the batch), in two submissions, against the
    single-threaded BitSet oracle on the same records: alive count and every bit of the set."""
    hc = ht if state == "table" else hc
    sp, _ = kta.synth_preset(preset)
    n = 1 << log2n
    host = kta.synth_fill_host(sp, 0, n, with_keys=True)
    o = Oracle(NOW, True)
    o.run_soa(host)
    hc.reset()
    hc.set_tuning(alive_variant=13)      # partitioned whatever the previous batch looked like
    b = hc.device_batch_alloc(n, n * 16)
    assert hc.synth_fill_device(sp, 0, n, b) == host["n_key_bytes"]
    half = n // 2 + 12345
    hc.submit_device(b, half, 0, which=2)
    rest = kta.KtaBatch()
    for f, sz in (("key_len", 4), ("val_len", 4), ("key_off", 4)):
        setattr(rest, f, getattr(b, f) + half * sz)
    rest.key_bytes = b.key_bytes
    hc.submit_device(rest, n - half, half, which=2)
    res, _ = hc.finish()
    hc.set_tuning()
    assert res.alive_keys == o.alive_keys()
    assert np.array_equal(hc.export_alive_bitmap(), o.alive_words())
    hc.device_batch_free(b)


@pytest.mark.parametrize("state", ["bitset", "table"])
@pytest.mark.parametrize("P,n,runs", [(1, 70_001, False), (8, 300_000, True), (64, 2_500_000, False), (256, 3_000_003, True),
                                      (257, 200_000, False)])
def test_both_handlers_in_one_pass_vs_oracle(P, n, runs, state):
    """which = 3 with at most 256 partitions: the partition kernel of the alive-key pass also does
    MessageMetrics::handle_message (/root/reference/src/kafka.rs:107-109: every handler sees every message) — counters,
    extrema, averages and panics, alive count and every bit against the oracle; null / empty keys and values, -1
    timestamps, sizes up to 2^31 - 1, batch lengths that are multiples of nothing, two batches.  257 partitions take the
    two passes (the fused pass keeps 256 partitions' sums in LDS).  Table state (what a rank of a sharded run keeps): the
    partitioned pass forced onto these small batches, the second batch with a seq column of global sequence numbers —
    submitted FIRST: the largest sequence number wins whatever the order of the batches — and the library's own counters
    say that the fused pass ran."""
    rng = np.random.default_rng(4000 + P)
    o = Oracle(NOW, True)
    table = state == "table"
    with kta.HipMetricHandler(P, count_alive_keys=True, now=NOW, alive_table=table) as h:
        if table:
            h.set_tuning(alive_variant=13)
        parts, at = [], 0
        for part in range(2):
            cols = random_cols(rng, n // (part + 1), P, key_space=50_000, tomb=0.3, runs=runs, big_sizes=(P == 8))
            o.run_soa(cols)
            m = len(cols["partition"])
            if table and part == 1:
                cols["seq"] = at + 2 * np.arange(m, dtype=np.uint64)       # ascending, with gaps
            parts.append((cols, at))
            at += m
        for cols, base in (reversed(parts) if table else parts):
            b, nb = h.upload_batch(cols, with_keys=True)
            h.submit_device(b, nb, base, which=3)
            h.device_batch_free(b)
        _compare(h, o, P, check_bitmap=True)
        info = h.alive_pass_info()
        assert info["slices"] == 2 and info["fused"] == (2 if P <= 256 else 0)
    o.close()


@pytest.mark.parametrize("state", ["bitset", "table"])
def test_both_handlers_in_one_pass_equals_two_passes_with_bad_partition_ids(state):
    """The fused pass and the two passes (kta_set_fuse(ctx, 0)) leave the same vector, bit for bit, also for what the reference
    has no word for: partition ids outside [0, P) (counted and reported, never accumulated) — while the alive set, which
    ignores the partition (metric.rs:289-304), takes those records' keys either way.  Table state: with a seq column."""
    P, n = 64, 1_200_007
    rng = np.random.default_rng(4100)
    cols = random_cols(rng, n, P, key_space=20_000, tomb=0.4)
    cols["partition"][::1001] = 64
    cols["partition"][5::7777] = -3
    cols["partition"][11::9001] = 2**31 - 1
    table = state == "table"
    if table:
        cols["seq"] = 1000 + 3 * np.arange(len(cols["partition"]), dtype=np.uint64)
    got = []
    for fuse in (True, False):
        h = kta.HipMetricHandler(P, count_alive_keys=True, now=NOW, alive_table=table)
        h.set_fuse(fuse)
        if table:
            h.set_tuning(alive_variant=13)
        b, nb = h.upload_batch(cols, with_keys=True)
        h.submit_device(b, nb, 0, which=3)
        h.device_batch_free(b)
        res, c = h.finish(allow_bad_partition=True)      # (waits for the batch)
        vec = h.result_vector_host().copy()
        assert h.alive_pass_info()["fused"] == (1 if fuse else 0)
        got.append((vec, c.copy(), int(res.bad_partition_records), int(res.alive_keys), h.export_alive_bitmap()))
        h.close()
    bad = int(((cols["partition"] < 0) | (cols["partition"] >= P)).sum())
    assert got[0][2] == got[1][2] == bad > 0
    assert np.array_equal(got[0][0], got[1][0]) and np.array_equal(got[0][1], got[1][1])
    assert got[0][3] == got[1][3] and np.array_equal(got[0][4], got[1][4])


def _fnv32_np(keys16):
    """The reference's FNV variant (src/fnv32.rs:92-101: multiplier = offset basis) over the rows of a [n, 16] byte array."""
    h = np.full(len(keys16), 0x811c9dc5, np.uint64)
    for j in range(keys16.shape[1]):
        h = ((h ^ keys16[:, j]) * np.uint64(0x811c9dc5)) & np.uint64(0xFFFFFFFF)
    return h.astype(np.uint32)


@pytest.mark.parametrize("state", ["bitset", "table"])
@pytest.mark.parametrize("shape,log2n", [("one_key", 26), ("forty_keys", 26), ("one_bucket", 24)])
def test_adversarial_alive_shapes_at_scale(hc, ht, state, shape, log2n):
    """What defeats the sizes of the partitioned pass, at scale, against the single BitSet oracle (count + every bit):
    ONE key over 2^26 records and 40 keys over 2^26 records (a compacted topic with dominant keys: the guard of the
    partition kernel drops what the same wave instruction supersedes, the rest overflows the hot buckets' segments into
    the pool — bit set state: kta_alive_fallback; table state: the direct path), and 4096 keys that all hash into ONE
    bucket (same top 10 hash bits) over 2^24 records: every pair of the batch in one region, nothing for the other 1023
    workgroups of pass 2.  Half of the records are tombstones, so the order of the records decides every bit
    (/root/reference/src/metric.rs:289-304)."""
    h = ht if state == "table" else hc
    n = 1 << log2n
    if shape == "one_bucket":
        rng = np.random.default_rng(77)
        cand = rng.integers(0, 256, size=(6_000_000, 16), dtype=np.uint8)
        hh = _fnv32_np(cand)
        pick = cand[(hh >> 22) == 0x2A5][:4096]
        assert len(pick) == 4096 and fnv32(pick[0].tobytes()) >> 22 == 0x2A5
        kid = rng.integers(0, 4096, size=n)
        cols = {"partition": (kid % 64).astype(np.int32), "key_len": np.full(n, 16, np.int32),
                "val_len": np.where(rng.random(n) < 0.5, -1, 100).astype(np.int32),
                "ts_ms": np.full(n, 1_600_000_000_000, np.int64), "key_off": (np.arange(n, dtype=np.uint64) * 16).astype(np.uint32),
                "key_bytes": pick[kid].reshape(-1), "n_key_bytes": 16 * n}
    else:
        sp, _ = kta.synth_preset("c3")
        sp.n_distinct_keys = 1 if shape == "one_key" else 40
        sp.tombstone_permille = 500
        cols = kta.synth_fill_host(sp, 0, n, with_keys=True)
    o = Oracle(NOW, True)
    o.run_soa(cols)
    h.reset()
    h.set_tuning(alive_variant=13)       # the partitioned pass in either state
    b, nb = h.upload_batch(cols, with_keys=True)
    h.submit_device(b, nb, 0, which=2)
    h.submit_device(b, nb, nb, which=2)  # and once more: the same records on top of what they left
    o.run_soa(cols)
    res, _ = h.finish()
    h.set_tuning()
    h.device_batch_free(b)
    assert res.alive_keys == o.alive_keys()
    assert np.array_equal(h.export_alive_bitmap(), o.alive_words())


def _keys_with_hash_prefix(rng, want_prefix, shift, count):
    """`count` distinct random 16-byte keys whose reference hash (src/fnv32.rs:92-101) has `want_prefix` above bit `shift`:
    random candidates, hashed by the C oracle's fnv1a on a few threads (numpy's 16 passes over 64-bit arrays are too slow for the
    hundreds of millions of candidates an 18-bit prefix asks for)."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle_c import lib
    L, T, m = lib(), (16 if (os.cpu_count() or 8) >= 32 else 8), 1 << 24
    pool = ThreadPoolExecutor(T)
    kl = np.full(m, 16, np.int32)
    ko = (np.arange(m, dtype=np.uint64) * 16).astype(np.uint32)
    out = np.empty(m, np.uint32)
    found = np.zeros((0, 16), np.uint8)
    while len(found) < count:
        cand = rng.integers(0, 256, size=(m, 16), dtype=np.uint8)
        list(pool.map(lambda j: L.kto_fnv1a_soa(j * (m // T), m // T, kl.ctypes.data, ko.ctypes.data, cand.ctypes.data, out.ctypes.data),
                      range(T)))
        found = np.concatenate([found, cand[(out >> shift) == want_prefix]])
    pool.shutdown()
    keys = found[:count]
    assert fnv32(keys[0].tobytes()) >> shift == want_prefix and len(np.unique(keys, axis=0)) == count
    return keys


def _special_keys_batch(rng, n, special, where, pool_keys=1 << 20):
    keys = rng.integers(0, 256, size=(pool_keys, 16), dtype=np.uint8)[rng.integers(0, pool_keys, size=n)]   # the ordinary keys
    keys[where] = special[np.arange(len(where)) % len(special)]
    return {"partition": rng.integers(0, 64, size=n).astype(np.int32), "key_len": np.full(n, 16, np.int32),
            "val_len": np.where(rng.random(n) < 0.4, -1, 100).astype(np.int32),
            "ts_ms": np.full(n, 1_600_000_000_000, np.int64), "key_off": (np.arange(n, dtype=np.uint64) * 16).astype(np.uint32),
            "key_bytes": keys.reshape(-1), "n_key_bytes": 16 * n}


def test_bucket_given_up_in_careful_mode_is_resolved_exactly(hc):
    """A bucket that pass 2 does not attempt and kta_alive_fallback resolves: 2 000 keys whose hashes fall into one bucket, every
    one of them twice within the first sixteenth of the batch — sixteen segments take 250 of them each on top of their 64
    ordinary pairs, more than a segment's 256, so the bucket's pairs overflow into the pool (the name is round 5's, when the
    case was thought to reach careful mode; the bit set state has none any more).  Count and every bit against the oracle;
    twice, the second batch on top of what the first left (/root/reference/src/metric.rs:289-304: the order decides); the
    library's counter says that exactly one bucket went to the fallback kernel in either batch."""
    n = 1 << 24
    rng = np.random.default_rng(4242)
    special = _keys_with_hash_prefix(rng, (0x155 << 4) | 7, 18, 2000)            # hash >> 18: bucket 0x155, its slots' seventh sixteenth
    cols = _special_keys_batch(rng, n, special, rng.choice(n // 16 - 1000, size=4000, replace=False))
    o = Oracle(NOW, True)
    hc.reset()
    b, nb = hc.upload_batch(cols, with_keys=True)
    for k in range(2):
        o.run_soa(cols)
        hc.submit_device(b, nb, k * nb, which=2)
        hc.sync()
    info = hc.alive_pass_info()
    res, _ = hc.finish()
    hc.device_batch_free(b)
    assert info["failed_buckets"] == 2, info             # one bucket, in either batch
    assert res.alive_keys == o.alive_keys()
    assert np.array_equal(hc.export_alive_bitmap(), o.alive_words())


def test_bucket_that_sixteen_slot_ranges_do_not_hold_goes_to_the_fallback_kernel(hc):
    """What defeats the slot-range passes of pass 2 (kta_alive_apply<.., RANGES>): 1 500 distinct keys whose slots lie in ONE
    window of 2^14 consecutive slots, spread evenly over the batch (six per segment: nothing overflows into the pool).  At
    every number of parts, 1 to 16, the window lies inside one slice of 128 sets, which holds 128 x 8 entries + the 128 of
    its side table: the plain attempt fails, the passes with 2, 4, 8 and 16 parts fail, and the bucket is handed to
    kta_alive_fallback from its first segment on — after parts of its slot range may already have been applied, which the
    fallback kernel must find as they should be.  Count, running count and every bit against the oracle, over two batches."""
    n = 1 << 24
    rng = np.random.default_rng(777)
    special = _keys_with_hash_prefix(rng, (0x2A3 << 8) | 0x5C, 14, 1500)        # hash >> 14: bucket 0x2A3, window 0x5C of its 256
    cols = _special_keys_batch(rng, n, special, rng.choice(n, size=3000, replace=False))
    o = Oracle(NOW, True)
    hc.reset()
    b, nb = hc.upload_batch(cols, with_keys=True)
    for k in range(2):
        o.run_soa(cols)
        hc.submit_device(b, nb, k * nb, which=2)
        hc.sync()
    info = hc.alive_pass_info()
    res, _ = hc.finish()
    hc.device_batch_free(b)
    assert info["failed_buckets"] == 2, info             # that one bucket, in either batch
    assert res.alive_keys == o.alive_keys()
    words = hc.export_alive_bitmap()
    assert np.array_equal(words, o.alive_words())
    assert int(np.bitwise_count(words).sum(dtype=np.uint64)) == res.alive_keys


def test_max_partitions_uses_large_dynamic_lds():
    P = 4096  # 96 KiB of LDS partials per workgroup
    rng = np.random.default_rng(41)
    cols = random_cols(rng, 150000, P, key_space=500)
    o = Oracle(NOW)
    o.run_soa(cols)
    with kta.HipMetricHandler(P, now=NOW) as h:
        h.submit_columns(cols["partition"], cols["key_len"], cols["val_len"], cols["ts_ms"])
        _compare(h, o, P)
    with pytest.raises(kta.KtaError):
        kta.HipMetricHandler(4097)


# ------------------------------------------------------------------------------------- generator
@pytest.mark.parametrize("preset", ["c1", "c2", "c3", "c4"])
def test_device_generator_matches_host_generator(hc, preset):
    sp, _ = kta.synth_preset(preset)
    n, first = 50000, 123456789
    host = kta.synth_fill_host(sp, first, n, with_keys=True, with_seq=True)
    b = hc.device_batch_alloc(n, max(host["n_key_bytes"], 16), with_seq=True)
    kb = hc.synth_fill_device(sp, first, n, b)
    assert kb == host["n_key_bytes"]
    dev = hc.download_batch(b, n, kb)
    for k in ("partition", "key_len", "val_len", "ts_ms", "key_off", "key_bytes", "seq"):
        assert np.array_equal(dev[k], host[k]), k
    hc.device_batch_free(b)


# ------------------------------------------------------------------------------------- BASELINE-scale
def _vector(h):
    h.finish_device()
    return h.result_vector_host()


def test_full_size_properties_metrics_scan():
    """2^27 records of the 256-partition mixed topic (BASELINE config 4's per-GPU shard size):
    oracle on a window, then properties that do not need the oracle at full size."""
    sp, _ = kta.synth_preset("c4")
    n = 1 << 27
    P = 256
    with kta.HipMetricHandler(P, now=NOW) as h:
        b = h.device_batch_alloc(n)
        h.synth_fill_device(sp, 0, n, b)
        # (a) oracle on a 2^21 window in the middle
        w0, wn = (n // 2) + 1, 1 << 21
        host = kta.synth_fill_host(sp, w0, wn)
        o = Oracle(NOW)
        o.run_soa(host)
        win = kta.KtaBatch()
        for f, sz in (("partition", 4), ("key_len", 4), ("val_len", 4), ("ts_ms", 8)):
            setattr(win, f, getattr(b, f) + ((w0 + 3) // 4 * 4) * sz)  # keep 16 B alignment
        skip = (w0 + 3) // 4 * 4 - w0
        o2 = Oracle(NOW)
        o2.run_soa({k: v[skip:] for k, v in host.items() if isinstance(v, np.ndarray)})
        h.submit_device(win, wn - skip, 0, which=1)
        _compare(h, o2, P)
        # (b) whole batch: identities + invariance to launch geometry and kernel variant
        vecs = []
        for variant, wgs in ((0, 0), (16, 0), (16, 300), (0, 2048), (16, 77)):
            h.reset()
            h.set_tuning(scan_workgroups=wgs, scan_variant=variant)
            h.submit_device(b, n, 0, which=1)
            vecs.append(_vector(h))
        for v in vecs[1:]:
            assert np.array_equal(v, vecs[0])
        v = vecs[0]
        c = v[:P * 7].reshape(P, 7)
        assert int(c[:, N.KTA_C_TOTAL].sum()) == n == int(v[P * 7 + N.KTA_G_RECORDS])
        assert np.array_equal(c[:, N.KTA_C_TOMBSTONES] + c[:, N.KTA_C_ALIVE], c[:, N.KTA_C_TOTAL])
        assert np.array_equal(c[:, N.KTA_C_KEY_NULL] + c[:, N.KTA_C_KEY_NON_NULL], c[:, N.KTA_C_TOTAL])
        assert (c[:, N.KTA_C_ALIVE] > 0).all()
        # (c) linearity: two halves accumulated into fresh state and merged == the whole
        lib = N.load()
        halves = []
        for lo in (0, n // 2):
            h.reset()
            hb = kta.KtaBatch()
            for f, sz in (("partition", 4), ("key_len", 4), ("val_len", 4), ("ts_ms", 8)):
                setattr(hb, f, getattr(b, f) + lo * sz)
            h.submit_device(hb, n // 2, 0, which=1)
            halves.append(_vector(h))
        assert lib.kta_merge_vectors(halves[0].ctypes.data, halves[1].ctypes.data, P) == N.KTA_OK
        assert np.array_equal(halves[0], v)
        h.device_batch_free(b)


@pytest.mark.parametrize("state", ["bitset", "table"])
def test_full_size_properties_alive_pass(hc, ht, state):
    """2^26 records, 16 B keys, 10M distinct (BASELINE config 3 shape): oracle on a prefix, then
    idempotence (replaying the same records with the same seq changes nothing) and the
    A-then-all-tombstones round trip (every key dead => 0 alive)."""
    hc = ht if state == "table" else hc
    sp, _ = kta.synth_preset("c3")
    n = 1 << 26
    hc.reset()
    b = hc.device_batch_alloc(n, n * 16)
    kb = hc.synth_fill_device(sp, 0, n, b)
    assert kb == n * 16
    m = 1 << 21
    host = kta.synth_fill_host(sp, 0, m, with_keys=True)
    o = Oracle(NOW, True)
    o.run_soa(host)
    hc.submit_device(b, m, 0, which=2)
    res, _ = hc.finish()
    assert res.alive_keys == o.alive_keys()
    assert np.array_equal(hc.export_alive_bitmap(), o.alive_words())
    hc.reset()
    hc.submit_device(b, n, 0, which=2)
    r1, _ = hc.finish()
    hc.submit_device(b, n, 0, which=2)  # idempotent under replay with identical sequence numbers
    r2, _ = hc.finish()
    assert r1.alive_keys == r2.alive_keys and 0 < r1.alive_keys <= 10_000_000
    words1 = hc.export_alive_bitmap()
    hc.reset()
    hc.set_tuning(alive_variant=2)      # the filtered kernel (backwards walk + pre-read): same set, same count
    hc.submit_device(b, n, 0, which=2)
    rf, _ = hc.finish()
    hc.set_tuning()
    assert rf.alive_keys == r1.alive_keys and np.array_equal(hc.export_alive_bitmap(), words1)
    hc.alive_table_modified()
    assert hc.finish()[0].alive_keys == r1.alive_keys       # its running count equals a recount
    # kill everything: same keys, later sequence numbers, all tombstones
    sp2, _ = kta.synth_preset("c3")
    sp2.tombstone_permille = 1000
    hc.synth_fill_device(sp2, 0, n, b)
    hc.submit_device(b, n, n, which=2)
    r3, _ = hc.finish()
    assert r3.alive_keys == 0
    hc.device_batch_free(b)


# ------------------------------------------------------------------------------------- additive analytics
@pytest.mark.parametrize("P,n,runs", [(1, 30000, False), (8, 120000, True), (256, 250000, False), (1000, 90000, False)])
def test_analytics_histograms_and_partition_extrema(P, n, runs):
    """KTA_FLAG_ANALYTICS (no reference counterpart): the kernel's histograms / per-partition extrema
    against the oracle's restatement of the same definition — and the reference counters unchanged."""
    from oracle_c import analytics
    rng = np.random.default_rng(P + n)
    cols = random_cols(rng, n, P, key_space=500, runs=runs, big_sizes=True)
    if P >= 8:
        cols["val_len"][cols["partition"] == 5] = -1  # a tombstone-only partition
        keep = cols["partition"] != 6                  # and an empty one
        cols = {k: (v[keep] if k != "key_bytes" else v) for k, v in cols.items()}
    want = analytics(cols, P)
    o = Oracle(NOW)
    o.run_soa(cols)
    with kta.HipMetricHandler(P, now=NOW, analytics=True, batch_capacity=1 << 16) as h:
        h.submit_columns(cols["partition"], cols["key_len"], cols["val_len"], cols["ts_ms"])
        res, c = h.finish()
        assert np.array_equal(c, o.counters(P))
        got = h.analytics()
        for k in want:
            assert np.array_equal(got[k], want[k]), k
        assert int(got["key_size_hist"].sum()) == len(cols["partition"]) == res.overall_count
        h.reset()
        z = h.analytics()
        assert not z["key_size_hist"].any() and (z["part_largest"] == 0).all()
    with kta.HipMetricHandler(4) as plain:
        with pytest.raises(kta.KtaError):
            plain.analytics()


# ------------------------------------------------------------------------------------- compact table exchange
def test_alive_export_import_merges_partition_shards(ht):
    """Table state.  Two partition shards with GLOBAL sequence numbers in two contexts; exporting one shard's written
    entries (compact) and importing them into the other reproduces the unsharded alive set exactly."""
    rng = np.random.default_rng(31)
    P = 6
    cols = random_cols(rng, 150000, P, key_space=4000, tomb=0.4)
    o = Oracle(NOW, True)
    o.run_soa(cols)
    n = len(cols["partition"])
    seq = np.arange(n, dtype=np.uint64)

    def shard(mask):
        idx = np.nonzero(mask)[0]
        kl = np.maximum(cols["key_len"][idx], 0).astype(np.int64)
        off = np.zeros(len(idx), np.int64)
        off[1:] = np.cumsum(kl)[:-1]
        kb = np.zeros(max(int(kl.sum()), 1), np.uint8)
        src = cols["key_off"][idx].astype(np.int64)
        for j in np.nonzero(kl)[0]:
            kb[off[j]:off[j] + kl[j]] = cols["key_bytes"][src[j]:src[j] + kl[j]]
        return {"partition": cols["partition"][idx], "key_len": cols["key_len"][idx], "val_len": cols["val_len"][idx],
                "ts_ms": cols["ts_ms"][idx], "key_off": off.astype(np.uint32), "key_bytes": kb[:int(kl.sum())],
                "seq": seq[idx]}

    a_cols, b_cols = shard(cols["partition"] % 2 == 0), shard(cols["partition"] % 2 == 1)
    hc = ht
    hc.reset()
    ba, na = hc.upload_batch(a_cols, with_keys=True)
    hc.submit_device(ba, na, 0)
    with kta.HipMetricHandler(P, count_alive_keys=True, now=NOW, alive_table=True) as hb:
        bb, nb = hb.upload_batch(b_cols, with_keys=True)
        hb.submit_device(bb, nb, 0)
        rb, _ = hb.finish()
        ps, pv, ne = hb.alive_export_entries()              # shard B's entries, on the device
        slots, vals = hb.alive_export_entries_host()
        assert ne == len(slots) == len(np.unique(slots)) and (vals != 0).all()
        assert int((vals & np.uint64(1)).sum()) == rb.alive_keys
        hc.alive_import_entries(ps, pv, ne)                  # merge into shard A's context
        hc.sync()
        hb.device_batch_free(bb)
    res, _ = hc.finish()                                     # running count stayed exact through the import
    assert res.alive_keys == o.alive_keys()
    assert np.array_equal(hc.export_alive_bitmap(), o.alive_words())
    hc.alive_table_modified()
    res2, _ = hc.finish()                                    # and equals a recount of the merged table
    assert res2.alive_keys == o.alive_keys()
    # per-hash-range counts (what the owners of a hash-range exchange report) tile the table, any bounds
    from kafka_topic_analyzer_amd import distributed as D
    for world in (1, 3, 8):
        assert sum(hc.alive_count_range(*D.hash_range(r, world)) for r in range(world)) == o.alive_keys()
    words = o.alive_words()
    lo, hi = 0x40000001, 0x40000001 + 64 * 1000 + 7
    bits = np.unpackbits(words[lo // 32:(hi + 31) // 32 + 1].view(np.uint8), bitorder="little")
    assert hc.alive_count_range(lo, hi) == int(bits[lo % 32:lo % 32 + (hi - lo)].sum())
    assert hc.alive_count_range(5, 5) == 0
    hc.device_batch_free(ba)
    # an empty table exports nothing
    hc.reset()
    assert hc.alive_export_entries()[2] == 0
