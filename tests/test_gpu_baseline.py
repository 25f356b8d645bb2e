"""Parity at the shapes BASELINE.json names (configs[0..4]), against the C oracle on the SAME records:

    c1   1 partition, 1 M records, 64-byte keys, -c                    counters + alive count + all 2^32 bits, both
                                                                       states of the alive set
    c2   8 partitions, 100 M records, mixed key / value sizes          counters + extrema, bit-exact
    c3   64 partitions, 2^30 records, --count-alive-keys, 10 M keys    alive count + all 2^32 bits of the set
    c4   256 partitions, 2^30 records (one GPU's worth of config 4)    counters + extrema, bit-exact
    c5   256 partitions, 2^26 records of config 5's key law (100 M     two partition-sharded ranks with seq columns
         distinct keys, 50 % tombstones), -c                           + kta_exchange (RCCL test double): the
                                                                       unsharded oracle's counters, count and bits

The oracle is single-threaded like the reference (src/kafka.rs:92-135).  For the counter configs it runs as
T independent instances over consecutive chunks of the topic (its state is sums and extrema, so the
instances merge by + / min / max — done here in numpy, nothing of the product involved); the BitSet of c3 is
order dependent PER SLOT: it runs as K instances over disjoint ranges of the 2^32 slots, every instance fed the whole
topic chunk by chunk in consumption order and taking the records whose key hashes into its range (oracle_c.
SlotRangeAliveOracle; tests/test_oracle.py holds it against the single instance) — one instance took 207 s of the GPU
step for 2^30 records.  Records come from the
counter-based generator (include/kta_synth.h), which the device and the host evaluate identically
(test_device_generator_matches_host_generator)."""
import os
import threading

import numpy as np
import pytest

import kafka_topic_analyzer_amd as kta
from kafka_topic_analyzer_amd import _native as N
from helpers import NOW
from oracle_c import Oracle, SlotRangeAliveOracle

pytestmark = pytest.mark.gpu

CHUNK = 1 << 22


def _oracle_counters_threaded(spec, n, P):
    """-> (counters[P,7], earliest (s, ns), latest, smallest, largest) of the oracle over records [0, n)."""
    threads = max(1, min(64, (os.cpu_count() or 8) - 2))
    chunks = [(lo, min(CHUNK, n - lo)) for lo in range(0, n, CHUNK)]
    oracles = [Oracle(NOW) for _ in range(threads)]
    nxt = iter(range(len(chunks)))
    lock = threading.Lock()
    errors = []

    def work(t):
        try:
            while True:
                with lock:
                    k = next(nxt, None)
                if k is None:
                    return
                lo, m = chunks[k]
                oracles[t].run_soa(kta.synth_fill_host(spec, lo, m))     # ctypes calls release the GIL
        except BaseException as e:   # noqa: BLE001
            errors.append(repr(e))

    ts = [threading.Thread(target=work, args=(t,)) for t in range(threads)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errors, errors
    counters = sum(o.counters(P) for o in oracles)
    earliest = min(o.earliest() for o in oracles)
    latest = max(o.latest() for o in oracles)
    largest = max(o.get("largest_message") for o in oracles)
    live = [o for o in oracles if int(o.counters(P)[:, N.KTA_C_ALIVE].sum())]      # smallest_message() maps "none" to 0
    smallest = min(o.get("smallest_message") for o in live) if live else 0
    for o in oracles:
        o.close()
    return counters, earliest, latest, smallest, largest


def _check_counters(preset, n, P):
    sp, _ = kta.synth_preset(preset)
    assert sp.n_partitions == P
    want_c, earliest, latest, smallest, largest = _oracle_counters_threaded(sp, n, P)
    with kta.HipMetricHandler(P, now=NOW) as h:
        b = h.device_batch_alloc(n)
        h.synth_fill_device(sp, 0, n, b)
        h.submit_device(b, n, 0, which=1)
        res, c = h.finish()
        h.device_batch_free(b)
        m = h.metrics()
    assert np.array_equal(c, want_c)
    assert res.overall_count == n == int(want_c[:, N.KTA_C_TOTAL].sum())
    assert res.overall_size == int(want_c[:, N.KTA_C_KEY_SIZE_SUM].sum() + want_c[:, N.KTA_C_VALUE_SIZE_SUM].sum())
    assert (m.earliest_message(), m.latest_message()) == (earliest, latest)
    assert (m.smallest_message(), m.largest_message()) == (smallest, largest)


def test_baseline_config_2_eight_partitions_100m_records():
    _check_counters("c2", 100_000_000, 8)


def test_baseline_config_4_256_partitions_2e30_records():
    _check_counters("c4", 1 << 30, 256)


@pytest.mark.parametrize("state,variant", [("bitset", 3), ("table", 3), ("table", 13), ("table", 2)])
def test_baseline_config_1_one_partition_1m_records_64_byte_keys(state, variant):
    """BASELINE.json configs[0] through the GPU path with -c: 64-byte keys take the general-length hash path of the
    partition kernel (fnv32_more behind the 16 prefetched bytes).  Bit set state, and the table state with the
    automatic choice (3: a batch of 10^6 records takes the single-kernel update), the partitioned pass forced
    (13) and the filtered kernel (2)."""
    sp, n = kta.synth_preset("c1")
    assert n == 1_000_000 and sp.n_partitions == 1
    host = kta.synth_fill_host(sp, 0, n, with_keys=True)
    assert int(host["key_len"].min()) == int(host["key_len"].max()) == 64
    o = Oracle(NOW, True)
    o.run_soa(host)
    with kta.HipMetricHandler(1, count_alive_keys=True, now=NOW, alive_table=(state == "table")) as h:
        h.set_tuning(alive_variant=variant)
        b = h.device_batch_alloc(n, n * 64)
        assert h.synth_fill_device(sp, 0, n, b) == n * 64
        cut = 600_003                                 # two submissions, the second at an odd record offset
        h.submit_device(b, cut, 0, which=2)
        rest = kta.KtaBatch()
        for f, sz in (("partition", 4), ("key_len", 4), ("val_len", 4), ("key_off", 4), ("ts_ms", 8)):
            setattr(rest, f, getattr(b, f) + cut * sz)
        rest.key_bytes = b.key_bytes
        with pytest.raises(kta.KtaError):             # the metric columns must be 16-byte aligned ...
            h.submit_device(rest, n - cut, cut)
        h.submit_device(rest, n - cut, cut, which=2)  # ... the alive pass takes any common alignment
        res, c = h.finish()
        assert res.alive_keys == o.alive_keys() and res.alive_keys > 0
        assert np.array_equal(h.export_alive_bitmap(), o.alive_words())
        h.device_batch_free(b)
    # counters of the config, through a fresh context
    with kta.HipMetricHandler(1, now=NOW) as h:
        h.submit_columns(host["partition"], host["key_len"], host["val_len"], host["ts_ms"])
        res, c = h.finish()
        assert np.array_equal(c, o.counters(1)) and res.overall_count == n
    o.close()


def _alive_oracle_threaded(sp, n):
    """-> SlotRangeAliveOracle over the records [0, n) of the topic, in consumption order (generator threads a few chunks ahead)."""
    from concurrent.futures import ThreadPoolExecutor
    o = SlotRangeAliveOracle(16 if (os.cpu_count() or 8) >= 24 else 8)
    chunks = [(lo, min(CHUNK * 4, n - lo)) for lo in range(0, n, CHUNK * 4)]
    pool = ThreadPoolExecutor(6)
    ahead = [pool.submit(kta.synth_fill_host, sp, lo, m, True) for lo, m in chunks[:8]]
    for k in range(len(chunks)):
        cols = ahead.pop(0).result()
        if k + 8 < len(chunks):
            ahead.append(pool.submit(kta.synth_fill_host, sp, chunks[k + 8][0], chunks[k + 8][1], True))
        o.run_soa(cols)
    pool.shutdown()
    return o


def test_baseline_config_3_alive_keys_2e30_records():
    sp, _ = kta.synth_preset("c3")
    n, P = 1 << 30, 64
    with kta.HipMetricHandler(P, count_alive_keys=True, now=NOW) as h:
        slice_n = 1 << 27                      # one device batch at a time: 2^27 records, 2 GiB of keys (key_off is u32)
        b = h.device_batch_alloc(slice_n, slice_n * 16)
        for lo in range(0, n, slice_n):
            assert h.synth_fill_device(sp, lo, slice_n, b) == slice_n * 16
            h.submit_device(b, slice_n, lo)    # both handlers, as the reference runs them
        # (the GPU works while the oracles do)
        want_c, earliest, latest, smallest, largest = _oracle_counters_threaded(sp, n, P)
        o = _alive_oracle_threaded(sp, n)
        res, c = h.finish()
        assert res.alive_keys == o.alive_keys() and 0 < res.alive_keys <= 10_000_000
        assert np.array_equal(c, want_c) and res.overall_count == n
        # (both handlers of a slice ran as ONE pass over it — the fused partition kernel: the extrema come from there too)
        mm = kta.MessageMetrics(res, c, NOW)
        assert (mm.earliest_message(), mm.latest_message()) == (earliest, latest)
        assert (mm.smallest_message(), mm.largest_message()) == (smallest, largest)
        assert mm.overall_size() == int(want_c[:, N.KTA_C_KEY_SIZE_SUM].sum() + want_c[:, N.KTA_C_VALUE_SIZE_SUM].sum())
        assert np.array_equal(h.export_alive_bitmap(), o.alive_words())
        h.device_batch_free(b)
    o.close()



def test_compacted_topic_of_20m_keys_takes_slot_range_passes_vs_oracle():
    """More distinct keys per batch than pass 2's tables hold in one piece: 20 M keys over 2^27 records are 19.5 k slots per
    bucket against the table's 16 k + 2 k.  The bucket's plain attempt fails, it is listed for kta_alive_apply<.., RANGES>
    and applied in two slot-range passes (rounds 4-5: an instalment after every group of segments, 7.6 instead of 2.6 ms
    at 2^28 records); nothing reaches the fallback kernel.  Two batches — the second revisits the first's keys — against
    the oracle: alive count, every bit, the fused pass's counters."""
    sp, _ = kta.synth_preset("c3")
    sp.n_distinct_keys = 20_000_000
    P, nb, batches = 64, 1 << 27, 2
    with kta.HipMetricHandler(P, count_alive_keys=True, now=NOW) as h:
        b = h.device_batch_alloc(nb, nb * 16)
        for k in range(batches):
            assert h.synth_fill_device(sp, k * nb, nb, b) == nb * 16
            h.submit_device(b, nb, k * nb, which=3)
            h.sync()
        want_c, earliest, latest, smallest, largest = _oracle_counters_threaded(sp, nb * batches, P)
        o = _alive_oracle_threaded(sp, nb * batches)
        res, c = h.finish()
        info = h.alive_pass_info()
        assert info["slices"] == batches == info["fused"] and info["failed_buckets"] == 0, info
        assert res.alive_keys == o.alive_keys() and 10_000_000 < res.alive_keys <= 20_000_000
        assert np.array_equal(c, want_c) and res.overall_count == nb * batches
        assert np.array_equal(h.export_alive_bitmap(), o.alive_words())
        h.device_batch_free(b)
    o.close()


_C5_ONE_GPU = {}


def _c5_one_gpu_oracle(n):
    """The oracle (both handlers) over the first n records of config 5's topic, in consumption order; kept for the two
    parametrisations of the test below.  MessageMetrics as independent instances over chunks (sums and extrema merge),
    the BitSet as instances over slot ranges (order matters per slot only)."""
    if n in _C5_ONE_GPU:
        return _C5_ONE_GPU[n]
    sp, _ = kta.synth_preset("c5")
    P = int(sp.n_partitions)
    counters, earliest, latest, smallest, largest = _oracle_counters_threaded(sp, n, P)
    o = _alive_oracle_threaded(sp, n)
    want = {"alive_keys": o.alive_keys(), "words": o.alive_words(), "counters": counters,
            "earliest": earliest, "latest": latest, "smallest": smallest, "largest": largest,
            "overall_size": int(counters[:, N.KTA_C_KEY_SIZE_SUM].sum() + counters[:, N.KTA_C_VALUE_SIZE_SUM].sum())}
    o.close()
    _C5_ONE_GPU[n] = want
    return want


@pytest.mark.parametrize("which", [2, 3])
def test_config_5_key_law_on_one_gpu_full_batches_vs_oracle(which):
    """The path a single GPU takes on config 5's key law (100 M distinct 16-byte keys, 50 % tombstones) in the bit set
    state: a bucket of a 15 x 2^24-record batch holds six times the distinct slots of pass 2's LDS table, so every bucket
    fails its plain attempt and is applied in eight slot-range passes (kta_alive.hip: kta_alive_apply<.., RANGES>; rounds
    4-5 applied it in instalments in segment order, round 4 sent all 1024 buckets of such a batch to kta_alive_fallback and
    applied the following batches in slices, a path no test reached and whose running count was wrong).  Two
    consecutive batches against ONE oracle fed in consumption order (/root/reference/src/metric.rs:288-305 is order
    dependent, kafka.rs:107-109 runs both handlers per message): every bit of the set, the running alive count, and for
    which = 3 the counters and extrema of the fused pass; the library's own counters say which path ran."""
    sp, _ = kta.synth_preset("c5")
    P, nb, batches = int(sp.n_partitions), 15 << 24, 2     # (15 x 2^24: the most a batch's u32 key offsets address with 16-byte keys)
    want = _c5_one_gpu_oracle(nb * batches)
    with kta.HipMetricHandler(P, count_alive_keys=True, now=NOW) as h:
        b = h.device_batch_alloc(nb, nb * 16)
        infos = []
        for k in range(batches):
            assert h.synth_fill_device(sp, k * nb, nb, b) == nb * 16
            h.submit_device(b, nb, k * nb, which=which)
            h.sync()
            infos.append(h.alive_pass_info())
        res, c = h.finish()
        # one launch pair per batch, no bucket given up
        assert [i["slices"] for i in infos] == [1, 2] and infos[-1]["failed_buckets"] == 0, infos
        assert infos[-1]["fused"] == (2 if which == 3 else 0) and infos[-1]["scanned"] == 0, infos
        words = h.export_alive_bitmap()
        assert np.array_equal(words, want["words"])                       # the set, bit for bit
        # ... and sum_all_alive, which the library keeps as a running count (kta_finish copies one word)
        assert int(np.bitwise_count(words).sum(dtype=np.uint64)) == want["alive_keys"]
        assert res.alive_keys == want["alive_keys"] and 0 < res.alive_keys
        if which == 3:
            assert np.array_equal(c, want["counters"]) and res.overall_count == nb * batches
            mm = kta.MessageMetrics(res, c, NOW)
            assert mm.earliest_message() == want["earliest"] and mm.latest_message() == want["latest"]
            assert mm.smallest_message() == want["smallest"] and mm.largest_message() == want["largest"]
            assert mm.overall_size() == want["overall_size"]
        h.reset()
        assert h.alive_pass_info()["slices"] == 0
        h.device_batch_free(b)


_C5_WORKER = r'''
import os, sys, threading
root, log2n = sys.argv[1], int(sys.argv[2])
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
import kafka_topic_analyzer_amd as kta
from helpers import NOW
from oracle_c import Oracle
sp, _ = kta.synth_preset("c5")
P, n, nranks = int(sp.n_partitions), 1 << log2n, 2
host = kta.synth_fill_host(sp, 0, n, with_keys=True, with_seq=True)
assert int(host["key_len"].min()) == int(host["key_len"].max()) == 16 and np.array_equal(host["seq"], np.arange(n, dtype=np.uint64))
o = Oracle(NOW, True); o.run_soa(host)
keys = host["key_bytes"].reshape(n, 16)
def shard_of(rank):       # partition p on rank p % nranks, every record with its GLOBAL sequence number
    idx = np.nonzero(host["partition"] % nranks == rank)[0]
    return {"partition": host["partition"][idx], "key_len": host["key_len"][idx], "val_len": host["val_len"][idx],
            "ts_ms": host["ts_ms"][idx], "key_off": (np.arange(len(idx), dtype=np.uint64) * 16).astype(np.uint32),
            "key_bytes": np.ascontiguousarray(keys[idx]).reshape(-1), "seq": host["seq"][idx]}
uid = kta.HipMetricHandler.comm_unique_id()
errors, stats = [], [None] * nranks
def at(rank, stage):           # where a rank was when the process died (stderr is shown by the failing assert)
    print("rank %d: %s" % (rank, stage), file=sys.stderr, flush=True)
def run(rank):
    try:
        h = kta.HipMetricHandler(P, count_alive_keys=True, now=NOW, seq_column=True)
        h.comm_create(nranks, rank, uid)
        sh = shard_of(rank)
        assert len(sh["partition"]) >= 1 << 21          # large enough for the partitioned pass (seq column: order checked on the device)
        b, nb = h.upload_batch(sh, with_keys=True)
        at(rank, "uploaded %d records" % nb)
        h.submit_device(b, nb, 0)
        at(rank, "submitted")
        h.exchange()
        at(rank, "exchanged")
        res, c = h.exchange_result()
        assert res.alive_keys == o.alive_keys(), (rank, res.alive_keys, o.alive_keys())
        assert np.array_equal(c, o.counters(P)) and res.overall_count == n
        lo, hi = -((-rank * (1 << 32)) // nranks), -((-(rank + 1) * (1 << 32)) // nranks)
        words, want = h.export_alive_bitmap(), o.alive_words()
        assert np.array_equal(words[(lo + 31) // 32:hi // 32], want[(lo + 31) // 32:hi // 32]), rank
        stats[rank] = h.comm_info()
        at(rank, "verified")
        h.device_batch_free(b); h.comm_destroy(); h.close()
    except BaseException as e:
        errors.append((rank, repr(e)))
        os._exit(2)
ts = [threading.Thread(target=run, args=(r,)) for r in range(nranks)]
[t.start() for t in ts]; [t.join() for t in ts]
assert not errors, errors
print("OK", o.alive_keys(), stats)
'''


def test_baseline_config_5_key_law_sharded_over_two_ranks_with_exchange(tmp_path):
    """BASELINE.json configs[4]'s key law (100 M distinct 16-byte keys, 50 % tombstones, 256 partitions) at 2^26
    records: two partition-sharded contexts on the one reachable GPU, each fed its partitions' records with a
    seq column of GLOBAL sequence numbers (>= 2^21 records per rank: the partitioned pass with the device's order
    check), then kta_exchange over tests/mock_rccl.cpp — every rank must hold the UNSHARDED oracle's counters and
    alive count, and the oracle's BitSet on the hash range it owns."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = tmp_path / "libmock_rccl.so"
    r = subprocess.run(["/opt/rocm/bin/hipcc", "-O1", "-shared", "-fPIC", "-std=c++17", os.path.join(root, "tests", "mock_rccl.cpp"),
                        "-o", str(lib), "-lrt", "-lpthread"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    script = tmp_path / "w.py"
    script.write_text(_C5_WORKER)
    env = dict(os.environ, KTA_RCCL_LIBRARY=str(lib))
    r = subprocess.run([sys.executable, str(script), root, "26"], capture_output=True, text=True, timeout=1200, env=env)
    assert r.returncode == 0 and "OK" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])
