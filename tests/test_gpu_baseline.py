"""Parity at the sizes BASELINE.json names (configs[1..3]), against the C oracle on the SAME records:

    c2   8 partitions, 100 M records, mixed key / value sizes          counters + extrema, bit-exact
    c3   64 partitions, 2^28 records, --count-alive-keys, 10 M keys    alive count + all 2^32 bits of the set
    c4   256 partitions, 2^30 records (one GPU's worth of config 4)    counters + extrema, bit-exact

The oracle is single-threaded like the reference (src/kafka.rs:92-135).  For the counter configs it runs as
T independent instances over consecutive chunks of the topic (its state is sums and extrema, so the
instances merge by + / min / max — done here in numpy, nothing of the product involved); the BitSet of c3 is
order dependent and runs in one instance, fed chunk by chunk in consumption order.  Records come from the
counter-based generator (include/kta_synth.h), which the device and the host evaluate identically
(test_device_generator_matches_host_generator)."""
import os
import threading

import numpy as np
import pytest

import kafka_topic_analyzer_amd as kta
from kafka_topic_analyzer_amd import _native as N
from helpers import NOW
from oracle_c import Oracle

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


def test_baseline_config_3_alive_keys_2e28_records():
    sp, _ = kta.synth_preset("c3")
    n, P = 1 << 28, 64
    o = Oracle(NOW, True)
    # the oracle's BitSet consumes the topic in order; three generator threads run a few chunks ahead of it
    from concurrent.futures import ThreadPoolExecutor
    chunks = [(lo, min(CHUNK * 4, n - lo)) for lo in range(0, n, CHUNK * 4)]
    pool = ThreadPoolExecutor(3)
    ahead = [pool.submit(kta.synth_fill_host, sp, lo, m, True) for lo, m in chunks[:4]]
    with kta.HipMetricHandler(P, count_alive_keys=True, now=NOW) as h:
        slice_n = 1 << 26                      # one device batch at a time: 2^26 records, 1 GiB of keys
        b = h.device_batch_alloc(slice_n, slice_n * 16)
        for lo in range(0, n, slice_n):
            assert h.synth_fill_device(sp, lo, slice_n, b) == slice_n * 16
            h.submit_device(b, slice_n, lo)    # both handlers, as the reference runs them
            h.sync()
        for k in range(len(chunks)):
            cols = ahead.pop(0).result()
            if k + 4 < len(chunks):
                ahead.append(pool.submit(kta.synth_fill_host, sp, chunks[k + 4][0], chunks[k + 4][1], True))
            o.run_soa(cols)
        pool.shutdown()
        res, c = h.finish()
        assert res.alive_keys == o.alive_keys() and 0 < res.alive_keys <= 10_000_000
        assert np.array_equal(c, o.counters(P)) and res.overall_count == n
        assert np.array_equal(h.export_alive_bitmap(), o.alive_words())
        h.device_batch_free(b)
    o.close()
