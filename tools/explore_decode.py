"""Decode kernel variants (kta_kafka_set_variant) x batch sizes: kernel time from HIP events.
Run on the GPU box: python tools/explore_decode.py [n_records]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import kafka_topic_analyzer_amd as kta  # noqa: E402
from kafka_topic_analyzer_amd import _native as N  # noqa: E402

lib = N.load()
n_records = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
spec, _ = kta.synth_preset("c4")
for rpb in (8, 60, 500, 4000):
    ln = C.c_uint64()
    lib.kta_kafka_encode_synth_host(C.byref(spec), 0, n_records, rpb, None, 0, C.byref(ln))
    buf = np.zeros(ln.value + 64, np.uint8)
    lib.kta_kafka_encode_synth_host(C.byref(spec), 0, n_records, rpb, buf.ctypes.data, ln.value, C.byref(ln))
    nb_cap = n_records // rpb + 2
    descs = (N.KtaKafkaBatchDesc * nb_cap)()
    st = N.KtaKafkaIndexStats()
    assert lib.kta_kafka_index_host(buf.ctypes.data_as(C.c_char_p), ln.value, 0, 0, 0, 0, descs, nb_cap, C.byref(st)) == 0
    h = kta.HipMetricHandler(256)
    d_blob = h.device_batch_alloc((ln.value + 3) // 4 + 32)
    h._check(lib.kta_copy_to_device(h._ctx, d_blob.partition, buf.ctypes.data, (ln.value + 63) // 64 * 64))
    out = h.device_batch_alloc(n_records, 16)
    ref = kta.synth_fill_host(spec, 0, 1 << 16)
    for variant in (1, 2, 3, 4, 5, 0):
        lib.kta_kafka_set_variant(variant)
        for _ in range(2):
            h._check(lib.kta_kafka_decode_device(h._ctx, d_blob.partition, ln.value, descs, st.n_batches, n_records,
                                                 C.byref(out), None, None))
        h.sync()
        h.set_timing(True)
        for _ in range(5):
            h._check(lib.kta_kafka_decode_device(h._ctx, d_blob.partition, ln.value, descs, st.n_batches, n_records,
                                                 C.byref(out), None, None))
        h.sync()
        a, c = (C.c_float * 2)(), (C.c_uint64 * 2)()
        h._check(lib.kta_kafka_time_stats(h._ctx, C.byref(a), C.byref(c)))
        h.set_timing(False)
        cols = h.download_batch(out, 1 << 16)
        ok = all(np.array_equal(cols[k], ref[k]) for k in ("key_len", "val_len", "ts_ms"))
        print(f"rpb={rpb:5d} batches={st.n_batches:7d} variant={variant}: {a[1]:.4f} ms  "
              f"{ln.value / (a[1] * 1e-3) / 1e9:8.1f} GB/s  {n_records / (a[1] * 1e-3) / 1e9:6.2f} G rec/s  parity={ok}",
              flush=True)
    lib.kta_kafka_set_variant(0)
    h.device_batch_free(out)
    h.device_batch_free(d_blob)
    h.close()
