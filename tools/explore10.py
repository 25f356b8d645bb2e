"""Developer tool: Snappy-compressed record sets: inflate + decode throughput."""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import kafka_topic_analyzer_amd as kta
from kafka_topic_analyzer_amd import _native as N

lib = N.load()
n = int(os.environ.get("N", 2_000_000))
for codec, rpb in ((2, 60), (2, 600), (3, 60), (3, 600)):
    sp, _ = kta.synth_preset("c4")
    ln = C.c_uint64()
    lib.kta_kafka_encode_synth_host_ex(C.byref(sp), 0, n, rpb, codec, None, 0, C.byref(ln))
    buf = np.zeros(ln.value + 128, np.uint8)
    lib.kta_kafka_encode_synth_host_ex(C.byref(sp), 0, n, rpb, codec, buf.ctypes.data, ln.value, C.byref(ln))
    st = N.KtaKafkaIndexStats()
    nb_cap = n // rpb + 2
    descs = (N.KtaKafkaBatchDesc * nb_cap)()
    inflate_at = (ln.value + 127) & ~63
    t0 = time.time(); rc = lib.kta_kafka_index_host(buf.ctypes.data_as(C.c_char_p), ln.value, 0, 0, 0, inflate_at, descs, nb_cap, C.byref(st)); ti = time.time() - t0
    assert rc == 0 and st.n_records == n and st.n_snappy + st.n_lz4 == st.n_batches
    total = inflate_at + st.inflate_bytes + 128
    h = kta.HipMetricHandler(256)
    d_blob = h.device_batch_alloc(total // 4 + 1)
    h._check(lib.kta_copy_to_device(h._ctx, d_blob.partition, buf.ctypes.data, (ln.value + 63) // 64 * 64))
    out = h.device_batch_alloc(n, 16)
    for it in range(5):
        h.sync(); t0 = time.perf_counter()
        h._check(lib.kta_kafka_decode_device(h._ctx, d_blob.partition, ln.value, descs, st.n_batches, n, C.byref(out), None, None))
        h.sync(); dt = time.perf_counter() - t0
    cols = h.download_batch(out, 1000)
    ref = kta.synth_fill_host(sp, 0, 1000)
    assert np.array_equal(cols["val_len"], ref["val_len"]) and np.array_equal(cols["ts_ms"], ref["ts_ms"])
    print(f"{'snappy' if codec == 2 else 'lz4'} rpb={rpb}: compressed {ln.value/1e6:.0f} MB -> inflated {st.inflate_bytes/1e6:.0f} MB; inflate+decode {dt*1e3:.2f} ms "
          f"= {st.inflate_bytes/dt/1e9:.1f} GB/s of inflated log, {ln.value/dt/1e9:.1f} GB/s of compressed log, {n/dt/1e6:.0f} M records/s; host index {ti*1e3:.1f} ms", flush=True)
    h.device_batch_free(out); h.device_batch_free(d_blob); h.close()
