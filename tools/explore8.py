"""Developer tool: throughput of the Kafka record-batch decode kernels."""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import kafka_topic_analyzer_amd as kta
from kafka_topic_analyzer_amd import _native as N

lib = N.load()
n = int(os.environ.get("N", 2_000_000))
for preset, rpb in (("c4", 60), ("c4", 600), ("c4", 4000), ("c3", 60)):
    sp, _ = kta.synth_preset(preset)
    ln = C.c_uint64()
    lib.kta_kafka_encode_synth_host(C.byref(sp), 0, n, rpb, None, 0, C.byref(ln))
    buf = np.zeros(ln.value + 16, np.uint8)
    t0 = time.time(); lib.kta_kafka_encode_synth_host(C.byref(sp), 0, n, rpb, buf.ctypes.data, ln.value, C.byref(ln)); te = time.time() - t0
    blob = buf[:ln.value]
    st = N.KtaKafkaIndexStats()
    nb_cap = n // rpb + 2
    descs = (N.KtaKafkaBatchDesc * nb_cap)()
    t0 = time.time(); rc = lib.kta_kafka_index_host(blob.ctypes.data_as(C.c_char_p), ln.value, 0, 0, 0, 0, descs, nb_cap, C.byref(st)); ti = time.time() - t0
    assert rc == 0 and st.n_records == n
    for variant, alive in ((0, False), (0, True), (2, True)):
        waves = 0
        lib.kta_kafka_set_variant(0)
        h = kta.HipMetricHandler(256, count_alive_keys=False)
        lib.kta_kafka_set_check_crcs(h._ctx, 1 if variant == 2 else 0)
        d_blob = h.device_batch_alloc((ln.value + 3) // 4 + 32)
        h._check(lib.kta_copy_to_device(h._ctx, d_blob.partition, buf.ctypes.data, (ln.value + 3) // 4 * 4))
        out = h.device_batch_alloc(n, ln.value if alive else 0)
        h.set_timing(True)
        kb, bad = C.c_uint64(), C.c_uint64()
        for it in range(6):
            h._check(lib.kta_kafka_decode_device(h._ctx, d_blob.partition, ln.value, descs, st.n_batches, n, C.byref(out), C.byref(kb), C.byref(bad)))
            if it == 0:
                a, c = (C.c_float * 2)(), (C.c_uint64 * 2)(); lib.kta_kafka_time_stats(h._ctx, C.byref(a), C.byref(c))
        a, c = (C.c_float * 2)(), (C.c_uint64 * 2)(); lib.kta_kafka_time_stats(h._ctx, C.byref(a), C.byref(c))
        tot = (a[0] if a[0] > 0 else 0) + a[1]
        print(f"{preset} rpb={rpb:5d} batches={st.n_batches:6d} blob={ln.value/1e6:.0f} MB waves={waves} keys={alive!s:5s}: crc {a[0]:.3f} ms decode {a[1]:.3f} ms "
              f"-> {n/tot/1e6:.2f} Grec/s, {ln.value/tot/1e6:.1f} GB/s of raw log  (host: encode {te:.2f}s index {ti*1e3:.1f} ms = {ln.value/ti/1e9:.1f} GB/s) bad={bad.value}", flush=True)
        h.device_batch_free(out); h.device_batch_free(d_blob); h.close()
