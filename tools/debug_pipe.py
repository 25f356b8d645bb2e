import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import kafka_topic_analyzer_amd as kta
from kafka_topic_analyzer_amd import _native as N
lib = N.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000
alive = len(sys.argv) > 2 and sys.argv[2] == "1"
sp, _ = kta.synth_preset("c4")
ln = C.c_uint64()
lib.kta_kafka_encode_synth_host(C.byref(sp), 0, n, 60, None, 0, C.byref(ln))
buf = np.zeros(ln.value + 64, np.uint8)
lib.kta_kafka_encode_synth_host(C.byref(sp), 0, n, 60, buf.ctypes.data, ln.value, C.byref(ln))
print("encoded", ln.value, flush=True)
h = kta.HipMetricHandler(256, count_alive_keys=alive)
h._check(lib.kta_kafka_configure(h._ctx, ln.value + 4096, 2))
st = N.KtaKafkaIndexStats()
for k in range(5):
    p, cap = C.c_void_p(), C.c_uint64()
    h._check(lib.kta_kafka_blob_acquire(h._ctx, C.byref(p), C.byref(cap)))
    print("acquired", k, hex(p.value), cap.value, flush=True)
    C.memmove(p, buf.ctypes.data, ln.value)
    h._check(lib.kta_kafka_blob_submit(h._ctx, ln.value, k % 256, C.byref(st)))
    print("submitted", k, st.n_batches, st.n_records, st.bytes_consumed, flush=True)
    h.sync()
    print("synced", k, flush=True)
res, c = h.finish()
print("finish", res.overall_count, res.bad_partition_records, flush=True)
h.close()
