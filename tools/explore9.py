"""Developer tool: end-to-end raw-log pipeline rate (pinned blob -> PCIe -> decode -> metrics [-> alive])."""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import kafka_topic_analyzer_amd as kta
from kafka_topic_analyzer_amd import _native as N

lib = N.load()
n = 900_000  # ~240 MB of raw log per blob
sp, _ = kta.synth_preset("c4")
ln = C.c_uint64()
codec = int(os.environ.get("CODEC", "0"))
lib.kta_kafka_encode_synth_host_ex(C.byref(sp), 0, n, 60, codec, None, 0, C.byref(ln))
buf = np.zeros(ln.value + 64, np.uint8)
lib.kta_kafka_encode_synth_host_ex(C.byref(sp), 0, n, 60, codec, buf.ctypes.data, ln.value, C.byref(ln))
for alive in (False, True):
    for stages in (3,):
        h = kta.HipMetricHandler(256, count_alive_keys=alive)
        h._check(lib.kta_kafka_configure(h._ctx, ln.value + 4096, stages))
        st = N.KtaKafkaIndexStats()
        for k in range(stages):  # fill every stage once (fetcher cost is not what we measure)
            p, cap = C.c_void_p(), C.c_uint64()
            h._check(lib.kta_kafka_blob_acquire(h._ctx, C.byref(p), C.byref(cap)))
            C.memmove(p, buf.ctypes.data, ln.value)
            h._check(lib.kta_kafka_blob_submit(h._ctx, ln.value, k % 256, C.byref(st)))
        h.sync()
        reps = 24
        t0 = time.perf_counter()
        for r in range(reps):
            p, cap = C.c_void_p(), C.c_uint64()
            h._check(lib.kta_kafka_blob_acquire(h._ctx, C.byref(p), C.byref(cap)))
            h._check(lib.kta_kafka_blob_submit(h._ctx, ln.value, r % 256, C.byref(st)))
        h.sync()
        dt = time.perf_counter() - t0
        res, c = h.finish()
        assert res.overall_count == n * (reps + stages), (res.overall_count, n * (reps + stages))
        print(f"codec={codec} raw-log pipeline alive={alive!s:5s} stages={stages}: {ln.value*reps/dt/1e9:.1f} GB/s of Kafka log end to end, "
              f"{n*reps/dt/1e6:.1f} M records/s  (blob {ln.value/1e6:.0f} MB, {st.n_batches} batches)", flush=True)
        h.close()
