"""gzip batches (zlib level 6 over the patterned synthetic records): inflate + decode wall time.
python tools/explore_gzip.py [n_records] [rpb]"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import bench  # noqa: E402
import kafka_topic_analyzer_amd as kta  # noqa: E402
from kafka_topic_analyzer_amd import _native as N  # noqa: E402

lib = N.load()
nc = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
rpb = int(sys.argv[2]) if len(sys.argv) > 2 else 60
spec, _ = kta.synth_preset("c4")
ln = C.c_uint64()
lib.kta_kafka_encode_synth_host_ex(C.byref(spec), 0, nc, rpb, 0x100, None, 0, C.byref(ln))
raw = np.zeros(ln.value + 128, np.uint8)
lib.kta_kafka_encode_synth_host_ex(C.byref(spec), 0, nc, rpb, 0x100, raw.ctypes.data, ln.value, C.byref(ln))
raw_len = ln.value
gz = bench._gzip_batches(lib, raw[:raw_len].tobytes())
cbuf = np.zeros(len(gz) + 128, np.uint8)
cbuf[:len(gz)] = np.frombuffer(gz, np.uint8)
inflate_at = (len(gz) + 127) & ~63
cap = nc // rpb + 2
descs = (N.KtaKafkaBatchDesc * cap)()
st = N.KtaKafkaIndexStats()
assert lib.kta_kafka_index_host(cbuf.ctypes.data_as(C.c_char_p), len(gz), 0, 0, 0, inflate_at, descs, cap, C.byref(st)) == 0
ref = kta.synth_fill_host(spec, 0, min(nc, 1 << 16))
for lanes in (16,):
    h = kta.HipMetricHandler(256)
    blob = h.device_batch_alloc((inflate_at + st.inflate_bytes + 256) // 4 + 1)
    h._check(lib.kta_copy_to_device(h._ctx, blob.partition, cbuf.ctypes.data, (len(gz) + 63) // 64 * 64))
    out = h.device_batch_alloc(nc, 16)
    best = 1e9
    for _ in range(4):
        h.sync()
        t0 = time.perf_counter()
        bad = C.c_uint64()
        h._check(lib.kta_kafka_decode_device(h._ctx, blob.partition, len(gz), descs, st.n_batches, nc, C.byref(out), None,
                                             C.byref(bad)))
        h.sync()
        best = min(best, time.perf_counter() - t0)
    cols = h.download_batch(out, len(ref["key_len"]))
    ok = all(np.array_equal(cols[k], ref[k]) for k in ("key_len", "val_len", "ts_ms")) and bad.value == 0
    print(f"lanes={lanes}: {st.n_gzip} gzip batches, {len(gz)} -> {raw_len} bytes: {best * 1e3:.2f} ms = "
          f"{len(gz) / best / 1e9:.1f} GB/s compressed, {raw_len / best / 1e9:.1f} GB/s inflated, parity={ok}", flush=True)
    h.device_batch_free(out)
    h.device_batch_free(blob)
    h.close()
