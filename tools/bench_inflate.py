"""Compressed record sets on the device: inflate + decode of gzip / zstd (and snappy / lz4 for reference) batches,
the workload of bench.py's `kafka_decode.compressed` block on its own, per variant of the inflate kernels.

    python tools/bench_inflate.py [--records 1000000] [--rpb 60] [--codecs gzip,zstd] [--variants 0,1]

Prints one JSON line per (codec, variant): ms of the device work (best of 5), GB/s of compressed input.  The
decoded columns are checked against the generator's columns for every run.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (the batch re-encoder)
import kafka_topic_analyzer_amd as kta  # noqa: E402
from kafka_topic_analyzer_amd import _native as N  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--records", type=int, default=1_000_000)
ap.add_argument("--rpb", type=int, default=60)
ap.add_argument("--codecs", default="gzip,zstd")
ap.add_argument("--variants", default="0,1")
ap.add_argument("--values", default="pattern", choices=["pattern", "text"],
                help="value bytes: the 24-byte periodic pattern of bench.py, or words / numbers / punctuation (JSON-like)")
args = ap.parse_args()
lib = N.load()
spec, _ = kta.synth_preset("c4")
nc, rpb = args.records, args.rpb
ids = {"snappy": 2, "lz4": 3, "gzip": 1, "zstd": 4}
ref = kta.synth_fill_host(spec, 0, min(nc, 1 << 16))
h = kta.HipMetricHandler(256)
for name in args.codecs.split(","):
    codec = ids[name]
    enc = (codec if codec in (2, 3) else 0x100) | (0x200 if args.values == "text" else 0)
    ln = C.c_uint64()
    lib.kta_kafka_encode_synth_host_ex(C.byref(spec), 0, nc, rpb, enc, None, 0, C.byref(ln))
    cbuf = np.zeros(ln.value + 128, np.uint8)
    lib.kta_kafka_encode_synth_host_ex(C.byref(spec), 0, nc, rpb, enc, cbuf.ctypes.data, ln.value, C.byref(ln))
    if codec in (1, 4):
        gz = bench._recompress_batches(lib, cbuf[:ln.value].tobytes(), codec)
        cbuf = np.zeros(len(gz) + 128, np.uint8)
        cbuf[:len(gz)] = np.frombuffer(gz, np.uint8)
        ln.value = len(gz)
    inflate_at = (ln.value + 127) & ~63
    cap = nc // rpb + 2
    descs = (N.KtaKafkaBatchDesc * cap)()
    st = N.KtaKafkaIndexStats()
    assert lib.kta_kafka_index_host(cbuf.ctypes.data_as(C.c_char_p), ln.value, 0, 0, 0, inflate_at, descs, cap, C.byref(st)) == 0
    assert st.n_records == nc
    blob = h.device_batch_alloc((inflate_at + st.inflate_bytes + 256) // 4 + 1)
    h._check(lib.kta_copy_to_device(h._ctx, blob.partition, cbuf.ctypes.data, (ln.value + 63) // 64 * 64))
    out = h.device_batch_alloc(nc, 16)
    for variant in [int(v) for v in args.variants.split(",")]:
        h._check(lib.kta_kafka_set_variant(h._ctx, variant))
        best = 1e9
        bad = C.c_uint64()
        for _ in range(5):
            h.sync()
            t0 = time.perf_counter()
            h._check(lib.kta_kafka_decode_device(h._ctx, blob.partition, ln.value, descs, st.n_batches, nc, C.byref(out), None, None))
            h.sync()
            best = min(best, time.perf_counter() - t0)
        h._check(lib.kta_kafka_decode_device(h._ctx, blob.partition, ln.value, descs, st.n_batches, nc, C.byref(out), None, C.byref(bad)))
        cols = h.download_batch(out, len(ref["partition"]))
        ok = bad.value == 0 and all(np.array_equal(cols[k], ref[k]) for k in ("key_len", "val_len", "ts_ms"))
        print(json.dumps({"codec": name, "values": args.values, "variant": variant, "batches": int(st.n_batches),
                          "compressed_bytes": int(ln.value), "inflate_area": int(st.inflate_bytes), "ms": round(best * 1e3, 3),
                          "compressed_GBps": round(ln.value / best / 1e9, 2), "ok": bool(ok), "bad_batches": int(bad.value)}), flush=True)
    h._check(lib.kta_kafka_set_variant(h._ctx, 0))
    h.device_batch_free(out)
    h.device_batch_free(blob)
h.close()
