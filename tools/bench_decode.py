"""The Kafka v2 decode kernel on its own (bench.py's `kafka_decode.by_batch_size` block without torch, so a fresh box
starts it in a second): raw uncompressed record batches resident in HBM -> columns, per batch size.

    python tools/bench_decode.py [--records 2000000] [--rpb 8,60,500] [--variants 0] [--reps 7]
                                 [--lib kafka_topic_analyzer_amd/libkta_hip.so.<tag>[,<another>]]

--lib: one process per library (a library is loaded once per process): the tool re-executes itself for every entry,
so one gpurun call can time several builds of the library (A/B of a kernel change: build the candidate into
libkta_hip.so.<tag> with `python -m kafka_topic_analyzer_amd.build --output ...` or by hand, both files travel).

Prints one JSON line per (library, batch size, variant): kernel ms (mean of --reps timed launches after two warm-up
launches, HIP events of the library's timing hooks), GB/s of raw log, fraction of the 8 TB/s HBM peak.  Every run
checks the decoded columns against the generator's."""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)

ap = argparse.ArgumentParser()
ap.add_argument("--records", type=int, default=2_000_000)
ap.add_argument("--rpb", default="8,60,500", help="records per batch: 8 ~2 KiB, 60 ~16 KiB, 500 ~134 KiB")
ap.add_argument("--variants", default="0")
ap.add_argument("--reps", type=int, default=7)
ap.add_argument("--lib", default="")
ap.add_argument("--sweeps", type=int, default=1, help="repeat the sweep over the variants this many times (A B C A B C: drift between variants shows)")
ap.add_argument("--val-mean", type=int, default=0, help="mean value length of the records (0: the c4 law's 208 B); e.g. 10240 with --rpb 6 --records 100000: records of ~10 KiB, whose 3-byte length prefixes the chain leaves to the byte loop")
args = ap.parse_args()

libs = [l for l in args.lib.split(",") if l]
if len(libs) > 1:
    for l in libs:
        argv = [sys.executable, os.path.abspath(__file__), "--records", str(args.records), "--rpb", args.rpb,
                "--variants", args.variants, "--reps", str(args.reps), "--lib", l, "--val-mean", str(args.val_mean),
                "--sweeps", str(args.sweeps)]
        subprocess.run(argv, check=False)
    sys.exit(0)

from kafka_topic_analyzer_amd import _native as N  # noqa: E402
if libs:
    N.LIB_PATH = os.path.abspath(libs[0])
import kafka_topic_analyzer_amd as kta  # noqa: E402

HBM_PEAK_GBS = 8000.0
lib = N.load()
spec, _ = kta.synth_preset("c4")
if args.val_mean:
    spec.val_mean = args.val_mean
    spec.val_cap = max(int(spec.val_cap), 8 * args.val_mean)
n = args.records
ref = kta.synth_fill_host(spec, 0, min(n, 1 << 18))
h = kta.HipMetricHandler(256)
for rpb in [int(x) for x in args.rpb.split(",")]:
    ln = C.c_uint64()
    room = n * (int(spec.val_mean) + 256) * 2 + (1 << 20)          # one pass: room to spare (untouched pages cost nothing)
    buf = np.zeros(room, np.uint8)
    lib.kta_kafka_encode_synth_host(C.byref(spec), 0, n, rpb, buf.ctypes.data, room - 64, C.byref(ln))
    if ln.value > room - 64:                                         # (the law's tail was longer: size it, then fill)
        buf = np.zeros(ln.value + 64, np.uint8)
        lib.kta_kafka_encode_synth_host(C.byref(spec), 0, n, rpb, buf.ctypes.data, ln.value, C.byref(ln))
    cap = n // rpb + 2
    descs = (N.KtaKafkaBatchDesc * cap)()
    st = N.KtaKafkaIndexStats()
    assert lib.kta_kafka_index_host(buf.ctypes.data_as(C.c_char_p), ln.value, 0, 0, 0, 0, descs, cap, C.byref(st)) == 0
    assert st.n_records == n
    blob = h.device_batch_alloc((ln.value + 3) // 4 + 32)
    h._check(lib.kta_copy_to_device(h._ctx, blob.partition, buf.ctypes.data, (ln.value + 63) // 64 * 64))
    out = h.device_batch_alloc(n, 16)
    for variant in [int(v) for v in args.variants.split(",")] * args.sweeps:
        h._check(lib.kta_kafka_set_variant(h._ctx, variant))
        a, c = (C.c_float * 2)(), (C.c_uint64 * 2)()

        def step():
            h._check(lib.kta_kafka_decode_device(h._ctx, blob.partition, ln.value, descs, st.n_batches, n, C.byref(out),
                                                 None, None))
        for _ in range(2):
            step()
        h.sync()
        h._check(lib.kta_kafka_time_stats(h._ctx, C.byref(a), C.byref(c)))      # drain what the warm-up recorded
        h.set_timing(True)
        for _ in range(args.reps):
            step()
        h.sync()
        h._check(lib.kta_kafka_time_stats(h._ctx, C.byref(a), C.byref(c)))
        h.set_timing(False)
        cols = h.download_batch(out, len(ref["partition"]))
        ok = all(np.array_equal(cols[k], ref[k]) for k in ("key_len", "val_len", "ts_ms"))
        print(json.dumps({"lib": os.path.basename(N.LIB_PATH), "records_per_batch": rpb, "variant": variant, "val_mean": int(spec.val_mean),
                          "batches": int(st.n_batches), "raw_log_bytes": int(ln.value), "kernel_ms": round(a[1], 4),
                          "launches": int(c[1]), "GBps": round(ln.value / (a[1] * 1e-3) / 1e9, 1),
                          # (a fraction of the HBM peak only where the kernel touches the whole log: with --val-mean the values
                          # beyond a window are never loaded, and log bytes / time says nothing about the memory — round 5's
                          # rows of 10 KiB records read 1.0 - 2.2 that way)
                          "frac": None if args.val_mean else round(ln.value / (a[1] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                          "columns_ok": bool(ok)}),
              flush=True)
    h.device_batch_free(out)
    h.device_batch_free(blob)
h.close()
