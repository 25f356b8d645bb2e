"""Host-fed staging rate: kta_batch_acquire / kta_batch_submit over the pinned staging ring (PCIe included), the
batches' contents left in place between submits so that the host does no work but the submit itself.  Metrics
only, and with --count-alive-keys (16-byte keys).  One H2D copy per batch (a slab): compare `copies per batch`
in a rocprofv3 --kernel-trace of this script (__amd_rocclr_copyBuffer calls / batches).

    python tools/bench_hostfed.py [--batches 40] [--log2-batch 22]
"""
import argparse
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import kafka_topic_analyzer_amd as kta  # noqa: E402
from kafka_topic_analyzer_amd import _native as N  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batches", type=int, default=40)
ap.add_argument("--log2-batch", type=int, default=22)
args = ap.parse_args()
n = 1 << args.log2_batch
lib = N.load()
for alive in (False, True):
    sp, _ = kta.synth_preset("c3" if alive else "c4")
    h = kta.HipMetricHandler(int(sp.n_partitions), count_alive_keys=alive, batch_capacity=n, key_bytes_capacity=16 * n)
    cols = kta.synth_fill_host(sp, 0, n, with_keys=alive)
    stages = 2
    for k in range(args.batches + stages):
        if k == stages:
            h.sync()
            t0 = time.perf_counter()
        b = N.KtaBatch()
        h._check(lib.kta_batch_acquire(h._ctx, C.byref(b)))
        if k < stages:      # fill each staging batch once; afterwards only the submit is timed
            for name in ("partition", "key_len", "val_len", "ts_ms"):
                C.memmove(getattr(b, name), cols[name].ctypes.data, cols[name].nbytes)
            if alive:
                C.memmove(b.key_off, cols["key_off"].ctypes.data, cols["key_off"].nbytes)
                C.memmove(b.key_bytes, cols["key_bytes"].ctypes.data, cols["key_bytes"].nbytes)
        h._check(lib.kta_batch_submit(h._ctx, n, cols["n_key_bytes"] if alive else 0, k * n))
    h.sync()
    dt = time.perf_counter() - t0
    res, _ = h.finish()
    per = (20 + (4 + 16 if alive else 0)) * n
    print(f"{'-c ' if alive else ''}host-fed: {args.batches} batches of 2^{args.log2_batch} records in {dt * 1e3:.1f} ms = "
          f"{args.batches * n / dt / 1e6:.0f} M records/s, {args.batches * per / dt / 1e9:.1f} GB/s over PCIe "
          f"({per / 1e6:.0f} MB per batch, one copy), records seen {res.overall_count}", flush=True)
    h.close()
