"""Developer tool: returning-atomic running count, P=4096, PCIe-inclusive staging rate."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import ctypes as C
import kafka_topic_analyzer_amd as kta
from kafka_topic_analyzer_amd import _native as N

n = 1 << 26
h = kta.HipMetricHandler(64, count_alive_keys=True)
b = h.device_batch_alloc(n, n * 16)
spec, _ = kta.synth_preset("c3")
kb = h.synth_fill_device(spec, 0, n, b); h.sync()
h.set_timing(True)
for variant in (0, 1):
    for wgs in (2048, 8192):
        h.reset(); h.set_tuning(alive_workgroups=wgs, alive_variant=variant)
        for it in range(2): h.submit_device(b, n, 0, which=2)
        h.kernel_time_stats()
        for it in range(5): h.submit_device(b, n, 0, which=2)
        ms, _ = h.kernel_time_stats()
        t0 = time.perf_counter(); res, _ = h.finish(); tf = time.perf_counter() - t0
        print(f"alive variant={variant} wgs={wgs}: {ms[2]:.3f} ms {n/ms[2]/1e6:.2f} Grec/s alive_keys={res.alive_keys} finish {tf*1e3:.2f} ms", flush=True)
h.set_timing(False)
h.device_batch_free(b); h.close()

# P = 4096 (96 KiB of dynamic LDS)
from helpers import random_cols, NOW
from oracle_c import Oracle
rng = np.random.default_rng(1)
cols = random_cols(rng, 200000, 4096, key_space=1000)
o = Oracle(NOW); o.run_soa(cols)
with kta.HipMetricHandler(4096, now=NOW) as h4:
    h4.submit_columns(cols["partition"], cols["key_len"], cols["val_len"], cols["ts_ms"])
    res, c = h4.finish()
    print("P=4096 parity", np.array_equal(c, o.counters(4096)), res.overall_count)

# PCIe-inclusive: pinned staging ring -> H2D -> kernels, buffers already filled (no host fill cost)
for alive in (False, True):
    cap = 1 << 22
    hh = kta.HipMetricHandler(256, count_alive_keys=alive, batch_capacity=cap, key_bytes_capacity=cap * 16, n_staging=3)
    sp, _ = kta.synth_preset("c3"); sp.n_partitions = 256
    lib = N.load()
    kbs = 0
    for st in range(3):  # fill each stage once with valid records
        bb = kta.KtaBatch(); lib.kta_batch_acquire(hh._ctx, C.byref(bb))
        k = C.c_uint64(0)
        if not alive: bb.key_off, bb.key_bytes = None, None
        lib.kta_synth_fill_host(C.byref(sp), st * cap, cap, C.byref(bb), C.byref(k)); kbs = k.value if alive else 0
        lib.kta_batch_submit(hh._ctx, cap, kbs, st * cap)
    hh.sync()
    reps = 30
    t0 = time.perf_counter()
    for r in range(reps):
        bb = kta.KtaBatch(); lib.kta_batch_acquire(hh._ctx, C.byref(bb))
        lib.kta_batch_submit(hh._ctx, cap, kbs, (3 + r) * cap)
    hh.sync()
    dt = time.perf_counter() - t0
    byts = cap * (20 + (4 if alive else 0)) + kbs
    print(f"host-fed (pinned, alive={alive}): {cap*reps/dt/1e9:.2f} Grec/s  {byts*reps/dt/1e9:.1f} GB/s over PCIe", flush=True)
    hh.close()
