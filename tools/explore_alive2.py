"""Alive-key pass: counting kernel (variant 1) vs the filtered kernel (variant 2: backwards walk + pre-read)
on a compacted-topic shape (config 3: 10 M distinct keys over 2^26 records) and on a mostly-unique shape
(config 5 keys: 100 M distinct).  Kernel time from HIP events.  python tools/explore_alive2.py"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import kafka_topic_analyzer_amd as kta  # noqa: E402

n = 1 << 26
for preset in ("c3", "c5"):
    sp, _ = kta.synth_preset(preset)
    h = kta.HipMetricHandler(256, count_alive_keys=True)
    b = h.device_batch_alloc(n, n * 16)
    h.synth_fill_device(sp, 0, n, b)
    for variant in (1, 2, 1, 2):
        h.set_tuning(alive_variant=variant)
        h.reset()
        h.submit_device(b, n, 0, which=2)      # warm: first touch of the table
        h.reset()
        h.sync()
        h.set_timing(True)
        h.submit_device(b, n, 0, which=2)      # a fresh table: every key's last record wins
        h.submit_device(b, n, n, which=2)      # the same keys again with later sequence numbers
        h.sync()
        ms, cnt = h.kernel_time_stats()
        h.set_timing(False)
        res, _ = h.finish()
        print(f"{preset} variant={variant}: {ms[2]:.3f} ms avg over {cnt[2]} launches = {n / ms[2] / 1e6:.1f} G records/s, "
              f"alive={res.alive_keys}", flush=True)
    h.device_batch_free(b)
    h.close()
