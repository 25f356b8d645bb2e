"""Developer tool: alive-pass throughput vs distinct-key working set."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import kafka_topic_analyzer_amd as kta

n = int(os.environ.get("N", 1 << 26))
h = kta.HipMetricHandler(64, count_alive_keys=True)
b = h.device_batch_alloc(n, n * 16)
h.set_timing(True)
for D in (10_000, 100_000, 1_000_000, 4_000_000, 10_000_000, 30_000_000, 100_000_000, 0):
    spec, _ = kta.synth_preset("c3")
    spec.n_distinct_keys = D
    kb = h.synth_fill_device(spec, 0, n, b); h.sync()
    for it in range(2):
        h.submit_device(b, n, 0, which=2)
    h.kernel_time_stats()
    for it in range(5):
        h.submit_device(b, n, 0, which=2)
    ms, cnt = h.kernel_time_stats()
    print(f"D={D:>11d}: alive {ms[2]:.3f} ms  {n / ms[2] / 1e6:7.2f} Grec/s   lines*64B={min(D or n, n) * 64 / 1e6:.0f} MB", flush=True)
