"""Developer tool: alive-pass ablation (hash only / table update only / fused)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kafka_topic_analyzer_amd as kta

n = int(os.environ.get("N", 1 << 26))
h = kta.HipMetricHandler(64, count_alive_keys=True)
b = h.device_batch_alloc(n, n * 16)
h.set_timing(True)
for D in (10_000, 1_000_000, 10_000_000, 100_000_000, 0):
    spec, _ = kta.synth_preset("c3")
    spec.n_distinct_keys = D
    kb = h.synth_fill_device(spec, 0, n, b); h.sync()
    for variant in (1, 2):
        for wgs in (2048, 8192):
            h.set_tuning(alive_workgroups=wgs, alive_variant=variant)
            for it in range(2):
                h.submit_device(b, n, 0, which=2)
            h.kernel_time_stats()
            for it in range(5):
                h.submit_device(b, n, 0, which=2)
            ms, cnt = h.kernel_time_stats()
            print(f"D={D:>9d} variant={variant} wgs={wgs:5d}: {ms[2]:.3f} ms  {n / ms[2] / 1e6:7.2f} Grec/s", flush=True)
