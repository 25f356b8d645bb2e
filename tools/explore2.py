"""Developer tool: scan-kernel launch-geometry sweep at full size."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import kafka_topic_analyzer_amd as kta

n = int(os.environ.get("N", 1 << 30))
spec, _ = kta.synth_preset("c4")
h = kta.HipMetricHandler(256)
b = h.device_batch_alloc(n)
h.synth_fill_device(spec, 0, n, b); h.sync()
h.set_timing(True)
for variant in (1, 17, 0, 16, 9, 25):
    for wgs in (256, 384, 512, 640, 768, 1024, 1280, 2048):
        h.set_tuning(scan_workgroups=wgs, scan_variant=variant)
        for it in range(2):
            h.submit_device(b, n, 0, which=1)
        h.kernel_time_stats()
        for it in range(6):
            h.submit_device(b, n, 0, which=1)
        ms, cnt = h.kernel_time_stats()
        print(f"variant={variant:2d} wgs={wgs:5d}: scan {ms[0]:.3f} ms  {n * 20 / ms[0] / 1e6:7.1f} GB/s  fold {ms[1]*1e3:.1f} us", flush=True)
