"""Developer tool: scan-kernel variant choice across partition-order / partition-count regimes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import kafka_topic_analyzer_amd as kta
from kafka_topic_analyzer_amd import _native as N

n = int(os.environ.get("N", 1 << 29))
for preset, P, mode in (("c4", 256, "random"), ("c4", 256, "runs"), ("c2", 8, "random"), ("c2", 8, "runs"), ("c1", 1, "random"), ("c3", 64, "random")):
    spec, _ = kta.synth_preset(preset)
    if mode == "runs":
        spec.part_mode, spec.part_run_len = N.KTA_PART_RUNS, 500
    h = kta.HipMetricHandler(P)
    b = h.device_batch_alloc(n)
    h.synth_fill_device(spec, 0, n, b); h.sync()
    h.set_timing(True)
    for variant in (16, 17):
        for wgs in (512, 768):
            h.set_tuning(scan_workgroups=wgs, scan_variant=variant)
            for it in range(3):
                h.submit_device(b, n, 0, which=1)
            h.kernel_time_stats()
            for it in range(12):
                h.submit_device(b, n, 0, which=1)
            ms, cnt = h.kernel_time_stats()
            print(f"{preset} P={P:3d} {mode:6s} variant={variant:2d} wgs={wgs:5d}: scan {ms[0]:.3f} ms  {n * 20 / ms[0] / 1e6:7.1f} GB/s", flush=True)
    h.device_batch_free(b); h.close()
