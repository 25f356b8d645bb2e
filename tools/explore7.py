"""Developer tool: cost of the additive analytics in the scan kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kafka_topic_analyzer_amd as kta
from kafka_topic_analyzer_amd import _native as N

n = 1 << 29
for preset, P, mode in (("c4", 256, "random"), ("c4", 256, "runs"), ("c2", 8, "random"), ("c1", 1, "random")):
    spec, _ = kta.synth_preset(preset)
    if mode == "runs":
        spec.part_mode, spec.part_run_len = N.KTA_PART_RUNS, 500
    for an in (False, True):
        h = kta.HipMetricHandler(P, analytics=an)
        b = h.device_batch_alloc(n)
        h.synth_fill_device(spec, 0, n, b); h.sync()
        h.set_timing(True)
        for wgs in (0, 512, 768):
            h.set_tuning(scan_workgroups=wgs)
            for it in range(3): h.submit_device(b, n, 0, which=1)
            h.kernel_time_stats()
            for it in range(10): h.submit_device(b, n, 0, which=1)
            ms, _ = h.kernel_time_stats()
            print(f"{preset} P={P:3d} {mode:6s} analytics={an!s:5s} wgs={wgs}: scan {ms[0]:.3f} ms {n*20/ms[0]/1e6:7.1f} GB/s fold {ms[1]*1e3:.1f} us", flush=True)
        h.device_batch_free(b); h.close()
