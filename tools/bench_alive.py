"""Alive-key pass (--count-alive-keys) kernel comparison on HBM-resident batches: the single-kernel filtered
update (variant 2) against the partitioned pass (3 automatic, 13 forced) in the TABLE state, on the config-3
shape (10 M distinct keys, a compacted topic) and the config-5 key law (100 M distinct, mostly unique per
batch).  Kernel time from HIP events on the compute stream; every variant must report the same alive count.

    python tools/bench_alive.py [--n LOG2] [--variants 2,3,4] [--wgs 0,512] [--presets c3,c5]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import kafka_topic_analyzer_amd as kta  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=26)
ap.add_argument("--variants", default="2,3,13")
ap.add_argument("--wgs", default="0")
ap.add_argument("--presets", default="c3,c5")
ap.add_argument("--reps", type=int, default=8)
args = ap.parse_args()
n = 1 << args.n
for preset in args.presets.split(","):
    sp, _ = kta.synth_preset(preset)
    h = kta.HipMetricHandler(256, count_alive_keys=True, alive_table=True)
    b = h.device_batch_alloc(n, n * 16)
    h.synth_fill_device(sp, 0, n, b)
    want = None
    for variant in [int(v) for v in args.variants.split(",")]:
        for wgs in [int(w) for w in args.wgs.split(",")]:
            if wgs and variant < 3:
                continue
            h.set_tuning(alive_workgroups=wgs, alive_variant=variant)
            h.reset()
            h.submit_device(b, n, 0, which=2)      # warm: first touch of the table, workspace allocation
            h.reset()
            h.sync()
            import time
            h.set_timing(True)
            t0 = time.perf_counter()
            for r in range(args.reps):             # a fresh table first, then the same keys with later sequence numbers
                h.submit_device(b, n, r * n, which=2)
            h.sync()
            wall_ms = (time.perf_counter() - t0) * 1e3 / args.reps
            ms, cnt = h.kernel_time_stats()
            h.set_timing(False)
            res, _ = h.finish()
            if want is None:
                want = res.alive_keys
            ok = "ok" if res.alive_keys == want else f"MISMATCH (want {want})"
            print(f"{preset} n=2^{args.n} variant={variant} wgs={wgs}: {ms[2]:.3f} ms avg over {cnt[2]} launches = "
                  f"{n / ms[2] / 1e6:.1f} G records/s (wall {wall_ms:.3f} ms per batch), alive={res.alive_keys} {ok}", flush=True)
    h.device_batch_free(b)
    h.close()
