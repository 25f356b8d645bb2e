"""Developer tool: time kernel variants on the GPU (run through gpurun).  Not part of the product."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import kafka_topic_analyzer_amd as kta
from oracle_c import Oracle


def time_scan(h, b, n, variant, wgs, iters=5):
    h.set_tuning(scan_workgroups=wgs, scan_variant=variant)
    h.set_timing(True)
    ms = []
    for it in range(iters + 2):
        h.submit_device(b, n, 0, which=1)
        ms.append(h.last_kernel_ms())
    h.set_timing(False)
    a = np.array(ms[2:])
    return a[:, 0].mean(), a[:, 1].mean()


def main():
    n = int(os.environ.get("N", 1 << 26))
    for preset, P in (("c4", 256), ("c2", 8), ("c1", 1)):
        spec, _ = kta.synth_preset(preset)
        h = kta.HipMetricHandler(P)
        b = h.device_batch_alloc(n)
        t0 = time.time(); h.synth_fill_device(spec, 0, n, b); h.sync(); print(preset, "gen s", time.time() - t0)
        # parity on the first 1M records
        m = 1 << 20
        cols = h.download_batch(b, m)
        o = Oracle(); o.run_soa(cols)
        for variant in (0, 1):
            h.reset(); h.set_tuning(scan_variant=variant)
            h.submit_device(b, m, 0, which=1)
            res, c = h.finish()
            ok = np.array_equal(c, o.counters(P)) and res.largest_message == o.get("largest_message") and \
                (res.smallest_message if res.any_live else 0) == o.get("smallest_message")
            print(preset, "variant", variant, "parity", ok, "min/max ts", res.min_ts_sec, res.max_ts_sec, o.earliest(), o.latest())
        for variant in (9, 0, 1):
            for wgs in (256 * 2, 256 * 4, 256 * 5, 256 * 8):
                s, f = time_scan(h, b, n, variant, wgs)
                print(f"{preset} P={P} n={n} variant={variant} wgs={wgs}: scan {s:.3f} ms fold {f:.3f} ms  "
                      f"{n * 20 / s / 1e6:.1f} GB/s  {n / s / 1e6:.2f} Grec/s", flush=True)
        h.device_batch_free(b); h.close()

    # alive pass: c3 (16 B keys, 10M distinct)
    na = int(os.environ.get("NA", 1 << 26))
    spec, _ = kta.synth_preset("c3")
    h = kta.HipMetricHandler(64, count_alive_keys=True)
    b = h.device_batch_alloc(na, na * 16)
    t0 = time.time(); kb = h.synth_fill_device(spec, 0, na, b); h.sync(); print("c3 gen s", time.time() - t0, "key bytes", kb)
    m = 1 << 20
    cols = h.download_batch(b, m, m * 16)
    o = Oracle(count_alive_keys=True); o.run_soa(cols)
    h.submit_device(b, m, 0, which=3)
    res, c = h.finish()
    print("alive parity", res.alive_keys, o.alive_keys(), np.array_equal(c, o.counters(64)))
    h.set_timing(True)
    for wgs in (1024, 2048, 4096, 8192):
        h.set_tuning(alive_workgroups=wgs)
        ms = []
        for it in range(4):
            h.submit_device(b, na, 0, which=2)
            ms.append(h.last_kernel_ms()[2])
        t = np.mean(ms[1:])
        print(f"c3 alive n={na} wgs={wgs}: {t:.3f} ms  {na / t / 1e6:.2f} Grec/s  {(na * 24 + kb) / t / 1e6:.1f} GB/s", flush=True)
    t0 = time.time(); res, _ = h.finish(); print("finish (count 2^32 table) s", time.time() - t0, res.alive_keys)
    h.close()


if __name__ == "__main__":
    main()
