#!/usr/bin/env python3
"""bench.py — the headline benchmark of the MI355X metric-accumulation path.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Metric (BASELINE.json): records/sec whole-node + achieved HBM GB/s on the 256-partition,
256 B-mean-record synthetic topic (config "c4").  One process per GPU; partition p lives on
rank p % N; every rank holds `--records-per-gpu` records (default 2^30: config 4's 1 B-record topic
on ONE GPU) of its partitions in HBM — weak scaling: per-GPU work is fixed as N grows.

A step = one pass of the metric-accumulation hot path (MessageMetrics::handle_message for every
record, src/metric.rs:207-252) over the rank's HBM-resident batch: metrics-scan kernel + partial
fold, then — for N > 1 — the exchange step: ONE all-reduce SUM of the counter vector prefix and
ONE all-reduce MAX of the four extrema (RCCL over xGMI).  Inputs are resident in HBM before the
timed region; nothing is skipped inside it.

Prints ONE JSON line (rank 0).  `roofline` describes the dominant kernel (kta_metrics_scan):
achieved = 20 algorithmic bytes/record x records per launch / mean kernel duration measured with
HIP events on the library's compute stream during the timed steps.  `cpu_baseline` is the C oracle
(oracle/kta_oracle.c, 1 thread — the reference is single threaded) on a bounded sample of the same
records; `alive_pass` reports the --count-alive-keys pass (config "c3" shape) the same way.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

METRIC = "records/sec whole-node + achieved HBM GB/s, 256-part 256B-mean synthetic topic"
HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6290 GB/s measured achievable
BYTES_PER_RECORD = 20        # partition i32 + key_len i32 + val_len i32 + ts_ms i64


TRAFFIC_SOURCE = "profiles/traffic.json (replayed: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of tools/profile_round.sh, " \
                 "not measured in this run; null when the kernel's source file changed since those passes)"


def _source_hash(rel):
    """sha256 of a kernel's source as it lies in the tree (what tools/make_traffic.py recorded): of the file, or — for
    a kernel that lives in several files, "a.hip+b.h+c.h" — of the files' digests one after the other."""
    import hashlib
    try:
        digests = [hashlib.sha256(open(os.path.join(ROOT, r), "rb").read()).hexdigest() for r in rel.split("+")]
        return (digests[0] if len(digests) == 1 else hashlib.sha256("".join(digests).encode()).hexdigest())[:16]
    except OSError:
        return None


def _traffic(kernels, records_per_launch):
    """HBM bytes per launch, summed over `kernels`, from the committed PMC passes (profiles/traffic.json).
    Replayed, not measured here: the counters need their own rocprofv3 passes.  None when the file lacks one of
    the kernels at this launch size, or when the kernel's source file is not the one the passes were taken with
    (every entry carries the file's hash): a changed kernel must not inherit an old number."""
    try:
        table = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
        total = 0
        for k in ([kernels] if isinstance(kernels, str) else kernels):
            e = table[k]
            if e["records_per_launch"] != records_per_launch:
                return None
            if not e.get("source_sha256_16") or e["source_sha256_16"] != _source_hash(e.get("source_file", "")):
                return None
            total += e["hbm_bytes_per_launch"]
        return total
    except Exception:
        return None


def _alive_checked(got, preset, distinct, n_upto):
    """The alive-key count a `-c` leg reports, held against the C oracle's count for the same records
    (tests/golden/bench_alive_counts.json, made on the CPU by tests/golden/make_bench_alive_counts.py from the generator's
    records in consumption order).  A leg whose shape is not in the file says so; a count that differs ends the run: a
    bench that times a wrong result times nothing."""
    key = f"{preset}:{distinct}:0:{n_upto}"
    try:
        want = json.load(open(os.path.join(ROOT, "tests", "golden", "bench_alive_counts.json")))["alive_keys"].get(key)
    except Exception:
        want = None
    if want is not None and int(got) != int(want):
        raise AssertionError(f"alive keys of {key}: the GPU says {int(got)}, the oracle {int(want)}")
    return {"alive_keys": int(got), "alive_keys_checked": want is not None}


def cpu_baseline_metrics(h, batch, n_sample, P, min_seconds=10.0):
    """Time the C oracle (1 thread) over the first n_sample records of the resident batch."""
    from oracle_c import Oracle
    cols = h.download_batch(batch, n_sample)
    passes, t_total = 0, 0.0
    while t_total < min_seconds and passes < 64:
        o = Oracle()
        t0 = time.perf_counter()
        o.run_soa(cols)
        t_total += time.perf_counter() - t0
        passes += 1
        o.close()
    out = {"value": n_sample * passes / t_total, "unit": "records/s", "cores": 1, "kind": "port",
           "sample": f"first {n_sample} records of the resident c4 shard x {passes} passes "
                     f"({t_total:.1f} s), oracle/kta_oracle.c MessageMetrics::handle_message loop, "
                     f"host has {os.cpu_count()} cores"}
    # context only: the same loop on every host core (each thread owns a contiguous slice of the sample and
    # its own maps; the reference itself is single threaded, so `value` above stays the 1-thread number)
    import threading
    T = max(1, min(os.cpu_count() or 1, 64))
    per = n_sample // T
    slices = [{k: v[t * per:(t + 1) * per] for k, v in cols.items()} for t in range(T)]
    oracles = [Oracle() for _ in range(T)]
    threads = [threading.Thread(target=oracles[t].run_soa, args=(slices[t],)) for t in range(T)]
    t0 = time.perf_counter()
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    t_all = time.perf_counter() - t0
    for o in oracles:
        o.close()
    out["all_cores_context"] = {"value": per * T / t_all, "unit": "records/s", "threads": T}
    return out


def boundary_per_message_report(kta, device, n_records=1 << 26, seconds=6.0):
    """What the drop-in costs per message.  /root/reference/src/kafka.rs:107-109 calls handle_message once per polled message;
    its replacement is kta_handle_message (copies partition, timestamp, lengths and — with -c — the key bytes into a pinned
    staging batch, submits the batch when it is full).  ONE host thread replays host-resident records through that entry
    in a native loop (kta_replay_messages: an indirect call per message, no ctypes), the GPU's work — H2D copy, both
    handlers — overlapped behind it; records/s, with the host-side breakdown the library keeps (kta_handle_message_stats).
    Two rows: config 4's records (metrics handler), config 3's with 16-byte keys and -c (both handlers)."""
    rows = {}
    for name, preset, alive, P in (("c4", "c4", False, 256), ("c3_alive_keys", "c3", True, 64)):
        spec, _ = kta.synth_preset(preset)
        cols = kta.synth_fill_host(spec, 0, n_records, with_keys=alive)
        h = kta.HipMetricHandler(P, count_alive_keys=alive, device=device, batch_capacity=1 << 22,
                                 key_bytes_capacity=(16 << 22) if alive else 0, n_staging=3)
        h.replay_messages(cols, min(n_records, 1 << 24))      # warm-up: staging slabs pinned, kernels loaded
        h.sync()
        h.reset()
        passes, t0 = 0, time.perf_counter()
        while passes < 64:
            h.replay_messages(cols)
            passes += 1
            if time.perf_counter() - t0 > seconds:
                break
        h.flush()
        t_host = time.perf_counter() - t0                     # the caller's thread is free again here
        h.sync()
        t_all = time.perf_counter() - t0
        st = h.handle_message_stats()
        res, _ = h.finish()
        assert res.overall_count == n_records * passes == st["messages"], "the per-message entry lost records"
        rows[name] = {"records": n_records * passes, "value": n_records * passes / t_all, "unit": "records/s",
                      "host_thread_records_per_s": n_records * passes / t_host, "ns_per_message": t_host / (n_records * passes) * 1e9,
                      "host_breakdown": {"staging_batches": st["batches"], "submit_ms_total": st["submit_ns"] / 1e6,
                                         "ring_wait_ms_total": st["ring_wait_ns"] / 1e6,
                                         "copy_into_pinned_columns_ms_total": (t_host * 1e9 - st["submit_ns"] - st["ring_wait_ns"]) / 1e6},
                      "bytes_per_record_over_pcie": 20 + (4 + 16 if alive else 0),
                      **(_alive_checked(res.alive_keys, "c3", 0, n_records) if alive else {})}
        h.close()
    return {"what": "kta_handle_message per record on ONE host thread (native replay loop, indirect call per message), GPU work "
                    "overlapped; reference boundary: MetricHandler::handle_message, kafka.rs:107-109", "host_cores_used": 1,
            "rows": rows, "note": "compare with cpu_baseline (the reference's handlers on one core): this is the same thread's "
                                   "cost when the handlers' work is the GPU's"}


def alive_pass_report(kta, device, steps, warmup, n_records, cpu_seconds, extras=True):
    """The --count-alive-keys pass (FNV + the reference's bit set) on the config-3 shape.  The batch is as large as
    the ABI takes with 16-byte keys (key_off is a u32: < 4 GiB of key bytes): the bit set's 512 MiB are streamed
    through LDS once per batch, whatever its size."""
    from oracle_c import Oracle
    spec, _ = kta.synth_preset("c3")
    h = kta.HipMetricHandler(64, count_alive_keys=True, device=device)
    b = h.device_batch_alloc(n_records, n_records * 16)
    kb = h.synth_fill_device(spec, 0, n_records, b)
    # (bit set state: batches are applied in submission order, the base sequence number is not looked at)
    for k in range(warmup):
        h.submit_device(b, n_records, k * n_records, which=2)
    h.sync()
    h.set_timing(True)
    t0 = time.perf_counter()
    for k in range(steps):
        h.submit_device(b, n_records, (warmup + k) * n_records, which=2)
    h.sync()
    wall = time.perf_counter() - t0
    avg_ms, cnt = h.kernel_time_stats()
    h.set_timing(False)
    res, _ = h.finish()
    # this kernel reads key_len, val_len, key_off and the key bytes (partition and timestamp belong
    # to the metrics scan): 12 B + len(key) per record
    algo_bytes = (4 + 4 + 4) * n_records + kb
    out = {"workload": f"c3 shape: 64 partitions, {n_records} records, 16 B keys, 10M distinct, 10% tombstones",
           "value": n_records * steps / wall, "unit": "records/s", "ms_per_step": wall / steps * 1e3,
           **_alive_checked(res.alive_keys, "c3", 0, n_records),
           "roofline": {"bound": "hbm", "kernel": "kta_alive_partition32 + kta_alive_apply",
                        "achieved": algo_bytes / (avg_ms[2] * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": algo_bytes / (avg_ms[2] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                        "bytes_per_launch": algo_bytes, "kernel_ms": avg_ms[2], "launches": int(cnt[2]),
                        "traffic": _traffic(["kta_alive_partition32", "kta_alive_apply"], n_records),
                        "traffic_source": TRAFFIC_SOURCE,
                        "note": "kernel_ms is the HIP-event time of the kernels of one batch (hash + partition, then "
                                "per-bucket merge + the bucket's region of the 512 MiB bit set streamed through LDS; "
                                "the fallback kernel returns at once); the 4 bytes per record of partitioned pairs "
                                "written and read back and the bit set's 2 x 512 MiB per batch are traffic, not "
                                "algorithmic bytes"}}
    # ---- both handlers over the same resident batch: the product's actual -c step (/root/reference/src/kafka.rs:107-109
    # calls every handler for every message) — reset-free, like a topic that keeps growing.  Twice: as the library runs it
    # (ONE pass: the partition kernel of the alive-key pass also does the metrics handler's work) and, in a second context
    # with kta_set_fuse(ctx, 0), as two passes (scan + fold, then the alive-key pass).
    def both(hh):
        for k in range(warmup):
            hh.submit_device(b, n_records, 0, which=3)
        hh.sync()
        hh.kernel_time_stats()
        hh.set_timing(True)
        t0 = time.perf_counter()
        for k in range(steps):
            hh.submit_device(b, n_records, 0, which=3)
        hh.sync()
        wall = time.perf_counter() - t0
        avg, cnt = hh.kernel_time_stats()
        hh.set_timing(False)
        return wall, [a if c else 0.0 for a, c in zip(avg, cnt)], cnt
    wall3, avg3, cnt3 = both(h)
    two = None
    if extras:
        h2 = kta.HipMetricHandler(64, count_alive_keys=True, device=device)
        h2.set_fuse(False)
        wall2, avg2, cnt2 = both(h2)
        r1, c1 = h.finish()
        r2, c2 = h2.finish()
        h2.close()
        # (h saw the alive-only steps before: the alive sets agree, the counters of the which=3 steps must be equal)
        assert r1.alive_keys == r2.alive_keys and (c1 == c2).all(), "the fused pass and the two passes disagree"
        two = {"value": n_records * steps / wall2, "kernel_ms": avg2[0] + avg2[1] + avg2[2], "scan_ms": avg2[0],
               "fold_ms": avg2[1], "alive_ms": avg2[2]}
    both_ms = avg3[0] + avg3[1] + avg3[2]
    # the scan reads partition, key_len, val_len, ts_ms (20 B), the alive pass key_len, val_len, key_off and the key
    # (12 B + key): 8 B of the two are the same columns — 40 B per record with 16-byte keys, read once by the fused pass
    algo3 = (20 + 12 - 8) * n_records + kb
    both = {"workload": out["workload"] + "; MessageMetrics + LogCompactionInMemoryMetrics per record (which=3)",
            "value": n_records * steps / wall3, "unit": "records/s", "ms_per_step": wall3 / steps * 1e3,
            "roofline": {"bound": "hbm", "kernel": "kta_alive_partition32<fused> + kta_fold_partials + kta_alive_apply",
                         "achieved": algo3 / (both_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": algo3 / (both_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "bytes_per_launch": algo3,
                         "kernel_ms": both_ms, "scan_launches": int(cnt3[0]), "launches": int(cnt3[2]),
                         "traffic": _traffic(["kta_alive_partition32_fused", "kta_alive_apply"], n_records),
                         "traffic_source": TRAFFIC_SOURCE,
                         "note": "algorithmic bytes = the union of the two handlers' columns (40 B per record with 16-byte "
                                 "keys), which the fused pass reads once; kernel_ms = HIP events around partition (with the "
                                 "metrics handler's sums) + fold + apply"},
            "two_passes": None if two is None else dict(
                two, frac=algo3 / (two["kernel_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                note="kta_set_fuse(ctx, 0): kta_metrics_scan + kta_fold_partials, then the alive-key pass (48 B per record touched: "
                     "key_len and val_len twice)")}
    m = min(n_records, 1 << 24)
    cols = h.download_batch(b, m, m * 16)
    passes, t_total = 0, 0.0
    while t_total < cpu_seconds and passes < 64:
        o = Oracle(count_alive_keys=True)
        t0 = time.perf_counter()
        o.L.kto_run_soa(None, o.lc, m, cols["partition"].ctypes.data, cols["key_len"].ctypes.data,
                        cols["val_len"].ctypes.data, cols["ts_ms"].ctypes.data, cols["key_off"].ctypes.data,
                        cols["key_bytes"].ctypes.data)
        t_total += time.perf_counter() - t0
        passes += 1
        o.close()
    out["cpu_baseline"] = {"value": m * passes / t_total, "unit": "records/s", "cores": 1, "kind": "port",
                           "sample": f"first {m} records x {passes} passes ({t_total:.1f} s), "
                                     "LogCompactionInMemoryMetrics::handle_message loop of oracle/kta_oracle.c"}
    h.device_batch_free(b)
    h.close()
    return out, both


def alive_table_report(kta, device, steps, n_records, extras=True):
    """What a rank of a partition-sharded -c run executes: the sequence-numbered table (32 GiB), batches with a seq column
    (the global consumption index of every record), config 3's key law — 10 M distinct keys, about what one of eight ranks of
    config 5 sees (key-affine partitions: 12.5 M).  Every step is a NEW batch of the topic (records and sequence numbers that
    follow the last one's): resubmitting a batch would find its own entries in the table and write nothing.  Three contexts over
    the same resident batches: the alive-key pass alone (which = 2), both handlers as the library runs them — ONE pass,
    kta_alive_partition48<seq, fused> (which = 3) — and, with kta_set_fuse(ctx, 0), as two (kta_metrics_scan, then the pass).
    Returns (alive_pass_table, both_handlers_table)."""
    spec, _ = kta.synth_preset("c3")
    h = kta.HipMetricHandler(64, count_alive_keys=True, device=device, alive_table=True)
    batches = []
    for k in range(steps + 1):
        b = h.device_batch_alloc(n_records, n_records * 16, with_seq=True)
        kb = h.synth_fill_device(spec, k * n_records, n_records, b)
        batches.append(b)

    def run(hh, which):
        hh.submit_device(batches[0], n_records, 0, which=which)   # warm-up: the first batch also writes 10 M new slots
        hh.sync()
        hh.kernel_time_stats()
        hh.set_timing(True)
        t0 = time.perf_counter()
        for k in range(steps):
            hh.submit_device(batches[k + 1], n_records, 0, which=which)
        hh.sync()
        wall = time.perf_counter() - t0
        avg, cnt = hh.kernel_time_stats()
        hh.set_timing(False)
        return wall, [a if c else 0.0 for a, c in zip(avg, cnt)], cnt

    wall, avg_ms, cnt = run(h, 2)
    res, _ = h.finish()
    # (a sharded rank's batches carry the global consumption index of every record: + 8 B per record, SURVEY §8e — read
    # by pass 1, whose consumer waves check that the column ascends inside the batch, and for the survivors)
    algo = (4 + 4 + 4 + 8) * n_records + kb
    what = f"c3 law, table state, seq column: {steps} consecutive batches of {n_records} records, 16 B keys, 10M distinct, 10% tombstones"
    rep = {"workload": what + " (the partitioned pass: kta_alive_partition48<seq> + kta_alive_apply<table>)",
           "value": n_records * steps / wall, "unit": "records/s", "ms_per_step": wall / steps * 1e3,
           **_alive_checked(res.alive_keys, "c3", 0, n_records * (steps + 1)),
           "roofline": {"bound": "hbm", "kernel": "kta_alive_partition48 + kta_alive_apply (table state)",
                        "achieved": algo / (avg_ms[2] * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": algo / (avg_ms[2] * 1e-3) / 1e9 / HBM_PEAK_GBS, "bytes_per_launch": algo,
                        "kernel_ms": avg_ms[2], "launches": int(cnt[2]),
                        "traffic": _traffic(["kta_alive_partition48", "kta_alive_apply_table"], n_records),
                        "traffic_source": TRAFFIC_SOURCE,
                        "note": "algorithmic bytes = alive_pass's 12 B + key per record + the 8-byte seq column of a sharded "
                                "rank's batches (SURVEY 8e): 36 B per record with 16-byte keys; 6-byte pairs since round 6"}}
    # ---- both handlers: 20 + 12 - 8 + 8 (seq) + key = 48 B per record with 16-byte keys, read once by the fused pass
    h3 = kta.HipMetricHandler(64, count_alive_keys=True, device=device, alive_table=True)
    wall3, avg3, cnt3 = run(h3, 3)
    r3, c3 = h3.finish()
    info3 = h3.alive_pass_info()
    h3.close()
    assert r3.alive_keys == res.alive_keys and info3["fused"] == steps + 1, "the fused table pass did not run, or disagrees"
    two = None
    if extras:
        h2 = kta.HipMetricHandler(64, count_alive_keys=True, device=device, alive_table=True)
        h2.set_fuse(False)
        wall2, avg2, cnt2 = run(h2, 3)
        r2, c2 = h2.finish()
        h2.close()
        assert r2.alive_keys == r3.alive_keys and (c2 == c3).all(), "the fused pass and the two passes disagree (table state)"
        two = {"value": n_records * steps / wall2, "kernel_ms": avg2[0] + avg2[1] + avg2[2], "scan_ms": avg2[0],
               "fold_ms": avg2[1], "alive_ms": avg2[2]}
    algo3 = (20 + 12 - 8 + 8) * n_records + kb
    both_ms = avg3[0] + avg3[1] + avg3[2]
    both = {"workload": what + "; MessageMetrics + LogCompactionInMemoryMetrics per record (which=3): a sharded -c rank's step",
            "value": n_records * steps / wall3, "unit": "records/s", "ms_per_step": wall3 / steps * 1e3,
            "alive_keys": int(r3.alive_keys), "alive_keys_checked": rep["alive_keys_checked"],
            "roofline": {"bound": "hbm", "kernel": "kta_alive_partition48<seq, fused> + kta_fold_partials + kta_alive_apply (table state)",
                         "achieved": algo3 / (both_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": algo3 / (both_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "bytes_per_launch": algo3,
                         "kernel_ms": both_ms, "scan_launches": int(cnt3[0]), "launches": int(cnt3[2]),
                         "traffic": _traffic(["kta_alive_partition48_fused", "kta_alive_apply_table"], n_records),
                         "traffic_source": TRAFFIC_SOURCE,
                         "note": "algorithmic bytes = the union of the two handlers' columns + the seq column (48 B per record with "
                                 "16-byte keys), which the fused pass reads once; kernel_ms = HIP events around partition (with the "
                                 "metrics handler's sums) + fold + apply + the pool's direct path"},
            "two_passes": None if two is None else dict(
                two, frac=algo3 / (two["kernel_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                note="kta_set_fuse(ctx, 0): kta_metrics_scan + kta_fold_partials, then the alive-key pass (56 B per record touched: "
                     "key_len and val_len twice)")}
    for b in batches:
        h.device_batch_free(b)
    h.close()
    return rep, both


def alive_hot_key_report(kta, device, n_records):
    """Adversarial shapes of the alive-key pass (bit set state): what a compacted topic with a dominant key costs.  One
    key / 40 keys over the whole batch: most records die in the partition kernel's guard (one record per hash and wave
    instruction survives), the rest overflow their segments into the pool (handed out in chunks, every block tagged with its
    bucket) and the bucket is resolved by kta_alive_fallback (exact; it reads the tags and its own blocks)."""
    rows = []
    for distinct in (1, 40):
        spec, _ = kta.synth_preset("c3")
        spec.n_distinct_keys = distinct
        h = kta.HipMetricHandler(64, count_alive_keys=True, device=device)
        b = h.device_batch_alloc(n_records, n_records * 16)
        h.synth_fill_device(spec, 0, n_records, b)
        h.submit_device(b, n_records, 0, which=2)
        h.sync()
        h.kernel_time_stats()
        h.set_timing(True)
        for k in range(3):
            h.submit_device(b, n_records, 0, which=2)
        h.sync()
        avg_ms, cnt = h.kernel_time_stats()
        h.set_timing(False)
        res, _ = h.finish()
        rows.append({"distinct_keys": distinct, "records": n_records, "kernel_ms": avg_ms[2], "launches": int(cnt[2]),
                     "records_per_s": n_records / (avg_ms[2] * 1e-3), **_alive_checked(res.alive_keys, "c3", distinct, n_records)})
        h.device_batch_free(b)
        h.close()
    # mostly unique keys (config 5's law: 100 M distinct, 50 % tombstones) on ONE GPU in the bit set state: a bucket holds
    # ten times the distinct slots of pass 2's LDS table, so every bucket is applied in instalments (careful mode, groups of
    # segments sized from the fills so that a group always fits: kta_alive.hip, pick_group_size)
    spec, _ = kta.synth_preset("c5")
    n5 = 15 << 24
    h = kta.HipMetricHandler(256, count_alive_keys=True, device=device)
    b = h.device_batch_alloc(n5, n5 * 16)
    h.synth_fill_device(spec, 0, n5, b)
    per_launch = []
    h.set_timing(True)
    for k in range(4):
        h.submit_device(b, n5, 0, which=2)
        h.sync()
        per_launch.append(h.kernel_time_stats()[0][2])
    h.set_timing(False)
    res, _ = h.finish()
    h.device_batch_free(b)
    h.close()
    unique = {"workload": f"c5 law (100 M distinct 16 B keys, 50 % tombstones), bit set state, {n5} records per batch, 4 batches",
              "kernel_ms_per_batch": per_launch, "records_per_s_last": n5 / (per_launch[-1] * 1e-3), **_alive_checked(res.alive_keys, "c5", 0, n5),
              "note": "a bucket holds six times the slots of pass 2's LDS table: it is applied in eight slot-range passes over all its "
                      "pairs, each sweeping an eighth of the bucket's 512 KiB of bit set (kta_alive_apply<.., RANGES>, round 6); round 5 "
                      "applied it in about 32 instalments in segment order, each a sweep of the whole 512 KiB (7.9-8.1 ms)"}
    # a compacted topic of MORE keys than pass 2's tables hold in one piece (20 M distinct: 19.5 k slots per bucket against 16 k):
    # two slot-range passes per bucket (rounds 4-5: an instalment after every group of segments, 7.6 ms)
    spec, _ = kta.synth_preset("c3")
    spec.n_distinct_keys = 20_000_000
    h = kta.HipMetricHandler(64, count_alive_keys=True, device=device)
    b = h.device_batch_alloc(n5, n5 * 16)
    h.synth_fill_device(spec, 0, n5, b)
    h.submit_device(b, n5, 0, which=2)
    h.sync()
    h.kernel_time_stats()
    h.set_timing(True)
    for k in range(3):
        h.submit_device(b, n5, 0, which=2)
    h.sync()
    avg_ms, cnt = h.kernel_time_stats()
    h.set_timing(False)
    res, _ = h.finish()
    info = h.alive_pass_info()
    h.device_batch_free(b)
    h.close()
    many = {"workload": f"c3 shape with 20 M distinct keys, bit set state, {n5} records per batch", "kernel_ms": avg_ms[2], "launches": int(cnt[2]),
            "records_per_s": n5 / (avg_ms[2] * 1e-3), "buckets_to_the_fallback_kernel": info["failed_buckets"],
            **_alive_checked(res.alive_keys, "c3", 20_000_000, n5)}
    return {"workload": f"c3 shape with 1 / 40 distinct keys over {n_records} records (bit set state)", "rows": rows,
            "mostly_unique_keys": unique, "many_keys": many}


def _recompress_batches(lib, raw, codec):
    """Re-encode uncompressed v2 batches as gzip (codec 1: zlib level 6) or zstd (codec 4: libzstd level 3
    through pyarrow) batches: records section compressed, header lengths and CRC-32C redone."""
    import zlib
    if codec == 4:
        import pyarrow as pa
        zstd = pa.Codec("zstd", compression_level=3)
    out, pos = bytearray(), 0
    while pos + 61 <= len(raw):
        total = 12 + int.from_bytes(raw[pos + 8:pos + 12], "big")
        b = raw[pos:pos + total]
        if codec == 1:
            co = zlib.compressobj(6, zlib.DEFLATED, 15 + 16)
            comp = co.compress(b[61:]) + co.flush()
        else:
            comp = zstd.compress(b[61:], asbytes=True)
        attrs = int.from_bytes(b[21:23], "big") | codec
        after_crc = attrs.to_bytes(2, "big") + b[23:61] + comp
        crc = lib.kta_crc32c_host(after_crc, len(after_crc))
        out += b[0:8] + (49 + len(comp)).to_bytes(4, "big") + b[12:17] + crc.to_bytes(4, "big") + after_crc
        pos += total
    return bytes(out)


def _gzip_batches(lib, raw):
    return _recompress_batches(lib, raw, 1)


def kafka_decode_report(kta, device, steps, warmup, n_records, cpu_seconds):
    """The step before the hot path: raw Kafka v2 record batches -> columns, on the GPU."""
    import numpy as np
    from kafka_topic_analyzer_amd import _native as N
    from oracle_c import kafka_decode
    lib = N.load()
    spec, _ = kta.synth_preset("c4")
    rpb = 60  # ~16 KiB batches: the producer default batch.size
    ln = C.c_uint64()
    lib.kta_kafka_encode_synth_host(C.byref(spec), 0, n_records, rpb, None, 0, C.byref(ln))
    buf = np.zeros(ln.value + 64, np.uint8)
    lib.kta_kafka_encode_synth_host(C.byref(spec), 0, n_records, rpb, buf.ctypes.data, ln.value, C.byref(ln))
    nb_cap = n_records // rpb + 2
    descs = (N.KtaKafkaBatchDesc * nb_cap)()
    st = N.KtaKafkaIndexStats()
    t0 = time.perf_counter()
    rc = lib.kta_kafka_index_host(buf.ctypes.data_as(C.c_char_p), ln.value, 0, 0, 0, 0, descs, nb_cap, C.byref(st))
    t_index = time.perf_counter() - t0
    assert rc == 0 and st.n_records == n_records
    h = kta.HipMetricHandler(256, device=device)
    d_blob = h.device_batch_alloc((ln.value + 3) // 4 + 32)
    h._check(lib.kta_copy_to_device(h._ctx, d_blob.partition, buf.ctypes.data, (ln.value + 63) // 64 * 64))
    out = h.device_batch_alloc(n_records, 16)  # key_off wanted: keys stay in the blob (zero-copy)
    kb, bad = C.c_uint64(), C.c_uint64()

    def step():
        h._check(lib.kta_kafka_decode_device(h._ctx, d_blob.partition, ln.value, descs, st.n_batches, n_records,
                                             C.byref(out), None, None))
    for _ in range(warmup):
        step()
    h.sync()
    h.set_timing(True)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    h.sync()
    wall = time.perf_counter() - t0
    a, c = (C.c_float * 2)(), (C.c_uint64 * 2)()
    h._check(lib.kta_kafka_time_stats(h._ctx, C.byref(a), C.byref(c)))
    h.set_timing(False)
    # parity inside the bench: decoded columns == the generator's columns
    cols = h.download_batch(out, min(n_records, 1 << 18))
    ref = kta.synth_fill_host(spec, 0, len(cols["partition"]))
    assert np.array_equal(cols["key_len"], ref["key_len"]) and np.array_equal(cols["val_len"], ref["val_len"])
    assert np.array_equal(cols["ts_ms"], ref["ts_ms"])
    blob = buf[:ln.value].tobytes()
    passes, t_total = 0, 0.0
    sample = blob[:descs[min(st.n_batches, 4000) - 1].byte_off + descs[min(st.n_batches, 4000) - 1].batch_bytes]
    n_sample = 0
    while t_total < cpu_seconds / 2 and passes < 64:
        t0 = time.perf_counter()
        ccols, _ = kafka_decode(sample, 0, transcode_zstd=False)
        t_total += time.perf_counter() - t0
        n_sample = len(ccols["partition"])
        passes += 1
    rep = {"workload": f"c4 records as Kafka v2 record batches: {n_records} records, {st.n_batches} batches of {rpb} "
                       f"(~16 KiB), {ln.value} bytes of raw log resident in HBM; keys zero-copy",
           "value": n_records * steps / wall, "unit": "records/s", "ms_per_step": wall / steps * 1e3,
           "raw_log_GBps": ln.value * steps / wall / 1e9,
           "roofline": {"bound": "hbm", "kernel": "kafka_decode_coop", "achieved": ln.value / (a[1] * 1e-3) / 1e9,
                        "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ln.value / (a[1] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                        "bytes_per_launch": ln.value, "kernel_ms": a[1], "launches": int(c[1]),
                        "traffic": _traffic("kafka_decode_coop", n_records), "traffic_source": TRAFFIC_SOURCE,
                        "note": "algorithmic bytes = the raw log (every window is streamed through LDS)"},
           "host_index": {"ms": t_index * 1e3, "GBps": ln.value / t_index / 1e9},
           "cpu_baseline": {"value": n_sample * passes / t_total, "unit": "records/s", "cores": 1, "kind": "port",
                            "sample": f"first {n_sample} records ({len(sample)} bytes) x {passes} passes "
                                      f"({t_total:.1f} s), oracle/kta_kafka_oracle.c sequential decoder"}}
    h.device_batch_free(out)
    h.device_batch_free(d_blob)
    # the same kernel family at other batch sizes (the geometry is chosen per call from the mean batch size):
    # one row per size, each with its own launch time — a profile's per-kernel average mixes them
    rep["by_batch_size"] = []
    # (the last two rows: few, very large batches — 500 / 125 batches of 4 000 / 16 000 records — are bound by the serial chain
    # of ONE batch, a dependent LDS read per record whatever the geometry: rows for the record, not a target)
    for rpb2, label in ((8, "~2 KiB"), (60, "~16 KiB"), (500, "~134 KiB"), (4000, "~1 MiB"), (16000, "~4 MiB")):
        n2 = min(n_records, 2_000_000)
        lib.kta_kafka_encode_synth_host(C.byref(spec), 0, n2, rpb2, None, 0, C.byref(ln))
        buf2 = np.zeros(ln.value + 64, np.uint8)
        lib.kta_kafka_encode_synth_host(C.byref(spec), 0, n2, rpb2, buf2.ctypes.data, ln.value, C.byref(ln))
        cap2 = n2 // rpb2 + 2
        descs2 = (N.KtaKafkaBatchDesc * cap2)()
        st2 = N.KtaKafkaIndexStats()
        assert lib.kta_kafka_index_host(buf2.ctypes.data_as(C.c_char_p), ln.value, 0, 0, 0, 0, descs2, cap2, C.byref(st2)) == 0
        blob2 = h.device_batch_alloc((ln.value + 3) // 4 + 32)
        h._check(lib.kta_copy_to_device(h._ctx, blob2.partition, buf2.ctypes.data, (ln.value + 63) // 64 * 64))
        out2 = h.device_batch_alloc(n2, 16)
        for _ in range(2):
            h._check(lib.kta_kafka_decode_device(h._ctx, blob2.partition, ln.value, descs2, st2.n_batches, n2, C.byref(out2), None, None))
        h.sync()
        h._check(lib.kta_kafka_time_stats(h._ctx, C.byref(a), C.byref(c)))      # drain
        h.set_timing(True)
        for _ in range(5):
            h._check(lib.kta_kafka_decode_device(h._ctx, blob2.partition, ln.value, descs2, st2.n_batches, n2, C.byref(out2), None, None))
        h.sync()
        h._check(lib.kta_kafka_time_stats(h._ctx, C.byref(a), C.byref(c)))
        h.set_timing(False)
        rep["by_batch_size"].append({"batch": label, "records_per_batch": rpb2, "batches": int(st2.n_batches), "records": n2,
                                     "raw_log_bytes": int(ln.value), "kernel_ms": a[1], "launches": int(c[1]),
                                     "achieved_GBps": ln.value / (a[1] * 1e-3) / 1e9,
                                     "frac": ln.value / (a[1] * 1e-3) / 1e9 / HBM_PEAK_GBS})
        h.device_batch_free(out2)
        h.device_batch_free(blob2)
    # compressed record sets: inflate + decode (wall time of the device work, keys zero-copy)
    rep["compressed"] = {}
    nc = min(n_records, 1_000_000)
    codecs = [(2, "snappy"), (3, "lz4"), (1, "gzip")]
    try:
        import pyarrow  # noqa: F401  (libzstd for the zstd sample)
        codecs.append((4, "zstd"))
    except ImportError:
        pass
    for codec, name in codecs:
        enc = codec if codec in (2, 3) else 0x100   # gzip / zstd: the real libraries over the uncompressed, patterned batches
        lib.kta_kafka_encode_synth_host_ex(C.byref(spec), 0, nc, rpb, enc, None, 0, C.byref(ln))
        cbuf = np.zeros(ln.value + 128, np.uint8)
        lib.kta_kafka_encode_synth_host_ex(C.byref(spec), 0, nc, rpb, enc, cbuf.ctypes.data, ln.value, C.byref(ln))
        if codec in (1, 4):
            gz = _recompress_batches(lib, cbuf[:ln.value].tobytes(), codec)
            cbuf = np.zeros(len(gz) + 128, np.uint8)
            cbuf[:len(gz)] = np.frombuffer(gz, np.uint8)
            ln.value = len(gz)
        inflate_at = (ln.value + 127) & ~63
        cdescs = (N.KtaKafkaBatchDesc * (nc // rpb + 2))()
        cst = N.KtaKafkaIndexStats()
        rc = lib.kta_kafka_index_host(cbuf.ctypes.data_as(C.c_char_p), ln.value, 0, 0, 0, inflate_at, cdescs,
                                      nc // rpb + 2, C.byref(cst))
        assert rc == 0 and cst.n_records == nc
        cblob = h.device_batch_alloc((inflate_at + cst.inflate_bytes + 256) // 4 + 1)
        h._check(lib.kta_copy_to_device(h._ctx, cblob.partition, cbuf.ctypes.data, (ln.value + 63) // 64 * 64))
        cout = h.device_batch_alloc(nc, 16)
        best = 1e9
        for _ in range(4):
            h.sync()
            t0 = time.perf_counter()
            h._check(lib.kta_kafka_decode_device(h._ctx, cblob.partition, ln.value, cdescs, cst.n_batches, nc,
                                                 C.byref(cout), None, None))
            h.sync()
            best = min(best, time.perf_counter() - t0)
        ccols = h.download_batch(cout, 4096)
        assert np.array_equal(ccols["val_len"], ref["val_len"][:4096]) and np.array_equal(ccols["ts_ms"], ref["ts_ms"][:4096])
        rep["compressed"][name] = {"records": nc, "compressed_bytes": int(ln.value), "ms": best * 1e3,
                                   "records_per_s": nc / best, "compressed_GBps": ln.value / best / 1e9}
        h.device_batch_free(cout)
        h.device_batch_free(cblob)
    h.close()
    return rep


def host_fed_report(kta, device, batches=24, log2_batch=22):
    """PCIe-inclusive: kta_batch_acquire / kta_batch_submit over the pinned staging ring (the path a Rust
    MetricHandler shim feeds), the batches' contents left in place between submits so that the host does no work
    but the submit itself: staging slab -> one H2D copy -> scan + fold (+ the alive pass).  Never `value`."""
    from kafka_topic_analyzer_amd import _native as N
    lib = N.load()
    n = 1 << log2_batch
    out = {}
    for alive in (False, True):
        sp, _ = kta.synth_preset("c3" if alive else "c4")
        h = kta.HipMetricHandler(int(sp.n_partitions), count_alive_keys=alive, device=device, batch_capacity=n,
                                 key_bytes_capacity=16 * n)
        cols = kta.synth_fill_host(sp, 0, n, with_keys=alive)
        stages = 2
        for k in range(batches + stages):
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
        assert res.overall_count == (batches + stages) * n
        per = (20 + (4 + 16 if alive else 0)) * n
        out["count_alive_keys" if alive else "metrics"] = {
            "workload": ("c3 shape, 16 B keys, -c" if alive else "c4") + f": {batches} staging batches of 2^{log2_batch} records",
            "records": batches * n, "bytes_over_pcie": batches * per, "ms": dt * 1e3,
            "records_per_s": batches * n / dt, "GBps_over_pcie": batches * per / dt / 1e9}
        h.close()
    return out


def raw_log_e2e_report(kta, device, n_records=4_000_000, passes=4):
    """PCIe-inclusive, the step before the path included: raw Kafka v2 record batches (what a Fetch response or a
    broker *.log segment holds) in the pinned blobs of kta_kafka_blob_acquire / submit -> header index on the host
    -> H2D -> decode kernel -> scan + fold (-> alive pass).  The blobs are filled once and resubmitted, so the host
    does what a fetcher that writes straight into the blobs would do: index and submit.  Never `value`."""
    import numpy as np
    from kafka_topic_analyzer_amd import _native as N
    lib = N.load()
    spec, _ = kta.synth_preset("c4")
    rpb = 60
    ln = C.c_uint64()
    lib.kta_kafka_encode_synth_host(C.byref(spec), 0, n_records, rpb, None, 0, C.byref(ln))
    buf = np.zeros(ln.value + 64, np.uint8)
    lib.kta_kafka_encode_synth_host(C.byref(spec), 0, n_records, rpb, buf.ctypes.data, ln.value, C.byref(ln))
    out = {}

    def leg(log, log_len, alive, n_passes, codec=None):
        """The log cut into the ring's three pinned blobs once, then the blobs resubmitted n_passes times; returns the row."""
        h = kta.HipMetricHandler(256, count_alive_keys=alive, device=device)
        h._check(lib.kta_kafka_configure(h._ctx, 0, 3))
        sizes, recs = [], []
        total_bytes = total_recs = 0
        stages, at = 3, 0
        t0 = None
        for k in range(stages + n_passes * stages):
            if k == stages:
                h.sync()
                t0 = time.perf_counter()
            ptr, cap = C.c_void_p(), C.c_uint64()
            h._check(lib.kta_kafka_blob_acquire(h._ctx, C.byref(ptr), C.byref(cap)))
            if k < stages:      # fill each pinned blob once with the next stretch of the log (a short log: all of it, again)
                if at >= log_len:
                    at = 0
                take = min(cap.value, log_len - at)
                C.memmove(ptr, log.ctypes.data + at, take)
                sizes.append(take)
            st = N.KtaKafkaIndexStats()
            h._check(lib.kta_kafka_blob_submit(h._ctx, sizes[k % stages], k % 256, C.byref(st)))
            if k < stages:
                sizes[k] = st.bytes_consumed          # whole batches only; the next blob starts at the tail
                recs.append(st.n_records)
                at += st.bytes_consumed
            else:
                total_bytes += st.bytes_consumed
                total_recs += st.n_records
        h.sync()
        dt = time.perf_counter() - t0
        res, _ = h.finish(allow_bad_partition=True)
        assert res.overall_count == sum(recs) * (n_passes + 1) and res.bad_partition_records == 0
        h.close()
        what = "uncompressed" if codec is None else codec
        return {"workload": f"c4 records as v2 record batches of {rpb} (~16 KiB), {what}; {n_passes * stages} pinned blobs of "
                            f"{sizes[0]} bytes resubmitted" + (", -c (keys zero-copy from the blob)" if alive else ""),
                "records": total_recs, "raw_log_bytes": total_bytes, "ms": dt * 1e3,
                "records_per_s": total_recs / dt, "raw_log_GBps": total_bytes / dt / 1e9}

    # (the first leg of the process pays what the second does not — the blob ring's pinned pages touched for the first time, the
    # decode kernels loaded: round 5's line had `metrics` at 46 GB/s behind `count_alive_keys` at 52 on the same blobs — so an
    # untimed leg goes first and the order of the timed ones decides nothing)
    leg(buf, ln.value, False, 1)
    out["metrics"] = leg(buf, ln.value, False, passes)
    out["count_alive_keys"] = leg(buf, ln.value, True, passes)
    # The same pipeline fed COMPRESSED logs (1 M records per codec; `raw_log_GBps` is then GB/s of compressed log over the link):
    # where the inflate kernel keeps up with PCIe the row sits at the link's rate, where it does not the kernel is the bound.
    nc = min(n_records, 1_000_000)
    codecs = [(2, "snappy"), (3, "lz4"), (1, "gzip")]
    try:
        import pyarrow  # noqa: F401  (libzstd for the zstd sample)
        codecs.append((4, "zstd"))
    except ImportError:
        pass
    out["compressed"] = {}
    for codec, name in codecs:
        enc = codec if codec in (2, 3) else 0x100   # gzip / zstd: the real libraries over the uncompressed, patterned batches
        cl = C.c_uint64()
        lib.kta_kafka_encode_synth_host_ex(C.byref(spec), 0, nc, rpb, enc, None, 0, C.byref(cl))
        cbuf = np.zeros(cl.value + 128, np.uint8)
        lib.kta_kafka_encode_synth_host_ex(C.byref(spec), 0, nc, rpb, enc, cbuf.ctypes.data, cl.value, C.byref(cl))
        if codec in (1, 4):
            z = _recompress_batches(lib, cbuf[:cl.value].tobytes(), codec)
            cbuf = np.zeros(len(z) + 128, np.uint8)
            cbuf[:len(z)] = np.frombuffer(z, np.uint8)
            cl.value = len(z)
        leg(cbuf, cl.value, False, 1, name)                       # (the codec's kernels loaded, the inflate area allocated)
        # (a compressed blob is a third of an uncompressed one: three times the passes for a timed stretch of the same length —
        # with 12 blobs, 25 ms, the pipeline's fill and drain were a tenth of the row: tools/e2e_probe.py)
        out["compressed"][name] = leg(cbuf, cl.value, False, 3 * passes, name)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--preroll", type=int, default=150,
                    help="untimed steps before the warmup steps (GPU clock ramp; ~0.5 s)")
    ap.add_argument("--records-per-gpu", type=int, default=1 << 30,
                    help="records resident per GPU (default 2^30 = 1.07 B: BASELINE config 4's 1 B-record "
                         "256-partition topic fits one MI355X: 21.5 GB of 288 GB)")
    ap.add_argument("--config", choices=["c4", "c5"], default="c4",
                    help="c4 (default, the headline): 256-partition topic, metrics handler; c5: config 5's key law "
                         "(100 M distinct 16-byte keys, 50 %% tombstones, 256 partitions) with --count-alive-keys: both "
                         "handlers per step and, for N > 1 (or KTA_BENCH_FORCE_COLLECTIVES=1), the whole kta_exchange "
                         "(hash-range exchange of the alive entries + the counter all-reduces)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak (default): --records-per-gpu on every rank; strong: that many records in TOTAL, "
                         "1/N of them per rank (config 4 as stated: 1 B records over 8 GPUs)")
    ap.add_argument("--part-mode", choices=["random", "runs"], default="random",
                    help="partition interleaving of the synthetic topic (random = worst case)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-alive", action="store_true", help="skip the --count-alive-keys sub-benchmark")
    ap.add_argument("--alive-records", type=int, default=15 << 24,
                    help="records per batch of the alive-key pass (default 15 x 2^24: 3.75 GiB of 16-byte keys, the most a "
                         "batch's u32 key offsets address)")
    ap.add_argument("--no-alive-extras", action="store_true",
                    help="skip the legs of the alive rows that launch the same kernels at other sizes or on other data (the two-pass "
                         "comparison of both_handlers, alive_pass_hot_key): a counter pass then sees every kernel at ONE launch size")
    ap.add_argument("--no-decode", action="store_true", help="skip the Kafka record-batch decode sub-benchmark")
    ap.add_argument("--no-hostfed", action="store_true", help="skip the PCIe-inclusive legs (host_fed, raw_log_e2e)")
    ap.add_argument("--decode-records", type=int, default=4_000_000)
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    args = ap.parse_args()

    # stdout carries exactly ONE line (the JSON, rank 0): native libraries (RCCL prints a version banner
    # on stdout when the first communicator is created) are pointed at stderr at the fd level.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    # (multi-process GPU work on this pool needs dmabuf IPC: without it RCCL fails with hipIpcGetMemHandle: invalid argument)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch  # first: its bundled HIP runtime must be the one this process loads
    import torch.distributed as dist
    import numpy as np
    import kafka_topic_analyzer_amd as kta
    from kafka_topic_analyzer_amd import _native as N
    from kafka_topic_analyzer_amd import distributed as D

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: libkta_hip has no CPU fallback")
    # KTA_BENCH_SHARE_DEVICE=1 (tests on a 1-GPU box, with KTA_RCCL_LIBRARY pointing at the test double of RCCL — real RCCL
    # refuses two ranks on one device): every rank on device 0; the line then carries "shared_device": true
    share_device = os.environ.get("KTA_BENCH_SHARE_DEVICE") == "1"
    if share_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    # KTA_BENCH_FORCE_COLLECTIVES=1: run the exchange step even with one rank (exercises the RCCL path
    # on a 1-GPU box; the line then carries "forced_collectives": true and is not a headline number)
    force_coll = os.environ.get("KTA_BENCH_FORCE_COLLECTIVES") == "1" and "RANK" in os.environ
    exchange = world > 1 or force_coll
    if exchange:
        # torch.distributed is the CONTROL plane only — the rendezvous (RCCL's 128-byte id), the barriers around the timed
        # region, the max over ranks of the elapsed time — and runs over gloo on CPU tensors: every rank process holds ONE
        # RCCL, the library's own (csrc/kta_comm.hip), which carries every byte of the data path.  (Rounds 1-5 initialised
        # torch's bundled RCCL as well: two RCCLs in a process, and a path no 2-rank run had ever executed.)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if os.environ["MASTER_ADDR"] in ("127.0.0.1", "localhost"):
            os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")   # one node: no need to resolve the host's own name
        dist.init_process_group(backend="gloo")

    P = 256
    c5 = args.config == "c5"
    n = args.records_per_gpu
    if c5:
        n = min(n, 15 << 24)                               # 16-byte keys: a batch's u32 key offsets address < 4 GiB
    if args.scaling == "strong":
        n //= world
    spec, _ = kta.synth_preset(args.config)
    spec = D.shard_spec(spec, rank, world)                 # partition p -> rank p % world
    if args.part_mode == "runs":
        spec.part_mode, spec.part_run_len = N.KTA_PART_RUNS, 500
    # c5: a rank of a sharded -c run keeps the sequence-numbered table (global sequence numbers across ranks)
    h = kta.HipMetricHandler(P, count_alive_keys=c5, device=local_rank, alive_table=c5)
    batch = h.device_batch_alloc(n, n * 16 if c5 else 0)
    h.synth_fill_device(spec, rank * n, n, batch)          # rank-disjoint record index ranges
    h.sync()

    n_sum = D.sum_prefix_len(P)

    if exchange:
        # The exchange is the library's own (csrc/kta_comm.hip: RCCL bound at run time, one grouped launch of
        # all-reduce SUM + all-reduce MAX on the context's compute stream, stream-ordered behind the fold
        # kernel: a step has no host synchronisation).  torch.distributed only carries the rendezvous (the
        # 128-byte RCCL id), the barriers and the max-over-ranks of the elapsed time.
        uid = torch.zeros(N.KTA_COMM_ID_BYTES, dtype=torch.uint8)
        if rank == 0:
            uid.copy_(torch.frombuffer(bytearray(kta.HipMetricHandler.comm_unique_id()), dtype=torch.uint8))
        dist.broadcast(uid, 0)
        if world == 1:
            os.environ["KTA_COMM_FORCE_RCCL"] = "1"        # forced mode: a real one-rank communicator
        h.comm_create(world, rank, bytes(uid.numpy().tobytes()))

    passes = [0]

    def step():
        """c4: one whole job — fresh state, scan + fold of the resident shard, cross-GPU exchange.
        c5: one more batch of the job — both handlers over the resident shard, every pass with later global sequence
        numbers than the table holds (the topic keeps growing; the state is NOT reset: clearing the 32 GiB table
        is no part of a batch), then the whole exchange: alive entries to their hash-range owners + all-reduces."""
        if c5:
            h.submit_device(batch, n, (passes[0] * world + rank) * n, which=3)
            passes[0] += 1
        else:
            h.reset()                                      # MessageMetrics::new state (tiny kernel)
            h.submit_device(batch, n, 0, which=1)          # scan + fold
        if exchange:
            h.exchange()                                   # snapshot + [c5: alive entries] + C1 SUM (counters) + C2 MAX (extrema), same stream

    def barrier():
        if exchange:
            dist.barrier()
        h.sync()
        torch.cuda.synchronize()

    # Pre-roll (untimed, not counted in --warmup): a fresh MI355X needs a few hundred ms under load to leave
    # its idle power state; without it the first bench process on a box measures the clock ramp (3.41 ms per
    # scan instead of 3.25 ms, same binary, same box).
    for _ in range(args.preroll):
        step()
    for _ in range(args.warmup):
        step()
    barrier()
    h.set_timing(True)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    avg_ms, cnt = h.kernel_time_stats()
    h.set_timing(False)

    if exchange:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    alive_total = None
    if exchange:
        # the exchange result itself: after one more step every rank must hold the whole job's totals
        step()
        barrier()
        xres, xc = h.exchange_result()
        got = int(xc[:, N.KTA_C_TOTAL].sum())
        want = n * world * (passes[0] if c5 else 1)
        assert got == want == xres.overall_count, "exchange step lost records: %d != %d" % (got, want)
        assert (xc[:, N.KTA_C_TOTAL] > 0).all(), "a partition of another rank is missing from the exchanged result"
        alive_total = int(xres.alive_keys) if c5 else None
        assert not c5 or 0 < alive_total <= 100_000_000
        h.comm_destroy()
    elif c5:
        alive_total = int(h.finish()[0].alive_keys)
    # sanity inside the bench: a single fresh pass must count exactly n records on this rank
    h.reset()
    h.submit_device(batch, n, 0, which=1)
    res, counters = h.finish()
    assert res.overall_count == n and int(counters[:, 0].sum()) == n, "scan lost records"
    own = counters[rank::world, 0]
    assert int(own.sum()) == n, "records outside this rank's partitions"

    if rank == 0:
        total_records = n * world * args.steps
        scan_ms = avg_ms[0]
        achieved = BYTES_PER_RECORD * n / (scan_ms * 1e-3) / 1e9
        traffic = _traffic("kta_metrics_scan", n)
        line = {
            "metric": METRIC, "value": total_records / elapsed, "unit": "records/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "preroll_steps": args.preroll, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "u64",
            "data": "synthetic",
            "config": {"workload": ("c5: 256-partition log-compacted topic, 100 M distinct 16-byte keys, 50 % tombstones, "
                                    "--count-alive-keys (both handlers + the alive-set exchange), partitions sharded p % n_gpus"
                                    if c5 else
                                    "c4: 256-partition synthetic topic, mixed key/value sizes (mean record "
                                    "~256 B), partitions sharded p % n_gpus"), "records_per_gpu": n,
                       "total_records_per_step": n * world, "partitions": P, "partition_order": args.part_mode,
                       "bytes_per_record": BYTES_PER_RECORD, "parallelism": f"partition-sharded x{world}",
                       "forced_collectives": bool(force_coll), "shared_device": bool(share_device),
                       "control_plane": "torch.distributed over gloo (rendezvous, barriers, max of the elapsed time)" if exchange else None,
                       "exchange": "none (1 GPU)" if not exchange else
                                   ("per step: kta_exchange = alive entries to their hash-range owners (count all-gather, "
                                    "grouped ncclSend / ncclRecv, owner merge + range count), then " if c5 else "per step: kta_exchange = ") +
                                   "one grouped RCCL launch of all-reduce SUM u64[%d] + all-reduce "
                                   "MAX i64[4] on the compute stream (csrc/kta_comm.hip)" % n_sum},
            "roofline": {"bound": "hbm", "kernel": "kta_metrics_scan", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": TRAFFIC_SOURCE,
                         "bytes_per_launch": BYTES_PER_RECORD * n, "kernel_ms": scan_ms, "launches": int(cnt[0]),
                         "fold_kernel_ms": avg_ms[1], "frac_of_measured_achievable_6290": achieved / 6290.0},
        }
        if c5:
            alive_ms = avg_ms[2]
            algo = (12 + 16) * n
            line["alive_pass"] = {"alive_keys": alive_total, "kernel_ms": alive_ms, "launches": int(cnt[2]),
                                  "roofline": {"bound": "hbm", "kernel": "kta_alive_partition + kta_alive_apply (table state)",
                                               "achieved": algo / (alive_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                               "frac": algo / (alive_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                               "bytes_per_launch": algo, "traffic": None}}
        if world == 1 and not args.no_cpu_baseline and not c5:
            line["cpu_baseline"] = cpu_baseline_metrics(h, batch, min(n, 1 << 26), P, args.cpu_seconds)
            line["cpu_baseline"]["gpu_over_cpu"] = line["value"] / line["cpu_baseline"]["value"]
    h.device_batch_free(batch)
    h.close()
    if rank == 0:
        if world == 1 and not args.no_alive and not c5:
            line["alive_pass"], line["both_handlers"] = alive_pass_report(kta, local_rank, max(3, args.steps // 5), 2,
                                                                          args.alive_records, args.cpu_seconds,
                                                                          extras=not args.no_alive_extras)
            line["alive_pass_table"], line["both_handlers_table"] = alive_table_report(kta, local_rank, 5, args.alive_records,
                                                                                       extras=not args.no_alive_extras)
            if not args.no_alive_extras:
                line["alive_pass_hot_key"] = alive_hot_key_report(kta, local_rank, 1 << 26)
        if world == 1 and not args.no_decode and not c5:
            line["kafka_decode"] = kafka_decode_report(kta, local_rank, max(3, args.steps // 5), 2,
                                                       args.decode_records, args.cpu_seconds)
        if world == 1 and not args.no_hostfed and not c5:
            # PCIe-inclusive legs (the link is the bound: never `value`)
            line["boundary_per_message"] = boundary_per_message_report(kta, local_rank)
            line["host_fed"] = host_fed_report(kta, local_rank)
            line["raw_log_e2e"] = raw_log_e2e_report(kta, local_rank)
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(line) + "\n").encode())
    if exchange:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
