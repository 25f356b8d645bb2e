"""Alive-key counts of bench.py's `-c` legs, from the C oracle (oracle/kta_oracle.c: LogCompactionInMemoryMetrics,
/root/reference/src/metric.rs:288-305) fed the generator's records in consumption order on the CPU.

    python tests/golden/make_bench_alive_counts.py > tests/golden/bench_alive_counts.json

bench.py compares what the GPU reports for these legs with the numbers in that file (a leg whose shape is not in the file
says "alive_keys_checked": false).  Needs no GPU: the generator's host side lives in libkta_hip.so (include/kta_synth.h).
Keys of the table: "<preset>:<distinct keys or 0 = the preset's>:<first record>:<records>"."""
import json
import os
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import kafka_topic_analyzer_amd as kta   # noqa: E402
from oracle_c import Oracle              # noqa: E402

CHUNK = 1 << 24


def alive_after(preset, distinct, n, checkpoints=()):
    """Alive keys after the first n records of the topic (and after each of `checkpoints` records on the way)."""
    sp, _ = kta.synth_preset(preset)
    if distinct:
        sp.n_distinct_keys = distinct
    o = Oracle(count_alive_keys=True)
    chunks = []
    cuts = sorted(set(list(checkpoints) + [n]))
    lo = 0
    for c in cuts:
        while lo < c:
            m = min(CHUNK, c - lo)
            chunks.append((lo, m))
            lo += m
    out = {}
    pool = ThreadPoolExecutor(6)
    ahead = [pool.submit(kta.synth_fill_host, sp, a, m, True) for a, m in chunks[:8]]
    for k, (a, m) in enumerate(chunks):
        cols = ahead.pop(0).result()
        if k + 8 < len(chunks):
            ahead.append(pool.submit(kta.synth_fill_host, sp, chunks[k + 8][0], chunks[k + 8][1], True))
        o.L.kto_run_soa(None, o.lc, m, cols["partition"].ctypes.data, cols["key_len"].ctypes.data, cols["val_len"].ctypes.data,
                        cols["ts_ms"].ctypes.data, cols["key_off"].ctypes.data, cols["key_bytes"].ctypes.data)
        if a + m in cuts:
            out[a + m] = int(o.alive_keys())
    pool.shutdown()
    o.close()
    return out


table = {}
n_hot, n_big = 1 << 26, 15 << 24
for d in (1, 40):
    table[f"c3:{d}:0:{n_hot}"] = alive_after("c3", d, n_hot)[n_hot]
table[f"c5:0:0:{n_big}"] = alive_after("c5", 0, n_big)[n_big]
table[f"c3:20000000:0:{n_big}"] = alive_after("c3", 20_000_000, n_big)[n_big]      # alive_pass_hot_key's "many_keys" row
# alive_pass / both_handlers resubmit records [0, n_big) (idempotent); alive_pass_table walks six consecutive batches
steps = [n_big * (k + 1) for k in range(6)]
for upto, v in alive_after("c3", 0, steps[-1], [n_hot] + steps).items():     # (n_hot: boundary_per_message's c3 row)
    table[f"c3:0:0:{upto}"] = v
json.dump({"source": "tests/golden/make_bench_alive_counts.py (C oracle over the generator's records, CPU)", "alive_keys": table},
          sys.stdout, indent=1)
print()
